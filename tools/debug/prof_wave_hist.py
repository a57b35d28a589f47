#!/usr/bin/env python3
"""Lifetimes of the analysis kernel's waves, by SIMD (debug build with -DSX_PROF): is there a straggler class?
  SOLO_LIB_OVERRIDE=build/libsolo_prof.so python tools/debug/prof_wave_hist.py [streams] [packets]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, solo_amd
from solo_amd.synth import synth_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 20
b = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512)
pcm = torch.from_numpy(synth_batch(0, N, P)).cuda()
b.encode(pcm); torch.cuda.synchronize()
lib = solo_amd.load_library()
buf = (ctypes.c_ulonglong * 256)()
lib.solo_debug_hist(buf, 1)
b.encode(pcm); torch.cuda.synchronize()
lib.solo_debug_hist(buf, 1)
print("analysis waves by SIMD and lifetime (50 us bins), %d streams x %d packets, SOLO_ENC_CHUNK=%s SOLO_EXP_SKIP=%s" % (N, P, os.environ.get("SOLO_ENC_CHUNK", "1"), os.environ.get("SOLO_EXP_SKIP", "0")))
for simd in range(4):
    row = [buf[simd * 64 + i] for i in range(64)]
    tot = sum(row)
    mean = sum((i + 0.5) * 0.05 * c for i, c in enumerate(row)) / max(tot, 1)
    print("SIMD %d: %7d waves, mean %.3f ms | " % (simd, tot, mean) + " ".join("%d:%d" % (i, c) for i, c in enumerate(row) if c))
