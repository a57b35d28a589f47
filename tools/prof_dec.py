#!/usr/bin/env python3
"""Per-section cycle breakdown of the decoder (debug build with -DSX_PROF).
  SOLO_LIB_OVERRIDE=build/libsolo_prof.so python tools/prof_dec.py [streams] [packets]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, solo_amd
from solo_amd.synth import synth_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10
NAMES = {0: "parse (2 lanes)", 1: "merge + inverse NSQ", 2: "decode_core", 3: "plc update", 4: "outBuf + glue", 5: "cng", 6: "hb (both frames)",
         7: "qmf synthesis", 8: "  hb: nlsf2a (dup)"}
b = solo_amd.SoloBatch(N, encoder=True, decoder=True, slot_bytes=512)
pcm = torch.from_numpy(synth_batch(0, N, P)).cuda()
bits, nb, st = b.encode(pcm)
b.decode(bits, nb, None); torch.cuda.synchronize()
lib = solo_amd.load_library()
buf = (ctypes.c_ulonglong * 64)()
lib.solo_debug_prof(buf, 1)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); b.decode(bits, nb, None); t1.record(); torch.cuda.synchronize()
lib.solo_debug_prof(buf, 1)
tot = sum(buf[i] for i in range(8))
print("decode %.2f ms for %d packets -> %.0f packets/s" % (t0.elapsed_time(t1), N * P, N * P / t0.elapsed_time(t1) * 1e3))
for i in range(64):
    if buf[i]:
        print("%-22s %8.0f cycles/packet  %5.1f %%" % (NAMES.get(i, str(i)), buf[i] / (N * P), 100.0 * buf[i] / tot))
print("total %.0f cycles/packet" % (tot / (N * P)))
