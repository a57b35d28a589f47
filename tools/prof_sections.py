#!/usr/bin/env python3
"""Per-section cycle breakdown of the encoder (debug build with -DSX_PROF, see solo_enc.h SX_T).
  SOLO_LIB_OVERRIDE=build/libsolo_prof.so python tools/prof_sections.py [streams] [packets]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, solo_amd
from solo_amd.synth import synth_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10
NAMES = {0: "qmf", 1: "vad+hp", 2: "pitch", 3: "noise_shape", 4: "prefilter", 5: "find_pred_coefs", 6: "process_gains", 7: "nsq",
         8: "frame_end", 9: "hb", 10: "range_coding", 11: "output", 15: "(frame call overhead)",
         16: "  fpc: ltp+filter", 17: "  fpc: find_LPC total(dup)", 18: "  fpc: interp_search", 19: "  fpc: msvq", 20: "  fpc: res_energy", 21: "  fpc: burg2+a2nlsf", 22: "  fpc: burg2 only", 23: "    burg(all 3): sum_sqr", 24: "    burg(all 3): first row", 25: "    burg(all 3): main loop", 26: "  noise_shape: per-subframe lanes (schur .. limit)"}
b = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512)
pcm = torch.from_numpy(synth_batch(0, N, P)).cuda()
b.encode(pcm); torch.cuda.synchronize()
lib = solo_amd.load_library()
buf = (ctypes.c_ulonglong * 64)()
lib.solo_debug_prof_enc(buf, 1)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); b.encode(pcm); t1.record(); torch.cuda.synchronize()
lib.solo_debug_prof_enc(buf, 1)
tot = sum(buf)
print("encode %.2f ms for %d packets -> %.0f packets/s" % (t0.elapsed_time(t1), N * P, N * P / t0.elapsed_time(t1) * 1e3))
for i in range(64):
    if buf[i]:
        print("%-22s %8.0f cycles/packet  %5.1f %%" % (NAMES.get(i, str(i)), buf[i] / (N * P), 100.0 * buf[i] / tot))
print("total %.0f cycles/packet" % (tot / (N * P)))
