#!/usr/bin/env python3
"""Per-phase cycle breakdown of the quantiser kernel (debug build with -DSX_PROF).
  SOLO_LIB_OVERRIDE=build/libsolo_prof.so python tools/prof_nsq.py [streams] [packets]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, solo_amd
from solo_amd.synth import synth_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10
NAMES = {1: "setup / subframe prologue / frame end", 2: "A predict+shape+residual (3 tracks)", 3: "B+C candidates (in-lane)", 4: "E judge: winner, expiry",
         5: "E replace-worst-by-best rounds (index registers)", 6: "survivor gather (bpermute)", 7: "D undo + F emit", 8: "G update + tap rotation", 9: "frame epilogue"}
b = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512)
pcm = torch.from_numpy(synth_batch(0, N, P)).cuda()
b.encode(pcm); torch.cuda.synchronize()
lib = solo_amd.load_library()
buf = (ctypes.c_ulonglong * 32)()
lib.solo_debug_prof_nsq(buf, 1)
b.encode(pcm); torch.cuda.synchronize()
lib.solo_debug_prof_nsq(buf, 1)
tot = sum(buf)
waves = N / 16.0
print("quantiser: %.0f cycles per wave-packet (16 streams)" % (tot / (waves * P)))
for i in range(32):
    if buf[i]:
        print("%-40s %9.0f cycles/wave-sample  %5.1f %%" % (NAMES.get(i, str(i)), buf[i] / (waves * P * 320), 100.0 * buf[i] / tot))
