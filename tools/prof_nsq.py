#!/usr/bin/env python3
"""Per-phase cycle breakdown of the quantiser kernel (debug build with -DSX_PROF).
  SOLO_LIB_OVERRIDE=build/libsolo_prof.so python tools/prof_nsq.py [streams] [packets]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, solo_amd
from solo_amd.synth import synth_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10
NAMES = {1: "frame setup + prologue: tap windows to LDS + frame end", 2: "A predict+shape+residual (3 tracks)", 3: "B+C candidates (in-lane)", 4: "E judge: winner, expiry",
         5: "E replace-worst-by-best rounds (index registers)", 6: "survivor gather (bpermute)", 7: "F emit (stores)", 8: "G update + tap rotation", 9: "frame epilogue",
         10: "D undo + joint winner", 11: "wait for the ring cells of this sample (vmcnt 0)", 12: "ring refill requests (4 loads)",
         13: "prologue: coefficients, gains", 14: "prologue: flush of the winner's lineage (k = 2)", 15: "prologue: re-whitening", 16: "prologue: history rescaling"}
b = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512)
pcm = torch.from_numpy(synth_batch(0, N, P)).cuda()
b.encode(pcm); torch.cuda.synchronize()
lib = solo_amd.load_library()
buf = (ctypes.c_ulonglong * 32)()
lib.solo_debug_prof_nsq(buf, 1)
b.encode(pcm); torch.cuda.synchronize()
lib.solo_debug_prof_nsq(buf, 1)
core, real = buf[30], buf[31]
buf[30] = buf[31] = 0
tot = sum(buf)
if real:
    print("shader clock seen by the quantiser's waves: %.0f MHz (s_memtime ticks per 100 MHz tick x 100)" % (100.0 * core / real))
waves = N / 16.0
print("quantiser: %.0f cycles per wave-packet (16 streams)" % (tot / (waves * P)))
for i in range(32):
    if buf[i]:
        print("%-40s %9.0f cycles per sample step  %5.1f %%" % (NAMES.get(i, str(i)), buf[i] / (waves * P * 640), 100.0 * buf[i] / tot))
