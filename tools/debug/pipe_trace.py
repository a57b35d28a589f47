#!/usr/bin/env python3
"""Where the wavefronts of the persistent encoder pipeline spend a call (a -DSX_PIPE_TRACE build: tools/build_variant_all.sh trace -DSX_PIPE_TRACE):
SOLO_LIB_OVERRIDE=build/libsolo_trace.so python tools/debug/pipe_trace.py [streams] [packets]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, solo_amd
from solo_amd.synth import synth_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 50
lib = solo_amd.load_library()
b = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512)
x = torch.from_numpy(synth_batch(0, N, P, workers=8)).cuda()
bits, nb, st = b.encode(x); torch.cuda.synchronize()
for rep in range(2):
    b.encode(x, bits, nb, st); torch.cuda.synchronize()
f = np.zeros((N, 6), np.uint64); q = np.zeros(((N + 3) // 4, 6), np.uint64)
assert lib.solo_debug_front_trace(f.ctypes.data_as(ctypes.c_void_p), N) == 0
assert lib.solo_debug_nsq_trace(q.ctypes.data_as(ctypes.c_void_p), (N + 3) // 4) == 0
t0 = min(f[:, 0].min(), q[:, 0].min())
ms = lambda a: (a.astype(np.int64) - np.int64(t0)) / 1e5
pc = lambda a: " ".join("%7.2f" % v for v in np.percentile(a, [0, 5, 25, 50, 75, 95, 100]))
print("percentiles (ms since the first wavefront started):      min      5      25      50      75      95     max")
print("front start                                        ", pc(ms(f[:, 0])))
print("front analysis of the last packet done             ", pc(ms(f[:, 1])))
print("front exit                                         ", pc(ms(f[:, 2])))
print("front packets coded before the last analysis ended ", pc(f[:, 3].astype(np.float64)))
print("quantiser start                                    ", pc(ms(q[:, 0])))
print("quantiser first packet done                        ", pc(ms(q[:, 1])))
print("quantiser exit                                     ", pc(ms(q[:, 2])))
print("quantiser ms spent waiting for analysis flags      ", pc(q[:, 3].astype(np.float64) / 1e5))

# placement: (XCC, SE, SH, CU) -> wavefronts per SIMD (HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID [3:0])
def place(a, name):
    hw = a[:, 4].astype(np.int64); xcc = a[:, 5].astype(np.int64) & 15
    cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    simd = (hw >> 4) & 3
    from collections import Counter
    per = Counter()
    for c, sd in zip(cu, simd): per[(c, sd)] += 1
    cus = sorted(set(cu))
    pat = Counter(tuple(per[(c, k)] for k in range(4)) for c in cus)
    print("%s: %d compute units; wavefronts per SIMD (pattern: units): %s" % (name, len(cus), ", ".join("%s: %d" % (p, n) for p, n in pat.most_common(12))))
place(f, "front"); place(q, "quantiser")
# how many front wavefronts had exited when the k-th quantiser wavefront started
fe = np.sort(ms(f[:, 2])); qs = np.sort(ms(q[:, 0]))
for k in (0, len(qs) // 100, len(qs) // 10, len(qs) // 4, len(qs) // 2, 3 * len(qs) // 4, len(qs) - 1):
    print("quantiser wavefront #%4d started at %7.3f ms: %4d front wavefronts had exited" % (k + 1, qs[k], int(np.searchsorted(fe, qs[k]))))
