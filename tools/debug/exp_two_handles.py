#!/usr/bin/env python3
"""Experiment: the 4096 streams of the benchmark as G handles of 4096 / G streams, every handle's encode call on a HIP stream of its own
(the launch-per-chunk pipelines of the handles then run side by side: a chunk's analysis launch only waits for the stragglers of ITS
group).   python tools/debug/exp_two_handles.py [groups=2] [streams=4096] [packets=50]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, solo_amd
from solo_amd.synth import synth_batch
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
P = int(sys.argv[3]) if len(sys.argv) > 3 else 50
x = torch.from_numpy(synth_batch(0, N, P, workers=16)).cuda()
ref = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512)
rb, rn, rs = ref.encode(x)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t = 0.0
for _ in range(3):
    ev0.record(); ref.encode(x, rb, rn, rs); ev1.record(); torch.cuda.synchronize(); t += ev0.elapsed_time(ev1)
print("one handle of %d streams: encode %.2f ms" % (N, t / 3))
n = N // G
hs = [solo_amd.SoloBatch(n, encoder=True, decoder=False, slot_bytes=512) for _ in range(G)]
ss = [torch.cuda.Stream() for _ in range(G)]
xs = [x[g * n:(g + 1) * n].contiguous() for g in range(G)]
outs = [h.encode(xg) for h, xg in zip(hs, xs)]
torch.cuda.synchronize()
ok = all(np.array_equal(o[1].cpu().numpy(), rn_.cpu().numpy()) for o, rn_ in zip(outs, [ref.encode(x)[1][g * n:(g + 1) * n] for g in range(G)][:0] or []))
first = ref2 = None
# parity of the split run against the single handle (first call of fresh handles on both sides)
ref_f = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512)
fb, fn, fs = ref_f.encode(x); torch.cuda.synchronize()
same = all(np.array_equal(outs[g][1].cpu().numpy(), fn[g * n:(g + 1) * n].cpu().numpy()) and np.array_equal(outs[g][0].cpu().numpy(), fb[g * n:(g + 1) * n].cpu().numpy()) for g in range(G))
t = 0.0
cur = torch.cuda.current_stream()
for _ in range(3):
    ev0.record()
    for g in range(G):
        ss[g].wait_stream(cur)
        with torch.cuda.stream(ss[g]):
            hs[g].encode(xs[g], *outs[g])
    for g in range(G): cur.wait_stream(ss[g])
    ev1.record(); torch.cuda.synchronize(); t += ev0.elapsed_time(ev1)
print("%d handles of %d streams side by side: encode %.2f ms (%.0f pkt/s)  equal to the single handle: %s" % (G, n, t / 3, N * P * 3 / t * 1e3, same))
