#!/usr/bin/env python3
"""How much of an encoder launch is its tail?  Encode time of N distinct streams against N copies of ONE stream (every wave of a launch
then has the same work: no early finishers, no stragglers), for a few choices of that stream.   python tools/debug/uniform_streams.py [N] [P]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, solo_amd
from solo_amd.synth import synth_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x = synth_batch(0, N, P, workers=16)
b = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]


def run(pcm, label):
    d = torch.from_numpy(pcm).cuda()
    b.reset(); bits, nb, st = b.encode(d); torch.cuda.synchronize()
    t = []
    for _ in range(3):
        b.reset(); torch.cuda.synchronize()
        ev[0].record(); b.encode(d, bits, nb, st); ev[1].record(); torch.cuda.synchronize()
        t.append(ev[0].elapsed_time(ev[1]))
    b.set_timing(True); b.reset(); b.encode(d, bits, nb, st); torch.cuda.synchronize()
    k = b.last_kernel_ms(); b.set_timing(False)
    print("%-34s encode %.2f ms (%.3f ms / packet)  mean payload %.1f B  kernels: %s" % (
        label, min(t), min(t) / P, float(nb[:, :, 0].float().mean()), "  ".join("%s %.2f" % kv for kv in k.items())))


run(x, "%d distinct streams" % N)
# per-stream cost proxy: payload bytes (voiced, active streams code more); pick the cheapest, the median and the most expensive one
bits, nb, st = b.encode(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
tot = nb[:, :, 0].float().sum(1).cpu().numpy()
order = np.argsort(tot)
for name, i in (("smallest payload", order[0]), ("median payload", order[N // 2]), ("largest payload", order[-1])):
    run(np.repeat(x[i:i + 1], N, axis=0).copy(), "%d copies of stream %d (%s)" % (N, i, name))
