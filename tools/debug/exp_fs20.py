#!/usr/bin/env python3
"""Timing experiment: the same 2 s of audio per stream as 50 packets of 40 ms and as 100 packets of 20 ms (framesize_ms = 20: one frame per
chunk of the encoder pipeline) -- does a finer pipeline granularity change the encode rate?   python tools/debug/exp_fs20.py [streams]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, solo_amd
from solo_amd.synth import synth_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x40 = torch.from_numpy(synth_batch(0, N, 50, workers=16)).cuda()
x20 = x40.reshape(N, 100, -1).contiguous()
for name, x, fs in (("40 ms", x40, 40), ("20 ms", x20, 20), ("40 ms", x40, 40), ("20 ms", x20, 20)):
    b = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512, framesize_ms=fs)
    b.encode(x); torch.cuda.synchronize()
    t = []
    for r in range(3):
        t0 = time.perf_counter(); b.encode(x); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    print("%s packets: encode of %d x %d: %.2f ms (best of 3)" % (name, N, x.shape[1], 1e3 * min(t)))
    del b
