#!/usr/bin/env python3
"""How long the HOST needs to enqueue one encode / decode call (all launches, events) vs how long the GPU needs to run it."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, solo_amd
from solo_amd.synth import synth_batch
N, P = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 50
b = solo_amd.SoloBatch(N, encoder=True, decoder=True, slot_bytes=512)
x = torch.from_numpy(synth_batch(0, N, P, workers=16)).cuda()
bits, nb, st = b.encode(x); out, st2 = b.decode(bits, nb); torch.cuda.synchronize()
for it in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    he = hd = 0.0
    for _ in range(5):
        a = time.perf_counter(); b.encode(x, bits, nb, st); c = time.perf_counter(); b.decode(bits, nb, None, out, st2); d = time.perf_counter()
        he += c - a; hd += d - c
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue per step: encode %.2f ms decode %.2f ms | 5 steps: enqueue done after %.1f ms, GPU done after %.1f ms -> %.2f ms/step" % (
        he / 5 * 1e3, hd / 5 * 1e3, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) / 5 * 1e3))
for it in range(2):
    te = td = 0.0
    for _ in range(5):
        torch.cuda.synchronize(); a = time.perf_counter(); b.encode(x, bits, nb, st); torch.cuda.synchronize(); c = time.perf_counter()
        b.decode(bits, nb, None, out, st2); torch.cuda.synchronize(); d = time.perf_counter()
        te += c - a; td += d - c
    print("synchronised after every call: encode %.2f ms decode %.2f ms" % (te / 5 * 1e3, td / 5 * 1e3))
for it in range(2):
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(5): b.encode(x, bits, nb, st)
    torch.cuda.synchronize(); c = time.perf_counter()
    for _ in range(5): b.decode(bits, nb, None, out, st2)
    torch.cuda.synchronize(); d = time.perf_counter()
    print("5 encodes back to back: %.2f ms each; 5 decodes back to back: %.2f ms each" % ((c - a) / 5 * 1e3, (d - c) / 5 * 1e3))
