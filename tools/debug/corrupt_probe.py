#!/usr/bin/env python3
"""Debug: the corrupted-payload scenario of tests/test_pinned_corners.py on the GPU; reports, per stream, the first packet whose
PCM / return code differs from the compiled reference, and saves the scenario for analysis with the host emulation."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, solo_amd
import refcodec as R
import solo_testlib as T
z = np.load(T.GOLDEN + "/synth8x25.npz")
bits, nb = z["bits"], z["nbytes"]
N, P, S = 48, 25, bits.shape[2]
rng = np.random.default_rng(5)
cb = np.zeros((N, P, S), np.uint8); cn = np.zeros((N, P, 2), np.int16); recv = np.zeros((N, P), np.uint8); hit = np.zeros((N, P), bool)
for t in range(N):
    s = t % 8
    for p in range(P):
        n0 = int(nb[s, p, 0]); pl = bits[s, p].copy()
        hit[t, p] = rng.random() < 0.25
        if hit[t, p]:
            for _ in range(rng.integers(1, 4)):
                pl[rng.integers(0, n0)] = rng.integers(0, 256)
        cb[t, p] = pl; cn[t, p] = nb[s, p]
        mode = rng.integers(0, 4)
        recv[t, p] = (0 if mode == 1 else 1) | (0 if mode == 2 else 2)
d = solo_amd.SoloBatch(N, encoder=False, decoder=True, slot_bytes=S)
dev = d.device
got = np.zeros((N, P, 640), np.int16); rets = np.zeros((N, P), np.int32)
for p in range(P):
    pcm, st = d.decode(torch.from_numpy(np.ascontiguousarray(cb[:, p:p + 1])).to(dev), torch.from_numpy(np.ascontiguousarray(cn[:, p:p + 1])).to(dev),
                       torch.from_numpy(np.ascontiguousarray(recv[:, p:p + 1])).to(dev))
    got[:, p] = pcm.cpu().numpy()[:, 0]; rets[:, p] = st.cpu().numpy()
np.savez(os.path.join(ROOT, "gpurun_out", "corrupt_probe.npz"), cb=cb, cn=cn, recv=recv, hit=hit, got=got, rets=rets)
for t in range(N):
    dr, de = R.RefDecoder("fix"), T.EmuDecoder()
    for p in range(P):
        n0, n1 = int(cn[t, p, 0]), int(cn[t, p, 1]); m = int(recv[t, p])
        a = R.map_loss(cb[t, p, :n0].tobytes(), n0, n1, not (m & 1), not (m & 2))
        x, r1 = dr.decode(*a); y, r3 = de.decode(*a)
        r2 = int(rets[t, p])
        if r1 != r2 or (r1 == 0 and not np.array_equal(x, got[t, p])) or r1 != r3 or (r1 == 0 and not np.array_equal(x, y)):
            bad = np.nonzero(x != got[t, p])[0]
            print("stream %d packet %d hit=%s recv=%d ref_ret=%d gpu_ret=%d emu_ret=%d emu==ref:%s first_bad_sample=%s n_bad=%d" % (
                t, p, hit[t, p], m, r1, r2, r3, np.array_equal(x, y), bad[:1], bad.size))
            break
        if r1 < 0:
            break
print("done")
