#!/usr/bin/env python3
"""Long-running fuzz of the decoder's host emulation (the kernel source compiled for the host, tests/emu) against the compiled
reference (oracle/_ref, this container only): bit errors, byte bursts, random description loss; both decode paths
(single kernel / extraction + synthesis).  Same acceptance rule as tests/test_emu_decoder.py::test_corrupted_payloads_vs_reference.
    python tools/debug/fuzz_decoder_emu.py [trials] [seed] [wb]      (wb: the 32 kHz mode, tests/golden/wb4x20.npz)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import solo_testlib as T
import refcodec as R

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
wb = len(sys.argv) > 3 and sys.argv[3] == "wb"
z = np.load(T.GOLDEN + ("/wb4x20.npz" if wb else "/synth8x25.npz"))
bits, nb = z["bits"], z["nbytes"]
NS, NP = nb.shape[0], nb.shape[1]
rng = np.random.default_rng(seed)
stats = dict(rejected=0, garbage=0, other_rate=0, packets=0)
bad = []
for trial in range(trials):
    s = trial % NS
    split = (trial // NS) & 1
    T.EmuDecoder.SPLIT = split
    joint = 1 if (wb and s == 3) else 0                       # (stream 3 of the wide-band fixture was coded with joint_mode 1)
    dr = R.RefDecoder("fix", joint=joint, samplerate=32000 if wb else 16000)
    de = T.EmuDecoder(2 if joint else 0, wb=wb)
    p_hit = rng.choice([0.03, 0.1, 0.25])
    for p in range(NP):
        n0, n1 = int(nb[s, p, 0]), int(nb[s, p, 1])
        pl = bytearray(bits[s, p, :n0].tobytes())
        hit = rng.random() < p_hit
        if hit:
            kind = rng.integers(0, 3)
            if kind == 0:
                for _ in range(rng.integers(1, 4)):
                    pl[rng.integers(0, n0)] = rng.integers(0, 256)
            elif kind == 1:                                   # burst
                a0 = rng.integers(0, n0); ln = rng.integers(1, 9)
                for i in range(a0, min(n0, a0 + ln)):
                    pl[i] = rng.integers(0, 256)
            else:                                             # single bit flip
                i = rng.integers(0, n0); pl[i] ^= 1 << rng.integers(0, 8)
        mode = rng.integers(0, 4)
        a = R.map_loss(bytes(pl), n0, n1, mode == 1, mode == 2)
        x, r1 = dr.decode(*a)
        y, r2 = de.decode(*a)
        stats["packets"] += 1
        if r1 == 0 and r2 == -12 and hit:
            stats["other_rate"] += 1
            break
        if r1 != r2:
            bad.append((trial, split, p, "rc", r1, r2)); break
        if r1 < 0:
            stats["rejected"] += 1
            break
        if not np.array_equal(x, y):
            bad.append((trial, split, p, "pcm", int(np.abs(x.astype(int) - y.astype(int)).max()))); break
        stats["garbage"] += int(hit)
    if trial % 200 == 199:
        print(trial + 1, stats, "mismatches:", len(bad), flush=True)
print("FUZZ", "OK" if not bad else "MISMATCH", stats, bad[:10])
sys.exit(1 if bad else 0)
