#!/usr/bin/env python3
"""Decoder fuzz on freshly generated streams: every trial encodes a synthetic stream with the compiled reference under a random
configuration (16 / 32 kHz mode, rate, description index symbol, 20 / 40 ms high-band frame, speech-like or edge input), corrupts
packets (byte errors, bursts, bit flips, truncated / extended length records) under random description loss and decodes them
with the compiled reference and with the host emulation (both decoder paths).  A stream is compared up to and including its first
rejected packet (see tests/test_emu_decoder.py::test_corrupted_payloads_vs_reference).  One process per trial.
    python tools/debug/fuzz_decoder_gen.py [trials] [first_seed]"""
import os, pickle, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import refcodec as R
import solo_testlib as T
from solo_amd.synth import edge_stream, synth_stream

P = 14
COLD = bool(os.environ.get("FUZZ_COLD"))          # FUZZ_COLD=1: the first seed % 4 packets of every stream are lost


def sequence(seed):
    """-> (configuration, [(mapped decoder arguments (payload, nBytes0, nBytes1, lostflag), packet was corrupted)]) of trial `seed`"""
    rng = np.random.default_rng(0xDEC0DE00 + seed)
    wb = bool(rng.integers(0, 2))
    mdi, joint = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    rate = int(rng.choice([16000, 24000, 32000, 40000] if wb else [10000, 13600, 20000, 24000]))
    src = edge_stream if rng.random() < 0.3 else synth_stream
    pcm = src(seed, 2 * P).reshape(P, 1280) if wb else src(seed, P)
    fs = 32000 if wb else 16000
    enc = R.RefEncoder("fix", rate=rate, joint=joint, use_md_index=mdi, samplerate=fs)
    recs = [enc.encode(pcm[p]) for p in range(P)]
    split = int(rng.integers(0, 2))
    p_hit = rng.choice([0.05, 0.15, 0.4])
    seq = []
    for p, (pl, n0, n1) in enumerate(recs):
        if n0 <= 0 or n0 > 1000:
            break
        pl = bytearray(pl[:n0])
        hit = rng.random() < p_hit
        if hit:
            kind = rng.integers(0, 4)
            if kind == 0:
                for _ in range(rng.integers(1, 4)):
                    pl[rng.integers(0, n0)] = rng.integers(0, 256)
            elif kind == 1:
                a0 = rng.integers(0, n0)
                for i in range(a0, min(n0, a0 + int(rng.integers(1, 9)))):
                    pl[i] = rng.integers(0, 256)
            elif kind == 2:
                i = rng.integers(0, n0)
                pl[i] ^= 1 << rng.integers(0, 8)
            else:                                              # the length record lies about where the second description starts
                n1 = int(np.clip(n1 + rng.integers(-3, 4), (4 if joint else 8) + 1, n0 - 1))
        mode = int(rng.integers(0, 4)) if p > 0 else 0
        if COLD and p < seed % 4:                               # leading packets never arrive: the decoder is still at its initial rate
            mode = 3
        a = R.map_loss(bytes(pl), n0, n1, mode == 1, mode == 2) if mode != 3 else (bytes(pl), n0, n1, 1)
        seq.append((a, bool(hit), (bytes(pl), n0, n1, {0: 3, 1: 2, 2: 1, 3: 0}[mode])))       # + the whole packet and its receive mask
    return dict(wb=wb, mdi=mdi, joint=joint, rate=rate, split=split, fs=fs), seq


def one(seed):
    cfg, seq = sequence(seed)
    wb, mdi, joint = cfg["wb"], cfg["mdi"], cfg["joint"]
    dr = R.RefDecoder("fix", joint=joint, use_md_index=mdi, samplerate=cfg["fs"])
    de = T.EmuDecoder(mdi | (2 if joint else 0), wb=wb, split=cfg["split"])
    hits = 0
    for p, (a, hit, _) in enumerate(seq):
        x, r1 = dr.decode(*a)
        y, r2 = de.decode(*a)
        if r1 == 0 and r2 == -12 and hit:
            return ("other_rate", hits)
        if r1 != r2:
            return ("BAD", seed, p, "rc", r1, r2, cfg)
        if r1 < 0:
            return ("rejected", hits)
        if not np.array_equal(x, y):
            return ("BAD", seed, p, "pcm", int(np.abs(x.astype(int) - y.astype(int)).max()), cfg)
        hits += int(hit)
    return ("ok", hits)


def run_isolated(seeds, workers):
    results, active, nxt = {}, {}, 0
    while nxt < len(seeds) or active:
        while nxt < len(seeds) and len(active) < workers:
            r, w = os.pipe()
            pid = os.fork()
            if pid == 0:
                os.close(r)
                try:
                    os.write(w, pickle.dumps(one(seeds[nxt])))
                finally:
                    os._exit(0)
            os.close(w)
            active[pid] = (seeds[nxt], r)
            nxt += 1
        pid, status = os.wait()
        seed, r = active.pop(pid)
        data = b""
        while True:
            chunk = os.read(r, 65536)
            if not chunk:
                break
            data += chunk
        os.close(r)
        results[seed] = pickle.loads(data) if status == 0 and data else ("crash", seed, status)
    return results


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    res = run_isolated(list(range(s0, s0 + n)), min(16, os.cpu_count() or 1))
    kinds = {}
    for v in res.values():
        kinds[v[0]] = kinds.get(v[0], 0) + 1
    bad = [v for v in res.values() if v[0] == "BAD"]
    crashed = [v[1] for v in res.values() if v[0] == "crash"]
    print("DECODER GEN FUZZ", "OK" if not bad else "MISMATCH", kinds, "accepted corrupted packets:", sum(v[1] for v in res.values() if v[0] in ("ok", "rejected", "other_rate")),
          bad[:8], ("crashed seeds %s" % crashed[:10]) if crashed else "")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
