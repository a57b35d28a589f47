#!/usr/bin/env python3
"""Encoder + decoder host emulation (the kernel source compiled for the host, tests/emu) against the compiled reference
(oracle/_ref, this container only) on the un-speech-like signal families of solo_amd.synth.edge_stream, several encoder
configurations, random description loss.   python tools/debug/fuzz_encoder_emu.py [streams] [packets] [first_seed] [wb]
(wb: the 32 kHz mode -- 1280-sample packets, 24 kbps, SILK wide band inside; random: speech-like streams at a random level under a
random configuration -- sampling mode, rate, high-band framing, DTX, description index -- encoder only)"""
import multiprocessing as mp, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import refcodec as R
import solo_testlib as T
from solo_amd.synth import edge_stream, EDGE_FAMILIES

CFGS = [dict(rate=13600, joint=0, dtx=0, mdi=0), dict(rate=13600, joint=1, dtx=0, mdi=1), dict(rate=24000, joint=0, dtx=0, mdi=0),
        dict(rate=13600, joint=0, dtx=1, mdi=0), dict(rate=10000, joint=0, dtx=0, mdi=0)]


WB_CFGS = [dict(rate=24000, joint=0, dtx=0, mdi=0), dict(rate=24000, joint=1, dtx=0, mdi=0), dict(rate=32000, joint=0, dtx=0, mdi=1),
           dict(rate=18000, joint=0, dtx=1, mdi=0)]


def one_random(seed, P):
    """random configuration (sampling mode, rate, high-band framing, DTX, description index) on a speech-like stream at a random level"""
    from solo_amd.synth import synth_stream
    rng = np.random.default_rng(0xE0C0DE00 + seed)
    wb = bool(rng.integers(0, 2))
    joint, dtx, mdi = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
    rate = int(rng.integers(15600, 48000)) if wb else int(rng.integers(5000, 40000))
    gain = float(10.0 ** rng.uniform(-2.0, 0.45))
    pcm = synth_stream(seed, 2 * P if wb else P).astype(np.float64) * gain
    if dtx and rng.random() < 0.7:                                      # long near-silent stretches so that DTX engages
        a0 = int(rng.integers(0, pcm.shape[0] // 2))
        pcm[a0:a0 + int(rng.integers(8, 40))] *= 0.0005
    pcm = np.clip(np.rint(pcm), -32768, 32767).astype(np.int16).reshape(P, 1280 if wb else 640)
    flags = mdi | (2 if joint else 0) | (4 if dtx else 0)
    e = T.EmuEncoder(rate, flags, wb=wb)
    r = R.RefEncoder("fix", rate=rate, joint=joint, dtx=dtx, use_md_index=mdi, samplerate=32000 if wb else 16000)
    for p in range(P):
        a, b = e.encode(pcm[p]), r.encode(pcm[p])
        if a != b:
            return (seed, p, "enc", dict(wb=wb, rate=rate, joint=joint, dtx=dtx, mdi=mdi, gain=gain), a[1:], b[1:])
    return None


def one(args):
    seed, P, wb = args
    if wb == "random":
        return one_random(seed, P)
    if wb:
        return one_wb(seed, P)
    cfg = CFGS[(seed // EDGE_FAMILIES) % len(CFGS)]
    pcm = edge_stream(seed, P)
    rng = np.random.default_rng(seed)
    flags = cfg["mdi"] | (2 if cfg["joint"] else 0) | (4 if cfg["dtx"] else 0)          # flag word of the emulation: MD index, joint mode, DTX
    e = T.EmuEncoder(cfg["rate"], flags)
    r = R.RefEncoder("fix", rate=cfg["rate"], joint=cfg["joint"], dtx=cfg["dtx"], use_md_index=cfg["mdi"])
    T.EmuDecoder.SPLIT = seed & 1
    de = T.EmuDecoder(flags & 3)
    dr = R.RefDecoder("fix", joint=cfg["joint"], use_md_index=cfg["mdi"])
    for p in range(P):
        a, b = e.encode(pcm[p]), r.encode(pcm[p])
        if a != b:
            return (seed, p, "enc", cfg)
        pl, n0, n1 = b
        m = int(rng.integers(0, 4)) if p > 0 else 3
        if n0 <= 0:
            args_ = (b"", 16, 0, 1)                                   # DTX: nothing was sent, the receiver conceals
        else:
            args_ = (pl, n0, n1, 1) if m == 0 else R.map_loss(pl, n0, n1, not (m & 1), not (m & 2))
        x, r1 = dr.decode(*args_)
        y, r2 = de.decode(*args_)
        if r1 != r2 or not np.array_equal(x, y):
            return (seed, p, "dec", cfg, r1, r2)
    return None


def one_wb(seed, P):
    cfg = WB_CFGS[(seed // EDGE_FAMILIES) % len(WB_CFGS)]
    pcm = edge_stream(seed, 2 * P).reshape(P, 1280)
    rng = np.random.default_rng(seed)
    flags = cfg["mdi"] | (2 if cfg["joint"] else 0) | (4 if cfg["dtx"] else 0)
    e = T.EmuEncoder(cfg["rate"], flags, wb=True)
    r = R.RefEncoder("fix", rate=cfg["rate"], joint=cfg["joint"], dtx=cfg["dtx"], use_md_index=cfg["mdi"], samplerate=32000)
    T.EmuDecoder.SPLIT = seed & 1
    de = T.EmuDecoder(flags & 3, wb=True)
    dr = R.RefDecoder("fix", joint=cfg["joint"], use_md_index=cfg["mdi"], samplerate=32000)
    for p in range(P):
        a, b = e.encode(pcm[p]), r.encode(pcm[p])
        if a != b:
            return (seed, p, "enc", cfg)
        pl, n0, n1 = b
        m = int(rng.integers(0, 4)) if p > 0 else 3
        if n0 <= 0:
            args_ = (b"", 16, 0, 1)
        else:
            args_ = (pl, n0, n1, 1) if m == 0 else R.map_loss(pl, n0, n1, not (m & 1), not (m & 2))
        x, r1 = dr.decode(*args_)
        y, r2 = de.decode(*args_)
        if r1 != r2 or not np.array_equal(x, y):
            return (seed, p, "dec", cfg, r1, r2)
    return None


def run_isolated(tasks, workers):
    """one forked process per task (a crash of the compiled reference -- it has stack overflows of its own on some extreme inputs --
    must not take the sweep down): -> list of results, ("crash", wait status) for a task whose process died"""
    import pickle
    results = [None] * len(tasks)
    active = {}
    nxt = 0
    while nxt < len(tasks) or active:
        while nxt < len(tasks) and len(active) < workers:
            r, w = os.pipe()
            pid = os.fork()
            if pid == 0:
                os.close(r)
                try:
                    os.write(w, pickle.dumps(one(tasks[nxt])))
                finally:
                    os._exit(0)
            os.close(w)
            active[pid] = (nxt, r)
            nxt += 1
        pid, status = os.wait()
        idx, r = active.pop(pid)
        data = b""
        while True:
            chunk = os.read(r, 65536)
            if not chunk:
                break
            data += chunk
        os.close(r)
        results[idx] = pickle.loads(data) if status == 0 and data else ("crash", tasks[idx][0], status)
    return results


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 280
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    s0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    wb = (len(sys.argv) > 4 and sys.argv[4] == "wb") or (len(sys.argv) > 4 and sys.argv[4] == "random" and "random")
    res = run_isolated([(s0 + i, P, wb) for i in range(N)], min(16, mp.cpu_count()))
    crashed = [r[1] for r in res if r and r[0] == "crash"]
    bad = [r for r in res if r and r[0] != "crash"]
    print("EDGE FUZZ", "OK" if not bad else "MISMATCH", "%d streams x %d packets" % (N, P), bad[:12],
          ("; process died (compiled reference) on seeds %s" % crashed[:20]) if crashed else "")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
