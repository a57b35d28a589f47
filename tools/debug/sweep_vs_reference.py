#!/usr/bin/env python3
"""Wide GPU-vs-compiled-reference sweep (needs oracle/_ref on the box): fresh random seeds, several configurations, every stream
compared bit for bit -- encoder payloads, decoder PCM with description loss.  Not part of the test suite (minutes of CPU work on the
box); run through gpurun after larger kernel changes:   python tools/debug/sweep_vs_reference.py [streams] [packets] [seed0] [edge]
(edge: the un-speech-like signal families of solo_amd.synth.edge_stream instead of the speech-like generator)"""
import multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import refcodec as R


def ref_stream(args):
    seed, P, cfg, recv, edge = args
    from solo_amd.synth import edge_stream
    pcm = edge_stream(seed, P) if edge else R.synth_stream(seed, P)
    fms = cfg.get("framesize", 40)
    if fms == 20:                                       # the same signal as twice as many half-size packets
        pcm = pcm.reshape(2 * P, -1)[:P]
    e = R.RefEncoder("fix", rate=cfg["rate"], joint=cfg["joint"], dtx=cfg["dtx"], use_md_index=cfg["mdi"], framesize_ms=fms)
    recs = [e.encode(pcm[p]) for p in range(P)]
    e.close()
    d = R.RefDecoder("fix", joint=cfg["joint"], use_md_index=cfg["mdi"], framesize_ms=fms)
    out = []
    for (pl, n0, n1), m in zip(recs, recv):
        if n0 <= 0 or m == 0:
            x, ret = d.decode(pl if n0 > 0 else b"\0" * 16, n0 if n0 > 0 else 16, n1 if n0 > 0 else 0, 1)
        else:
            x, ret = d.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
        out.append(x)
    d.close()
    return recs, np.stack(out)


CFGS = [dict(rate=13600, joint=0, dtx=0, mdi=0, loss=0.3), dict(rate=13600, joint=1, dtx=0, mdi=1, loss=0.2),
        dict(rate=24000, joint=0, dtx=0, mdi=0, loss=0.1), dict(rate=13600, joint=0, dtx=1, mdi=0, loss=0.15),
        dict(rate=13600, joint=0, dtx=0, mdi=1, loss=0.2, framesize=20)]


def sweep(N, P, seed0, edge=False, cfgs=CFGS, log=print):
    """-> number of streams (over all configurations) whose encoder payloads or decoder PCM differ from the compiled reference"""
    import torch, solo_amd
    from solo_amd.synth import synth_batch, edge_batch
    bad = 0
    for ci, cfg in enumerate(cfgs):
        t0 = time.time()
        s0 = seed0 + ci * 100000
        pcm = edge_batch(s0, N, P) if edge else synth_batch(s0, N, P, workers=16)
        fms = cfg.get("framesize", 40)
        if fms == 20:
            pcm = np.ascontiguousarray(pcm.reshape(N, 2 * P, -1)[:, :P])
        rng = np.random.default_rng(s0)
        recv = ((rng.random((N, P)) >= cfg["loss"]).astype(np.uint8) | ((rng.random((N, P)) >= cfg["loss"]).astype(np.uint8) << 1))
        b = solo_amd.SoloBatch(N, rate=cfg["rate"], encoder=True, decoder=True, slot_bytes=512, joint=cfg["joint"], dtx=cfg["dtx"], use_md_index=cfg["mdi"], framesize_ms=fms)
        bits, nb, st = b.encode(torch.from_numpy(pcm).cuda())
        out, st2 = b.decode(bits, nb, torch.from_numpy(recv).cuda())
        torch.cuda.synchronize()
        hb, hn, ho = bits.cpu().numpy(), nb.cpu().numpy(), out.cpu().numpy()
        with mp.get_context("fork").Pool(min(32, mp.cpu_count())) as pool:
            ref = pool.map(ref_stream, [(s0 + i, P, cfg, [int(m) for m in recv[i]], edge) for i in range(N)], chunksize=4)
        nbad = 0
        first_bad = []
        for i, (recs, dec) in enumerate(ref):
            ok = all(hn[i, p, 0] == max(r[1], 0) * (1 if r[1] > 0 else 0) or (r[1] <= 0 and hn[i, p, 0] == 0) for p, r in enumerate(recs))
            ok = ok and all(hb[i, p, :r[1]].tobytes() == r[0] for p, r in enumerate(recs) if r[1] > 0 and hn[i, p, 0] == r[1])
            ok = ok and all(hn[i, p, 0] == r[1] and hn[i, p, 1] == r[2] for p, r in enumerate(recs) if r[1] > 0)
            ok = ok and np.array_equal(ho[i], dec)
            nbad += 0 if ok else 1
            if not ok and len(first_bad) < 8:
                first_bad.append(s0 + i)
        bad += nbad
        log("config %d %s: %d streams x %d packets (%s, first seed %d), %d mismatching streams, status enc %d dec %d (%.0f s) %s" % (
            ci, cfg, N, P, "edge" if edge else "speech-like", s0, nbad, int(st.abs().max()), int(st2.abs().max()), time.time() - t0, first_bad if nbad else ""))
        b.close()
    return bad


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 900000
    edge = len(sys.argv) > 4 and sys.argv[4] == "edge"
    bad = sweep(N, P, seed0, edge, log=lambda s: print(s, flush=True))
    print("SWEEP", "OK" if bad == 0 else "FAILED")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
