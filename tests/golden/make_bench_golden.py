#!/usr/bin/env python3
"""Reference hashes of bench.py's first step: binds the benchmarked batch to parity.

For every 4096-stream block b of the synthetic workload (global streams [4096 b, 4096 (b + 1)), 50 packets per stream, 13.6 kbps,
both descriptions received) the COMPILED REFERENCE (oracle/_ref/libsolo_ref_fix.so) encodes and decodes every stream from a
fresh state; the block's hashes are

    payload_md5 = md5( nbytes[4096, 50, 2] int16  ||  bits[4096, 50, 512] uint8 (zero past each payload) )
    pcm_md5     = md5( pcm[4096, 50, 640] int16 )

exactly what solo_amd.dist.block_hashes computes from the GPU's output buffers.  bench.py compares its first step (freshly reset
streams) block by block: N = 1 covers block 0 (BASELINE configs[1] / [2]), the 8192-stream legs blocks 0-1, rank r of an 8-GPU
run blocks 2 r and 2 r + 1 (configs[4]).

Blocks 0 and 1 (the 8192 streams of BASELINE configs[3]) also carry

    pcm_loss30_md5 = md5( pcm[4096, 50, 640] )   decoded by a fresh reference decoder under bench.loss_mask(8192, 50): every description
                                                lost with probability 0.3 (numpy default_rng(4242)), packet 0 kept; mapping of the
                                                mask to (payload, nBytes, lostflag) as in the reference CLI (refcodec.map_loss)

    python tests/golden/make_bench_golden.py [n_blocks=16] [workers=8]      (this container only: needs oracle/_ref; ~10 min)
    python tests/golden/make_bench_golden.py loss [workers=8]               (adds pcm_loss30_md5 to blocks 0-1 of an existing file)
    python tests/golden/make_bench_golden.py extra [workers=8]              (adds to block 0: the hashes of the THIRD step -- the streams' state
                                                continued over two earlier steps of the same 50 input packets, what bench.py hashes after running
                                                steps 2 and 3 under its pipelined schedule -- and of the 32 kHz leg: the same samples taken as
                                                4096 x 25 packets of 1280 samples at 24 kbps, `samplerate = 32000`)
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
BLOCK, PACKETS, SLOT, RATE = 4096, 50, 512, 13600


LOSS_STREAMS = 8192
_MASK = None


def _mask():
    global _MASK
    if _MASK is None:
        import bench
        _MASK = bench.loss_mask(LOSS_STREAMS, PACKETS)
    return _MASK


def _stream(i):
    import refcodec as R
    from solo_amd.synth import synth_stream
    x = synth_stream(i, PACKETS)
    e, d = R.RefEncoder("fix", rate=RATE), R.RefDecoder("fix")
    dl = R.RefDecoder("fix") if i < LOSS_STREAMS else None
    nb = np.zeros((PACKETS, 2), np.int16)
    bits = np.zeros((PACKETS, SLOT), np.uint8)
    pcm = np.zeros((PACKETS, 640), np.int16)
    pcm_l = np.zeros((PACKETS, 640), np.int16) if dl else None
    for p in range(PACKETS):
        pl, n0, n1 = e.encode(x[p])
        nb[p] = (n0, n1)
        bits[p, :n0] = np.frombuffer(pl, np.uint8)
        y, ret = d.decode(*R.map_loss(pl, n0, n1, False, False))
        assert ret == 0
        pcm[p] = y
        if dl:
            m = int(_mask()[i, p])
            y, ret = dl.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
            assert ret == 0
            pcm_l[p] = y
    e.close(); d.close()
    if dl:
        dl.close()
    return nb, bits, pcm, pcm_l


def _stream_step3(i):
    import refcodec as R
    from solo_amd.synth import synth_stream
    x = synth_stream(i, PACKETS)
    e, d = R.RefEncoder("fix", rate=RATE), R.RefDecoder("fix")
    nb = np.zeros((PACKETS, 2), np.int16); bits = np.zeros((PACKETS, SLOT), np.uint8); pcm = np.zeros((PACKETS, 640), np.int16)
    for step in range(3):
        for p in range(PACKETS):
            pl, n0, n1 = e.encode(x[p])
            y, ret = d.decode(*R.map_loss(pl, n0, n1, False, False))
            assert ret == 0
            if step == 2:
                nb[p] = (n0, n1); bits[p, :n0] = np.frombuffer(pl, np.uint8); pcm[p] = y
    e.close(); d.close()
    return nb, bits, pcm


def _stream_wb(i):
    import refcodec as R
    from solo_amd.synth import synth_stream
    x = synth_stream(i, PACKETS).reshape(PACKETS // 2, 1280)
    e, d = R.RefEncoder("fix", rate=24000, samplerate=32000), R.RefDecoder("fix", samplerate=32000)
    nb = np.zeros((PACKETS // 2, 2), np.int16); bits = np.zeros((PACKETS // 2, SLOT), np.uint8); pcm = np.zeros((PACKETS // 2, 1280), np.int16)
    for p in range(PACKETS // 2):
        pl, n0, n1 = e.encode(x[p])
        y, ret = d.decode(*R.map_loss(pl, n0, n1, False, False))
        assert ret == 0
        nb[p] = (n0, n1); bits[p, :n0] = np.frombuffer(pl, np.uint8); pcm[p] = y
    e.close(); d.close()
    return nb, bits, pcm


def main():
    path = os.path.join(HERE, "bench_blocks.json")
    if len(sys.argv) > 1 and sys.argv[1] == "extra":
        workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        out = json.load(open(path))
        with mp.get_context("fork").Pool(workers) as pool:
            for fn, key in ((_stream_step3, "step3"), (_stream_wb, "wb")):
                rows = pool.map(fn, range(0, BLOCK), chunksize=16)
                nb = np.stack([r[0] for r in rows]); bits = np.stack([r[1] for r in rows]); pcm = np.stack([r[2] for r in rows])
                h = hashlib.md5(); h.update(nb.tobytes()); h.update(bits.tobytes())
                out["blocks"][0][key + "_payload_md5"] = h.hexdigest()
                out["blocks"][0][key + "_pcm_md5"] = hashlib.md5(pcm.tobytes()).hexdigest()
                print(key, "done", flush=True)
        json.dump(out, open(path, "w"), indent=1)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "loss":
        workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        out = json.load(open(path))
        _mask()
        with mp.get_context("fork").Pool(workers) as pool:
            for b in range(LOSS_STREAMS // BLOCK):
                rows = pool.map(_stream, range(b * BLOCK, (b + 1) * BLOCK), chunksize=16)
                nb = np.stack([r[0] for r in rows]); bits = np.stack([r[1] for r in rows])
                h = hashlib.md5(); h.update(nb.tobytes()); h.update(bits.tobytes())
                assert h.hexdigest() == out["blocks"][b]["payload_md5"]
                out["blocks"][b]["pcm_loss30_md5"] = hashlib.md5(np.stack([r[3] for r in rows]).tobytes()).hexdigest()
                print("block %d loss pass done" % b, flush=True)
        json.dump(out, open(path, "w"), indent=1)
        return
    n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    _mask()
    out = {"comment": "reference-generated (compiled fixed-point tree) hashes of the synthetic benchmark workload, see make_bench_golden.py",
           "block_streams": BLOCK, "packets": PACKETS, "slot_bytes": SLOT, "rate_bps": RATE, "blocks": []}
    t0 = time.time()
    with mp.get_context("fork").Pool(workers) as pool:
        for b in range(n_blocks):
            rows = pool.map(_stream, range(b * BLOCK, (b + 1) * BLOCK), chunksize=16)
            nb = np.stack([r[0] for r in rows]); bits = np.stack([r[1] for r in rows]); pcm = np.stack([r[2] for r in rows])
            h = hashlib.md5(); h.update(nb.tobytes()); h.update(bits.tobytes())
            out["blocks"].append({"block": b, "first_stream": b * BLOCK, "payload_md5": h.hexdigest(), "pcm_md5": hashlib.md5(pcm.tobytes()).hexdigest(),
                                  "payload_bytes": int(nb[:, :, 0].astype(np.int64).sum())})
            if rows[0][3] is not None:
                out["blocks"][-1]["pcm_loss30_md5"] = hashlib.md5(np.stack([r[3] for r in rows]).tobytes()).hexdigest()
            print("block %d done, %.0f s" % (b, time.time() - t0), flush=True)
            json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
