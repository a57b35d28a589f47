"""The N > 1 path of bench.py on CPU: two gloo ranks shard the stream index space,
each encodes AND decodes its own streams (host emulation of the kernel source stands in for the GPU), builds the per-rank record
{packets, seconds, payload_bytes, payload_md5, pcm_md5} with the same helper bench.py uses (solo_amd.dist.result_record) and ONE
all_gather collects the records.  Every rank's hashes must equal a single-process run over the same global stream indices.
No collective touches codec data."""
import multiprocessing as mp
import os
import socket

import numpy as np

import refcodec as R
import solo_testlib as T

PER_RANK, PACKETS, SLOT = 3, 4, 512


def _round_trip_streams(indices):
    """-> (nbytes [N,P,2], bits [N,P,slot], pcm [N,P,640]) of freshly reset streams"""
    from solo_amd.synth import synth_stream
    n = len(indices)
    nb = np.zeros((n, PACKETS, 2), np.int16)
    bits = np.zeros((n, PACKETS, SLOT), np.uint8)
    pcm = np.zeros((n, PACKETS, 640), np.int16)
    for k, i in enumerate(indices):
        e, d = T.EmuEncoder(), T.EmuDecoder()
        x = synth_stream(i, PACKETS)
        for p in range(PACKETS):
            pl, n0, n1 = e.encode(x[p])
            nb[k, p] = (n0, n1)
            bits[k, p, :n0] = np.frombuffer(pl, np.uint8)
            y, ret = d.decode(*R.map_loss(pl, n0, n1, False, False))
            assert ret == 0
            pcm[k, p] = y
    return nb, bits, pcm


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from solo_amd import dist as sdist
    dist = sdist.init("gloo")
    mine = sdist.stream_range(rank, PER_RANK)
    dist.barrier()
    nb, bits, pcm = _round_trip_streams(list(mine))
    dt = sdist.max_over_ranks(0.5 + rank, dist)
    rec = sdist.result_record(rank, mine[0], len(mine), len(mine) * PACKETS, 0.5 + rank, nb, bits, pcm)
    recs = sdist.gather_records(rec, dist)
    dist.barrier()
    if rank == 0:
        q.put((dt, recs))
    dist.destroy_process_group()


def test_two_ranks_shard_streams_and_gather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    dt, recs = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert dt == 1.5                                   # max over ranks of (0.5, 1.5)
    assert [r["rank"] for r in recs] == [0, 1] and [r["first_stream"] for r in recs] == [0, PER_RANK]
    from solo_amd import dist as sdist
    for r in recs:                                     # each shard equals a single-process run of the same global indices
        assert r["packets"] == PER_RANK * PACKETS and r["packets_per_s"] == round(r["packets"] / r["seconds"], 1)
        nb, bits, pcm = _round_trip_streams(list(range(r["first_stream"], r["first_stream"] + r["streams"])))
        want = sdist.result_record(0, r["first_stream"], r["streams"], r["packets"], r["seconds"], nb, bits, pcm)
        for k in ("payload_md5", "pcm_md5", "payload_bytes_per_step"):
            assert r[k] == want[k] and r[k] is not None, k
    assert recs[0]["payload_md5"] != recs[1]["payload_md5"]          # different streams, different hashes
