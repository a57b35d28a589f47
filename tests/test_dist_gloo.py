"""The N > 1 path of bench.py on CPU: two gloo ranks shard the stream index space, each encodes its own streams
(host emulation of the kernel source stands in for the GPU), and the gathered per-rank records must equal a
single-process run over the same global stream indices.  No collective touches codec data."""
import hashlib
import multiprocessing as mp
import os
import socket

import numpy as np

import solo_testlib as T

PER_RANK, PACKETS = 3, 4


def _encode_streams(indices):
    from solo_amd.synth import synth_stream
    h = hashlib.md5()
    nbytes = 0
    for i in indices:
        e = T.EmuEncoder()
        x = synth_stream(i, PACKETS)
        for p in range(PACKETS):
            pl, n0, n1 = e.encode(x[p])
            h.update(pl)
            nbytes += n0
    return h.hexdigest(), nbytes


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from solo_amd import dist as sdist
    dist = sdist.init("gloo")
    mine = sdist.stream_range(rank, PER_RANK)
    dist.barrier()
    digest, nbytes = _encode_streams(mine)
    dt = sdist.max_over_ranks(0.5 + rank, dist)
    recs = sdist.gather_records({"rank": rank, "first": mine[0], "n": len(mine), "md5": digest, "bytes": nbytes}, dist)
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
    assert [r["rank"] for r in recs] == [0, 1] and [r["first"] for r in recs] == [0, PER_RANK]
    for r in recs:                                     # each shard equals a single-process run of the same global indices
        assert (r["md5"], r["bytes"]) == _encode_streams(range(r["first"], r["first"] + r["n"]))
