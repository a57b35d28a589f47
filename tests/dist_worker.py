"""One rank of the CPU stand-in for `bench.py --gpus N` (tests/test_dist_gloo.py starts N of these through the SAME launcher path
bench.py uses, solo_amd.dist.self_launch -> python -m torch.distributed.run): backend gloo, the host emulation of the kernel source
stands in for the GPU.  Mirrors bench.py's distributed skeleton step for step: env_world, check_world(--gpus), init, barrier, each
rank encodes AND decodes its own contiguous shard of the global stream index space, max-over-ranks time, ONE all_gather of the
per-rank record (solo_amd.dist.result_record), rank 0 prints one JSON line with n_gpus = WORLD_SIZE.  Test infrastructure only."""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

PER_RANK, PACKETS, SLOT = 3, 4, 512


def round_trip_streams(indices):
    """-> (nbytes [N,P,2], bits [N,P,slot], pcm [N,P,640]) of freshly reset streams (host emulation)"""
    import refcodec as R
    import solo_testlib as T
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    args = ap.parse_args()
    from solo_amd import dist as sdist
    world, rank, local_rank = sdist.env_world()
    sdist.check_world(args.gpus, world)                      # the same refusal bench.py makes
    dist = sdist.init("gloo")
    mine = sdist.stream_range(rank, PER_RANK)
    dist.barrier()
    t0 = time.perf_counter()
    nb, bits, pcm = round_trip_streams(list(mine))
    dist.barrier()
    dt_local = time.perf_counter() - t0
    dt = sdist.max_over_ranks(dt_local, dist)
    rec = sdist.result_record(rank, mine[0], len(mine), len(mine) * PACKETS, dt_local, nb, bits, pcm)
    recs = sdist.gather_records(rec, dist)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "value": world * PER_RANK * PACKETS / dt, "seconds_max": dt, "ranks": recs,
                          "self_launched": bool(os.environ.get("SOLO_SELF_LAUNCHED"))}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
