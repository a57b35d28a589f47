"""The N > 1 path of bench.py on CPU.

* `python bench.py --gpus 2` on a box without two GPUs must FAIL LOUDLY (it used to time one GPU silently), and a launched rank whose
  WORLD_SIZE differs from --gpus must refuse as well.
* The launcher bench.py uses for a plain `--gpus N` (solo_amd.dist.self_launch -> torch.distributed.run, one rank per device) starts two
  gloo ranks of tests/dist_worker.py -- bench.py's distributed skeleton with the host emulation of the kernel source standing in for
  the GPU: the ranks shard the stream index space, each encodes AND decodes its own streams, builds the per-rank record with the same
  helper (solo_amd.dist.result_record), ONE all_gather collects the records, rank 0 prints n_gpus = WORLD_SIZE.  Every rank's hashes
  must equal a single-process run over the same global stream indices.  No collective touches codec data."""
import json
import os
import subprocess
import sys

import numpy as np

import solo_testlib as T

ROOT = T.ROOT if hasattr(T, "ROOT") else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(kw)
    return e


def test_bench_refuses_to_misreport_n_gpus():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two devices present: the refusal is for boxes with fewer")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "HIP device(s) visible" in (r.stderr + r.stdout), (r.returncode, r.stderr[-400:])
    assert '"n_gpus"' not in r.stdout                          # no result line of any kind
    # a rank started by a launcher with another world size than --gpus refuses before touching the GPU
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in r.stderr, r.stderr[-400:]
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr, r.stderr[-400:]


def test_two_ranks_through_the_launcher_shard_streams_and_gather():
    from solo_amd import dist as sdist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker as W
    r = sdist.self_launch(WORKER, ["--gpus", "2"], 2, exec_=False, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    recs = j["ranks"]
    assert j["n_gpus"] == 2 and j["self_launched"] is True
    assert [x["rank"] for x in recs] == [0, 1] and [x["first_stream"] for x in recs] == [0, W.PER_RANK]
    assert abs(j["seconds_max"] - max(x["seconds"] for x in recs)) < 1e-3            # max over ranks
    for x in recs:                                     # each shard equals a single-process run of the same global indices
        assert x["packets"] == W.PER_RANK * W.PACKETS and abs(x["packets_per_s"] - x["packets"] / x["seconds"]) < 0.11
        nb, bits, pcm = W.round_trip_streams(list(range(x["first_stream"], x["first_stream"] + x["streams"])))
        want = sdist.result_record(0, x["first_stream"], x["streams"], x["packets"], x["seconds"], nb, bits, pcm)
        for k in ("payload_md5", "pcm_md5", "payload_bytes_per_step"):
            assert x[k] == want[k] and x[k] is not None, k
    assert recs[0]["payload_md5"] != recs[1]["payload_md5"]          # different streams, different hashes
    # a worker whose --gpus disagrees with the world the launcher made refuses (every rank exits non-zero)
    r = sdist.self_launch(WORKER, ["--gpus", "3"], 2, exec_=False, env=_env(), timeout=600)
    assert r.returncode != 0 and "--gpus 3 but WORLD_SIZE=2" in r.stderr


def test_block_hashes_and_golden_file():
    """solo_amd.dist.block_hashes is the unit bench.py compares with tests/golden/bench_blocks.json (reference-generated)."""
    from solo_amd import dist as sdist
    rng = np.random.default_rng(1)
    nb = rng.integers(0, 100, (8, 2, 2)).astype(np.int16); bits = rng.integers(0, 255, (8, 2, 16)).astype(np.uint8); pcm = rng.integers(-5, 5, (8, 2, 640)).astype(np.int16)
    h = sdist.block_hashes(nb, bits, pcm, block=4)
    assert len(h) == 2 and h[0] != h[1] and sdist.block_hashes(nb[:6], bits[:6], pcm[:6], block=4) is None
    assert h[1] == sdist.block_hashes(nb[4:], bits[4:], pcm[4:], block=4)[0]
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_blocks.json")))
    assert g["block_streams"] == 4096 and g["packets"] == 50 and len(g["blocks"]) >= 2
    assert all(len(b["payload_md5"]) == 32 and len(b["pcm_md5"]) == 32 for b in g["blocks"]) and "pcm_loss30_md5" in g["blocks"][0]
    import bench
    m = bench.loss_mask(8192, 50)
    assert m.shape == (8192, 50) and (m[:, 0] == 3).all() and 0.25 < float(((m[:, 1:] & 1) == 0).mean()) < 0.35
