"""The N > 1 code path of bench.py ON HARDWARE: the driver's launch line
`python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P bench.py --gpus 1 ...`
with SOLO_FORCE_DIST=1, so that everything a multi-GPU run executes besides the codec -- process group over RCCL (backend nccl,
device-bound), the barriers around the timed region, all_reduce(MAX) of the elapsed time, ONE all_gather_object of the per-rank
record, destroy_process_group -- runs on the GPU box's one device with WORLD_SIZE = 1.  The record and the line must equal what the
plain single-process run of the same command reports (everything that is not a time), and both must equal the compiled
reference's hashes of block 0 (tests/golden/bench_blocks.json)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
ARGS = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra"]


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SOLO_FORCE_DIST")}
    e.update(kw)
    return e


def _line(r):
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
def test_rccl_collectives_of_the_multi_gpu_path_run_with_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH] + ARGS
    d = _line(subprocess.run(cmd, env=_env(SOLO_FORCE_DIST="1", MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=900, cwd=ROOT))
    p = _line(subprocess.run([sys.executable, BENCH] + ARGS, env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT))
    # the forced run went through the process group (one RCCL rank), the plain one did not
    assert d["rccl_ranks"] == 1 and p["rccl_ranks"] == 0
    assert d["n_gpus"] == 1 and p["n_gpus"] == 1
    for j in (d, p):
        assert j["parity_checked"] is True and j["parity"]["third_step_pipelined_checked"] is True
        assert len(j["ranks"]) == 1 and j["ranks"][0]["rank"] == 0 and j["ranks"][0]["first_stream"] == 0
    # the gathered record equals the local record of the plain run in everything that is not a time
    timeish = ("seconds", "packets_per_s", "shader_clock_mhz_under_vector_load")
    rd = {k: v for k, v in d["ranks"][0].items() if k not in timeish}
    rp = {k: v for k, v in p["ranks"][0].items() if k not in timeish}
    assert rd == rp, (rd, rp)
    assert rd["payload_md5"] and rd["pcm_md5"] and rd["packets"] == 4096 * 50 * 3
    # ... and the reference's hashes of block 0
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_blocks.json")))
    b0 = [b for b in gold["blocks"] if b["first_stream"] == 0][0]
    # (the printed line carries the first twelve hex digits of every md5: bench.py compact_line)
    assert len(rd["payload_md5"]) == 12 and rd["payload_md5"] == b0["payload_md5"][:12] and rd["pcm_md5"] == b0["pcm_md5"][:12]
    # max over ranks of ONE rank = that rank's time: value = packets / all_reduce(MAX)(seconds)
    assert abs(d["value"] - d["ranks"][0]["packets"] / d["ranks"][0]["seconds"]) / d["value"] < 1e-3
    assert d["config"]["workload"] == p["config"]["workload"] and d["metric"] == p["metric"]
