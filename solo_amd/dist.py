"""One process per GPU, streams sharded contiguously over ranks; no collective on the data path.

torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests) is used only for: a barrier around the
timed region, the max-over-ranks elapsed time, and one all_gather of per-rank result records."""
import os
import sys


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def stream_range(rank, streams_per_rank):
    """Global stream indices owned by `rank` (weak scaling: every rank owns the same count)."""
    return range(rank * streams_per_rank, (rank + 1) * streams_per_rank)


def init(backend, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    return dist


def max_over_ranks(seconds, dist, device="cpu"):
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(record, dist):
    """all_gather of one small picklable record per rank -> list ordered by rank."""
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, record)
    return out


def result_record(rank, first_stream, n_streams, packets, seconds, nbytes, bits, pcm):
    """The per-rank record of SURVEY 8(e) / BASELINE.md section 4 row 5: {packets, seconds, payload_bytes, hashes}.
    nbytes int16 [N,P,2], bits uint8 [N,P,slot] (bytes past a payload must be zero), pcm int16 [N,P,640] -- numpy arrays of ONE
    step from freshly reset streams, or None to skip the hashes.  The hashes depend only on the streams' global indices, so a rank
    of an N-GPU run and a single-GPU run over the same `first_stream .. first_stream + n_streams` must agree."""
    import hashlib
    rec = {"rank": int(rank), "first_stream": int(first_stream), "streams": int(n_streams), "packets": int(packets),
           "seconds": round(float(seconds), 6), "packets_per_s": round(packets / seconds, 1) if seconds > 0 else None,
           "payload_bytes": None, "payload_md5": None, "pcm_md5": None}
    if nbytes is not None:
        rec["payload_bytes_per_step"] = int(nbytes[:, :, 0].astype("int64").sum())
        h = hashlib.md5()
        h.update(nbytes.tobytes())
        h.update(bits.tobytes())
        rec["payload_md5"] = h.hexdigest()
    if pcm is not None:
        rec["pcm_md5"] = hashlib.md5(pcm.tobytes()).hexdigest()
    return rec


BLOCK_STREAMS = 4096


def block_hashes(nbytes, bits, pcm, block=BLOCK_STREAMS):
    """Hashes of ONE step from freshly reset streams, per aligned block of `block` streams (the unit tests/golden/bench_blocks.json
    holds reference-generated hashes for): [{"payload_md5": md5(nbytes || bits), "pcm_md5": md5(pcm)}, ...].  Arrays as in
    result_record; pcm may be None.  A stream count that is not a multiple of `block` gives None (nothing to compare with)."""
    import hashlib
    n = nbytes.shape[0]
    if n % block:
        return None
    out = []
    for b in range(n // block):
        sl = slice(b * block, (b + 1) * block)
        h = hashlib.md5()
        h.update(nbytes[sl].tobytes())
        h.update(bits[sl].tobytes())
        out.append({"payload_md5": h.hexdigest(), "pcm_md5": hashlib.md5(pcm[sl].tobytes()).hexdigest() if pcm is not None else None})
    return out


def torchrun_command(script, argv, nproc, port=None):
    """The command line that runs `script argv` as `nproc` ranks of one node (one process per GPU), the way the benchmark driver
    does it: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P script argv."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nproc)), "--master-addr", "127.0.0.1",
            "--master-port", str(port), script] + list(argv)


def self_launch(script, argv, nproc, exec_=True, env=None, timeout=None):
    """`python bench.py --gpus N` started WITHOUT a launcher: start the N ranks ourselves.  exec_=True replaces this process
    (bench.py); exec_=False runs the launcher as a child and returns its CompletedProcess (tests)."""
    cmd = torchrun_command(script, argv, nproc)
    e = dict(os.environ if env is None else env)
    e.setdefault("MASTER_ADDR", "127.0.0.1")
    e["SOLO_SELF_LAUNCHED"] = "1"
    if exec_:
        sys.stdout.flush()
        sys.stderr.flush()
        os.execve(sys.executable, cmd, e)
    import subprocess
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout)


def check_world(requested_gpus, world):
    """A run that was asked for N GPUs must BE N ranks: anything else would time a different job than the one it reports."""
    if int(world) != int(requested_gpus):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus %d` (self-launching) or under "
                         "`python -m torch.distributed.run --nnodes=1 --nproc-per-node %d ... bench.py --gpus %d`"
                         % (requested_gpus, world, requested_gpus, requested_gpus, requested_gpus))
