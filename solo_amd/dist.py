"""One process per GPU, streams sharded contiguously over ranks; no collective on the data path.

torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests) is used only for: a barrier around the
timed region, the max-over-ranks elapsed time, and one all_gather of per-rank result records."""
import os


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
