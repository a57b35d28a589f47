#!/usr/bin/env python3
"""The collectives bench.py uses for N > 1 (barrier, all_reduce MAX, all_gather_object over RCCL), next to the codec's own HIP streams
and under the same runtime environment (GPU_MAX_HW_QUEUES=2): a one-rank check that the process does not stall.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 tools/debug/rccl_smoke.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import solo_amd
from solo_amd import dist as sdist
from solo_amd.synth import synth_batch

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
dist = sdist.init("nccl", dev)
N, P = 4096, 6
pcm = torch.from_numpy(synth_batch(0, N, P, workers=4)).to(dev)
batch = solo_amd.SoloBatch(N, rate=13600, encoder=True, decoder=True, slot_bytes=512)
bits = torch.zeros((N, P, 512), dtype=torch.uint8, device=dev)
nb = torch.zeros((N, P, 2), dtype=torch.int16, device=dev)
st = torch.zeros((N,), dtype=torch.int32, device=dev)
out = torch.zeros((N, P, 640), dtype=torch.int16, device=dev)
t0 = time.perf_counter()
for k in range(4):
    dist.barrier(); torch.cuda.synchronize()
    batch.encode(pcm, bits, nb, st)
    batch.decode(bits, nb, None, out, st)
    torch.cuda.synchronize()
    t = sdist.max_over_ranks(time.perf_counter() - t0, dist, dev)
    recs = sdist.gather_records({"rank": dist.get_rank(), "t": t}, dist)
print("RCCL SMOKE OK", recs, "payload", float(nb[:, :, 0].float().mean()))
dist.destroy_process_group()
