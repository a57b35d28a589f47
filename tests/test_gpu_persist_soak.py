"""Soak test of the PERSISTENT encoder schedule (SOLO_ENC_PERSIST=1: solo_enc_front_kernel + solo_nsq_persist_kernel hand packets to
each other through per-stream flags in HBM while both run -- a lock-free protocol across the XCDs' L2s) against the launch-per-chunk
schedule, whose batches are reference-hashed elsewhere (test_gpu_fullsize.py, bench.py): more than a million packets on seeds derived
from the kernel sources, four state-continued calls per handle, every payload byte, length and status equal; then once more with the
calls issued back to back under asynchronous joins (the next call's kernels are enqueued while the previous call's still run)."""
import hashlib
import os

import numpy as np
import pytest

import solo_testlib as T


def _speechish(torch, seed, n, p, samples):
    """Band-limited noise bursts with per-stream level and a pitch-like comb: enough to visit voiced / unvoiced frames, gain changes and
    silence; generated on the device (a million packets of host-side synthesis would take minutes)."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    L = p * samples
    x = torch.randn((n, L), generator=g, device="cuda")
    x = torch.nn.functional.avg_pool1d(x[:, None, :], 5, 1, 2)[:, 0, :]                      # low-pass
    lag = torch.randint(40, 200, (n, 1), generator=g, device="cuda")
    idx = (torch.arange(L, device="cuda")[None, :] - lag).clamp_(min=0)
    x = x + 0.8 * torch.gather(x, 1, idx)                                                     # comb = a pitch
    env = torch.rand((n, L // 800 + 1), generator=g, device="cuda")
    env = torch.where(env < 0.25, torch.zeros_like(env), env)                                 # a quarter of the 50 ms blocks silent
    env = torch.repeat_interleave(env, 800, dim=1)[:, :L]
    lvl = 10 ** (torch.rand((n, 1), generator=g, device="cuda") * 2.0 + 1.5)                  # 30 ... 3000
    return (x * env * lvl).clamp_(-32768, 32767).to(torch.int16).reshape(n, p, samples).contiguous()


@pytest.mark.gpu
@pytest.mark.parametrize("n_streams,calls,packets", [(4096, 4, 64), (1000, 3, 7)], ids=["4096x4x64", "ragged-1000x3x7"])
def test_persistent_schedule_equals_launch_per_chunk(n_streams, calls, packets):
    import torch
    import solo_amd
    seed = int(hashlib.sha256((solo_amd.kernel_source_hash() + "soak").encode()).hexdigest()[:8], 16)
    print("soak seed %d" % seed)
    outs = {}
    old = os.environ.get("SOLO_ENC_PERSIST")
    try:
        for persist, async_join in ((0, False), (1, False), (1, True)):
            os.environ["SOLO_ENC_PERSIST"] = str(persist)          # (read when the handle's first encode call sets its pipeline up)
            b = solo_amd.SoloBatch(n_streams, encoder=True, decoder=False, slot_bytes=512)
            b.set_async_join(async_join)
            res, keep = [], []
            for c in range(calls):
                x = _speechish(torch, seed + c, n_streams, packets, 640)
                keep.append(x)                                  # (asynchronous joins: the kernels may still read it when the loop moves on)
                bits, nb, st = b.encode(x)
                res.append((bits, nb, st))
            if async_join:
                b.wait_encode(0)
                b.wait_encode(1)
            torch.cuda.synchronize()
            outs[(persist, async_join)] = res
            b.close()
    finally:
        if old is None:
            os.environ.pop("SOLO_ENC_PERSIST", None)
        else:
            os.environ["SOLO_ENC_PERSIST"] = old
    ref = outs[(0, False)]
    total = 0
    for key in ((1, False), (1, True)):
        for c in range(calls):
            for a, r, what in zip(outs[key][c], ref[c], ("payload slots", "lengths", "status")):
                if not torch.equal(a, r):
                    bad = (a != r).reshape(a.shape[0], -1).any(dim=1).nonzero().flatten()
                    raise AssertionError("persistent schedule (persist, async) = %s differs in %s of call %d: %d streams, first %s" % (key, what, c, bad.numel(), bad[:8].tolist()))
    for c in range(calls):
        assert int(ref[c][2].abs().max()) == 0
        total += n_streams * packets
        assert int(ref[c][1][:, :, 0].min()) > 0                     # every packet was coded
    print("%d packets per schedule equal" % total)
    assert total >= 1_000_000 or n_streams < 4096
