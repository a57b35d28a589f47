"""Receiver staging ring (solo_recv_*, solo_amd/csrc/solo_recv.h): the reference README's "cache queue" (README.md:52-58) with a time
axis -- descriptions arrive separately, in any order, possibly after their partner, possibly too late.

CPU: the bookkeeping functions the insert kernel is made of (window check, slot claim), compiled for the host, against an independent
model in this file.  GPU: a play-out simulation through the C ABI -- every description of every packet gets a network delay (or is
lost), arrivals are filed tick by tick, one packet per stream is decoded per tick -- whose PCM must equal the COMPILED REFERENCE decoder
called with the (ptr, nBytes, lostflag) of test/dec_main.c:255-378 for exactly the descriptions that had arrived by the packet's turn."""
import ctypes as C

import numpy as np
import pytest

import refcodec as R
import solo_testlib as T

need_ref = pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not present")
INSERTED, LATE, AHEAD, DUP, BAD = range(5)


def _model_file(arr, n_streams, depth, slot, payload_bytes, use_md_index, true_desc, play, lens):
    """Independent restatement of the filing rules (lens: dict entry -> [lenA, lenB]; true_desc: offset -> description index)."""
    verdict, slots = [], []
    for s, q, d, off, ln in arr:
        if not (0 <= s < n_streams) or d not in (-1, 0, 1) or ln <= 0 or ln > slot or off < 0 or off + ln > payload_bytes or q < 0:
            verdict.append(BAD); slots.append(-1); continue
        if q < play[s]:
            verdict.append(LATE); slots.append(-1); continue
        if q >= play[s] + depth:
            verdict.append(AHEAD); slots.append(-1); continue
        if d < 0:
            if not use_md_index:
                verdict.append(BAD); slots.append(-1); continue
            d = true_desc[off]
        e = lens.setdefault((s, q % depth), [0, 0])
        if e[d]:
            verdict.append(DUP); slots.append(-1); continue
        e[d] = ln
        verdict.append(INSERTED); slots.append(d)
    return verdict, slots


@need_ref
@pytest.mark.parametrize("mdi", [0, 1])
def test_emu_filing_rules_against_model(mdi):
    lib = T.load_emu()
    lib.emu_recv_file.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5 + mdi)
    # a pool of real descriptions (with useMDIndex = 1 each carries its index as its first coded symbol)
    enc = R.RefEncoder("fix", rate=13600, use_md_index=mdi)
    pcm = R.synth_stream(11, 12)
    pool, true_desc, blob = [], {}, b""
    for p in range(12):
        pl, n0, n1 = enc.encode(pcm[p])
        for d, part in enumerate((pl[:n0 - n1], pl[n0 - n1:n0])):
            pool.append((len(blob), len(part), d)); true_desc[len(blob)] = d
            blob += part
    payload = np.frombuffer(blob, np.uint8).copy()
    N, D, SLOT, PB = 7, 5, 200, payload.size
    play = rng.integers(0, 50, N).astype(np.int32)
    lens = np.zeros((N, D), np.uint32)
    model_lens = {}
    seen = set()
    for call in range(6):
        n = 400
        pick = rng.integers(0, len(pool), n)
        arr = np.stack([rng.integers(-1, N + 1, n), np.zeros(n, np.int64), np.zeros(n, np.int64), [pool[k][0] for k in pick],
                        [pool[k][1] for k in pick]], axis=1).astype(np.int32)
        # the transport's description index: the true one, or "unknown"; now and then a nonsense value / length / offset
        arr[:, 2] = np.where(rng.random(n) < 0.5, -1, [pool[k][2] for k in pick])
        arr[::23, 2] = 2
        arr[5::29, 4] = SLOT + 1
        arr[7::31, 3] = PB - 3
        ok = (arr[:, 0] >= 0) & (arr[:, 0] < N)
        arr[:, 1] = np.where(ok, play[np.clip(arr[:, 0], 0, N - 1)] + rng.integers(-3, D + 3, n), rng.integers(-2, 60, n))
        arr[::17, 1] = -1
        verdict = np.zeros(n, np.int32); slots = np.zeros(n, np.int32)
        lib.emu_recv_file(arr.ctypes.data, n, N, D, SLOT, payload.ctypes.data, PB, mdi, play.ctypes.data, lens.ctypes.data, verdict.ctypes.data,
                          slots.ctypes.data)
        mv, ms = _model_file(arr.tolist(), N, D, SLOT, PB, mdi, true_desc, play.tolist(), model_lens)
        assert verdict.tolist() == mv and slots.tolist() == ms
        seen |= set(mv)
        for (s, e), (la, lb) in model_lens.items():
            assert int(lens[s, e]) == la | (lb << 16)
    assert seen == {INSERTED, LATE, AHEAD, DUP, BAD}
    assert int((lens != 0).sum()) == sum(1 for v in model_lens.values() if v[0] or v[1])


@need_ref
def test_emu_unknown_description_with_damaged_payloads():
    """desc = -1 on payloads that are not descriptions (random bytes, bit errors, one-byte payloads): the index is whatever the range
    decoder makes of the first symbol -- 0 or 1 files the arrival in that slot, a coder error counts it as bad; nothing else may
    happen (no other verdict, no length word touched for a bad one), and an undamaged description is still recognised afterwards."""
    lib = T.load_emu()
    lib.emu_recv_file.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(99)
    enc = R.RefEncoder("fix", rate=13600, use_md_index=1)
    pl, n0, n1 = enc.encode(R.synth_stream(3, 1)[0])
    good = [np.frombuffer(pl[:n0 - n1], np.uint8), np.frombuffer(pl[n0 - n1:n0], np.uint8)]
    N, D, SLOT = 600, 1, 200
    blobs, arr, off = [], [], 0
    for i in range(N - 2):
        kind = i % 3
        if kind == 0:
            x = rng.integers(0, 256, int(rng.integers(1, 120)), dtype=np.uint8)
        elif kind == 1:
            x = good[i & 1].copy(); x[int(rng.integers(0, 3))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        else:
            x = good[i & 1][:int(rng.integers(1, 4))].copy()
        blobs.append(x); arr.append((i, 0, -1, off, x.size)); off += x.size
    for d in (0, 1):
        blobs.append(good[d]); arr.append((N - 2 + d, 0, -1, off, good[d].size)); off += good[d].size
    payload = np.concatenate(blobs)
    arr = np.array(arr, np.int32)
    play = np.zeros(N, np.int32); lens = np.zeros((N, D), np.uint32)
    verdict = np.zeros(N, np.int32); slots = np.zeros(N, np.int32)
    lib.emu_recv_file(arr.ctypes.data, N, N, D, SLOT, payload.ctypes.data, payload.size, 1, play.ctypes.data, lens.ctypes.data, verdict.ctypes.data,
                      slots.ctypes.data)
    assert set(verdict.tolist()) <= {INSERTED, BAD}
    for i in range(N):
        if verdict[i] == INSERTED:
            assert slots[i] in (0, 1) and int(lens[i, 0]) == int(arr[i, 4]) << (16 * int(slots[i]))
        else:
            assert slots[i] == -1 and lens[i, 0] == 0
    assert verdict[N - 2] == INSERTED and slots[N - 2] == 0 and verdict[N - 1] == INSERTED and slots[N - 1] == 1


# ---------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch


def _play_out(torch, mdi, samplerate=16000):
    import solo_amd
    N, P, D, BUDGET, SLOT = 24, 14, 4, 2, 256            # BUDGET: ticks between a packet's send time and its play-out turn
    if samplerate == 16000:
        pcm = np.stack([R.synth_stream(500 + i, P) for i in range(N)])
        rate = 13600
    else:
        pcm = np.stack([T.synth_stream_32k(500 + i, P) for i in range(N)])
        rate = 24000
    ref = []
    for i in range(N):
        e = R.RefEncoder("fix", rate=rate, use_md_index=mdi, samplerate=samplerate)
        ref.append([e.encode(pcm[i, p]) for p in range(P)])
    rng = np.random.default_rng(77 + mdi)
    # network: description d of packet (i, p) is sent at tick p and arrives at tick p + delay, or never; delays beyond the budget are
    # late; some descriptions are delivered twice; with useMDIndex the transport does not say which description a packet carries
    delay = rng.integers(0, BUDGET + 3, (N, P, 2))
    lost = rng.random((N, P, 2)) < 0.15
    twice = rng.random((N, P, 2)) < 0.1
    ticks = [[] for _ in range(P + BUDGET + 8)]
    blobs, off = [], 0
    for i in range(N):
        for p, (pl, n0, n1) in enumerate(ref[i]):
            parts = (pl[:n0 - n1], pl[n0 - n1:n0])
            for d in (0, 1):
                if lost[i, p, d]:
                    continue
                blobs.append(parts[d])
                for k in range(2 if twice[i, p, d] else 1):
                    ticks[p + int(delay[i, p, d]) + k].append((i, p, -1 if mdi else d, off, len(parts[d])))
                off += len(parts[d])
    payload = torch.from_numpy(np.frombuffer(b"".join(blobs), np.uint8).copy()).cuda()
    b = solo_amd.SoloBatch(N, rate=rate, encoder=False, decoder=True, slot_bytes=512, use_md_index=mdi, samplerate=samplerate)
    b.recv_create(D, SLOT, 0)
    got = np.zeros((N, P, b.packet_samples), np.int16)
    for t in range(len(ticks)):
        if ticks[t]:
            order = rng.permutation(len(ticks[t]))
            b.recv_insert(torch.from_numpy(np.array(ticks[t], np.int32)[order].copy()).cuda(), payload)
        if BUDGET <= t < P + BUDGET:
            x, st = b.recv_decode(1)
            torch.cuda.synchronize()
            assert int(st.abs().max()) == 0
            got[:, t - BUDGET] = x.cpu().numpy()[:, 0]
    stats = b.recv_stats()
    # expectation: description (i, p, d) counts iff it was not lost and arrived by tick p + BUDGET (the insert of a tick precedes its decode)
    have = (~lost) & (delay <= BUDGET)
    n_late = int(((~lost) & (delay > BUDGET)).sum() + ((~lost) & twice & (delay + 1 > BUDGET)).sum())
    n_dup = int(((~lost) & twice & (delay + 1 <= BUDGET)).sum())
    assert stats["inserted"] == int(have.sum()) and stats["late"] == n_late and stats["duplicate"] == n_dup, stats
    assert stats["ahead"] == 0 and stats["bad"] == 0 and n_late > 0 and n_dup > 0
    states = set()
    for i in range(N):
        dr = R.RefDecoder("fix", use_md_index=mdi, samplerate=samplerate)
        for p, (pl, n0, n1) in enumerate(ref[i]):
            x, ret = dr.decode(*R.map_loss(pl, n0, n1, not have[i, p, 0], not have[i, p, 1]))
            assert ret == 0 and np.array_equal(got[i, p], x), (i, p, have[i, p].tolist())
            states.add((bool(have[i, p, 0]), bool(have[i, p, 1])))
    assert len(states) == 4
    # the queue is empty again, and an arrival for a played packet is late
    b.recv_insert(torch.tensor([[0, 0, 0, 0, 10], [1, P + D, 0, 0, 10], [2, P, 1, 0, 10], [N, P, 0, 0, 10]], dtype=torch.int32).cuda(), payload)
    s2 = b.recv_stats()
    assert (s2["late"], s2["ahead"], s2["inserted"], s2["bad"]) == (n_late + 1, 1, stats["inserted"] + 1, 1)


@pytest.mark.gpu
@need_ref
@pytest.mark.parametrize("mdi", [0, 1])
def test_gpu_play_out_vs_compiled_reference(torch_cuda, mdi):
    _play_out(torch_cuda, mdi)


@pytest.mark.gpu
@need_ref
def test_gpu_play_out_32k_vs_compiled_reference(torch_cuda):
    _play_out(torch_cuda, 1, samplerate=32000)
