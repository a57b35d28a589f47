"""BASELINE.json's full-size configurations on the GPU (4096 streams encode / round trip, 8192 streams decode-only with
30 % description loss), checked through properties that do not need the whole batch on the CPU: a sample of the streams
goes through the compiled reference (when oracle/_ref travelled with the snapshot), every stream is checked against its
duplicate elsewhere in the batch (stream independence: no cross-talk between wavefronts / lane groups), and launches are
repeatable."""
import multiprocessing as mp

import numpy as np
import pytest

import refcodec as R
import solo_testlib as T

pytestmark = pytest.mark.gpu

DISTINCT = 512           # distinct synthetic streams; the batch repeats them


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch


def _ref_encode_stream(args):
    seed, P = args
    pcm = R.synth_stream(seed, P)
    e = R.RefEncoder("fix")
    out = [e.encode(pcm[p]) for p in range(P)]
    e.close()
    return out


def _ref_decode_stream(args):
    recs, recv = args
    d = R.RefDecoder("fix")
    out = []
    for (pl, n0, n1), m in zip(recs, recv):
        pcm, ret = d.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
        assert ret == 0
        out.append(pcm)
    d.close()
    return np.stack(out)


def _pool_map(fn, items):
    with mp.get_context("fork").Pool(min(32, mp.cpu_count())) as pool:
        return pool.map(fn, items, chunksize=1)


def _batch_pcm(N, P):
    from solo_amd.synth import synth_batch
    base = synth_batch(7000, DISTINCT, P, workers=16)            # streams 7000 .. 7000 + DISTINCT - 1
    return np.ascontiguousarray(np.tile(base, (N // DISTINCT, 1, 1)))


def test_4096_streams_encode_and_round_trip(torch_cuda):
    """configs[1] and configs[2]: 4096 streams, encode (bit-exact) and decode of both descriptions."""
    import solo_amd
    torch = torch_cuda
    N, P = 4096, 4
    pcm = _batch_pcm(N, P)
    b = solo_amd.SoloBatch(N, encoder=True, decoder=True, slot_bytes=512)
    x = torch.from_numpy(pcm).to(b.device)
    bits, nb, st = b.encode(x)
    out, st2 = b.decode(bits, nb, None)
    torch.cuda.synchronize()
    assert int(st.abs().max()) == 0 and int(st2.abs().max()) == 0
    hb, hn, ho = bits.cpu().numpy(), nb.cpu().numpy(), out.cpu().numpy()
    # every copy of a stream produces the same payload bytes and the same PCM
    for r in range(1, N // DISTINCT):
        sl = slice(r * DISTINCT, (r + 1) * DISTINCT)
        assert np.array_equal(hn[sl], hn[:DISTINCT])
        assert np.array_equal(ho[sl], ho[:DISTINCT])
    n0 = hn[:, :, 0].astype(np.int64)
    mask = np.arange(hb.shape[2])[None, None, :] < n0[:, :, None]
    pay = np.where(mask, hb, 0)
    for r in range(1, N // DISTINCT):
        assert np.array_equal(pay[r * DISTINCT:(r + 1) * DISTINCT], pay[:DISTINCT])
    # a second batch object repeats the launch bit for bit
    b2 = solo_amd.SoloBatch(N, encoder=True, decoder=True, slot_bytes=512)
    bits2, nb2, _ = b2.encode(x)
    torch.cuda.synchronize()
    assert np.array_equal(nb2.cpu().numpy(), hn)
    assert np.array_equal(np.where(mask, bits2.cpu().numpy(), 0), pay)
    if R.have_ref("fix"):
        idx = list(range(0, DISTINCT, 4))                           # 128 streams through the compiled reference
        ref = _pool_map(_ref_encode_stream, [(7000 + i, P) for i in idx])
        for i, recs in zip(idx, ref):
            for p, (pl, r0, r1) in enumerate(recs):
                assert hn[i, p, 0] == r0 and hn[i, p, 1] == r1
                assert hb[i, p, :r0].tobytes() == pl
        dec = _pool_map(_ref_decode_stream, [(recs, [3] * P) for recs in ref])
        for i, d in zip(idx, dec):
            assert np.array_equal(ho[i], d)


def test_8192_streams_decode_only_with_30pct_loss(torch_cuda):
    """configs[3]: 8192 streams, decode only, every description dropped with probability 0.3 (single-description decoding and
    concealment), vs the compiled reference on a sample and vs duplicates everywhere."""
    import solo_amd
    torch = torch_cuda
    N, P = 8192, 6
    pcm = _batch_pcm(DISTINCT, P)
    enc = solo_amd.SoloBatch(DISTINCT, encoder=True, decoder=False, slot_bytes=512)
    bits, nb, st = enc.encode(torch.from_numpy(pcm).to(enc.device))
    torch.cuda.synchronize()
    assert int(st.abs().max()) == 0
    reps = N // DISTINCT
    bits_all = bits.repeat(reps, 1, 1).contiguous()
    nb_all = nb.repeat(reps, 1, 1).contiguous()
    recv = T.bernoulli_recv(DISTINCT, P, 0.3, 4242)
    recv[:, 0] = 3                                                # first packet of every stream received
    recv_all = np.ascontiguousarray(np.tile(recv, (reps, 1)))
    dec = solo_amd.SoloBatch(N, encoder=False, decoder=True, slot_bytes=512)
    out, st2 = dec.decode(bits_all, nb_all, torch.from_numpy(recv_all).to(dec.device))
    torch.cuda.synchronize()
    assert int(st2.abs().max()) == 0
    ho = out.cpu().numpy()
    for r in range(1, reps):
        assert np.array_equal(ho[r * DISTINCT:(r + 1) * DISTINCT], ho[:DISTINCT])
    assert len({int(m) for m in recv.ravel()}) == 4               # all four loss states occur
    if R.have_ref("fix"):
        hb, hn = bits.cpu().numpy(), nb.cpu().numpy()
        idx = list(range(0, DISTINCT, 2))                           # 256 streams through the compiled reference decoder
        jobs = []
        for i in idx:
            recs = [(hb[i, p, :hn[i, p, 0]].tobytes(), int(hn[i, p, 0]), int(hn[i, p, 1])) for p in range(P)]
            jobs.append((recs, [int(m) for m in recv[i]]))
        ref = _pool_map(_ref_decode_stream, jobs)
        for i, d in zip(idx, ref):
            assert np.array_equal(ho[i], d), "stream %d" % i


def test_8192_streams_x_50_packets_encode_and_decode_config5_share(torch_cuda):
    """configs[4]: the per-GPU share of the 8-GPU configuration -- 8192 streams x 50 packets (2 s of audio per stream: voiced /
    unvoiced / pauses, long-run state), encode + decode on one GPU.  A sample of the streams goes through the compiled reference
    (bitstreams, clean decode, and a second decode pass with 10 % packet loss so that concealment and comfort noise run long);
    every stream equals its duplicates; the per-rank record of bench.py (solo_amd.dist.result_record) is repeatable."""
    import solo_amd
    from solo_amd import dist as sdist
    torch = torch_cuda
    N, P = 8192, 50
    pcm = _batch_pcm(N, P)
    b = solo_amd.SoloBatch(N, encoder=True, decoder=True, slot_bytes=512)
    x = torch.from_numpy(pcm).to(b.device)
    bits, nb, st = b.encode(x)
    out, st2 = b.decode(bits, nb, None)
    torch.cuda.synchronize()
    assert int(st.abs().max()) == 0 and int(st2.abs().max()) == 0
    hb, hn, ho = bits.cpu().numpy(), nb.cpu().numpy(), out.cpu().numpy()
    rec = sdist.result_record(0, 7000, N, N * P, 1.0, hn, hb, ho)
    for r in range(1, N // DISTINCT):
        sl = slice(r * DISTINCT, (r + 1) * DISTINCT)
        assert np.array_equal(hn[sl], hn[:DISTINCT]) and np.array_equal(hb[sl], hb[:DISTINCT]) and np.array_equal(ho[sl], ho[:DISTINCT])
    # lossy second pass (10 % of the packets lost entirely, 10 % reduced to one description)
    rng = np.random.default_rng(77)
    recv = np.full((DISTINCT, P), 3, np.uint8)
    u = rng.random((DISTINCT, P))
    recv[u < 0.10] = 0
    recv[(u >= 0.10) & (u < 0.15)] = 1
    recv[(u >= 0.15) & (u < 0.20)] = 2
    recv[:, 0] = 3
    recv_all = np.ascontiguousarray(np.tile(recv, (N // DISTINCT, 1)))
    b.reset()
    bits_b, nb_b, _ = b.encode(x)                                  # second run of the same handle after a reset: same record
    out_l, st3 = b.decode(bits_b, nb_b, torch.from_numpy(recv_all).to(b.device))
    torch.cuda.synchronize()
    assert int(st3.abs().max()) == 0
    hol = out_l.cpu().numpy()
    rec2 = sdist.result_record(0, 7000, N, N * P, 1.0, nb_b.cpu().numpy(), bits_b.cpu().numpy(), None)
    assert rec2["payload_md5"] == rec["payload_md5"] and rec["payload_bytes_per_step"] == int(hn[:, :, 0].astype(np.int64).sum())
    for r in range(1, N // DISTINCT):
        assert np.array_equal(hol[r * DISTINCT:(r + 1) * DISTINCT], hol[:DISTINCT])
    if R.have_ref("fix"):
        idx = list(range(0, DISTINCT, 8))                           # 64 streams x 50 packets through the compiled reference
        ref = _pool_map(_ref_encode_stream, [(7000 + i, P) for i in idx])
        for i, recs in zip(idx, ref):
            for p, (pl, r0, r1) in enumerate(recs):
                assert hn[i, p, 0] == r0 and hn[i, p, 1] == r1, (i, p)
                assert hb[i, p, :r0].tobytes() == pl, (i, p)
        dec = _pool_map(_ref_decode_stream, [(recs, [3] * P) for recs in ref])
        for i, d in zip(idx, dec):
            assert np.array_equal(ho[i], d), i
        dec = _pool_map(_ref_decode_stream, [(recs, [int(m) for m in recv[i]]) for i, recs in zip(idx, ref)])
        for i, d in zip(idx, dec):
            assert np.array_equal(hol[i], d), i
