"""Corners that round 1 only compared with the host emulation of the same source, now pinned to the COMPILED REFERENCE:

  * useMDIndex = 1 (SKP_Silk_encode_parameters.c:50-51, SKP_Silk_decode_parameters.c:55-57): encoder bytes, decoder with
    description loss, and the receiver front end (solo_batch_decode_split) with missing / swapped / duplicated arrival slots
    against RefDecoder fed the correctly ordered call;
  * the second narrow-band rate (24 kbps at 16 kHz), encoder and decoder;
  * corrupted payloads on the GPU (the CPU twin is tests/test_emu_decoder.py::test_corrupted_payloads_vs_reference);
  * what AGR_Sate_Decoder_Decode leaves in the caller's nBytes[] / *nSamplesOut (AGR_BWE_decode_frame_FIX.c:150-169,
    AGR_BWE_SDK_API.c:277) and AGR_Sate_Encoder_Encode in nBytesOut[].
CPU tests use the host emulation of the kernel source, `-m gpu` tests the gfx950 library through its C ABI."""
import ctypes as C

import numpy as np
import pytest

import refcodec as R
import solo_testlib as T

need_ref = pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not present")


# ---------------------------------------------------------------------------------------------------------------------
# CPU: kernel source (host emulation) vs compiled reference
# ---------------------------------------------------------------------------------------------------------------------
@need_ref
@pytest.mark.parametrize("rate,mdi", [(13600, 1), (24000, 0), (24000, 1), (9600, 1)])
def test_emu_md_index_and_rates_vs_reference(rate, mdi):
    P = 10
    for seed in (70, 71, 72):
        pcm = R.synth_stream(seed, P)
        er, ee = R.RefEncoder("fix", rate=rate, use_md_index=mdi), T.EmuEncoder(rate, mdi)
        recs = []
        for p in range(P):
            a, b = er.encode(pcm[p]), ee.encode(pcm[p])
            assert a == b, (rate, mdi, seed, p, a[1:], b[1:])
            recs.append(a)
        recv = T.bernoulli_recv(1, P, 0.3, seed)[0]
        dr, de = R.RefDecoder("fix", use_md_index=mdi), T.EmuDecoder(mdi)
        for p, (pl, n0, n1) in enumerate(recs):
            m = int(recv[p])
            args = R.map_loss(pl, n0, n1, not (m & 1), not (m & 2))
            x, r1 = dr.decode(*args)
            y, r2 = de.decode(*args)
            assert r1 == r2 == 0 and np.array_equal(x, y), (rate, mdi, seed, p, m)


# ---------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch


def _ref_streams(pcm, rate, mdi):
    out = []
    for i in range(pcm.shape[0]):
        e = R.RefEncoder("fix", rate=rate, use_md_index=mdi)
        out.append([e.encode(pcm[i, p]) for p in range(pcm.shape[1])])
    return out


@pytest.mark.gpu
@need_ref
@pytest.mark.parametrize("rate,mdi", [(13600, 1), (24000, 0), (24000, 1)])
def test_gpu_md_index_and_second_rate_vs_compiled_reference(torch_cuda, rate, mdi):
    """encoder bytes and masked decode (all four lostflag states) against the compiled reference"""
    import solo_amd
    torch = torch_cuda
    N, P = 16, 12
    pcm = np.stack([R.synth_stream(900 + i, P) for i in range(N)])
    b = solo_amd.SoloBatch(N, rate=rate, encoder=True, decoder=True, slot_bytes=512, use_md_index=mdi)
    bits, nb, st = b.encode(torch.from_numpy(pcm).to(b.device))
    recv = T.bernoulli_recv(N, P, 0.3, 21 + mdi)
    out, st2 = b.decode(bits, nb, torch.from_numpy(recv).to(b.device))
    torch.cuda.synchronize()
    assert int(st.abs().max()) == 0 and int(st2.abs().max()) == 0
    hb, hn, ho = bits.cpu().numpy(), nb.cpu().numpy(), out.cpu().numpy()
    ref = _ref_streams(pcm, rate, mdi)
    assert len({int(m) for m in recv.ravel()}) == 4
    for i in range(N):
        d = R.RefDecoder("fix", use_md_index=mdi)
        for p, (pl, n0, n1) in enumerate(ref[i]):
            assert (int(hn[i, p, 0]), int(hn[i, p, 1])) == (n0, n1), (i, p)
            assert hb[i, p, :n0].tobytes() == pl, (i, p)
            m = int(recv[i, p])
            x, ret = d.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
            assert ret == 0 and np.array_equal(ho[i, p], x), (i, p, m)


@pytest.mark.gpu
@need_ref
@pytest.mark.parametrize("mdi", [0, 1])
def test_gpu_split_arrivals_vs_compiled_reference(torch_cuda, mdi):
    """Receiver front end: descriptions in two arrival slots, missing / (with useMDIndex = 1) swapped / duplicated; the PCM must
    equal the compiled reference decoder called with the correctly ordered (ptr, nBytes, lostflag) of test/dec_main.c:255-378."""
    import solo_amd
    torch = torch_cuda
    N, P, S = 24, 12, 256
    pcm = np.stack([R.synth_stream(300 + i, P) for i in range(N)])
    ref = _ref_streams(pcm, 13600, mdi)
    rng = np.random.default_rng(31 + mdi)
    recv = T.bernoulli_recv(N, P, 0.3, 31 + mdi)
    dA = np.zeros((N, P, S), np.uint8); dB = np.zeros((N, P, S), np.uint8)
    lA = np.zeros((N, P), np.int16); lB = np.zeros((N, P), np.int16)
    kinds = set()
    for i in range(N):
        for p, (pl, n0, n1) in enumerate(ref[i]):
            md1, md2 = np.frombuffer(pl[:n0 - n1], np.uint8), np.frombuffer(pl[n0 - n1:n0], np.uint8)
            a = md1 if recv[i, p] & 1 else None
            b = md2 if recv[i, p] & 2 else None
            if mdi:
                r = int(rng.integers(0, 4))
                if r == 1:
                    a, b = b, a; kinds.add("swapped")
                elif r == 2 and a is not None and b is None:
                    b = a; kinds.add("dup1")
                elif r == 3 and b is not None and a is None:
                    a = b; kinds.add("dup2")
            if a is not None:
                dA[i, p, :a.size] = a; lA[i, p] = a.size
            if b is not None:
                dB[i, p, :b.size] = b; lB[i, p] = b.size
    if mdi:
        assert kinds == {"swapped", "dup1", "dup2"}
    d = solo_amd.SoloBatch(N, encoder=False, decoder=True, slot_bytes=512, use_md_index=mdi)
    dev = d.device
    got, st = d.decode_split(torch.from_numpy(dA).to(dev), torch.from_numpy(lA).to(dev), torch.from_numpy(dB).to(dev), torch.from_numpy(lB).to(dev))
    torch.cuda.synchronize()
    assert int(st.abs().max()) == 0
    g = got.cpu().numpy()
    for i in range(N):
        dr = R.RefDecoder("fix", use_md_index=mdi)
        for p, (pl, n0, n1) in enumerate(ref[i]):
            m = int(recv[i, p])
            x, ret = dr.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
            assert ret == 0 and np.array_equal(g[i, p], x), (i, p, m)


@pytest.mark.gpu
@need_ref
def test_gpu_corrupted_payloads_vs_compiled_reference(torch_cuda):
    """Bit errors on the GPU, packet by packet (one call per packet so that every packet's return code is seen): a corrupted
    packet the range decoder still accepts decodes to the reference's PCM and leaves the reference's state, a rejected one returns
    the reference's negative code; a stream is compared up to and including its first rejection (after one, the reference's state
    depends on an uninitialised stack buffer, SKP_Silk_decode_frame.c:358).  A corrupted rate index that claims another internal
    rate is decoded and resampled by the reference and rejected (-12) by this build (one internal rate per handle)."""
    import solo_amd
    torch = torch_cuda
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    bits, nb = z["bits"], z["nbytes"]
    N, P, S = 48, 25, bits.shape[2]
    rng = np.random.default_rng(5)
    cb = np.zeros((N, P, S), np.uint8)
    cn = np.zeros((N, P, 2), np.int16)
    recv = np.zeros((N, P), np.uint8)
    hit = np.zeros((N, P), bool)
    for t in range(N):
        s = t % 8
        for p in range(P):
            n0 = int(nb[s, p, 0])
            pl = bits[s, p].copy()
            hit[t, p] = rng.random() < 0.25
            if hit[t, p]:
                for _ in range(rng.integers(1, 4)):
                    pl[rng.integers(0, n0)] = rng.integers(0, 256)
            cb[t, p] = pl
            cn[t, p] = nb[s, p]
            mode = rng.integers(0, 4)                    # 0, 3: both, 1: MD1 lost, 2: MD2 lost
            recv[t, p] = (0 if mode == 1 else 1) | (0 if mode == 2 else 2)
    d = solo_amd.SoloBatch(N, encoder=False, decoder=True, slot_bytes=S)
    dev = d.device
    got = np.zeros((N, P, 640), np.int16)
    rets = np.zeros((N, P), np.int32)
    for p in range(P):
        pcm, st = d.decode(torch.from_numpy(np.ascontiguousarray(cb[:, p:p + 1])).to(dev), torch.from_numpy(np.ascontiguousarray(cn[:, p:p + 1])).to(dev),
                           torch.from_numpy(np.ascontiguousarray(recv[:, p:p + 1])).to(dev))
        got[:, p] = pcm.cpu().numpy()[:, 0]
        rets[:, p] = st.cpu().numpy()
    n_rejected = n_garbage = n_other_rate = 0
    for t in range(N):
        dr = R.RefDecoder("fix")
        for p in range(P):
            n0, n1 = int(cn[t, p, 0]), int(cn[t, p, 1])
            m = int(recv[t, p])
            x, r1 = dr.decode(*R.map_loss(cb[t, p, :n0].tobytes(), n0, n1, not (m & 1), not (m & 2)))
            r2 = int(rets[t, p])
            if r1 == 0 and r2 == -12 and hit[t, p]:
                n_other_rate += 1
                break
            assert r1 == r2, (t, p, r1, r2)
            if r1 < 0:
                n_rejected += 1
                break
            assert np.array_equal(x, got[t, p]), (t, p)
            n_garbage += int(hit[t, p])
    assert n_rejected >= 8 and n_garbage >= 40 and n_other_rate <= 4, (n_rejected, n_garbage, n_other_rate)


@pytest.mark.gpu
def test_gpu_inconsistent_length_records_are_not_dereferenced(torch_cuda):
    """Network-controlled lengths: total > slot, len(MD2) > total, negative lengths, a second description shorter than its
    high-band bytes.  Such packets are concealed as lost (status -11 / -12), nothing outside the slot is read, and the streams
    around them decode as if those packets had been lost."""
    import solo_amd
    torch = torch_cuda
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    bits, nb = z["bits"].copy(), z["nbytes"].copy()
    N, P, S = bits.shape
    bad = {0: (S + 40, 20), 1: (60, 90), 2: (70, -5), 3: (50, 3), 4: (32767, 32767), 5: (80, 81)}
    want_status = {0: -11, 1: -12, 2: -12, 3: -12, 4: -11, 5: -12}
    recv = np.full((N, P), 3, np.uint8)
    for s, (n0, n1) in bad.items():
        nb[s, 7] = (n0, n1)
    d = solo_amd.SoloBatch(N, encoder=False, decoder=True, slot_bytes=S)
    dev = d.device
    got, st = d.decode(torch.from_numpy(bits).to(dev), torch.from_numpy(nb).to(dev), torch.from_numpy(recv).to(dev))
    torch.cuda.synchronize()
    status = st.cpu().numpy()
    for s in range(N):
        assert int(status[s]) == want_status.get(s, 0), (s, int(status[s]))
    # reference behaviour for "this packet was lost": the same streams with packet 7 masked out
    recv2 = recv.copy()
    for s in bad:
        recv2[s, 7] = 0
    d2 = solo_amd.SoloBatch(N, encoder=False, decoder=True, slot_bytes=S)
    want, st2 = d2.decode(torch.from_numpy(bits).to(dev), torch.from_numpy(z["nbytes"]).to(dev), torch.from_numpy(recv2).to(dev))
    torch.cuda.synchronize()
    assert int(st2.abs().max()) == 0
    assert np.array_equal(got.cpu().numpy(), want.cpu().numpy())


@pytest.mark.gpu
@need_ref
def test_legacy_api_write_backs_vs_compiled_reference(torch_cuda):
    """AGR_Sate_Decoder_Decode rewrites the caller's nBytes[0..1] and *nSamplesOut, AGR_Sate_Encoder_Encode fills nBytesOut[0..1]
    and leaves [2..5] alone: compare what both libraries leave in the caller's arrays, call by call (CLI loss pattern, all four
    lostflag values incl. lostflag = 1 with the previous lengths like test/dec_main.c:377)."""
    import solo_amd
    lib = solo_amd.load_library()
    pcm = np.fromfile(T.GOLDEN + "/Ch_f1_raw.pcm", np.int16)
    P = 48
    # encoder side
    ctrl = solo_amd.default_enc_ctrl()
    h = lib.AGR_Sate_Encoder_Init(C.byref(ctrl))
    er = R.RefEncoder("fix")
    buf = np.zeros(1024, np.uint8)
    recs = []
    for p in range(P):
        x = np.ascontiguousarray(pcm[p * 640:(p + 1) * 640])
        nbv = np.full(6, 77, np.int16)
        n = lib.AGR_Sate_Encoder_Encode(h, x.ctypes.data, buf.ctypes.data, 1024, nbv.ctypes.data)
        er._nb[:] = 77
        nr = er.lib.AGR_Sate_Encoder_Encode(er.h, x.ctypes.data, er._bits.ctypes.data, 1024, er._nb.ctypes.data)
        assert n == nr and np.array_equal(nbv, er._nb), (p, n, nr, nbv, er._nb)
        assert buf[:n].tobytes() == er._bits[:n].tobytes()
        recs.append((buf[:n].tobytes(), int(nbv[0]), int(nbv[1])))
    lib.AGR_Sate_Encoder_Uninit(h)
    # decoder side
    dctrl = solo_amd.default_dec_ctrl()
    hd = lib.AGR_Sate_Decoder_Init(C.byref(dctrl))
    dr = R.RefDecoder("fix")
    pat = R.cli_loss_pattern(P, 30)
    out = np.zeros(1920, np.int16)
    ns = np.zeros(1, np.int16)
    flags = set()
    for p, (pl, n0, n1) in enumerate(recs):
        payload, a0, a1, flag = R.map_loss(pl, n0, n1, *pat[p])
        flags.add(flag)
        b = np.zeros(1100, np.uint8)
        b[:len(payload)] = np.frombuffer(payload, np.uint8)
        nbv = np.array([a0, a1, 55, 55, 55, 55], np.int16)
        ns[0] = -1
        ret = lib.AGR_Sate_Decoder_Decode(hd, out.ctypes.data, ns.ctypes.data, b.ctypes.data, nbv.ctypes.data, flag)
        x, r1 = dr.decode(payload, a0, a1, flag)
        assert ret == r1 == 0 and np.array_equal(out[:640], x), p
        assert (int(nbv[0]), int(nbv[1])) == dr.nbytes_after, (p, flag, nbv[:2], dr.nbytes_after)
        assert list(nbv[2:]) == [55] * 4 and int(ns[0]) == dr.nsamples_out == 640
    assert flags == {1, 2, 3, 4}
    # an empty packet: -1 and nothing is touched, in both libraries (AGR_BWE_SDK_API.c:266)
    nbv = np.array([0, 0, 55, 55, 55, 55], np.int16)
    ns[0] = -7
    assert lib.AGR_Sate_Decoder_Decode(hd, out.ctypes.data, ns.ctypes.data, b.ctypes.data, nbv.ctypes.data, 4) == -1
    assert list(nbv) == [0, 0, 55, 55, 55, 55] and int(ns[0]) == -7
    x, r1 = dr.decode(b"", 0, 0, 4)
    assert r1 == -1
    lib.AGR_Sate_Decoder_Uninit(hd)
