"""32 kHz mode of the decoder (`samplerate == 32000`, libBWE/AGR_BWE_SDK_API.c:197 -- 16 kHz bands, SILK wide
band: fs_kHz 16, LPC order 16, order-16 NLSF codebooks, stage-3 pitch contours, 1280-sample packets).  The kernel source is the
same as for the 16 kHz rate, compiled with SX_FS_KHZ = 16.  CPU tests: host emulation of that build against the committed
goldens (made by the compiled reference, tests/golden/make_golden.py) and against the reference itself; GPU tests: the HIP
kernels through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import refcodec as R
import solo_testlib as T


def _wb():
    return np.load(T.GOLDEN + "/wb4x20.npz")


def test_wb_goldens_pinned():
    z, g = _wb(), T.golden_json()
    assert T.md5(z["bits"]) == g["wb_bits_md5"] and T.md5(z["dec_clean"]) == g["wb_dec_clean_md5"] and T.md5(z["dec_loss"]) == g["wb_dec_loss_md5"]
    assert (z["recv"][1, :2] == 0).all()              # stream 1 starts with lost packets (decoder cold start at 24 kHz -> 16 kHz)


def test_wb_emulation_vs_goldens():
    z = _wb()
    bits, nb, recv = z["bits"], z["nbytes"], z["recv"]
    N, P = recv.shape
    for mask, key in ((None, "dec_clean"), (recv, "dec_loss")):
        for s in range(N):
            dec = T.EmuDecoder(2 if s == 3 else 0, wb=True)           # stream 3 was coded with joint_mode 1
            for p in range(P):
                n0, n1 = int(nb[s, p, 0]), int(nb[s, p, 1])
                m = 3 if mask is None else int(mask[s, p])
                x, ret = dec.decode(*R.map_loss(bits[s, p, :n0].tobytes(), n0, n1, not (m & 1), not (m & 2)))
                assert ret == 0 and x.size == 1280
                assert np.array_equal(x, z[key][s, p]), (key, s, p)


@pytest.mark.parametrize("split", [0, 1])
def test_wb_more_frames_than_carried_emulation(split):
    """The one known input on which the reference's output high-pass runs (SKP_Silk_decode_frame.c:381, nFramesDecoded > 2): a
    corrupted description whose termination symbol announces more frames than the packet carries, so that the next packet's first
    call decodes on in the OLD buffer as "frame 3" (tests/golden/make_more_frames_golden.py).  Six packets decode like the reference,
    the seventh is rejected with the reference's code."""
    z = np.load(T.GOLDEN + "/wb_more_frames.npz")
    d = T.EmuDecoder(0, wb=True, split=split)
    for p in range(z["recv"].shape[1]):
        n0, n1, m = int(z["nbytes"][0, p, 0]), int(z["nbytes"][0, p, 1]), int(z["recv"][0, p])
        x, ret = d.decode(*R.map_loss(z["bits"][0, p, :n0].tobytes(), n0, n1, not (m & 1), not (m & 2)))
        assert ret == int(z["ret"][0, p]), (p, ret)
        if ret == 0:
            assert np.array_equal(x, z["dec"][0, p]), p


@pytest.mark.parametrize("split", [0, 1])
def test_wb_edge_family_goldens_emulation(split):
    """Un-speech-like inputs in the 32 kHz mode (tests/golden/edge_wb13x8.npz, one stream per family of solo_amd.synth.edge_stream
    except the full-scale square wave, on which the compiled reference overflows its stack at 32 kHz): encoder emulation against the
    reference bitstreams, decoder emulation (both paths) against the CRC-32 of every packet the reference decoded under description loss."""
    import zlib
    z = np.load(T.GOLDEN + "/edge_wb13x8.npz")
    N, P, _ = z["pcm"].shape
    for i in range(N):
        e, d = T.EmuEncoder(24000, 0, wb=True), T.EmuDecoder(0, wb=True, split=split)
        for p in range(P):
            pl, n0, n1 = e.encode(z["pcm"][i, p])
            assert (n0, n1) == tuple(int(v) for v in z["nbytes"][i, p]) and pl == z["bits"][i, p, :n0].tobytes(), (i, p)
            m = int(z["recv"][i, p])
            x, ret = d.decode(pl, n0, n1, 1) if m == 0 else d.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
            assert ret == 0 and zlib.crc32(x.tobytes()) == int(z["dec_crc"][i, p]), (i, p, m)


@pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not built")
@pytest.mark.parametrize("rate,joint", [(16000, 0), (24000, 0), (32000, 1), (40000, 0)])
def test_wb_emulation_vs_reference(rate, joint):
    """Rates from the lowest that keeps the reference encoder at 16 kHz internally (SILK rate >= WB2MB_BITRATE_BPS,
    SKP_Silk_control_audio_bandwidth.c:44) upwards; 30 % description loss, leading losses, both high-band framings."""
    P = 30
    pcm = T.synth_stream_32k(900 + rate // 1000, P)
    enc = R.RefEncoder("fix", rate=rate, samplerate=32000, joint=joint)
    recs = [enc.encode(pcm[p]) for p in range(P)]
    for variant in range(2):
        lost = np.random.default_rng(rate + variant).random((P, 2)) < 0.3
        if variant == 1:
            lost[:3] = True
        dr, de = R.RefDecoder("fix", samplerate=32000, joint=joint), T.EmuDecoder(2 if joint else 0, wb=True)
        for p, (pl, n0, n1) in enumerate(recs):
            a = R.map_loss(pl, n0, n1, bool(lost[p, 0]), bool(lost[p, 1]))
            x, r1 = dr.decode(*a)
            y, r2 = de.decode(*a)
            assert r1 == r2 == 0 and np.array_equal(x, y), (rate, joint, variant, p)


def test_wb_encoder_emulation_vs_goldens():
    """The ENCODER source compiled for the 32 kHz mode (16 kHz pitch stage 3, order-16 Burg / NLSF quantiser, wide-band de-esser,
    order-16 prediction in the quantiser, high band on 320-sample frames) reproduces the reference bitstreams byte for byte."""
    z = _wb()
    N, P = z["recv"].shape
    for s in range(N):
        e = T.EmuEncoder(24000, 2 if s == 3 else 0, wb=True)
        for p in range(P):
            pl, n0, n1 = e.encode(z["pcm"][s, p])
            assert (n0, n1) == tuple(int(v) for v in z["nbytes"][s, p]), (s, p)
            assert pl == z["bits"][s, p, :n0].tobytes(), (s, p)


@pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not built")
@pytest.mark.parametrize("rate,joint,dtx", [(15600, 0, 0), (20000, 1, 1), (32000, 0, 0), (40000, 0, 1)])
def test_wb_encoder_emulation_vs_reference(rate, joint, dtx):
    P = 25
    pcm = T.synth_stream_32k(1200 + rate // 1000, P)
    rng = np.random.default_rng(rate)
    t = np.arange(1280 * 4) / 32000.0
    edge = [np.zeros(1280 * 4), rng.integers(-32768, 32767, 1280 * 4), 25000 * np.sin(2 * np.pi * 180 * t), 30000 * np.sign(np.sin(2 * np.pi * 110 * t))]
    pcm = np.concatenate([pcm] + [np.asarray(x).astype(np.int16).reshape(4, 1280) for x in edge])
    er = R.RefEncoder("fix", rate=rate, samplerate=32000, joint=joint, dtx=dtx)
    ee = T.EmuEncoder(rate, (2 if joint else 0) | (4 if dtx else 0), wb=True)
    for p in range(pcm.shape[0]):
        assert ee.encode(pcm[p]) == er.encode(pcm[p]), (rate, joint, dtx, p)


def test_wb_file_level_known_answers_emulation():
    """.bit container of a 32 kHz stream and its decode at 0 % / 30 % CLI loss against the md5s made with the compiled reference."""
    g, z = T.golden_json(), _wb()
    e = T.EmuEncoder(24000, wb=True)
    recs = [e.encode(z["pcm"][0, p]) for p in range(20)]
    assert T.md5(T.write_bit_container(recs)) == g["wb_kat_bit_md5"]
    for loss in (0, 30):
        d = T.EmuDecoder(wb=True)
        pat = R.cli_loss_pattern(20, loss, [(r[1], r[2]) for r in recs])
        out = np.concatenate([d.decode(*R.map_loss(pl, n0, n1, *pat[p]))[0] for p, (pl, n0, n1) in enumerate(recs)])
        assert T.md5(out) == g["wb_kat_dec_loss%d_md5" % loss]


def test_wb_decoder_rejects_other_internal_rates():
    """A narrow-band stream (16 kHz API rate) handed to the 32 kHz decoder: the reference would switch its SILK core to 8 kHz
    and resample; this build decodes one internal rate per handle and reports a payload error instead of producing audio."""
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    n0, n1 = int(z["nbytes"][0, 0, 0]), int(z["nbytes"][0, 0, 1])
    dec = T.EmuDecoder(wb=True)
    x, ret = dec.decode(z["bits"][0, 0, :n0].tobytes(), n0, n1, 4)
    assert ret < 0


# ---- GPU -------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch


def _gpu_decode(torch, bits, nb, recv, joint=0):
    import solo_amd
    N, P, S = bits.shape
    b = solo_amd.SoloBatch(N, encoder=False, decoder=True, slot_bytes=S, samplerate=32000, joint=joint)
    pcm, status = b.decode(torch.from_numpy(np.ascontiguousarray(bits)).to(b.device), torch.from_numpy(np.ascontiguousarray(nb)).to(b.device),
                           None if recv is None else torch.from_numpy(np.ascontiguousarray(recv)).to(b.device))
    torch.cuda.synchronize()
    assert tuple(pcm.shape) == (N, P, 1280) and int(status.abs().max()) == 0
    return pcm.cpu().numpy()


@pytest.mark.gpu
def test_wb_gpu_goldens(torch_cuda):
    z = _wb()
    for recv, key in ((None, "dec_clean"), (z["recv"], "dec_loss")):
        out = _gpu_decode(torch_cuda, z["bits"][:3], z["nbytes"][:3], None if recv is None else recv[:3])
        assert np.array_equal(out, z[key][:3]), key
        out = _gpu_decode(torch_cuda, z["bits"][3:], z["nbytes"][3:], None if recv is None else recv[3:], joint=1)
        assert np.array_equal(out, z[key][3:]), key + " joint"


@pytest.mark.gpu
def test_wb_gpu_more_frames_than_carried(torch_cuda):
    """tests/golden/wb_more_frames.npz on the GPU (see test_wb_more_frames_than_carried_emulation): the packets before the rejected
    one equal the reference's PCM -- packet 5 through the output high-pass -- and the stream's status is the reference's code."""
    import solo_amd
    torch = torch_cuda
    z = np.load(T.GOLDEN + "/wb_more_frames.npz")
    P = z["recv"].shape[1]
    b = solo_amd.SoloBatch(1, encoder=False, decoder=True, slot_bytes=z["bits"].shape[2], samplerate=32000)
    out, st = b.decode(torch.from_numpy(np.ascontiguousarray(z["bits"])).to(b.device), torch.from_numpy(np.ascontiguousarray(z["nbytes"])).to(b.device),
                       torch.from_numpy(np.ascontiguousarray(z["recv"])).to(b.device))
    torch.cuda.synchronize()
    assert int(st[0]) == int(z["ret"][0, P - 1])
    assert np.array_equal(out.cpu().numpy()[0, :P - 1], z["dec"][0, :P - 1])


@pytest.mark.gpu
def test_wb_gpu_packetwise_and_wrong_rate(torch_cuda):
    import solo_amd
    torch = torch_cuda
    z = _wb()
    bits, nb, recv = z["bits"][:3], z["nbytes"][:3], z["recv"][:3]
    N, P, S = bits.shape
    b = solo_amd.SoloBatch(N, encoder=False, decoder=True, slot_bytes=S, samplerate=32000)
    for p in range(P):                      # state carries in HBM between launches
        pcm, st = b.decode(torch.from_numpy(np.ascontiguousarray(bits[:, p:p + 1])).to(b.device),
                           torch.from_numpy(np.ascontiguousarray(nb[:, p:p + 1])).to(b.device),
                           torch.from_numpy(np.ascontiguousarray(recv[:, p:p + 1])).to(b.device))
        assert np.array_equal(pcm.cpu().numpy()[:, 0], z["dec_loss"][:3, p]), p
    nbz = np.load(T.GOLDEN + "/synth8x25.npz")      # narrow-band streams into the 32 kHz decoder: rejected, not decoded
    b2 = solo_amd.SoloBatch(8, encoder=False, decoder=True, slot_bytes=nbz["bits"].shape[2], samplerate=32000)
    pcm, st = b2.decode(torch.from_numpy(nbz["bits"][:, :1].copy()).to(b2.device), torch.from_numpy(nbz["nbytes"][:, :1].copy()).to(b2.device))
    assert (st.cpu().numpy() < 0).all()
    with pytest.raises(RuntimeError):       # below 15.6 kbps the reference leaves 16 kHz internally: refused, not approximated
        solo_amd.SoloBatch(4, rate=13600, encoder=True, decoder=False, samplerate=32000)


@pytest.mark.gpu
def test_wb_gpu_encoder_goldens_and_round_trip(torch_cuda):
    """The 32 kHz ENCODER on the GPU (three-kernel pipeline of the wide-band build): reference bitstreams byte for byte, then the
    round trip through the wide-band decoder; one call and packet-by-packet calls."""
    import solo_amd
    torch = torch_cuda
    z = _wb()
    for sl, joint in ((slice(0, 3), 0), (slice(3, 4), 1)):
        pcm, bits_ref, nb_ref = z["pcm"][sl], z["bits"][sl], z["nbytes"][sl]
        N, P, S = bits_ref.shape
        b = solo_amd.SoloBatch(N, rate=24000, encoder=True, decoder=True, slot_bytes=S, samplerate=32000, joint=joint)
        bits, nb, st = b.encode(torch.from_numpy(np.ascontiguousarray(pcm)).to(b.device))
        torch.cuda.synchronize()
        assert int(st.abs().max()) == 0
        nbh, bh = nb.cpu().numpy(), bits.cpu().numpy()
        assert np.array_equal(nbh, nb_ref), joint
        for i in range(N):
            for p in range(P):
                n0 = int(nb_ref[i, p, 0])
                assert np.array_equal(bh[i, p, :n0], bits_ref[i, p, :n0]), (joint, i, p)
        out, st2 = b.decode(bits, nb)
        torch.cuda.synchronize()
        assert int(st2.abs().max()) == 0 and np.array_equal(out.cpu().numpy(), z["dec_clean"][sl])
        b2 = solo_amd.SoloBatch(N, rate=24000, encoder=True, decoder=False, slot_bytes=S, samplerate=32000, joint=joint)
        for p in range(6):
            bits1, nb1, st1 = b2.encode(torch.from_numpy(np.ascontiguousarray(pcm[:, p:p + 1])).to(b2.device))
            assert np.array_equal(nb1.cpu().numpy()[:, 0], nb_ref[:, p])
            for i in range(N):
                n0 = int(nb_ref[i, p, 0])
                assert np.array_equal(bits1.cpu().numpy()[i, 0, :n0], bits_ref[i, p, :n0]), (joint, i, p)


@pytest.mark.gpu
@pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not present on this box")
def test_wb_gpu_encoder_many_streams_vs_reference(torch_cuda):
    """96 streams x 8 packets at 15.6 .. 40 kbps with DTX on half of them... one configuration per batch: two batches."""
    import solo_amd
    torch = torch_cuda
    for rate, dtx in ((15600, 0), (36000, 1)):
        N, P = 48, 8
        pcm = np.stack([T.synth_stream_32k(7000 + rate // 100 + i, P) for i in range(N)])
        b = solo_amd.SoloBatch(N, rate=rate, encoder=True, decoder=False, slot_bytes=512, samplerate=32000, dtx=dtx)
        bits, nb, st = b.encode(torch.from_numpy(pcm).to(b.device))
        torch.cuda.synchronize()
        assert int(st.abs().max()) == 0
        bh, nbh = bits.cpu().numpy(), nb.cpu().numpy()
        for i in range(N):
            e = R.RefEncoder("fix", rate=rate, samplerate=32000, dtx=dtx)
            for p in range(P):
                pl, n0, n1 = e.encode(pcm[i, p])
                assert (int(nbh[i, p, 0]), int(nbh[i, p, 1])) == (n0, n1), (rate, i, p)
                assert bh[i, p, :n0].tobytes() == pl[:n0], (rate, i, p)


@pytest.mark.gpu
@pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not present on this box")
def test_wb_gpu_many_streams_vs_reference_and_legacy_api(torch_cuda):
    """64 streams x 10 packets, 30 % description loss, against the compiled reference; and the legacy single-stream entry points
    (AGR_Sate_Decoder_Init / Decode with samplerate = 32000) for one stream."""
    import solo_amd
    N, P = 64, 10
    streams = []
    for i in range(N):
        e = R.RefEncoder("fix", rate=20000 + 1000 * (i % 8), samplerate=32000)
        pcm = T.synth_stream_32k(3000 + i, P)
        streams.append([e.encode(pcm[p]) for p in range(P)])
    bits, nb = T.pack_slots(streams)
    recv = T.bernoulli_recv(N, P, 0.3, 77)
    recv[::7, 0] = 0
    out = _gpu_decode(torch_cuda, bits, nb, recv)
    for i in range(N):
        d = R.RefDecoder("fix", samplerate=32000)
        for p, (pl, n0, n1) in enumerate(streams[i]):
            m = int(recv[i, p])
            x, ret = d.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
            assert ret == 0 and np.array_equal(out[i, p], x), (i, p)
    lib = solo_amd.load_library()
    ctrl = solo_amd.default_dec_ctrl(samplerate=32000)
    h = lib.AGR_Sate_Decoder_Init(C.byref(ctrl))
    assert h
    d = R.RefDecoder("fix", samplerate=32000)
    pcm = np.zeros(1920, np.int16)
    ns = C.c_int16(0)
    for p, (pl, n0, n1) in enumerate(streams[0]):
        m = int(recv[1, p])
        a = R.map_loss(pl, n0, n1, not (m & 1), not (m & 2))
        x, ret = d.decode(*a)
        buf = np.zeros(1100, np.uint8)
        buf[:len(a[0])] = np.frombuffer(a[0], np.uint8)
        nbv = (C.c_int16 * 6)(a[1], a[2], 0, 0, 0, 0)
        r = lib.AGR_Sate_Decoder_Decode(h, pcm.ctypes.data_as(C.c_void_p), C.byref(ns), buf.ctypes.data_as(C.c_void_p), nbv, a[3])
        assert r == ret == 0 and ns.value == 1280 and np.array_equal(pcm[:1280], x), p
    lib.AGR_Sate_Decoder_Uninit(h)


@pytest.mark.gpu
def test_wb_gpu_legacy_encoder_api(torch_cuda):
    """AGR_Sate_Encoder_Init / _Encode with samplerate = 32000 (1280 samples per call) against the golden bitstreams; a rate below
    15.6 kbps is refused by Init like any other unsupported configuration."""
    import solo_amd
    lib = solo_amd.load_library()
    z = _wb()
    ctrl = solo_amd.default_enc_ctrl(rate=24000, samplerate=32000)
    h = lib.AGR_Sate_Encoder_Init(C.byref(ctrl))
    assert h
    bits = np.zeros(1100, np.uint8)
    nb = np.zeros(6, np.int16)
    for p in range(6):
        pcm = np.ascontiguousarray(z["pcm"][0, p])
        n = lib.AGR_Sate_Encoder_Encode(h, pcm.ctypes.data_as(C.c_void_p), bits.ctypes.data_as(C.c_void_p), 1024, nb.ctypes.data_as(C.c_void_p))
        n0 = int(z["nbytes"][0, p, 0])
        assert n == n0 and (int(nb[0]), int(nb[1])) == (n0, int(z["nbytes"][0, p, 1]))
        assert np.array_equal(bits[:n0], z["bits"][0, p, :n0]), p
    lib.AGR_Sate_Encoder_Uninit(h)
    low = solo_amd.default_enc_ctrl(rate=13600, samplerate=32000)
    assert not lib.AGR_Sate_Encoder_Init(C.byref(low))
