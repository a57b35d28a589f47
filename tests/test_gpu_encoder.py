"""GPU parity of the encoder (rows E0-E9) through the C ABI: HIP kernels vs the committed golden bitstreams
and, when oracle/_ref travelled with the snapshot, vs the compiled reference itself.  Bit-exact."""
import numpy as np
import pytest

import refcodec as R
import solo_testlib as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch


def _gpu_encode(torch, pcm, slot=512, batch=None):
    import solo_amd
    N, P, _ = pcm.shape
    b = batch or solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=slot)
    bits, nb, status = b.encode(torch.from_numpy(np.ascontiguousarray(pcm)).to(b.device))
    torch.cuda.synchronize()
    assert int(status.abs().max()) == 0
    return bits.cpu().numpy(), nb.cpu().numpy()


def _assert_streams_equal(bits, nb, ref_bits, ref_nb):
    assert np.array_equal(nb, ref_nb)
    N, P, _ = bits.shape
    for i in range(N):
        for p in range(P):
            n = int(nb[i, p, 0])
            assert np.array_equal(bits[i, p, :n], ref_bits[i, p, :n]), (i, p)


def test_ch_f1_bitstream_md5(torch_cuda):
    """The reference's own sample file -> the byte-exact .bit container its CLI writes (known-answer md5)."""
    g = T.golden_json()
    pcm = np.fromfile(T.GOLDEN + "/Ch_f1_raw.pcm", np.int16)
    P = len(pcm) // 640
    bits, nb = _gpu_encode(torch_cuda, pcm[:P * 640].reshape(1, P, 640))
    recs = [(bits[0, p, :nb[0, p, 0]].tobytes(), int(nb[0, p, 0]), int(nb[0, p, 1])) for p in range(P)]
    assert T.md5(T.write_bit_container(recs)) == g["ch_f1_bit_md5"]


def test_synthetic_goldens(torch_cuda):
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    bits, nb = _gpu_encode(torch_cuda, z["pcm"])
    _assert_streams_equal(bits, nb, z["bits"], z["nbytes"])


def test_edge_family_goldens_encode_and_decode(torch_cuda):
    """Un-speech-like inputs (tests/golden/edge28x12.npz: silence with stray LSBs, full-scale noise / square waves / sweeps, DC,
    impulses, 90 dB level ramps, clipping, bursts, high-band-only tones, the Nyquist pattern, sub-audio sines, random walks):
    reference bitstreams byte for byte, and the decode under the fixture's description-loss mask packet by packet (CRC-32 of the
    compiled reference's PCM)."""
    import zlib
    import solo_amd
    torch = torch_cuda
    z = np.load(T.GOLDEN + "/edge28x12.npz")
    N, P, _ = z["pcm"].shape
    b = solo_amd.SoloBatch(N, encoder=True, decoder=True, slot_bytes=512)
    bits, nb, st = b.encode(torch.from_numpy(z["pcm"]).to(b.device))
    hb, hn = bits.cpu().numpy(), nb.cpu().numpy()
    assert int(st.abs().max()) == 0
    _assert_streams_equal(hb, hn, z["bits"], z["nbytes"])
    out, st2 = b.decode(bits, nb, torch.from_numpy(z["recv"]).to(b.device))
    assert int(st2.abs().max()) == 0
    ho = out.cpu().numpy()
    for i in range(N):
        for p in range(P):
            assert zlib.crc32(ho[i, p].tobytes()) == int(z["dec_crc"][i, p]), (i, p, int(z["recv"][i, p]))


def test_packetwise_calls_equal_one_call(torch_cuda):
    import solo_amd
    torch = torch_cuda
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    N, P, _ = z["pcm"].shape
    b = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512)
    for p in range(P):
        bits, nb = _gpu_encode(torch, z["pcm"][:, p:p + 1], batch=b)
        _assert_streams_equal(bits, nb, z["bits"][:, p:p + 1], z["nbytes"][:, p:p + 1])


@pytest.mark.parametrize("chunk,gate", [("0", "1"), ("3", "1"), ("7", "0"), ("1", "0")])
def test_pipeline_shapes_give_the_same_bits(torch_cuda, monkeypatch, chunk, gate):
    """The encoder's chunked three-stream pipeline (SOLO_ENC_CHUNK packets per chunk, 0 = sequential; residency gate on / off)
    is a schedule, not an algorithm: every shape must reproduce the golden bitstreams, also when a second call follows."""
    import solo_amd
    monkeypatch.setenv("SOLO_ENC_CHUNK", chunk)
    monkeypatch.setenv("SOLO_ENC_GATE", gate)
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    N, P, _ = z["pcm"].shape
    b = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512)      # the environment is read at the first encode
    bits, nb = _gpu_encode(torch_cuda, z["pcm"][:, :10], batch=b)
    _assert_streams_equal(bits, nb, z["bits"][:, :10], z["nbytes"][:, :10])
    bits, nb = _gpu_encode(torch_cuda, z["pcm"][:, 10:], batch=b)
    _assert_streams_equal(bits, nb, z["bits"][:, 10:], z["nbytes"][:, 10:])


def test_round_trip_on_gpu(torch_cuda):
    """encode -> erase descriptions -> decode, all on the GPU, equals the reference chain's PCM."""
    import solo_amd
    torch = torch_cuda
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    N, P, _ = z["pcm"].shape
    b = solo_amd.SoloBatch(N, encoder=True, decoder=True, slot_bytes=512)
    bits, nb, st = b.encode(torch.from_numpy(z["pcm"]).to(b.device))
    pcm, st2 = b.decode(bits, nb, torch.from_numpy(z["recv"]).to(b.device))
    torch.cuda.synchronize()
    assert int(st.abs().max()) == 0 and int(st2.abs().max()) == 0
    assert np.array_equal(pcm.cpu().numpy(), z["dec_loss"])


def _edge_streams(P):
    rng = np.random.default_rng(7)
    t = np.arange(640 * P) / 16000.0
    x = [np.zeros(640 * P), rng.integers(-32768, 32767, 640 * P), rng.integers(-40, 40, 640 * P),
         20000 * np.sin(2 * np.pi * 200 * t), 28000 * np.sin(2 * np.pi * (100 + 3000 * t) * t),
         30000 * np.sign(np.sin(2 * np.pi * 130 * t)),
         np.clip(80000 * np.sin(2 * np.pi * 300 * t) + rng.normal(0, 3000, t.size), -32768, 32767),
         np.full(640 * P, 12000), (np.arange(640 * P) % 97 == 0) * 30000]
    return np.stack([np.asarray(v).astype(np.int16).reshape(P, 640) for v in x])


@pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not present on this box")
def test_edge_signals_vs_compiled_reference(torch_cuda):
    """silence, full-scale noise, near-silence, tones, sweep, square, clipping, DC, impulse train"""
    P = 10
    pcm = _edge_streams(P)
    bits, nb = _gpu_encode(torch_cuda, pcm)
    for i in range(pcm.shape[0]):
        e = R.RefEncoder("fix")
        for p in range(P):
            pl, n0, n1 = e.encode(pcm[i, p])
            assert (int(nb[i, p, 0]), int(nb[i, p, 1])) == (n0, n1), (i, p)
            assert bits[i, p, :n0].tobytes() == pl, (i, p)


@pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not present on this box")
def test_many_streams_vs_compiled_reference(torch_cuda):
    """512 independent synthetic streams x 6 packets (config 2/3 shape, reduced count)."""
    N, P = 512, 6
    pcm = np.stack([R.synth_stream(5000 + i, P) for i in range(N)])
    bits, nb = _gpu_encode(torch_cuda, pcm)
    for i in range(N):
        e = R.RefEncoder("fix")
        for p in range(P):
            pl, n0, n1 = e.encode(pcm[i, p])
            assert (int(nb[i, p, 0]), int(nb[i, p, 1])) == (n0, n1), (i, p)
            assert bits[i, p, :n0].tobytes() == pl, (i, p)


def test_legacy_single_stream_api(torch_cuda):
    """AGR_Sate_Encoder_Init / _Encode / _Uninit driven exactly like test/enc_main.c does."""
    import ctypes as C
    import solo_amd
    lib = solo_amd.load_library()
    recs = T.parse_bit_container(open(T.GOLDEN + "/ch_f1.bit", "rb").read())
    pcm = np.fromfile(T.GOLDEN + "/Ch_f1_raw.pcm", np.int16)
    ctrl = solo_amd.default_enc_ctrl()
    h = lib.AGR_Sate_Encoder_Init(C.byref(ctrl))
    assert h
    buf = np.zeros(1024, np.uint8)
    nbv = np.zeros(6, np.int16)
    for p in range(40):
        x = np.ascontiguousarray(pcm[p * 640:(p + 1) * 640])
        n = lib.AGR_Sate_Encoder_Encode(h, x.ctypes.data, buf.ctypes.data, 1024, nbv.ctypes.data)
        assert (n, int(nbv[0]), int(nbv[1])) == (len(recs[p][0]), recs[p][1], recs[p][2]), p
        assert buf[:n].tobytes() == recs[p][0], p
    lib.AGR_Sate_Encoder_Uninit(h)


@pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not present on this box")
def test_joint_mode_1_round_trip_vs_compiled_reference(torch_cuda):
    """`-joint 1` (one 40 ms high-band frame per packet, 4 high-band bytes): encoder bitstreams and
    decoder PCM (with description loss) through the C ABI against the compiled reference."""
    import solo_amd
    torch = torch_cuda
    N, P = 12, 10
    pcm = np.stack([R.synth_stream(600 + i, P) for i in range(N)])
    b = solo_amd.SoloBatch(N, encoder=True, decoder=True, slot_bytes=512, joint=1)
    bits, nb, st = b.encode(torch.from_numpy(pcm).to(b.device))
    recv = T.bernoulli_recv(N, P, 0.3, 5)
    out, st2 = b.decode(bits, nb, torch.from_numpy(recv).to(b.device))
    torch.cuda.synchronize()
    assert int(st.abs().max()) == 0 and int(st2.abs().max()) == 0
    hb, hn, ho = bits.cpu().numpy(), nb.cpu().numpy(), out.cpu().numpy()
    for i in range(N):
        e, d = R.RefEncoder("fix", joint=1), R.RefDecoder("fix", joint=1)
        for p in range(P):
            pl, n0, n1 = e.encode(pcm[i, p])
            assert (int(hn[i, p, 0]), int(hn[i, p, 1])) == (n0, n1), (i, p)
            assert hb[i, p, :n0].tobytes() == pl, (i, p)
            m = int(recv[i, p])
            x, ret = d.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
            assert ret == 0 and np.array_equal(ho[i, p], x), (i, p, m)


@pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not present on this box")
def test_dtx_round_trip_vs_compiled_reference(torch_cuda):
    """SURVEY 8(f) rank 3 (second half): `-DTX 1` through the C ABI: empty packets where the reference sends none, identical bytes
    elsewhere, and the decoder treating empty packets as lost (test/dec_main.c:236-252)."""
    import solo_amd
    torch = torch_cuda
    N, P = 6, 40
    rng = np.random.default_rng(9)
    pcm = np.stack([R.synth_stream(800 + i, P) for i in range(N)])
    for i in range(N):
        a = 3 + i
        pcm[i, a:a + 24] = (rng.standard_normal((24, 640)) * 3).astype(np.int16)
    b = solo_amd.SoloBatch(N, encoder=True, decoder=True, slot_bytes=512, dtx=1)
    bits, nb, st = b.encode(torch.from_numpy(pcm).to(b.device))
    out, st2 = b.decode(bits, nb, None)
    torch.cuda.synchronize()
    assert int(st.abs().max()) == 0 and int(st2.abs().max()) == 0
    hb, hn, ho = bits.cpu().numpy(), nb.cpu().numpy(), out.cpu().numpy()
    assert int((hn[:, :, 0] == 0).sum()) >= 10 * N
    for i in range(N):
        e, d = R.RefEncoder("fix", dtx=1), R.RefDecoder("fix")
        for p in range(P):
            pl, n0, n1 = e.encode(pcm[i, p])
            assert (int(hn[i, p, 0]), int(hn[i, p, 1])) == (n0, n1), (i, p)
            assert hb[i, p, :n0].tobytes() == pl[:n0], (i, p)
            x, ret = d.decode(*((b"", 16, 0, 1) if n0 == 0 else (pl, n0, n1, 4)))
            assert ret == 0 and np.array_equal(ho[i, p], x), (i, p)


def test_async_join_pipelines_consecutive_calls(torch_cuda):
    """solo_batch_set_async_join: encode calls return without joining; two calls in flight write different output buffers and
    the consumer waits with solo_batch_wait_encode.  Same bits as the golden vectors, same PCM after decoding."""
    import solo_amd
    torch = torch_cuda
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    N, P, S = z["bits"].shape
    b = solo_amd.SoloBatch(N, encoder=True, decoder=True, slot_bytes=S)
    b.set_async_join(True)
    pcm = torch.from_numpy(z["pcm"]).to(b.device)
    cuts = [(0, 9), (9, 17), (17, 25)]
    outs, pcms = [], []
    prev = None
    ins = [pcm[:, a:e].contiguous() for a, e in cuts]              # inputs (and outputs) stay alive and untouched until wait_encode
    torch.cuda.synchronize()
    for x in ins:
        cur = b.encode(x)                                          # fresh output tensors per call
        if prev is not None:
            b.wait_encode(1)
            pcms.append(b.decode(prev[0], prev[1], None)[0])
        outs.append(cur)
        prev = cur
    b.wait_encode(0)
    pcms.append(b.decode(prev[0], prev[1], None)[0])
    torch.cuda.synchronize()
    bits = np.concatenate([o[0].cpu().numpy() for o in outs], axis=1)
    nb = np.concatenate([o[1].cpu().numpy() for o in outs], axis=1)
    _assert_streams_equal(bits, nb, z["bits"], z["nbytes"])
    assert np.array_equal(np.concatenate([p.cpu().numpy() for p in pcms], axis=1), z["dec_clean"])


def test_stream_groups_of_the_pipeline(torch_cuda, monkeypatch):
    """More streams than one launch group (default 4096, here forced to 16 through SOLO_ENC_GROUP): the pipeline walks the groups
    one after the other on offset pointers; 40 streams = 5 copies of the golden batch in groups of 16 / 16 / 8, two calls."""
    import solo_amd
    torch = torch_cuda
    monkeypatch.setenv("SOLO_ENC_GROUP", "16")
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    pcm = np.concatenate([z["pcm"]] * 5)
    N, P, S = pcm.shape[0], 12, z["bits"].shape[2]
    b = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=S)
    for half in range(2):
        x = torch.from_numpy(np.ascontiguousarray(pcm[:, half * P:(half + 1) * P])).to(b.device)
        bits, nb, st = b.encode(x)
        torch.cuda.synchronize()
        assert int(st.abs().max()) == 0
        bh, nh = bits.cpu().numpy(), nb.cpu().numpy()
        for i in range(N):
            assert np.array_equal(nh[i], z["nbytes"][i % 8, half * P:(half + 1) * P]), (half, i)
            for p in range(P):
                n0 = int(nh[i, p, 0])
                assert np.array_equal(bh[i, p, :n0], z["bits"][i % 8, half * P + p, :n0]), (half, i, p)


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["1", "0", "2"])
def test_payload_larger_than_its_slot_is_reported_not_written(torch_cuda, monkeypatch, order):
    """A payload that does not fit the caller's slot (AGR_Sate_Encoder_Encode's nBytesOut / buffer check, AGR_BWE_SDK_API.c:129 ->
    sx_enc_stage_c_out): both byte counts of the packet are 0, the stream's status carries the first error of the call, every other
    packet of the batch is unaffected -- under each launch order of the third stage (the assembly lives in a different kernel in each)."""
    import solo_amd
    torch = torch_cuda
    monkeypatch.setenv("SOLO_ENC_CORDER", order)
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    P, S = 12, 80                                              # golden payloads of these packets: 56 .. 114 bytes
    b = solo_amd.SoloBatch(8, encoder=True, decoder=False, slot_bytes=S)
    bits, nb, st = b.encode(torch.from_numpy(np.ascontiguousarray(z["pcm"][:, :P])).to(b.device))
    torch.cuda.synchronize()
    bh, nh, sh = bits.cpu().numpy(), nb.cpu().numpy(), st.cpu().numpy()
    too_big = z["nbytes"][:, :P, 0] > S
    assert too_big.any() and (~too_big).any()
    for i in range(8):
        assert int(sh[i]) == (-1 if too_big[i].any() else 0), (i, int(sh[i]))
        for p in range(P):
            if too_big[i, p]:
                assert nh[i, p, 0] == 0 and nh[i, p, 1] == 0, (i, p, nh[i, p])
            else:
                n0 = int(z["nbytes"][i, p, 0])
                assert np.array_equal(nh[i, p], z["nbytes"][i, p]), (i, p)
                assert np.array_equal(bh[i, p, :n0], z["bits"][i, p, :n0]), (i, p)
