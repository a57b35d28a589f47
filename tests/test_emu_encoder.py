"""CPU checks of the ENCODER kernel source (solo_amd/csrc/solo_enc*.h) compiled for the host (tests/emu):
bit-exact against the committed golden bitstreams and, where oracle/_ref exists, the compiled reference.
The GPU parity tests proper are tests/test_gpu_encoder.py."""
import numpy as np
import pytest

import refcodec as R
import solo_testlib as T


def test_ch_f1_bitstream_md5():
    g = T.golden_json()
    pcm = np.fromfile(T.GOLDEN + "/Ch_f1_raw.pcm", np.int16)
    e = T.EmuEncoder()
    recs = [e.encode(pcm[p * 640:(p + 1) * 640]) for p in range(len(pcm) // 640)]
    assert T.md5(T.write_bit_container(recs)) == g["ch_f1_bit_md5"]


def test_synthetic_goldens():
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    N, P, _ = z["pcm"].shape
    for i in range(N):
        e = T.EmuEncoder()
        for p in range(P):
            pl, n0, n1 = e.encode(z["pcm"][i, p])
            assert (n0, n1) == tuple(int(v) for v in z["nbytes"][i, p]), (i, p)
            assert pl == z["bits"][i, p, :n0].tobytes(), (i, p)


@pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not built")
def test_edge_signals_vs_compiled_reference():
    rng = np.random.default_rng(11)
    P = 8
    t = np.arange(640 * P) / 16000.0
    sigs = [np.zeros(640 * P), rng.integers(-32768, 32767, 640 * P), 25000 * np.sin(2 * np.pi * 180 * t),
            30000 * np.sign(np.sin(2 * np.pi * 110 * t)), np.full(640 * P, -20000), (np.arange(640 * P) % 131 == 0) * 32767]
    for k, s in enumerate(sigs):
        x = np.asarray(s).astype(np.int16)
        e, r = T.EmuEncoder(), R.RefEncoder("fix")
        for p in range(P):
            assert e.encode(x[p * 640:(p + 1) * 640]) == r.encode(x[p * 640:(p + 1) * 640]), (k, p)


def test_edge_family_goldens():
    """28 un-speech-like streams (two of every family of solo_amd.synth.edge_stream: silence with stray LSBs, full-scale noise,
    square waves, DC, impulses, 90 dB ramps, clipping, bursts, high-band-only tones, Nyquist pattern, sub-audio sines, random walks):
    the committed reference bitstreams (tests/golden/make_edge_golden.py), byte for byte."""
    from solo_amd.synth import edge_stream
    z = np.load(T.GOLDEN + "/edge28x12.npz")
    N, P, _ = z["pcm"].shape
    for i in range(N):
        assert np.array_equal(edge_stream(i, P), z["pcm"][i]), i           # (the generator itself is pinned by the fixture)
        e = T.EmuEncoder()
        for p in range(P):
            pl, n0, n1 = e.encode(z["pcm"][i, p])
            assert (n0, n1) == tuple(int(v) for v in z["nbytes"][i, p]), (i, p)
            assert pl == z["bits"][i, p, :n0].tobytes(), (i, p)


def test_round_trip_through_emu_decoder():
    """encode -> decode with the emulated kernels reproduces the committed reference PCM"""
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    for i in range(2):
        e, d = T.EmuEncoder(), T.EmuDecoder()
        for p in range(10):
            pl, n0, n1 = e.encode(z["pcm"][i, p])
            x, ret = d.decode(*R.map_loss(pl, n0, n1, False, False))
            assert ret == 0 and np.array_equal(x, z["dec_clean"][i, p])


@pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not present")
def test_joint_mode_1_encoder_and_decoder_vs_reference():
    """`-joint 1` of the reference CLI (one 40 ms high-band frame per packet, 4 high-band bytes, SILK rate = target - 800):
    kernel source (host emulation) against the compiled reference, encoder and decoder with description loss."""
    P = 12
    for seed in (500, 501, 502):
        pcm = R.synth_stream(seed, P)
        er, ee = R.RefEncoder("fix", joint=1), T.EmuEncoder(13600, 2)          # bit 1 of the emulation's flag word = joint
        recs = []
        for p in range(P):
            a, b = er.encode(pcm[p]), ee.encode(pcm[p])
            assert a == b, (seed, p, a[1:], b[1:])
            recs.append(a)
        recv = T.bernoulli_recv(1, P, 0.3, seed)[0]
        dr, de = R.RefDecoder("fix", joint=1), T.EmuDecoder(2)
        for p, (pl, n0, n1) in enumerate(recs):
            m = int(recv[p])
            args = R.map_loss(pl, n0, n1, not (m & 1), not (m & 2))
            x, r1 = dr.decode(*args)
            y, r2 = de.decode(*args)
            assert r1 == r2 == 0 and np.array_equal(x, y), (seed, p, m)


def _dtx_signal(seed, P):
    rng = np.random.default_rng(seed)
    pcm = R.synth_stream(seed, P).copy()
    pcm[4:30] = (rng.standard_normal((26, 640)) * 3).astype(np.int16)      # a second of near-silence: DTX engages, and times out once
    return pcm


@pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not present")
@pytest.mark.parametrize("joint", [0, 1])
def test_dtx_vs_reference(joint):
    """`-DTX 1` (SKP_Silk_encode_frame_FIX.c:155-171, SKP_Silk_enc_API.c:260-265): packets are analysed and quantised but not sent
    while the encoder is in DTX (nBytesOut = {0, 0}; Encode still returns the high-band bytes); empty packets decode as lost."""
    P = 40
    pcm = _dtx_signal(700, P)
    er, ee = R.RefEncoder("fix", dtx=1, joint=joint), T.EmuEncoder(13600, 4 | (2 if joint else 0))
    recs = []
    for p in range(P):
        a, b = er.encode(pcm[p]), ee.encode(pcm[p])
        assert a == b, (joint, p, a[1:], b[1:])
        recs.append(a)
    sizes = [r[1] for r in recs]
    assert sizes.count(0) >= 15 and sizes[-1] > 0 and any(s > 0 for s in sizes[8:30])     # DTX engaged, timed out once, ended
    dr, de = R.RefDecoder("fix", joint=joint), T.EmuDecoder(2 if joint else 0)
    for p, (pl, n0, n1) in enumerate(recs):
        args = (b"", 16, 0, 1) if n0 == 0 else (pl, n0, n1, 4)
        x, r1 = dr.decode(*args)
        y, r2 = de.decode(*args)
        assert r1 == r2 == 0 and np.array_equal(x, y), (joint, p)
