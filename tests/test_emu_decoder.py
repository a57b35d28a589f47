"""Kernel SOURCE (solo_amd/csrc/solo_dec.h) compiled for the host (tests/emu, 1 lane) against the
committed golden vectors -- runs without a GPU and without /root/reference."""
import numpy as np
import pytest

import refcodec as R
import solo_testlib as T


@pytest.fixture(autouse=True, params=[0, 1], ids=["serial-parse", "extract-then-decode"])
def _decoder_path(request):
    """Every case runs through the decoder that reads its symbols itself and through the batch path's two steps: history-free
    symbol extraction per description (solo_dec.h sx_extract_desc, interval-form range decoder) -> records -> decoder."""
    old = T.EmuDecoder.SPLIT
    T.EmuDecoder.SPLIT = request.param
    yield
    T.EmuDecoder.SPLIT = old


def _decode_all(recs, pattern):
    dec = T.EmuDecoder()
    out = []
    for p, (pl, n0, n1) in enumerate(recs):
        x, ret = dec.decode(*R.map_loss(pl, n0, n1, *pattern[p]))
        assert ret == 0
        out.append(x)
    return np.concatenate(out)


def test_ch_f1_decode_clean_and_cli_loss():
    g = T.golden_json()
    recs = T.parse_bit_container(open(T.GOLDEN + "/ch_f1.bit", "rb").read())
    assert len(recs) == g["ch_f1_packets"]
    for loss in (0, 30):
        pat = R.cli_loss_pattern(len(recs), loss)
        assert T.md5(_decode_all(recs, pat)) == g["ch_f1_dec_loss%d_md5" % loss]


def test_synthetic_streams_bernoulli_loss():
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    bits, nb, recv = z["bits"], z["nbytes"], z["recv"]
    N, P = recv.shape
    for mask, key in ((None, "dec_clean"), (recv, "dec_loss")):
        for s in range(N):
            dec = T.EmuDecoder()
            for p in range(P):
                n0, n1 = int(nb[s, p, 0]), int(nb[s, p, 1])
                m = 3 if mask is None else int(mask[s, p])
                x, ret = dec.decode(*R.map_loss(bits[s, p, :n0].tobytes(), n0, n1, not (m & 1), not (m & 2)))
                assert ret == 0
                assert np.array_equal(x, z[key][s, p]), (key, s, p)


def test_edge_family_goldens():
    """The edge-signal fixture (tests/golden/edge28x12.npz) through the decoder with its description-loss mask: CRC-32 of every
    packet the compiled reference decoded."""
    import zlib
    z = np.load(T.GOLDEN + "/edge28x12.npz")
    N, P, _ = z["pcm"].shape
    for i in range(N):
        d = T.EmuDecoder()
        for p in range(P):
            n0, n1, m = int(z["nbytes"][i, p, 0]), int(z["nbytes"][i, p, 1]), int(z["recv"][i, p])
            pl = z["bits"][i, p, :n0].tobytes()
            x, ret = d.decode(pl, n0, n1, 1) if m == 0 else d.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
            assert ret == 0 and zlib.crc32(x.tobytes()) == int(z["dec_crc"][i, p]), (i, p, m)


def test_old_buffers_of_both_slots_after_a_record_path_packet():
    """tests/golden/nb_stale_coder.npz (made by make_stale_coder_golden.py from fuzz trial 299538): a corrupted packet that announces
    more frames than it carries makes a later two-description packet decode on in the old buffers of BOTH description slots; the
    second slot's last description went through the batch path's records, which do not keep the coder registers -- the decoder
    rebuilds them from the shadow buffer.  Thirteen packets decode like the reference, the fourteenth is rejected with its code."""
    z = np.load(T.GOLDEN + "/nb_stale_coder.npz")
    d = T.EmuDecoder()
    for p in range(z["recv"].shape[1]):
        n0, n1, m = int(z["nbytes"][0, p, 0]), int(z["nbytes"][0, p, 1]), int(z["recv"][0, p])
        pl = z["bits"][0, p, :n0].tobytes()
        x, ret = d.decode(pl, n0, n1, 1) if m == 0 else d.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
        assert ret == int(z["ret"][0, p]), (p, ret)
        if ret == 0:
            assert np.array_equal(x, z["dec"][0, p]), p


def test_cold_start_leading_packets_lost():
    """Packets lost before anything was decoded: the reference decoder is still at its initial 24 kHz (create_init_destroy.c:41),
    emits zeros through the 24 -> 8 kHz resampler and fades the first decoded frame in with the 480-sample slope (decode_frame.c:303,
    PLC.c:405); the concealment LCG has advanced 480 steps per lost frame.  First arriving packet complete / MD1 only / MD2 only."""
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    c = np.load(T.GOLDEN + "/synth8x25_cold.npz")
    bits, nb, recv = z["bits"], z["nbytes"], c["recv"]
    N, P = recv.shape
    assert (recv[:, 0] == 0).all() and T.md5(c["dec"]) == T.golden_json()["synth_dec_cold_md5"]
    for s in range(N):
        dec = T.EmuDecoder()
        for p in range(P):
            n0, n1 = int(nb[s, p, 0]), int(nb[s, p, 1])
            m = int(recv[s, p])
            x, ret = dec.decode(*R.map_loss(bits[s, p, :n0].tobytes(), n0, n1, not (m & 1), not (m & 2)))
            assert ret == 0
            assert np.array_equal(x, c["dec"][s, p]), (s, p)
            if m == 0 and (recv[s, :p] == 0).all():
                assert not x.any()


def test_cold_start_joint_mode_vs_reference():
    """Same corner with one 40 ms high-band frame per packet (joint_mode 1), against the compiled reference directly."""
    import pytest
    if not R.have_ref("fix"):
        pytest.skip("oracle/_ref not built")
    P = 12
    pcm = R.synth_stream(4242, P)
    enc = R.RefEncoder("fix", joint=1)
    recs = [enc.encode(pcm[p]) for p in range(P)]
    for k, first in ((1, 3), (3, 1), (2, 2)):
        dr, de = R.RefDecoder("fix", joint=1), T.EmuDecoder(2)
        for p, (pl, n0, n1) in enumerate(recs):
            m = 0 if (p < k or p == 8) else (first if p == k else 3)
            a = R.map_loss(pl, n0, n1, not (m & 1), not (m & 2))
            x, r1 = dr.decode(*a)
            y, r2 = de.decode(*a)
            assert r1 == r2 == 0 and np.array_equal(x, y), (k, first, p)


def test_corrupted_payloads_vs_reference():
    """Bit errors in the payload: a corrupted packet that the range decoder still accepts must decode to the same (garbage) PCM and
    leave the same state as in the reference, and a rejected one must return the same negative code.  After a rejected packet the
    reference's own state is not defined by its inputs (AGR_Sate_decode_process copies an uninitialised stack buffer into the
    decoder's output history, SKP_Silk_decode_frame.c:358), so a stream is compared up to and including its first rejection.
    One documented difference: a corrupted rate index can make the payload claim another internal rate (12 / 16 / 24 kHz), which
    the reference then decodes and resamples; this build has one internal rate per handle and rejects it (payload error)."""
    import pytest
    if not R.have_ref("fix"):
        pytest.skip("oracle/_ref not built")
    z = np.load(T.GOLDEN + "/synth8x25.npz")
    bits, nb = z["bits"], z["nbytes"]
    rng = np.random.default_rng(5)
    n_rejected = n_garbage = n_other_rate = 0
    for trial in range(240):
        s = trial % 8
        dr, de = R.RefDecoder(), T.EmuDecoder()
        for p in range(25):
            n0, n1 = int(nb[s, p, 0]), int(nb[s, p, 1])
            pl = bytearray(bits[s, p, :n0].tobytes())
            hit = rng.random() < 0.25
            if hit:
                for _ in range(rng.integers(1, 4)):
                    pl[rng.integers(0, n0)] = rng.integers(0, 256)
            mode = rng.integers(0, 4)
            a = R.map_loss(bytes(pl), n0, n1, mode == 1, mode == 2)
            x, r1 = dr.decode(*a)
            y, r2 = de.decode(*a)
            if r1 == 0 and r2 == -12 and hit:
                n_other_rate += 1
                break
            assert r1 == r2, (trial, p, r1, r2)
            if r1 < 0:
                n_rejected += 1
                break
            assert np.array_equal(x, y), (trial, p)
            n_garbage += int(hit)
    assert n_rejected >= 50 and n_garbage >= 200 and n_other_rate <= 20, (n_rejected, n_garbage, n_other_rate)


def test_decoder_rejects_empty_payload():
    dec = T.EmuDecoder()
    x, ret = dec.decode(b"", 0, 0, 4)
    assert ret == -1      # AGR_BWE_SDK_API.c:266
