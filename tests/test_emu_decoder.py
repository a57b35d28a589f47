"""Kernel SOURCE (solo_amd/csrc/solo_dec.h) compiled for the host (tests/emu, 1 lane) against the
committed golden vectors -- runs without a GPU and without /root/reference."""
import numpy as np

import refcodec as R
import solo_testlib as T


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


def test_decoder_rejects_empty_payload():
    dec = T.EmuDecoder()
    x, ret = dec.decode(b"", 0, 0, 4)
    assert ret == -1      # AGR_BWE_SDK_API.c:266
