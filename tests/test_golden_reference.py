"""The compiled reference (oracle/_ref) against the known-answer md5s recorded in SURVEY.md 8(c) /
BASELINE.md 2 -- pins the oracle itself.  Skipped where oracle/_ref is not built."""
import numpy as np
import pytest

import refcodec as R
import solo_testlib as T

pytestmark = pytest.mark.skipif(not R.have_ref("fix"), reason="oracle/_ref not built (needs /root/reference)")

KAT = {  # SURVEY.md section 8(c), ARM (fixed-point) tree, stable across gcc -O0/-O2/-O3
    "bit": "4a38792082e2503d1edc5031b29ea405",
    "dec0": "149ad62fe193f19d6c80aacc7336ebce",
    "dec30": "2c54efdbd9530af9c0c4bb64d3fa2270",
    "pcm": "7070b51963afb0a419c48f8cecfaddc9",
}


def test_reference_known_answers():
    pcm = T.load_ch_f1()
    assert T.md5(pcm) == KAT["pcm"]
    npk = pcm.size // 640
    enc = R.RefEncoder("fix")
    recs = [enc.encode(pcm[p * 640:(p + 1) * 640]) for p in range(npk)]
    assert npk == 191
    bit = T.write_bit_container(recs)
    assert len(bit) == 16600 and T.md5(np.frombuffer(bit, np.uint8)) == KAT["bit"]
    # committed fixture is the same stream
    assert open(T.GOLDEN + "/ch_f1.bit", "rb").read() == bit
    for loss, key in ((0, "dec0"), (30, "dec30")):
        dec = R.RefDecoder("fix")
        pat = R.cli_loss_pattern(npk, loss)
        out = [dec.decode(*R.map_loss(pl, n0, n1, *pat[p]))[0] for p, (pl, n0, n1) in enumerate(recs)]
        assert T.md5(np.concatenate(out)) == KAT[key]


def test_golden_json_matches_kat():
    g = T.golden_json()
    assert g["ch_f1_bit_md5"] == KAT["bit"]
    assert g["ch_f1_dec_loss0_md5"] == KAT["dec0"]
    assert g["ch_f1_dec_loss30_md5"] == KAT["dec30"]
