"""the package's own mirror of the reference CLI harness (solo_amd/harness.py): `.bit` container, loss
simulator, and -- on a GPU -- the file-level encode / decode against the reference CLI's known-answer md5s."""
import hashlib
import os

import numpy as np
import pytest

import refcodec as R
import solo_testlib as T
from solo_amd import harness as H


def test_container_round_trip_and_loss_pattern():
    raw = open(os.path.join(T.GOLDEN, "ch_f1.bit"), "rb").read()
    recs = H.parse_bit_container(raw)
    assert H.write_bit_container(recs) == raw
    assert [(r[1], r[2]) for r in recs] == [(r[1], r[2]) for r in T.parse_bit_container(raw)]
    for loss in (0, 10, 30, 50, 100):
        assert H.cli_loss_pattern(len(recs), loss) == R.cli_loss_pattern(len(recs), loss)
    assert list(H.recv_mask([(0, 0), (0, 1), (1, 0), (1, 1)])) == [3, 1, 2, 0]
    with pytest.raises(ValueError):
        H.parse_bit_container(raw[:-3])


@pytest.mark.gpu
def test_file_level_known_answers():
    import torch
    assert torch.cuda.is_available()
    g = T.golden_json()
    recs = H.encode_pcm(T.load_ch_f1())
    assert hashlib.md5(H.write_bit_container(recs)).hexdigest() == g["ch_f1_bit_md5"]
    for loss in (0, 30):
        pcm = H.decode_records(recs, loss_perc=loss)
        assert T.md5(pcm) == g["ch_f1_dec_loss%d_md5" % loss]
