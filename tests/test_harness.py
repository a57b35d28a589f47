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


@pytest.mark.gpu
def test_file_level_known_answers_32k():
    """`-Fs_API 32000`: .bit container and decoded PCM (0 % and 30 % CLI loss) of a 32 kHz stream against the md5s obtained with
    the compiled reference (tests/golden/make_golden.py)."""
    import torch
    assert torch.cuda.is_available()
    g = T.golden_json()
    z = np.load(T.GOLDEN + "/wb4x20.npz")
    recs = H.encode_pcm(z["pcm"][0].reshape(-1), rate=24000, samplerate=32000)
    bit = H.write_bit_container(recs)
    assert T.md5(bit) == g["wb_kat_bit_md5"]
    for loss in (0, 30):
        pcm = H.decode_records(H.parse_bit_container(bit), loss_perc=loss, samplerate=32000)
        assert pcm.size == 20 * 1280 and T.md5(pcm) == g["wb_kat_dec_loss%d_md5" % loss]


REF_ENC = os.path.join(T.ROOT, "oracle", "_ref", "JC1Encoder_ref")
REF_DEC = os.path.join(T.ROOT, "oracle", "_ref", "JC1Decoder_ref")


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(REF_ENC) and os.path.exists(REF_DEC)), reason="oracle/_ref/JC1*_ref not built")
def test_dtx_file_against_the_reference_cli(tmp_path):
    """`-DTX 1` files: the reference CLI (test/enc_main.c / dec_main.c linked with the compiled reference) against this package's
    harness on the GPU.  Empty packets make the reference decoder return -1 untouched and the CLI repeat its previous output
    buffer (AGR_BWE_SDK_API.c:266, dec_main.c:365-381); the harness must produce the same file, with and without simulated loss."""
    import subprocess
    import torch
    assert torch.cuda.is_available()
    P = 60
    rng = np.random.default_rng(12)
    pcm = R.synth_stream(811, P).copy()
    pcm[5:31] = (rng.standard_normal((26, 640)) * 3).astype(np.int16)
    pcm[40:52] = (rng.standard_normal((12, 640)) * 2).astype(np.int16)
    src, bit = str(tmp_path / "in.pcm"), str(tmp_path / "ref.bit")
    pcm.tofile(src)
    subprocess.run([REF_ENC, src, bit, "-Fs_API", "16000", "-rate", "13600", "-DTX", "1"], check=True, stdout=subprocess.DEVNULL, timeout=300)
    raw = open(bit, "rb").read()
    recs = H.encode_pcm(pcm.reshape(-1), dtx=1)
    assert H.write_bit_container(recs) == raw
    assert sum(1 for r in recs if r[1] == 0) >= 12 and recs[0][1] > 0
    for loss in (0, 20):
        out = str(tmp_path / ("ref%d.pcm" % loss))
        subprocess.run([REF_DEC, bit, out, "-Fs_API", "16000", "-loss", str(loss)], check=True, stdout=subprocess.DEVNULL, timeout=300)
        want = np.fromfile(out, np.int16)
        got = H.decode_records(H.parse_bit_container(raw), loss_perc=loss)
        assert got.size == want.size and np.array_equal(got, want), loss


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(REF_ENC) and os.path.exists(REF_DEC)), reason="oracle/_ref/JC1*_ref not built")
def test_dec_mode_against_the_reference_cli(tmp_path):
    """`-dec_mode 1|2` of the decoder CLI (test/dec_main.c:123-145,341-361): the whole file decoded from ONE description.  The
    reference CLI on the reference's speech sample (and a DTX file, whose empty packets repeat the previous buffer) against the
    harness on the GPU."""
    import subprocess
    import torch
    assert torch.cuda.is_available()
    with pytest.raises(ValueError):
        H.decode_records([(b"\0" * 8, 8, 8)], loss_perc=10, dec_mode=1)
    raw = open(os.path.join(T.GOLDEN, "ch_f1.bit"), "rb").read()
    bit = str(tmp_path / "in.bit")
    open(bit, "wb").write(raw)
    outs = {}
    for mode in (1, 2):
        out = str(tmp_path / ("ref_m%d.pcm" % mode))
        subprocess.run([REF_DEC, bit, out, "-Fs_API", "16000", "-dec_mode", str(mode)], check=True, stdout=subprocess.DEVNULL, timeout=300)
        want = np.fromfile(out, np.int16)
        got = H.decode_records(H.parse_bit_container(raw), dec_mode=mode)
        assert got.size == want.size and np.array_equal(got, want), mode
        outs[mode] = want
    assert not np.array_equal(outs[1], outs[2])              # the two descriptions do decode differently
    # a DTX file (empty records)
    P = 40
    rng = np.random.default_rng(5)
    pcm = R.synth_stream(1311, P).copy()
    pcm[8:30] = (rng.standard_normal((22, 640)) * 3).astype(np.int16)
    src, dbit = str(tmp_path / "dtx.pcm"), str(tmp_path / "dtx.bit")
    pcm.tofile(src)
    subprocess.run([REF_ENC, src, dbit, "-Fs_API", "16000", "-rate", "13600", "-DTX", "1"], check=True, stdout=subprocess.DEVNULL, timeout=300)
    draw = open(dbit, "rb").read()
    assert sum(1 for r in H.parse_bit_container(draw) if r[1] == 0) >= 8
    for path, fs in ((dbit, 16000),):            # (the reference CLIs refuse -Fs_API 32000: "only support wideband")
        data = open(path, "rb").read()
        for mode in (1, 2):
            out = str(tmp_path / "o.pcm")
            subprocess.run([REF_DEC, path, out, "-Fs_API", str(fs), "-dec_mode", str(mode)], check=True, stdout=subprocess.DEVNULL, timeout=300)
            want = np.fromfile(out, np.int16)
            got = H.decode_records(H.parse_bit_container(data), dec_mode=mode, samplerate=fs)
            assert got.size == want.size and np.array_equal(got, want), (path, mode)
