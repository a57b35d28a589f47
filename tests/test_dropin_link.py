"""Link-level proof of the drop-in boundary: the reference's OWN command-line mains
(JC1_SDK_SRC_ARM/test/enc_main.c:190-274, test/dec_main.c:188-392) are compiled UNCHANGED against <repo>/include
(`#include "AGR_JC1_SDK_API.h"` resolves to include/AGR_JC1_SDK_API.h) and linked with libsolo_mi355x.so instead of the
reference library, with -Wl,--no-undefined (recipe: `make -C oracle dropin`; outputs in oracle/_ref/, git-ignored, they travel to
the GPU box like the other prebuilt checkers; no reference source is copied).

  * container (no GPU, /root/reference present): the two binaries build and every AGR_Sate_* symbol they import is exported
    by the product library;
  * GPU box: the binaries run the reference CLI's known-answer procedure on the GPU and reproduce its md5s."""
import hashlib
import os
import subprocess

import pytest

import solo_testlib as T

REF_MAIN = "/root/reference/JC1_SDK_SRC_ARM/test/enc_main.c"
ENC = os.path.join(T.ROOT, "oracle", "_ref", "JC1Encoder_solo")
DEC = os.path.join(T.ROOT, "oracle", "_ref", "JC1Decoder_solo")
LIB = os.path.join(T.ROOT, "solo_amd", "libsolo_mi355x.so")


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference sources are only present in the build container")
def test_reference_mains_compile_unchanged_and_link_against_the_product():
    for f in (ENC, DEC):
        if os.path.exists(f):
            os.remove(f)
    subprocess.check_call(["make", "-s", "-C", os.path.join(T.ROOT, "oracle"), "dropin"])
    assert os.path.exists(ENC) and os.path.exists(DEC)
    exported = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    exported = {l.split()[-1] for l in exported.splitlines() if l.strip()}
    for binary, want in ((ENC, {"AGR_Sate_Encoder_Init", "AGR_Sate_Encoder_Encode", "AGR_Sate_Encoder_Uninit"}),
                         (DEC, {"AGR_Sate_Decoder_Init", "AGR_Sate_Decoder_Decode", "AGR_Sate_Decoder_Uninit"})):
        und = subprocess.check_output(["nm", "-D", "--undefined-only", binary], text=True)
        und = {l.split()[-1].split("@")[0] for l in und.splitlines() if l.strip()}
        assert want <= und, (binary, want - und)                  # the mains really call the six entry points ...
        assert want <= exported                                    # ... and the product exports them
        needed = subprocess.check_output(["readelf", "-d", binary], text=True)
        assert "libsolo_mi355x.so" in needed and "JC1Codec" not in needed


def _md5_file(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(ENC) and os.path.exists(DEC)), reason="oracle/_ref/JC1*_solo not built (make -C oracle dropin)")
def test_reference_cli_mains_on_the_gpu_reproduce_the_known_answers(tmp_path):
    """JC1Encoder / JC1Decoder of the reference, linked with this library: bitstream, decoded and `-loss 30` md5s of SURVEY 8(c)."""
    g = T.golden_json()
    bit, dec, dec30 = str(tmp_path / "out.bit"), str(tmp_path / "dec.pcm"), str(tmp_path / "dec30.pcm")
    pcm = os.path.join(T.GOLDEN, "Ch_f1_raw.pcm")
    subprocess.run([ENC, pcm, bit, "-mode", "2", "-Fs_API", "16000", "-rate", "13600"], check=True, stdout=subprocess.DEVNULL, timeout=600)
    assert _md5_file(bit) == g["ch_f1_bit_md5"]
    subprocess.run([DEC, bit, dec, "-Fs_API", "16000"], check=True, stdout=subprocess.DEVNULL, timeout=600)
    assert _md5_file(dec) == g["ch_f1_dec_loss0_md5"]
    subprocess.run([DEC, bit, dec30, "-Fs_API", "16000", "-loss", "30"], check=True, stdout=subprocess.DEVNULL, timeout=600)
    assert _md5_file(dec30) == g["ch_f1_dec_loss30_md5"]
