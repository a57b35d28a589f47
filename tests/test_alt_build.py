"""The kernel source under the ALTERNATE compilation mode: every stage function inlined (-DSX_INLINE_ALL) and the range decoder's
symbol search in the reference's own ++/-- form (-DSX_RC_REFERENCE_LOOP), built by __graft_entry__.build() into
build/libsolo_mi355x_alt.so.  Round 1 reported that full inlining "broke decoder parity" and that hipcc mis-compiled the reference
form of the search loop; the cause turned out to be the range decoder's registers not being kept across packets (a corrupted payload
that announces a third frame then decoded from uninitialised registers -- garbage that depended on how the code was laid out).  With
that fixed both compilation modes must give identical, reference-exact results: this test runs the decoder / encoder / corrupted-payload
parity tests through the alternate library in a child process (the library is chosen at import time by SOLO_LIB_OVERRIDE)."""
import os
import subprocess
import sys

import pytest

import solo_testlib as T

ALT = os.path.join(T.ROOT, "build", "libsolo_mi355x_alt.so")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(ALT), reason="build/libsolo_mi355x_alt.so not built (python -c 'import __graft_entry__ as g; g.build()')")
def test_parity_suite_through_the_alternate_build():
    env = dict(os.environ, SOLO_LIB_OVERRIDE=ALT)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                        os.path.join(T.ROOT, "tests", "test_gpu_decoder.py"), os.path.join(T.ROOT, "tests", "test_gpu_encoder.py"),
                        os.path.join(T.ROOT, "tests", "test_pinned_corners.py"), os.path.join(T.ROOT, "tests", "test_gpu_receiver.py")],
                       env=env, cwd=T.ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


@pytest.mark.gpu
def test_decoder_suite_through_the_single_kernel_path():
    """The batched decode call normally runs two kernels (history-free symbol extraction, one lane per description; then the decoder
    proper); SOLO_DEC_SPLIT=0 keeps the single kernel that reads its symbols itself.  Both must give the reference's output: the decoder
    tests (goldens, description loss, corrupted payloads, receiver) once more through the single-kernel path."""
    env = dict(os.environ, SOLO_DEC_SPLIT="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                        os.path.join(T.ROOT, "tests", "test_gpu_decoder.py"), os.path.join(T.ROOT, "tests", "test_pinned_corners.py"),
                        os.path.join(T.ROOT, "tests", "test_gpu_receiver.py")],
                       env=env, cwd=T.ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


@pytest.mark.gpu
def test_decoder_suite_with_small_decode_chunks():
    """The batched decode call works through its packets in chunks (symbol extraction, then the decoder proper, per chunk; by
    default a call of up to 64 packets is ONE chunk).  With SOLO_DEC_CHUNK=3 and a first chunk of 2 even short calls cross several
    chunk boundaries: same output required (stream state, description shadow and the serial fall-back for unusual packets all
    carry over the boundaries; the two record buffers alternate)."""
    env = dict(os.environ, SOLO_DEC_CHUNK="3", SOLO_DEC_FIRST_CHUNK="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                        os.path.join(T.ROOT, "tests", "test_gpu_decoder.py"), os.path.join(T.ROOT, "tests", "test_pinned_corners.py")],
                       env=env, cwd=T.ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


@pytest.mark.gpu
@pytest.mark.parametrize("knobs", [{"SOLO_ENC_CHUNK": "2"}, {"SOLO_ENC_CHUNK": "0"}, {"SOLO_ENC_GATE": "1"}, {"SOLO_ENC_GROUP": "4", "SOLO_ENC_CHUNK": "3"},
                                   {"SOLO_ENC_PERSIST": "1"}, {"SOLO_ENC_PERSIST": "1", "SOLO_ENC_FINAL_WAIT_US": "0"},
                                   {"SOLO_ENC_PERSIST": "1", "SOLO_ENC_FINAL_WAIT_US": "-1"}, {"SOLO_ENC_PERSIST": "1", "SOLO_ENC_GROUP": "8", "SOLO_ENC_GATE": "0"}],
                         ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_encoder_suite_under_the_pipeline_knobs(knobs):
    """The encoder's host-side schedule has run-time knobs that change how a call is cut into launches -- chunks of
    several packets, no pipeline at all, the residency gate, small launch groups, and the PERSISTENT schedule (two kernels per call that hand
    packets to each other through flags in HBM while they run; with its bounded final wait set to zero, or with every packet left to its
    second launch, or cut into several launch groups without the gate) -- but must never change a bit of the output: the encoder parity tests
    (goldens, packet-wise vs batched calls, joint mode, DTX, both rates, framesize 20) once more under each."""
    env = dict(os.environ, **knobs)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                        os.path.join(T.ROOT, "tests", "test_gpu_encoder.py"), os.path.join(T.ROOT, "tests", "test_framesize20.py"), os.path.join(T.ROOT, "tests", "test_wb.py")],
                       env=env, cwd=T.ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
