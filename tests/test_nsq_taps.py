"""The quantiser ALONE (row E6 of SURVEY.md section 8a) against the reference's SKP_Silk_NSQ_del_dec (SKP_Silk_NSQ_del_dec.c:925), call by
call: tests/golden/nsq_taps.npz holds the arguments and outputs of 2 x 200 calls recorded from the compiled reference
(tests/golden/make_nsq_taps.py, oracle/ref_taps.c).  The recorded arguments are fed to the quantiser -- its host emulation here, the
gfx950 kernel through solo_debug_nsq in the GPU test -- on a freshly initialised stream; pulses of both descriptions, the centre
excitation and the seed must equal the reference's for every one of the 160 samples of every call.  When the encoder's end-to-end
parity breaks, this says in seconds whether the quantiser is the stage that broke."""
import ctypes as C
import os

import numpy as np
import pytest

import solo_testlib as T

OUT_DT = np.dtype([("Seed", "<i4"), ("r", "<i4", (160,)), ("q", "i1", (2, 164))])
REF_DT = np.dtype([("Seed", "<i4"), ("q", "i1", (2, 160)), ("r", "<i4", (160,))])


def _check(out, ref, what):
    out, ref = out.view(OUT_DT).reshape(-1), ref.view(REF_DT).reshape(-1)
    assert np.array_equal(out["Seed"], ref["Seed"]), what
    bad = np.nonzero((out["q"][:, :, :160] != ref["q"]).any(axis=(1, 2)) | (out["r"] != ref["r"]).any(axis=1))[0]
    assert bad.size == 0, (what, "first differing call", int(bad[0]))


def test_emulated_quantiser_equals_the_reference_call_by_call():
    z = np.load(os.path.join(T.GOLDEN, "nsq_taps.npz"))
    lib = C.CDLL(os.path.join(T.ROOT, "tests", "emu", "libsolo_emu.so"))
    lib.emu_nsq_frames.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    assert lib.emu_sizeof_nsq_in() == z["nsq_in"].shape[2]
    for s in range(z["nsq_in"].shape[0]):
        n = z["nsq_in"].shape[1]
        inp = np.ascontiguousarray(z["nsq_in"][s])
        out = np.zeros((n, OUT_DT.itemsize), np.uint8)
        assert lib.emu_nsq_frames(inp.ctypes.data, n, out.ctypes.data) == OUT_DT.itemsize
        _check(out, np.ascontiguousarray(z["nsq_out"][s]), "stream %d" % s)


@pytest.mark.gpu
def test_gpu_quantiser_kernel_equals_the_reference_call_by_call():
    import solo_amd
    z = np.load(os.path.join(T.GOLDEN, "nsq_taps.npz"))
    lib = solo_amd.load_library()
    S, n = z["nsq_in"].shape[:2]
    reps = 5                                   # 2 recorded streams x 5 copies: more than one wavefront (four streams each), ragged
    inp = np.ascontiguousarray(np.tile(z["nsq_in"], (reps, 1, 1)))
    out = np.zeros((S * reps, n, OUT_DT.itemsize), np.uint8)
    assert lib.solo_debug_nsq(S * reps, n // 2, inp.ctypes.data, out.ctypes.data) == OUT_DT.itemsize
    for i in range(S * reps):
        _check(out[i], np.ascontiguousarray(z["nsq_out"][i % S]), "stream %d" % i)
