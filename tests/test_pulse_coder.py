"""The pulse coder of the range-coding stage on its own (sx_encode_pulses, solo_amd/csrc/solo_enc.h -- host emulation of the kernel source)
against the compiled reference's SKP_Silk_encode_pulses (SKP_Silk_encode_pulses.c:55) on crafted frames of pulses: the cases the speech-like
and edge signals of the other suites reach rarely -- blocks that need one to eight down-shifts before their pairwise sums fit, |pulse| up to
128, empty frames, single pulses, sign patterns across the mask's word boundaries.  (Round 5 re-cut the coder: no copy of the pulses, unshifted
magnitudes + a sign mask in its work row, shifts applied where the reference uses its shifted copy.)"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import refcodec as R
import solo_testlib as T

pytestmark = pytest.mark.skipif(not R.have_ref("fix"), reason="compiled reference (oracle/_ref) not built")


class RefRC(C.Structure):                                   # SKP_Silk_range_coder_state, SKP_Silk_structs.h:85
    _fields_ = [("bufferLength", C.c_int32), ("bufferIx", C.c_int32), ("base_Q32", C.c_uint32), ("range_Q16", C.c_uint32),
                ("error", C.c_int32), ("buffer", C.c_uint8 * 1024)]


def ref_bytes(lib, sigtype, qot, q):
    rc = RefRC()
    lib.SKP_Silk_range_enc_init(C.byref(rc))
    qa = np.ascontiguousarray(q, np.int8)
    lib.SKP_Silk_encode_pulses(C.byref(rc), C.c_int(sigtype), C.c_int(qot), qa.ctypes.data_as(C.c_void_p), C.c_int(len(qa)))
    nb = C.c_int32(0)
    lib.SKP_Silk_range_coder_get_length(C.byref(rc), C.byref(nb))
    lib.SKP_Silk_range_enc_wrap_up(C.byref(rc))
    return bytes(rc.buffer[:nb.value]), rc.error


def emu_bytes(emu, sigtype, qot, q):
    qa = np.ascontiguousarray(q, np.int8)
    out = np.zeros(1100, np.uint8)
    err = C.c_int(0)
    nb = emu.emu_encode_pulses(sigtype, qot, qa.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.byref(err))
    return bytes(out[:max(nb, 0)]), err.value


def frames(n):
    rng = np.random.default_rng(20250928)
    yield "empty", np.zeros(n, np.int8)
    for pos in (0, 15, 16, 31, 32, 33, 63, 64, n - 1):
        for v in (1, -1, 18, -19, 127, -128):
            q = np.zeros(n, np.int8); q[pos] = v
            yield "single %d at %d" % (v, pos), q
    yield "all +1", np.ones(n, np.int8)
    yield "all -1", -np.ones(n, np.int8)
    yield "alternating +-2", (2 * (1 - 2 * (np.arange(n) & 1))).astype(np.int8)
    yield "all 127", np.full(n, 127, np.int8)
    yield "all -128", np.full(n, -128, np.int8)
    for scale in (0.4, 1.0, 2.5, 6.0, 15.0, 40.0, 100.0):                  # Laplacian pulses of growing size: zero to eight shifts per block
        for rep in range(6):
            q = np.clip(np.round(rng.laplace(0.0, scale, n)), -128, 127).astype(np.int8)
            yield "laplace %.1f #%d" % (scale, rep), q
    for rep in range(12):                                                 # loud blocks between silent ones, sparse sign patterns
        q = np.zeros(n, np.int8)
        for b in rng.choice(n // 16, size=rng.integers(1, n // 16), replace=False):
            q[16 * b:16 * b + 16] = np.clip(np.round(rng.laplace(0.0, rng.choice([1.0, 8.0, 60.0]), 16)), -128, 127)
        yield "blocks #%d" % rep, q


def test_pulse_coder_against_the_reference_on_crafted_frames():
    lib = R.load_ref("fix")
    emu = T.load_emu()
    emu.emu_encode_pulses.restype = C.c_int
    emu.emu_encode_pulses.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    n = emu.emu_frame_samples()
    checked = shifted = 0
    for name, q in frames(n):
        for sigtype in (0, 1):
            for qot in (0, 1):
                rb, re_ = ref_bytes(lib, sigtype, qot, q)
                eb, ee = emu_bytes(emu, sigtype, qot, q)
                assert (re_ != 0) == (ee != 0), (name, sigtype, qot, re_, ee)
                if re_ == 0:
                    assert eb == rb, (name, sigtype, qot, len(rb), len(eb))
                checked += 1
        a = np.abs(q.astype(np.int32)).reshape(-1, 16)
        shifted += int((a.sum(axis=1) > 18).sum())
    assert checked > 400 and shifted > 200          # (the crafted set does reach the down-shift path, many times)
