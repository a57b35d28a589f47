"""the fixed-point vocabulary (solo_amd/csrc/solo_fix.h, solo_common.h) operand by operand against the
reference's OWN macros and inlines (SKP_Silk_macros.h:33-122, SKP_Silk_SigProc_FIX.h, SKP_Silk_Inlines.h:43-220, lin2log / log2lin /
sigm_Q15 / sum_sqr_shift, libBWE/AGR_BWE_fixed_generic.h), made callable by oracle/ref_l0_shim*.c (which only include the
reference's headers in place; built into oracle/_ref/libsolo_ref_l0.so by `make -C oracle ref`).

CPU: the host emulation of the kernel source (tests/emu).  GPU (-m gpu): the same functions as compiled by hipcc for gfx950,
through the product library's conformance probe (solo_debug_l0 / solo_debug_sum_sqr_shift).  Bit-exact, edge + random operands."""
import ctypes as C
import os

import numpy as np
import pytest

import refcodec as R
import solo_testlib as T

L0_LIB = os.path.join(T.ROOT, "oracle", "_ref", "libsolo_ref_l0.so")
need_ref = pytest.mark.skipif(not os.path.exists(L0_LIB), reason="oracle/_ref/libsolo_ref_l0.so not built")

I32 = np.iinfo(np.int32)
EDGE = np.array([0, 1, -1, 2, -2, 3, 127, 128, 255, 256, 1023, 1024, 32767, 32768, -32767, -32768, -32769, 65535, 65536, -65536,
                 46340, 46341, (1 << 20), -(1 << 20), (1 << 24) - 1, (1 << 29), (1 << 30), (1 << 30) + 1, I32.max, I32.max - 1,
                 I32.min, I32.min + 1, 0x55555555, -0x55555556, 0x0000FFFF, 0x7FFF0000, -0x7FFF0000], np.int64)


def _operands(rng, n_rand, bmap=None, cmap=None, amap=None):
    """all edge x edge pairs (c cycling through the edges) + random triples; a/b/c maps restrict a domain"""
    ea, eb = np.meshgrid(EDGE, EDGE, indexing="ij")
    a = np.concatenate([ea.ravel(), rng.integers(I32.min, I32.max, n_rand, endpoint=True),
                        rng.integers(-70000, 70000, n_rand)])
    b = np.concatenate([eb.ravel(), rng.integers(I32.min, I32.max, n_rand, endpoint=True),
                        rng.integers(-70000, 70000, n_rand)])
    c = np.concatenate([np.resize(EDGE, ea.size), rng.integers(I32.min, I32.max, 2 * n_rand, endpoint=True)])
    a = a if amap is None else amap(a)
    b = b if bmap is None else bmap(b)
    c = c if cmap is None else cmap(c)
    return [np.ascontiguousarray(v.astype(np.int64).astype(np.int32)) for v in (a, b, c)]


def _nonzero(v):
    """divisor domain of DIV32_varQ / INVERSE32_varQ: non-zero and not INT32_MIN (|INT32_MIN| has no headroom: the reference shifts
    by -1 there, which is undefined in C)"""
    v = v.copy()
    v[v == 0] = 7
    v[v == I32.min] = I32.min + 1
    return v


def _no_min(v):
    v = v.copy()
    v[v == I32.min] = I32.min + 1
    return v


# op -> (name, operand restrictions): the domains on which the reference itself is defined
OPS = {
    0: ("SMULWB", {}), 1: ("SMULWT", {}), 2: ("SMULWW", {}), 3: ("SMLAWB", {}), 4: ("SMMUL", {}), 5: ("SMULBB", {}), 6: ("SMLABB", {}),
    7: ("SMULBT", {}), 8: ("SMULTT", {}),
    9: ("RSHIFT_ROUND", {"bmap": lambda b: 1 + (np.abs(b) % 31)}),
    10: ("SAT16", {}), 11: ("ADD_SAT32", {}), 12: ("SUB_SAT32", {}), 13: ("ADD_POS_SAT32", {}),
    14: ("LSHIFT_SAT32", {"bmap": lambda b: np.abs(b) % 32}),
    15: ("CLZ32", {}),
    16: ("ROR32", {"bmap": lambda b: (np.abs(b) % 63) - 31}),
    17: ("SQRT_APPROX", {}), 18: ("lin2log", {"amap": lambda a: np.where(a <= 0, 1 - (a % 1000003), a)}),
    19: ("log2lin", {"amap": lambda a: (a % 4200) - 100}),
    20: ("DIV32_varQ", {"amap": _no_min, "bmap": _nonzero, "cmap": lambda c: np.abs(c) % 33}),
    21: ("INVERSE32_varQ", {"amap": _nonzero, "bmap": lambda b: 1 + (np.abs(b) % 62)}),
    22: ("sigm_Q15", {"amap": lambda a: (a % 600) - 300}),
    23: ("RAND", {}), 24: ("LIMIT", {}), 25: ("SMLAWW", {}), 26: ("SMLAWT", {}), 27: ("CLZ16", {}), 28: ("abs", {"amap": lambda a: np.where(a == I32.min, 5, a)}),
    40: ("PSHR32", {"bmap": lambda b: 1 + (np.abs(b) % 30)}),
    41: ("SATURATE", {"bmap": lambda b: np.abs(b) % 40000}),
    42: ("ADD16", {}), 43: ("SUB16", {}), 44: ("MULT16_16", {}), 45: ("MAC16_16", {}),
}


def _ref_eval(op, a, b, c):
    lib = C.CDLL(L0_LIB)
    f = lib.ref_l0_bwe if op >= 40 else lib.ref_l0
    f.restype = C.c_int32
    f.argtypes = [C.c_int32] * 4
    # DIV32_varQ / INVERSE32_varQ divide by (b_nrm >> 16): the reference leaves a zero there undefined (cannot happen for |b| >= 1)
    return np.array([f(op, int(x), int(y), int(z)) for x, y, z in zip(a, b, c)], np.int32)


def _cases(seed=1, n_rand=1500):
    rng = np.random.default_rng(seed)
    for op, (name, dom) in OPS.items():
        a, b, c = _operands(rng, n_rand, **dom)
        yield op, name, a, b, c


@need_ref
def test_l0_vocabulary_host_emulation_vs_reference_macros():
    emu = T.load_emu()
    emu.emu_l0.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for op, name, a, b, c in _cases():
        want = _ref_eval(op, a, b, c)
        got = np.zeros_like(want)
        emu.emu_l0(op, a.size, a.ctypes.data, b.ctypes.data, c.ctypes.data, got.ctypes.data)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (name, int(a[bad[0]]), int(b[bad[0]]), int(c[bad[0]]), int(got[bad[0]]), int(want[bad[0]]))


def test_derived_forms_equal_their_definitions():
    """The one-instruction forms the kernels use in sample loops (pre-shifted SMULWB / SMLAWB) and the LCG jump-ahead equal the
    plain vocabulary they stand for (needs no reference: checked against the emulation's own SMULWB / RAND, which the test above pins)."""
    emu = T.load_emu()
    emu.emu_l0.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(3)
    a, b, c = _operands(rng, 2000)

    def run(op, a, b, c):
        out = np.zeros(a.size, np.int32)
        emu.emu_l0(op, a.size, a.ctypes.data, b.ctypes.data, c.ctypes.data, out.ctypes.data)
        return out
    assert np.array_equal(run(30, a, b, c), run(0, a, b, c))
    assert np.array_equal(run(31, a, b, c), run(3, a, b, c))
    # the two-instruction rounding shift equals RSHIFT_ROUND wherever a + 2^(s-1) cannot overflow (|a| < 2^30)
    aa = np.clip(a.astype(np.int64), -(1 << 30) + 1, (1 << 30) - 1).astype(np.int32)
    ss = (1 + (np.abs(b.astype(np.int64)) % 29)).astype(np.int32)
    assert np.array_equal(run(34, aa, ss, c), run(9, aa, ss, c))
    seeds = a[:64].copy()
    steps = np.arange(64, dtype=np.int32) * 8
    want = seeds.copy()
    for i in range(64):
        v = np.array([seeds[i]], np.int32)
        for _ in range(int(steps[i])):
            v = run(23, v, v, v)
        want[i] = v[0]
    assert np.array_equal(run(32, seeds, steps, steps), want)


def _sum_sqr_rows(rng):
    rows = []
    for amp in (0, 1, 30, 400, 5000, 20000, 32767):
        for length in (1, 2, 3, 16, 39, 40, 80, 159, 160, 161, 320, 336, 480):
            x = np.zeros(1024, np.int16)
            x[:length] = rng.integers(-amp, amp, length, endpoint=True)
            rows.append((x, length))
    x = np.full(1024, -32768, np.int16)
    rows += [(x, 480), (x, 17), (x, 2)]
    return rows


@need_ref
def test_sum_sqr_shift_vs_reference():
    ref = R.load_ref("fix")
    emu = T.load_emu()
    rng = np.random.default_rng(4)
    for x, length in _sum_sqr_rows(rng):
        for odd in (0, 1):
            # the reference's result depends on the 4-byte alignment of the int16 pointer: offset the buffer accordingly
            buf = np.zeros(1032, np.int16)
            base = buf.ctypes.data
            off = (0 if (base % 4 == 0) else 1) + odd
            buf[off:off + length] = x[:length]
            assert ((base + 2 * off) % 4 != 0) == bool(odd)
            e0, s0 = C.c_int32(), C.c_int32()
            ref.SKP_Silk_sum_sqr_shift(C.byref(e0), C.byref(s0), C.c_void_p(base + 2 * off), C.c_int32(length))
            e1, s1 = C.c_int32(), C.c_int32()
            xs = np.ascontiguousarray(x[:length])
            emu.emu_sum_sqr_shift(xs.ctypes.data_as(C.c_void_p), length, odd, C.byref(e1), C.byref(s1))
            assert (e0.value, s0.value) == (e1.value, s1.value), (length, odd, e0.value, s0.value, e1.value, s1.value)


@pytest.mark.gpu
@need_ref
def test_l0_vocabulary_gfx950_vs_reference_macros():
    """the same operands through the functions as hipcc compiled them for gfx950"""
    import torch
    import solo_amd
    lib = solo_amd.load_library()
    lib.solo_debug_l0.restype = C.c_int32
    lib.solo_debug_l0.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for op, name, a, b, c in _cases():
        want = _ref_eval(op, a, b, c)
        da, db, dc = (torch.from_numpy(v).cuda() for v in (a, b, c))
        out = torch.zeros(a.size, dtype=torch.int32, device="cuda")
        assert lib.solo_debug_l0(op, a.size, da.data_ptr(), db.data_ptr(), dc.data_ptr(), out.data_ptr()) == 0
        got = out.cpu().numpy()
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (name, int(a[bad[0]]), int(b[bad[0]]), int(c[bad[0]]), int(got[bad[0]]), int(want[bad[0]]))
    # pre-shifted forms and the LCG jump-ahead on the device
    rng = np.random.default_rng(3)
    a, b, c = _operands(rng, 2000)
    da, db, dc = (torch.from_numpy(v).cuda() for v in (a, b, c))
    outs = {}
    for op in (0, 3, 30, 31):
        out = torch.zeros(a.size, dtype=torch.int32, device="cuda")
        assert lib.solo_debug_l0(op, a.size, da.data_ptr(), db.data_ptr(), dc.data_ptr(), out.data_ptr()) == 0
        outs[op] = out.cpu().numpy()
    assert np.array_equal(outs[30], outs[0]) and np.array_equal(outs[31], outs[3])


@pytest.mark.gpu
def test_varq_reciprocal_gfx950_every_divisor():
    """sx_div_q29 (solo_fix.h): the device form -- reciprocal estimate + exact remainder correction -- of (INT32_MAX >> 2) / d, the C
    division inside SKP_DIV32_varQ / SKP_INVERSE32_varQ (Inlines.h:136, :182), for EVERY divisor the normalisation can produce"""
    import torch
    import solo_amd
    lib = solo_amd.load_library()
    lib.solo_debug_l0.restype = C.c_int32
    lib.solo_debug_l0.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    d = np.concatenate([np.arange(16384, 32769), -np.arange(16384, 32769)]).astype(np.int32)
    want = (np.sign(d.astype(np.int64)) * (0x1FFFFFFF // np.abs(d.astype(np.int64)))).astype(np.int32)      # C division truncates
    dd = torch.from_numpy(d).cuda()
    out = torch.zeros(d.size, dtype=torch.int32, device="cuda")
    assert lib.solo_debug_l0(33, d.size, dd.data_ptr(), dd.data_ptr(), dd.data_ptr(), out.data_ptr()) == 0
    got = out.cpu().numpy()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (int(d[bad[0]]), int(got[bad[0]]), int(want[bad[0]]))


@pytest.mark.gpu
def test_positive_inverse_gfx950_equals_inverse32_varq():
    """sx_inverse32_varQ_pos (solo_fix.h), the form of SKP_INVERSE32_varQ the step-down recursions call for a positive divisor and a
    right-shifted result, against sx_inverse32_varQ (pinned to the reference's inline by the vocabulary test) on that domain:
    every value the recursion can produce at the top (1 - rc^2 in Q30, Qres 46) is >= 2^19; swept with strides plus random values"""
    import torch
    import solo_amd
    lib = solo_amd.load_library()
    lib.solo_debug_l0.restype = C.c_int32
    lib.solo_debug_l0.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(11)
    a = np.concatenate([np.arange(1 << 19, 1 << 30, 40961), rng.integers(1 << 19, 1 << 30, 200000), (1 << np.arange(19, 30)), (1 << np.arange(20, 31)) - 1]).astype(np.int32)
    b = np.full(a.size, 46, np.int32)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    o1 = torch.zeros(a.size, dtype=torch.int32, device="cuda"); o2 = torch.zeros_like(o1)
    assert lib.solo_debug_l0(35, a.size, da.data_ptr(), db.data_ptr(), db.data_ptr(), o1.data_ptr()) == 0
    assert lib.solo_debug_l0(21, a.size, da.data_ptr(), db.data_ptr(), db.data_ptr(), o2.data_ptr()) == 0
    bad = torch.nonzero(o1 != o2).flatten().cpu().numpy()
    assert bad.size == 0, (int(a[bad[0]]), int(o1[bad[0]]), int(o2[bad[0]]))


@pytest.mark.gpu
@need_ref
def test_sum_sqr_shift_wave_form_gfx950_vs_reference():
    """the wave-cooperative saturating-scan form of SKP_Silk_sum_sqr_shift (solo_common.h) against the reference function"""
    import torch
    import solo_amd
    lib = solo_amd.load_library()
    lib.solo_debug_sum_sqr_shift.restype = C.c_int32
    lib.solo_debug_sum_sqr_shift.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    ref = R.load_ref("fix")
    rng = np.random.default_rng(4)
    rows = _sum_sqr_rows(rng)
    for length in sorted({l for _, l in rows}):
        xs = np.stack([x for x, l in rows if l == length])
        dx = torch.from_numpy(xs).cuda()
        for odd in (0, 1):
            de = torch.zeros(xs.shape[0], dtype=torch.int32, device="cuda")
            ds = torch.zeros_like(de)
            assert lib.solo_debug_sum_sqr_shift(dx.data_ptr(), xs.shape[0], length, 1024, odd, de.data_ptr(), ds.data_ptr()) == 0
            ge, gs = de.cpu().numpy(), ds.cpu().numpy()
            for r in range(xs.shape[0]):
                buf = np.zeros(1032, np.int16)
                base = buf.ctypes.data
                off = (0 if (base % 4 == 0) else 1) + odd
                buf[off:off + length] = xs[r, :length]
                e0, s0 = C.c_int32(), C.c_int32()
                ref.SKP_Silk_sum_sqr_shift(C.byref(e0), C.byref(s0), C.c_void_p(base + 2 * off), C.c_int32(length))
                assert (int(ge[r]), int(gs[r])) == (e0.value, s0.value), (length, odd, r)
