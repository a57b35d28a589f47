// solo_fix.h -- L0 fixed-point vocabulary of the SOLO hot path, written for the gfx950 kernels.
//
// Every function restates the arithmetic of one reference macro/inline bit-exactly (reference
// file:line cited per function; paths relative to JC1_SDK_SRC_ARM/src/libSATECodec unless noted).
// All 32-bit additions/multiplications wrap (two's complement) exactly like the reference
// binaries do; shifts of negative values are arithmetic.
//
// The header compiles both as HIP device code (hipcc, gfx950) and as plain host C++ -- the host
// build exists only for the kernel-source emulation used by the CPU-side tests (tests/emu), the
// product library never runs codec arithmetic on the host.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SX_HD __host__ __device__ __forceinline__
#define SX_DEV __device__
// large stage functions: real calls (keeps register pressure and code size per kernel in check)
#if defined(SX_INLINE_ALL)
#define SX_FN __host__ __device__ __forceinline__
#else
#define SX_FN static __host__ __device__ __attribute__((noinline))
#endif
#else
#define SX_HD static inline
#define SX_DEV
#define SX_FN static
#endif
// stage functions with ONE call site.  Inlining them there (-DSX_INLINE_SINGLE) costs no code, but their private arrays then all
// live in one frame (1008 B of scratch per lane instead of 380 along the deepest call chain) and the analysis kernel gets 14 % slower
// (measured); so they stay real calls.
#if defined(__HIPCC__) && defined(SX_INLINE_SINGLE)
#define SX_FN1 __host__ __device__ __forceinline__
#else
#define SX_FN1 SX_FN
#endif
// the frame / packet level wrappers between a kernel and its stages: always inlined into the kernel (their callee-saved registers
// would otherwise be saved and restored around every frame: ~100 B of scratch per lane and 2 x 24 scratch accesses per frame)
#if defined(__HIPCC__) && !defined(SX_OUTLINE_WRAPPERS)
#define SX_FNW __host__ __device__ __forceinline__
#else
#define SX_FNW SX_FN
#endif

// address-space hints for generic pointer parameters of non-inlined stage functions: lets the compiler emit ds_* / global_*
// instead of flat_* (LLVM InferAddressSpaces understands these assumes)
#if defined(__HIP_DEVICE_COMPILE__)
#define SX_IN_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void*)(p)))
#define SX_IN_GLOBAL(p) __builtin_assume(!__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void*)(p)) && \
                                         !__builtin_amdgcn_is_private((const __attribute__((address_space(0))) void*)(p)))
#else
#define SX_IN_LDS(p)
#define SX_IN_GLOBAL(p)
#endif

// keep a scalar search loop exactly as written (no vectorisation / interleaving / unrolling)
#if defined(__clang__)
#define SX_PLAIN_LOOP _Pragma("clang loop vectorize(disable) interleave(disable) unroll(disable)")
#else
#define SX_PLAIN_LOOP
#endif

typedef int32_t i32;
typedef uint32_t u32;
typedef int16_t i16;
typedef uint16_t u16;
typedef int64_t i64;
typedef uint64_t u64;
typedef uint8_t u8;
typedef int8_t i8;

#define SX_I32_MAX 0x7FFFFFFF
#define SX_I32_MIN ((i32)0x80000000)

// ---- wrapping primitives ---------------------------------------------------------------------
SX_HD i32 sx_add(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
SX_HD i32 sx_sub(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }
SX_HD i32 sx_mul(i32 a, i32 b) { return (i32)((u32)a * (u32)b); }
SX_HD i32 sx_shl(i32 a, int s) { return (i32)((u32)a << s); }
SX_HD i32 sx_neg(i32 a) { return (i32)(0u - (u32)a); }
SX_HD i32 sx_min(i32 a, i32 b) { return a < b ? a : b; }
SX_HD i32 sx_max(i32 a, i32 b) { return a > b ? a : b; }
SX_HD i32 sx_abs(i32 a) { return a > 0 ? a : sx_neg(a); }  // SKP_abs, SigProc_FIX.h:642
// SKP_LIMIT (SigProc_FIX.h:634): order-agnostic clamp
SX_HD i32 sx_limit(i32 a, i32 l1, i32 l2) {
    return l1 > l2 ? (a > l1 ? l1 : (a < l2 ? l2 : a)) : (a > l2 ? l2 : (a < l1 ? l1 : a));
}

// ---- 16x32 / 16x16 multiplies (SKP_Silk_macros.h:33-67) -----------------------------------------
// SKP_SMULWB: (a * (int16)b) >> 16, exact floor, 32-bit wrap on the (impossible) overflow
// (a * (int16)b) >> 16 written as the high word of a * (b << 16): ONE v_mul_hi_i32 on gfx950 instead of mul_hi + mul_lo + 64-bit shift
SX_HD i32 sx_smulwb(i32 a, i32 b) { return (i32)(((i64)a * (i64)(i32)((u32)b << 16)) >> 32); }
SX_HD i32 sx_smlawb(i32 acc, i32 a, i32 b) { return sx_add(acc, sx_smulwb(a, b)); }
// the same with the 16-bit factor already moved to the high half (hoisted out of sample loops): bs = (int16)b << 16
SX_HD i32 sx_pre16(i32 b) { return (i32)((u32)b << 16); }
SX_HD i32 sx_smulw_pre(i32 a, i32 bs) { return (i32)(((i64)a * (i64)bs) >> 32); }
SX_HD i32 sx_smlaw_pre(i32 acc, i32 a, i32 bs) { return sx_add(acc, sx_smulw_pre(a, bs)); }
// SKP_SMULWT: (a * (b >> 16)) >> 16
SX_HD i32 sx_smulwt(i32 a, i32 b) { return (i32)(((i64)a * (i64)(i32)((u32)b & 0xFFFF0000u)) >> 32); }
SX_HD i32 sx_smlawt(i32 acc, i32 a, i32 b) { return sx_add(acc, sx_smulwt(a, b)); }
SX_HD i32 sx_smulbb(i32 a, i32 b) { return (i32)(i16)a * (i32)(i16)b; }
SX_HD i32 sx_smlabb(i32 acc, i32 a, i32 b) { return sx_add(acc, sx_smulbb(a, b)); }
SX_HD i32 sx_smulbt(i32 a, i32 b) { return sx_mul((i32)(i16)a, b >> 16); }
SX_HD i32 sx_smultt(i32 a, i32 b) { return sx_mul(a >> 16, b >> 16); }
// SKP_RSHIFT_ROUND (SigProc_FIX.h:592); shift >= 1
SX_HD i32 sx_rshift_round(i32 a, int s) { return s == 1 ? sx_add(a >> 1, a & 1) : (sx_add(a >> (s - 1), 1) >> 1); }
// ... in two instructions where a + 2^(s-1) cannot overflow (|a| < 2^30): floor((floor(a / 2^(s-1)) + 1) / 2) = floor((a + 2^(s-1)) / 2^s)
SX_HD i32 sx_rshift_round_small(i32 a, int s) { return sx_add(a, 1 << (s - 1)) >> s; }
SX_HD i64 sx_rshift_round64(i64 a, int s) { return s == 1 ? (a >> 1) + (a & 1) : (((a >> (s - 1)) + 1) >> 1); }
// SKP_SMULWW = MLA(SMULWB(a,b), a, RSHIFT_ROUND(b,16))   (macros.h:61) -- NOT a plain 64-bit product
// SKP_SMULWW = SMULWB(a,b) + a * RSHIFT_ROUND(b,16) (32-bit wrapping) is exactly the low word of (a * b) >> 16
SX_HD i32 sx_smulww(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 16); }
SX_HD i32 sx_smlaww(i32 acc, i32 a, i32 b) { return sx_add(acc, sx_smulww(a, b)); }
SX_HD i32 sx_smmul(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 32); }  // macros.h:67
SX_HD i64 sx_smull(i32 a, i32 b) { return (i64)a * (i64)b; }

// ---- saturation (macros.h:70-76, SigProc_FIX.h:554-577) -------------------------------------------
SX_HD i32 sx_sat16(i32 a) { return a > 32767 ? 32767 : (a < -32768 ? -32768 : a); }
SX_HD i32 sx_add_sat32(i32 a, i32 b) {
    i64 s = (i64)a + (i64)b;
    return s > SX_I32_MAX ? SX_I32_MAX : (s < (i64)SX_I32_MIN ? SX_I32_MIN : (i32)s);
}
SX_HD i32 sx_sub_sat32(i32 a, i32 b) {
    i64 s = (i64)a - (i64)b;
    return s > SX_I32_MAX ? SX_I32_MAX : (s < (i64)SX_I32_MIN ? SX_I32_MIN : (i32)s);
}
SX_HD i32 sx_add_pos_sat32(i32 a, i32 b) { i32 s = sx_add(a, b); return (s & 0x80000000) ? SX_I32_MAX : s; }
// SKP_LSHIFT_SAT32 (SigProc_FIX.h:576)
SX_HD i32 sx_lshift_sat32(i32 a, int s) {
    i32 lo = SX_I32_MIN >> s, hi = SX_I32_MAX >> s;
    i32 c = a < lo ? lo : (a > hi ? hi : a);
    return sx_shl(c, s);
}

// ---- bit scans (macros.h:78-122, Inlines.h:43-66) --------------------------------------------------
SX_HD i32 sx_clz32(i32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return x == 0 ? 32 : __clz(x);
#else
    return x == 0 ? 32 : __builtin_clz((u32)x);
#endif
}
SX_HD i32 sx_clz16(i16 x) { return sx_clz32((i32)(u32)(u16)x) - 16; }
SX_HD i32 sx_ror32(i32 a, int rot) {  // SigProc_FIX.h:482 (generic variant; rot in [-31, 31])
    u32 x = (u32)a;
    u32 r = (u32)rot & 31u;
    return (i32)((x >> r) | (x << ((32u - r) & 31u)));
}
SX_HD void sx_clz_frac(i32 in, i32* lz, i32* frac_Q7) {
    i32 l = sx_clz32(in);
    *lz = l;
    *frac_Q7 = sx_ror32(in, 24 - l) & 0x7f;
}
// SKP_Silk_SQRT_APPROX (Inlines.h:71)
SX_HD i32 sx_sqrt_approx(i32 x) {
    if (x <= 0) return 0;
    i32 lz, frac;
    sx_clz_frac(x, &lz, &frac);
    i32 y = (lz & 1) ? 32768 : 46214;
    y >>= (lz >> 1);
    return sx_smlawb(y, y, sx_smulbb(213, frac));
}
// SKP_Silk_lin2log (SKP_Silk_lin2log.c:41): ~128*log2(x)
SX_HD i32 sx_lin2log(i32 x) {
    i32 lz, frac;
    sx_clz_frac(x, &lz, &frac);
    return sx_add(sx_shl(31 - lz, 7), sx_smlawb(frac, sx_mul(frac, 128 - frac), 179));
}
// SKP_Silk_log2lin (SKP_Silk_log2lin.c:40)
SX_HD i32 sx_log2lin(i32 inLog_Q7) {
    if (inLog_Q7 < 0) return 0;
    if (inLog_Q7 >= (31 << 7)) return SX_I32_MAX;
    i32 out = sx_shl(1, inLog_Q7 >> 7);
    i32 frac = inLog_Q7 & 0x7F;
    i32 p = sx_smlawb(frac, sx_mul(frac, 128 - frac), -174);
    if (inLog_Q7 < 2048) return sx_add(out, sx_mul(out, p) >> 7);
    return sx_add(out, sx_mul(out >> 7, p));
}
// (INT32_MAX >> 2) / d, C division, for the normalised divisor of DIV32_varQ / INVERSE32_varQ below (Inlines.h:136, :182): 16384 <= |d| <= 32768
SX_HD i32 sx_div_q29(i32 d) {
#if defined(__HIP_DEVICE_COMPILE__)
    // the compiler's generic signed 32-bit division is ~30 instructions.  The quotient is at most 2^15 here, so a single-precision
    // reciprocal estimate is off by far less than one, and one exact remainder check either way restores the truncated quotient
    // (tests/test_l0_primitives.py runs every divisor of the domain on the device)
    const i32 ad = d < 0 ? -d : d;
    i32 q = (i32)(536870912.0f * __builtin_amdgcn_rcpf((float)ad));
    const i32 r = (SX_I32_MAX >> 2) - q * ad;
    q += r >= ad ? 1 : 0;
    q -= r < 0 ? 1 : 0;
    return d < 0 ? -q : q;
#else
    return (SX_I32_MAX >> 2) / d;
#endif
}
// SKP_DIV32_varQ (Inlines.h:124)
SX_HD i32 sx_div32_varQ(i32 a32, i32 b32, int Qres) {
    int a_headrm = sx_clz32(sx_abs(a32)) - 1;
    i32 a_nrm = sx_shl(a32, a_headrm);
    int b_headrm = sx_clz32(sx_abs(b32)) - 1;
    i32 b_nrm = sx_shl(b32, b_headrm);
    i32 b_inv = sx_div_q29(b_nrm >> 16);
    i32 result = sx_smulwb(a_nrm, b_inv);
    a_nrm = sx_sub(a_nrm, sx_shl(sx_smmul(b_nrm, result), 3));
    result = sx_smlawb(result, a_nrm, b_inv);
    int lshift = 29 + a_headrm - b_headrm - Qres;
    if (lshift <= 0) return sx_lshift_sat32(result, -lshift);
    return lshift < 32 ? (result >> lshift) : 0;
}
// SKP_INVERSE32_varQ (Inlines.h:169)
SX_HD i32 sx_inverse32_varQ(i32 b32, int Qres) {
    int b_headrm = sx_clz32(sx_abs(b32)) - 1;
    i32 b_nrm = sx_shl(b32, b_headrm);
    i32 b_inv = sx_div_q29(b_nrm >> 16);
    i32 result = sx_shl(b_inv, 16);
    i32 err_Q32 = sx_shl(sx_neg(sx_smulwb(b_nrm, b_inv)), 3);
    result = sx_smlaww(result, err_Q32, b_inv);
    int lshift = 61 - b_headrm - Qres;
    if (lshift <= 0) return sx_lshift_sat32(result, -lshift);
    return lshift < 32 ? (result >> lshift) : 0;
}
// SKP_INVERSE32_varQ for a divisor known to be POSITIVE and a result that is shifted RIGHT (61 - headroom - Qres in 1 .. 31): what is
// left of it then -- no absolute values, no sign handling in the reciprocal, no saturating-left-shift branch.  (The step-down recursion
// of LPC_inverse_pred_gain: 1 - rc^2 in Q30 with |rc| <= 0.99975, Qres 46; lanes whose filter was found unstable compute garbage here and
// are masked by the caller.)
SX_HD i32 sx_inverse32_varQ_pos(i32 b32, int Qres) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int b_headrm = sx_clz32(b32) - 1;
    const i32 b_nrm = sx_shl(b32, b_headrm);
    const i32 d = b_nrm >> 16;                                   // 16384 .. 32767
    i32 b_inv = (i32)(536870912.0f * __builtin_amdgcn_rcpf((float)d));
    const i32 r = (SX_I32_MAX >> 2) - b_inv * d;
    b_inv += r >= d ? 1 : 0;
    b_inv -= r < 0 ? 1 : 0;
    i32 result = sx_shl(b_inv, 16);
    const i32 err_Q32 = sx_shl(sx_neg(sx_smulwb(b_nrm, b_inv)), 3);
    result = sx_smlaww(result, err_Q32, b_inv);
    return result >> (61 - b_headrm - Qres);
#else
    return sx_inverse32_varQ(b32, Qres);
#endif
}
// SKP_RAND (SigProc_FIX.h:650)
SX_HD i32 sx_rand(i32 seed) { return (i32)(907633515u + (u32)seed * 196314165u); }
// n-th iterate of sx_rand (n >= 0) by square-and-multiply on the affine map x -> A x + C (mod 2^32): lets every lane of a
// data-parallel loop regenerate "its" value of a serial LCG sequence
template <int BITS>
SX_HD i32 sx_rand_skip_bits(i32 seed, u32 n) {      // n < 2^BITS
    u32 Ar = 1u, Cr = 0u, Ab = 196314165u, Cb = 907633515u;
#pragma unroll
    for (int bit = 0; bit < BITS; bit++) {
        if (n & (1u << bit)) { Ar = Ab * Ar; Cr = Ab * Cr + Cb; }
        Cb = Ab * Cb + Cb;
        Ab = Ab * Ab;
    }
    return (i32)(Ar * (u32)seed + Cr);
}
SX_HD i32 sx_rand_skip(i32 seed, u32 n) { return sx_rand_skip_bits<9>(seed, n); }      // n < 512 (a 20 ms frame at 16 kHz has 320 samples)
// A lane-strided loop over a serial LCG sequence (SX_PAR: i = lane, lane + NLANES, ...): the lane's first value by jump-ahead, the
// following ones by the constant map "NLANES steps at once" (one multiply-add; compile-time constants)
struct SxLcgMap { u32 A, C; };
constexpr SxLcgMap sx_lcg_map(u32 n) {             // x -> A x + C = n applications of sx_rand
    u32 Ar = 1u, Cr = 0u, Ab = 196314165u, Cb = 907633515u;
    for (int bit = 0; bit < 32; bit++) {
        if (n & (1u << bit)) { Ar = Ab * Ar; Cr = Ab * Cr + Cb; }
        Cb = Ab * Cb + Cb;
        Ab = Ab * Ab;
    }
    return SxLcgMap{Ar, Cr};
}

// ---- Speex-derived 16-bit helpers of the QMF (libBWE/AGR_BWE_fixed_generic.h:40-80) ---------------
SX_HD i32 sx_pshr32(i32 a, int s) { return sx_add(a, (1 << s) >> 1) >> s; }
SX_HD i32 sx_saturate(i32 x, i32 a) { return x > a ? a : (x < -a ? -a : x); }
