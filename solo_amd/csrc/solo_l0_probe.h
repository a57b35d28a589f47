// solo_l0_probe.h -- one entry point over the L0 fixed-point vocabulary (solo_fix.h / solo_common.h) so that every primitive can
// be checked operand by operand against the reference's own macros ("restate once, test exhaustively").
// Used by the host emulation (tests/emu: emu_l0) and by the gfx950 build (solo_debug_l0 in solo_api.hip); the op numbers are
// shared with oracle/ref_l0_shim.c.  Not on the codec's path.
#pragma once
#include "solo_common.h"

SX_HD i32 sx_l0_probe(int op, i32 a, i32 b, i32 c) {
    switch (op) {
        case 0: return sx_smulwb(a, b);
        case 1: return sx_smulwt(a, b);
        case 2: return sx_smulww(a, b);
        case 3: return sx_smlawb(c, a, b);
        case 4: return sx_smmul(a, b);
        case 5: return sx_smulbb(a, b);
        case 6: return sx_smlabb(c, a, b);
        case 7: return sx_smulbt(a, b);
        case 8: return sx_smultt(a, b);
        case 9: return sx_rshift_round(a, b);
        case 10: return sx_sat16(a);
        case 11: return sx_add_sat32(a, b);
        case 12: return sx_sub_sat32(a, b);
        case 13: return sx_add_pos_sat32(a, b);
        case 14: return sx_lshift_sat32(a, b);
        case 15: return sx_clz32(a);
        case 16: return sx_ror32(a, b);
        case 17: return sx_sqrt_approx(a);
        case 18: return sx_lin2log(a);
        case 19: return sx_log2lin(a);
        case 20: return sx_div32_varQ(a, b, c);
        case 21: return sx_inverse32_varQ(a, b);
        case 22: return sx_sigm_Q15(a);
        case 23: return sx_rand(a);
        case 24: return sx_limit(a, b, c);
        case 25: return sx_smlaww(c, a, b);
        case 26: return sx_smlawt(c, a, b);
        case 27: return sx_clz16((i16)a);
        case 28: return sx_abs(a);
        // the pre-shifted one-instruction forms the kernels use in their sample loops must equal SMULWB / SMULWT
        case 30: return sx_smulw_pre(a, sx_pre16(b));
        case 31: return sx_smlaw_pre(c, a, sx_pre16(b));
        case 32: return sx_rand_skip(a, (u32)b & 511u);         // b-th iterate of SKP_RAND
        case 34: return sx_rshift_round_small(a, b);            // = RSHIFT_ROUND for |a| < 2^30
        case 35: return sx_inverse32_varQ_pos(a, b);            // = INVERSE32_varQ for a > 0 and 61 - headroom - b in 1 .. 31
        case 33: return sx_div_q29(a);                          // (INT32_MAX >> 2) / a on the divisor domain of the varQ divisions
        // Speex-derived helpers of the QMF (libBWE/AGR_BWE_fixed_generic.h)
        case 40: return sx_pshr32(a, b);
        case 41: return sx_saturate(a, b);
        case 42: return sx_smulbb(c, (i32)(i16)((i16)a + (i16)b));     // MULT16_16(c, ADD16(a, b)), AGR_BWE_qmf.c:71
        case 43: return sx_smulbb(c, (i32)(i16)((i16)a - (i16)b));     // MULT16_16(c, SUB16(a, b))
        case 44: return sx_smulbb(a, b);
        case 45: return sx_smlabb(c, a, b);
    }
    return 0;
}
