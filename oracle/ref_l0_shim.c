/* ref_l0_shim.c -- TEST INFRASTRUCTURE.  Exposes the reference's INLINE fixed-point vocabulary (macros and static inlines
 * that have no symbol of their own in libsolo_ref_fix.so) as callable functions, by including the reference's headers from where
 * they lie (oracle/Makefile target `ref`, output oracle/_ref/libsolo_ref_l0.so).  Nothing of the reference is copied: every
 * body below is a one-line call of the reference's own macro / inline.  tests/test_l0_primitives.py compares
 * solo_amd/csrc/solo_fix.h (host emulation and the gfx950 build) against these on edge and random operands.
 *   SKP_Silk_macros.h:33-122, SKP_Silk_SigProc_FIX.h:474-650, SKP_Silk_Inlines.h:43-220, libBWE/AGR_BWE_fixed_generic.h:40-111 */
#include "SKP_Silk_SigProc_FIX.h"
#include "SKP_Silk_Inlines.h"

/* op numbering shared with tests/emu/solo_emu.cpp (emu_l0) and solo_api.hip (solo_debug_l0_kernel) */
int ref_l0(int op, int a, int b, int c) {
    switch (op) {
        case 0: return SKP_SMULWB(a, b);
        case 1: return SKP_SMULWT(a, b);
        case 2: return SKP_SMULWW(a, b);
        case 3: return SKP_SMLAWB(c, a, b);
        case 4: return SKP_SMMUL(a, b);
        case 5: return SKP_SMULBB(a, b);
        case 6: return SKP_SMLABB(c, a, b);
        case 7: return SKP_SMULBT(a, b);
        case 8: return SKP_SMULTT(a, b);
        case 9: return SKP_RSHIFT_ROUND(a, b);                 /* b in [1, 31] */
        case 10: return SKP_SAT16(a);
        case 11: return SKP_ADD_SAT32(a, b);
        case 12: return SKP_SUB_SAT32(a, b);
        case 13: return SKP_ADD_POS_SAT32(a, b);
        case 14: return SKP_LSHIFT_SAT32(a, b);                /* b in [0, 31] */
        case 15: return SKP_Silk_CLZ32(a);
        case 16: return SKP_ROR32(a, b);                       /* b in [-31, 31] */
        case 17: return SKP_Silk_SQRT_APPROX(a);
        case 18: return SKP_Silk_lin2log(a);
        case 19: return SKP_Silk_log2lin(a);
        case 20: return SKP_DIV32_varQ(a, b, c);               /* b != 0, c = Qres */
        case 21: return SKP_INVERSE32_varQ(a, b);              /* a != 0, b = Qres */
        case 22: return SKP_Silk_sigm_Q15(a);
        case 23: return SKP_RAND(a);
        case 24: return SKP_LIMIT(a, b, c);
        case 25: return SKP_SMLAWW(c, a, b);
        case 26: return SKP_SMLAWT(c, a, b);
        case 27: return SKP_Silk_CLZ16((SKP_int16)a);
        case 28: return SKP_abs(a);
    }
    return 0;
}
