/* ref_l0_shim_bwe.c -- TEST INFRASTRUCTURE, see ref_l0_shim.c.  The Speex-derived 16-bit helpers of the QMF
 * (libBWE/AGR_BWE_fixed_generic.h:40-111) live in a header that redefines names of the SILK vocabulary, hence a second file. */
#include "AGR_BWE_defines.h"      /* FIXED_POINT, as the fixed-point tree compiles libBWE */
#include "AGR_BWE_arch.h"
int ref_l0_bwe(int op, int a, int b, int c) {
    (void)c;
    switch (op) {
        case 40: return PSHR32(a, b);                          /* b in [1, 30] */
        case 41: return SATURATE(a, b);
        /* ADD16 / SUB16 as the QMF uses them, inside MULT16_16 (AGR_BWE_qmf.c:71-75): SUB16 itself carries no outer cast */
        case 42: return MULT16_16((short)c, ADD16(a, b));
        case 43: return MULT16_16((short)c, SUB16(a, b));
        case 44: return MULT16_16((short)a, (short)b);
        case 45: return MAC16_16(c, (short)a, (short)b);
    }
    return 0;
}
