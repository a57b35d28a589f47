// solo_common.h -- small DSP building blocks shared by the encoder and decoder kernels
// (LPC/NLSF conversions, stabilisers, energy with the reference's block-shift rule, short filters).
// Wave-uniform unless a function says "wave-parallel".  Reference paths are relative to
// JC1_SDK_SRC_ARM/src/libSATECodec/.
#pragma once
#include "solo_wave.h"

#if defined(__HIPCC__)
#define SOLO_TAB static __device__ const
#else
#define SOLO_TAB static const
#endif
#include "solo_tables.inc"

// SILK's internal rate is a compile-time parameter of the whole kernel source: 8 (16 kHz API rate: the benchmark
// configuration) or 16 (32 kHz API rate, `samplerate == 32000` in the control structs, libBWE/AGR_BWE_SDK_API.c:86-89,197).
// The rate-dependent tables keep one set of names in the code below.
#ifndef SX_FS_KHZ
#define SX_FS_KHZ 8
#endif
#if SX_FS_KHZ == 8
#define SX_LPC 10          // predictLPCOrder at fs_kHz == 8   (SKP_Silk_control_codec_FIX.c:278)
#define SX_NLSF_STAGES 6   // stages of SKP_Silk_NLSF_CB0_10 / CB1_10
#define SX_NLSF_CB0_NVEC_TOTAL 120
#define SX_NLSF_CB1_NVEC_TOTAL 72
#define SX_NLSF_CB_MAXVEC 120
#define SX_MSVQ_ROW 16
#define SX_N_NLSF_CB0_CDF 126
#define SX_N_NLSF_CB1_CDF 78
#define SX_N_PITCH_LAG_CDF 130
#define SX_N_PITCH_CONTOUR_CDF 12
#define SX_PITCH_CB_N 11   // contours of SKP_Silk_CB_lags_stage2 (decode_pitch.c:43)
#define T_cdf_pitch_lag T_cdf_pitch_lag_nb
#define T_cdf_pitch_contour T_cdf_pitch_contour_nb
#define T_CDF_MID_PITCH_LAG T_CDF_MID_PITCH_LAG_NB
#define T_CDF_MID_PITCH_CONTOUR T_CDF_MID_PITCH_CONTOUR_NB
#define T_pitch_cb_dec T_pitch_cb_stage2
#define T_target_rate T_target_rate_nb       // control_codec_FIX.c:338-346
#define K_MU_LTP_QUANT_Q8 K_MU_LTP_QUANT_NB_Q8   // control_codec_FIX.c:300-316
#elif SX_FS_KHZ == 16
#include "solo_tables_wb.inc"
#define SX_LPC 16          // decoder_set_fs.c:45
#define SX_NLSF_STAGES 10  // stages of SKP_Silk_NLSF_CB0_16 / CB1_16
#define SX_NLSF_CB0_NVEC_TOTAL 216
#define SX_NLSF_CB1_NVEC_TOTAL 104
#define SX_NLSF_CB_MAXVEC 216
#define SX_MSVQ_ROW 32
#define SX_N_NLSF_CB0_CDF 226
#define SX_N_NLSF_CB1_CDF 114
#define SX_N_PITCH_LAG_CDF 258
#define SX_N_PITCH_CONTOUR_CDF 35
#define SX_PITCH_CB_N 34   // contours of SKP_Silk_CB_lags_stage3 (decode_pitch.c:52)
#define T_cdf_pitch_lag T_cdf_pitch_lag_wb
#define T_cdf_pitch_contour T_cdf_pitch_contour_wb
#define T_CDF_MID_PITCH_LAG T_CDF_MID_PITCH_LAG_WB
#define T_CDF_MID_PITCH_CONTOUR T_CDF_MID_PITCH_CONTOUR_WB
#define T_pitch_cb_dec T_pitch_cb_stage3
#define T_target_rate T_target_rate_wb
#define K_MU_LTP_QUANT_Q8 K_MU_LTP_QUANT_WB_Q8
#define T_nlsf_cb0_Q15 T_nlsf16_cb0_Q15
#define T_nlsf_cb1_Q15 T_nlsf16_cb1_Q15
#define T_nlsf_cb0_rates_Q5 T_nlsf16_cb0_rates_Q5
#define T_nlsf_cb1_rates_Q5 T_nlsf16_cb1_rates_Q5
#define T_nlsf_cb0_cdf T_nlsf16_cb0_cdf
#define T_nlsf_cb1_cdf T_nlsf16_cb1_cdf
#define T_nlsf_cb0_cdf_mid T_nlsf16_cb0_cdf_mid
#define T_nlsf_cb1_cdf_mid T_nlsf16_cb1_cdf_mid
#define T_nlsf_cb0_ndelta_min_Q15 T_nlsf16_cb0_ndelta_min_Q15
#define T_nlsf_cb1_ndelta_min_Q15 T_nlsf16_cb1_ndelta_min_Q15
#undef T_NLSF_CB0_NVEC
#undef T_NLSF_CB1_NVEC
#define T_NLSF_CB0_NVEC T_NLSF16_CB0_NVEC
#define T_NLSF_CB1_NVEC T_NLSF16_CB1_NVEC
#else
#error "SX_FS_KHZ must be 8 or 16"
#endif
// kernels / launchers of the two builds get distinct symbols
#if SX_FS_KHZ == 8
#define SX_K(name) name
#else
#define SX_K(name) name##_wb
#endif
#define SX_MAX_LPC 16      // MAX_LPC_ORDER                    (SKP_Silk_define.h:200)
#define SX_HB_LPC 8        // BWE_LPCOrder                     (libBWE/AGR_BWE_SDK_API.c:106)
#define SX_FRAME (20 * SX_FS_KHZ)   // 20 ms
#define SX_SUBFR (5 * SX_FS_KHZ)
#define SX_NB_SUBFR 4
#define SX_LTP_ORDER 5
#define SX_FCH ((SX_FRAME + 63) / 64)   // 64-lane chunks of a frame (lane-register streaming of per-sample recursions)

// SKP_Silk_bwexpander, SKP_Silk_bwexpander.c:31
SX_HD void sx_bwexpander(i16* ar, int d, i32 chirp_Q16) {
    i32 chirp_minus_one_Q16 = chirp_Q16 - 65536;
    for (int i = 0; i < d - 1; i++) {
        ar[i] = (i16)sx_rshift_round(sx_mul(chirp_Q16, ar[i]), 16);
        chirp_Q16 = sx_add(chirp_Q16, sx_rshift_round(sx_mul(chirp_Q16, chirp_minus_one_Q16), 16));
    }
    ar[d - 1] = (i16)sx_rshift_round(sx_mul(chirp_Q16, ar[d - 1]), 16);
}

// SKP_Silk_bwexpander_32, SKP_Silk_bwexpander_32.c:31
SX_HD void sx_bwexpander_32(i32* ar, int d, i32 chirp_Q16) {
    i32 tmp = chirp_Q16;
    for (int i = 0; i < d - 1; i++) {
        ar[i] = sx_smulww(ar[i], tmp);
        tmp = sx_smulww(chirp_Q16, tmp);
    }
    ar[d - 1] = sx_smulww(ar[d - 1], tmp);
}

// LPC_inverse_pred_gain_QA, SKP_Silk_LPC_inv_pred_gain.c:43.  A_QA holds the coefficients in Q16
// in row (order & 1).  Returns 1 if unstable.
SX_HD int sx_lpc_inv_pred_gain_QA(i32* invGain_Q30, i32 A_QA[2][SX_MAX_LPC], int order) {
    const i32 A_LIMIT = 65520;  // SKP_FIX_CONST( 0.99975, 16 ) = (int)(65519.616 + 0.5)
    i32* Anew = A_QA[order & 1];
    *invGain_Q30 = 1 << 30;
    for (int k = order - 1; k > 0; k--) {
        if (Anew[k] > A_LIMIT || Anew[k] < -A_LIMIT) return 1;
        i32 rc_Q31 = sx_neg(sx_shl(Anew[k], 31 - 16));
        i32 rc_mult1_Q30 = (SX_I32_MAX >> 1) - sx_smmul(rc_Q31, rc_Q31);
        i32 rc_mult2_Q16 = sx_inverse32_varQ(rc_mult1_Q30, 46);
        *invGain_Q30 = sx_shl(sx_smmul(*invGain_Q30, rc_mult1_Q30), 2);
        i32* Aold = Anew;
        Anew = A_QA[k & 1];
        int headrm = sx_clz32(rc_mult2_Q16) - 1;
        rc_mult2_Q16 = sx_shl(rc_mult2_Q16, headrm);
        for (int n = 0; n < k; n++) {
            i32 tmp = sx_sub(Aold[n], sx_shl(sx_smmul(Aold[k - n - 1], rc_Q31), 1));
            Anew[n] = sx_shl(sx_smmul(tmp, rc_mult2_Q16), 16 - headrm);
        }
    }
    if (Anew[0] > A_LIMIT || Anew[0] < -A_LIMIT) return 1;
    i32 rc_Q31 = sx_neg(sx_shl(Anew[0], 31 - 16));
    i32 rc_mult1_Q30 = (SX_I32_MAX >> 1) - sx_smmul(rc_Q31, rc_Q31);
    *invGain_Q30 = sx_shl(sx_smmul(*invGain_Q30, rc_mult1_Q30), 2);
    return 0;
}
// SKP_Silk_LPC_inverse_pred_gain (Q12 input), SKP_Silk_LPC_inv_pred_gain.c:113
SX_HD int sx_lpc_inv_pred_gain_ws(i32* invGain_Q30, const i16* A_Q12, int order, i32 (*A)[SX_MAX_LPC]) {
    for (int k = 0; k < order; k++) A[order & 1][k] = sx_shl((i32)A_Q12[k], 4);
    return sx_lpc_inv_pred_gain_QA(invGain_Q30, A, order);
}
SX_HD int sx_lpc_inv_pred_gain(i32* invGain_Q30, const i16* A_Q12, int order) {
    i32 A[2][SX_MAX_LPC];
    return sx_lpc_inv_pred_gain_ws(invGain_Q30, A_Q12, order, A);
}
// SKP_Silk_LPC_inverse_pred_gain_Q24, SKP_Silk_LPC_inv_pred_gain.c:134
SX_HD int sx_lpc_inv_pred_gain_Q24(i32* invGain_Q30, const i32* A_Q24, int order) {
    i32 A[2][SX_MAX_LPC];
    for (int k = 0; k < order; k++) A[order & 1][k] = sx_rshift_round(A_Q24[k], 8);
    return sx_lpc_inv_pred_gain_QA(invGain_Q30, A, order);
}

// SKP_Silk_NLSF2A_find_poly, SKP_Silk_NLSF2A.c:37
SX_HD void sx_nlsf2a_find_poly(i32* out, const i32* cLSF, int dd) {
    out[0] = 1 << 20;
    out[1] = sx_neg(cLSF[0]);
    for (int k = 1; k < dd; k++) {
        i32 ftmp = cLSF[2 * k];
        out[k + 1] = sx_sub(sx_shl(out[k - 1], 1), (i32)sx_rshift_round64(sx_smull(ftmp, out[k]), 20));
        for (int n = k; n > 1; n--)
            out[n] = sx_add(out[n], sx_sub(out[n - 2], (i32)sx_rshift_round64(sx_smull(ftmp, out[n - 1]), 20)));
        out[1] = sx_sub(out[1], ftmp);
    }
}

// SKP_Silk_NLSF2A, SKP_Silk_NLSF2A.c:59   (d even, <= 16)
#define SX_NLSF2A_WS (SX_MAX_LPC + 2 * (SX_MAX_LPC / 2 + 1) + SX_MAX_LPC + 2 * SX_MAX_LPC)    // words of workspace
SX_HD void sx_nlsf2a_ws(i16* a, const i32* NLSF, int d, i32* ws) {
    i32 *cos_LSF_Q20 = ws, *P = ws + SX_MAX_LPC, *Q = P + SX_MAX_LPC / 2 + 1, *a32 = Q + SX_MAX_LPC / 2 + 1;
    for (int k = 0; k < d; k++) {
        i32 f_int = NLSF[k] >> 8;
        i32 f_frac = NLSF[k] - (f_int << 8);
        i32 cos_val = T_lsf_cos_Q12[f_int];
        i32 delta = T_lsf_cos_Q12[f_int + 1] - cos_val;
        cos_LSF_Q20[k] = sx_add(sx_shl(cos_val, 8), sx_mul(delta, f_frac));
    }
    int dd = d >> 1;
    sx_nlsf2a_find_poly(P, &cos_LSF_Q20[0], dd);
    sx_nlsf2a_find_poly(Q, &cos_LSF_Q20[1], dd);
    for (int k = 0; k < dd; k++) {
        i32 Ptmp = sx_add(P[k + 1], P[k]);
        i32 Qtmp = sx_sub(Q[k + 1], Q[k]);
        a32[k] = sx_neg(sx_rshift_round(sx_add(Ptmp, Qtmp), 9));
        a32[d - k - 1] = sx_rshift_round(sx_sub(Qtmp, Ptmp), 9);
    }
    int i;
    for (i = 0; i < 10; i++) {
        i32 maxabs = 0, idx = 0;
        for (int k = 0; k < d; k++) {
            i32 av = sx_abs(a32[k]);
            if (av > maxabs) { maxabs = av; idx = k; }
        }
        if (maxabs > 32767) {
            maxabs = sx_min(maxabs, 98369);
            i32 sc_Q16 = 65470 - sx_mul(65470 >> 2, maxabs - 32767) / (sx_mul(maxabs, idx + 1) >> 2);
            sx_bwexpander_32(a32, d, sc_Q16);
        } else {
            break;
        }
    }
    if (i == 10) {
        for (int k = 0; k < d; k++) a32[k] = sx_sat16(a32[k]);
    }
    for (int k = 0; k < d; k++) a[k] = (i16)a32[k];
}

SX_HD void sx_nlsf2a(i16* a, const i32* NLSF, int d) {
    i32 ws[SX_NLSF2A_WS];
    sx_nlsf2a_ws(a, NLSF, d, ws);
}

// SKP_Silk_NLSF2A_stable, SKP_Silk_NLSF2A_stable.c:31 with caller-provided workspace (SX_NLSF2A_WS words, e.g. in LDS)
SX_HD void sx_nlsf2a_stable_ws(i16* pAR_Q12, const i32* pNLSF, int order, i32* ws) {
    i32 invGain_Q30;
    sx_nlsf2a_ws(pAR_Q12, pNLSF, order, ws);
    i32 (*A)[SX_MAX_LPC] = (i32 (*)[SX_MAX_LPC])(ws + SX_NLSF2A_WS - 2 * SX_MAX_LPC);
    int i;
    for (i = 0; i < 20; i++) {
        if (sx_lpc_inv_pred_gain_ws(&invGain_Q30, pAR_Q12, order, A) == 1)
            sx_bwexpander(pAR_Q12, order, 65536 - sx_smulbb(10 + i, i));
        else
            break;
    }
    if (i == 20) {
        for (i = 0; i < order; i++) pAR_Q12[i] = 0;
    }
}

// SKP_Silk_NLSF2A_stable, SKP_Silk_NLSF2A_stable.c:31
SX_FN void sx_nlsf2a_stable(i16* pAR_Q12, const i32* pNLSF, int order) {
    i32 invGain_Q30;
    sx_nlsf2a(pAR_Q12, pNLSF, order);
    int i;
    for (i = 0; i < 20; i++) {
        if (sx_lpc_inv_pred_gain(&invGain_Q30, pAR_Q12, order) == 1)
            sx_bwexpander(pAR_Q12, order, 65536 - sx_smulbb(10 + i, i));
        else
            break;
    }
    if (i == 20) {
        for (i = 0; i < order; i++) pAR_Q12[i] = 0;
    }
}

// SKP_Silk_NLSF_stabilize, SKP_Silk_NLSF_stabilize.c:42   (NDeltaMin has L+1 entries)
SX_FN void sx_nlsf_stabilize(i32* NLSF_Q15, const i32* NDeltaMin_Q15, int L) {
    int loops;
    for (loops = 0; loops < 20; loops++) {
        i32 min_diff = NLSF_Q15[0] - NDeltaMin_Q15[0];
        int I = 0;
        for (int i = 1; i <= L - 1; i++) {
            i32 diff = NLSF_Q15[i] - (NLSF_Q15[i - 1] + NDeltaMin_Q15[i]);
            if (diff < min_diff) { min_diff = diff; I = i; }
        }
        i32 diff = (1 << 15) - (NLSF_Q15[L - 1] + NDeltaMin_Q15[L]);
        if (diff < min_diff) { min_diff = diff; I = L; }
        if (min_diff >= 0) return;
        if (I == 0) {
            NLSF_Q15[0] = NDeltaMin_Q15[0];
        } else if (I == L) {
            NLSF_Q15[L - 1] = (1 << 15) - NDeltaMin_Q15[L];
        } else {
            i32 min_center = 0;
            for (int k = 0; k < I; k++) min_center += NDeltaMin_Q15[k];
            min_center += NDeltaMin_Q15[I] >> 1;
            i32 max_center = 1 << 15;
            for (int k = L; k > I; k--) max_center -= NDeltaMin_Q15[k];
            max_center -= (NDeltaMin_Q15[I] - (NDeltaMin_Q15[I] >> 1));
            i32 center = sx_limit(sx_rshift_round(NLSF_Q15[I - 1] + NLSF_Q15[I], 1), min_center, max_center);
            NLSF_Q15[I - 1] = center - (NDeltaMin_Q15[I] >> 1);
            NLSF_Q15[I] = NLSF_Q15[I - 1] + NDeltaMin_Q15[I];
        }
    }
    // fall-back (SKP_Silk_NLSF_stabilize.c:117-137): insertion sort, then clamp both ways
    for (int i = 1; i < L; i++) {
        i32 value = NLSF_Q15[i];
        int j;
        for (j = i - 1; j >= 0 && value < NLSF_Q15[j]; j--) NLSF_Q15[j + 1] = NLSF_Q15[j];
        NLSF_Q15[j + 1] = value;
    }
    NLSF_Q15[0] = sx_max(NLSF_Q15[0], NDeltaMin_Q15[0]);
    for (int i = 1; i < L; i++) NLSF_Q15[i] = sx_max(NLSF_Q15[i], NLSF_Q15[i - 1] + NDeltaMin_Q15[i]);
    NLSF_Q15[L - 1] = sx_min(NLSF_Q15[L - 1], (1 << 15) - NDeltaMin_Q15[L]);
    for (int i = L - 2; i >= 0; i--) NLSF_Q15[i] = sx_min(NLSF_Q15[i], NLSF_Q15[i + 1] - NDeltaMin_Q15[i + 1]);
}

// SKP_Silk_NLSF_VQ_weights_laroia, SKP_Silk_NLSF_VQ_weights_laroia.c:40   (D even)
SX_HD void sx_nlsf_weights_laroia(i32* pW_Q6, const i32* pNLSF_Q15, int D) {
    i32 t1 = sx_max(pNLSF_Q15[0], 3);
    t1 = (1 << 21) / t1;
    i32 t2 = sx_max(pNLSF_Q15[1] - pNLSF_Q15[0], 3);
    t2 = (1 << 21) / t2;
    pW_Q6[0] = sx_min(t1 + t2, 32767);
    for (int k = 1; k < D - 1; k += 2) {
        t1 = sx_max(pNLSF_Q15[k + 1] - pNLSF_Q15[k], 3);
        t1 = (1 << 21) / t1;
        pW_Q6[k] = sx_min(t1 + t2, 32767);
        t2 = sx_max(pNLSF_Q15[k + 2] - pNLSF_Q15[k + 1], 3);
        t2 = (1 << 21) / t2;
        pW_Q6[k + 1] = sx_min(t1 + t2, 32767);
    }
    t1 = sx_max((1 << 15) - pNLSF_Q15[D - 1], 3);
    t1 = (1 << 21) / t1;
    pW_Q6[D - 1] = sx_min(t1 + t2, 32767);
}

// SKP_Silk_sum_sqr_shift, SKP_Silk_sum_sqr_shift.c:40.  The reference's result depends on whether
// the int16 pointer is 4-byte aligned (it then accumulates in sample PAIRS and tests for overflow
// once per pair); `odd_start` = 1 reproduces the "pointer & 2" branch.  Wave-uniform, serial.
SX_FN void sx_sum_sqr_shift(i32* energy, i32* shift, const i16* x, int len, int odd_start) {
    i32 nrg, nrg_tmp;
    int i, shft = 0;
    if (odd_start) { nrg = sx_smulbb(x[0], x[0]); i = 1; } else { nrg = 0; i = 0; }
    len--;
    while (i < len) {
        nrg = sx_add(nrg, sx_smulbb(x[i], x[i]));
        nrg = sx_add(nrg, sx_smulbb(x[i + 1], x[i + 1]));
        i += 2;
        if (nrg < 0) { nrg = (i32)((u32)nrg >> 2); shft = 2; break; }
    }
    for (; i < len; i += 2) {
        nrg_tmp = sx_smulbb(x[i], x[i]);
        nrg_tmp = sx_add(nrg_tmp, sx_smulbb(x[i + 1], x[i + 1]));
        nrg = (i32)((u32)nrg + ((u32)nrg_tmp >> shft));
        if (nrg < 0) { nrg = (i32)((u32)nrg >> 2); shft += 2; }
    }
    if (i == len) {
        nrg_tmp = sx_smulbb(x[i], x[i]);
        nrg = (i32)((u32)nrg + ((u32)nrg_tmp >> shft));
    }
    if (nrg & 0xC0000000) { nrg = (i32)((u32)nrg >> 2); shft += 2; }
    *shift = shft;
    *energy = nrg;
}

// the same with the length known at compile time, inlined at the call site (a lane-serial call inside SX_PAR: the loops unroll)
template <int LEN>
SX_HD void sx_sum_sqr_shift_n(i32* energy, i32* shift, const i16* x, int odd_start) {
    i32 nrg, nrg_tmp;
    int i, shft = 0;
    if (odd_start) { nrg = sx_smulbb(x[0], x[0]); i = 1; } else { nrg = 0; i = 0; }
    constexpr int len = LEN - 1;
    while (i < len) {
        nrg = sx_add(nrg, sx_smulbb(x[i], x[i]));
        nrg = sx_add(nrg, sx_smulbb(x[i + 1], x[i + 1]));
        i += 2;
        if (nrg < 0) { nrg = (i32)((u32)nrg >> 2); shft = 2; break; }
    }
    for (; i < len; i += 2) {
        nrg_tmp = sx_smulbb(x[i], x[i]);
        nrg_tmp = sx_add(nrg_tmp, sx_smulbb(x[i + 1], x[i + 1]));
        nrg = (i32)((u32)nrg + ((u32)nrg_tmp >> shft));
        if (nrg < 0) { nrg = (i32)((u32)nrg >> 2); shft += 2; }
    }
    if (i == len) {
        nrg_tmp = sx_smulbb(x[i], x[i]);
        nrg = (i32)((u32)nrg + ((u32)nrg_tmp >> shft));
    }
    if (nrg & 0xC0000000) { nrg = (i32)((u32)nrg >> 2); shft += 2; }
    *shift = shft;
    *energy = nrg;
}

// The same function for wave-uniform callers: all lanes cooperate.  The reference's loop is a state machine (nrg, shift):
// add the next pair of squares >> shift; when bit 31 gets set, nrg >>= 2 and shift += 2.  Between two such events the
// accumulation is a plain sum, so each round takes one saturating prefix scan over the remaining pairs (lane l owns C
// consecutive pairs), finds the lane where the running total first reaches 2^31, and replays only that lane's pairs.
#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64
SX_HD u32 sx_uadd_sat(u32 a, u32 b) { const u32 r = a + b; return r < a ? 0xFFFFFFFFu : r; }
SX_HD u32 wv_scan_sat(u32 v) {      // inclusive saturating prefix sum over the 64 lanes
    v = sx_uadd_sat(v, (u32)__builtin_amdgcn_update_dpp(0, (i32)v, 0x111, 0xF, 0xF, true));
    v = sx_uadd_sat(v, (u32)__builtin_amdgcn_update_dpp(0, (i32)v, 0x112, 0xF, 0xF, true));
    v = sx_uadd_sat(v, (u32)__builtin_amdgcn_update_dpp(0, (i32)v, 0x114, 0xF, 0xF, true));
    v = sx_uadd_sat(v, (u32)__builtin_amdgcn_update_dpp(0, (i32)v, 0x118, 0xF, 0xF, true));
    const u32 r0 = (u32)__builtin_amdgcn_readlane((i32)v, 15), r1 = (u32)__builtin_amdgcn_readlane((i32)v, 31),
              r2 = (u32)__builtin_amdgcn_readlane((i32)v, 47);
    const u32 s01 = sx_uadd_sat(r0, r1), s012 = sx_uadd_sat(s01, r2);
    const int row = SX_LANE >> 4;
    const u32 off = row == 0 ? 0u : (row == 1 ? r0 : (row == 2 ? s01 : s012));
    return sx_uadd_sat(v, off);
}
// (MAXC: pairs per lane the instance is built for -- a caller whose vectors are at most 128 samples long passes 1 and gets a third of the code)
template <int MAXC>
SX_HD void sx_sum_sqr_shift_wv_inl(i32* energy, i32* shift, const i16* x, int len, int odd_start) {
    const int start = odd_start ? 1 : 0;
    const int npairs = (len - start) >> 1;
    const int tail = (len - start) & 1;
    if (npairs > MAXC * 64) { sx_sum_sqr_shift(energy, shift, x, len, odd_start); return; }
    u32 nrg = odd_start ? (u32)sx_smulbb(x[0], x[0]) : 0u;
    const int C = MAXC == 1 ? 1 : (npairs + 63) >> 6;
    u32 P[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; j++) {
        const int m = SX_LANE * C + j;
        u32 v = 0;
        if (j < C && m < npairs) {
            const i32 a = x[start + 2 * m], b = x[start + 2 * m + 1];
            v = (u32)(a * a) + (u32)(b * b);
        }
        P[j] = v;
    }
    int cur = 0, shft = 0;
    for (;;) {
        u32 sl = 0;
#pragma unroll
        for (int j = 0; j < MAXC; j++) {
            const int m = SX_LANE * C + j;
            if (m >= cur) sl = sx_uadd_sat(sl, P[j] >> shft);
        }
        const u32 pre = wv_scan_sat(sl);
        const u32 tot = sx_uadd_sat(nrg, pre);
        const unsigned long long cross = __builtin_amdgcn_ballot_w64(tot >= 0x80000000u);
        if (!cross) { nrg = (u32)__builtin_amdgcn_readlane((i32)tot, 63); break; }
        const int Lc = __builtin_ctzll(cross);
        u32 base = nrg;
        if (Lc > 0) base += (u32)__builtin_amdgcn_readlane((i32)pre, Lc - 1);
        bool found = false;
#pragma unroll
        for (int j = 0; j < MAXC; j++) {
            const int m = Lc * C + j;
            const u32 pj = (u32)__builtin_amdgcn_readlane((i32)P[j], Lc);
            if (!found && j < C && m >= cur) {
                base += pj >> shft;
                if (base >= 0x80000000u) { found = true; cur = m + 1; }
            }
        }
        nrg = base >> 2;
        shft += 2;
    }
    if (tail) nrg += (u32)sx_smulbb(x[len - 1], x[len - 1]) >> shft;
    if (nrg & 0xC0000000u) { nrg >>= 2; shft += 2; }
    *shift = shft;
    *energy = (i32)nrg;
}
// (a call of its own; short vectors in a loop of the caller: the _inl form)
template <int MAXC = 4>
SX_FN void sx_sum_sqr_shift_wv(i32* energy, i32* shift, const i16* x, int len, int odd_start) { sx_sum_sqr_shift_wv_inl<MAXC>(energy, shift, x, len, odd_start); }
#else
template <int MAXC = 4>
SX_HD void sx_sum_sqr_shift_wv(i32* energy, i32* shift, const i16* x, int len, int odd_start) { sx_sum_sqr_shift(energy, shift, x, len, odd_start); }
template <int MAXC>
SX_HD void sx_sum_sqr_shift_wv_inl(i32* energy, i32* shift, const i16* x, int len, int odd_start) { sx_sum_sqr_shift(energy, shift, x, len, odd_start); }
#endif

// SKP_Silk_LPC_analysis_filter, SKP_Silk_MA.c:70 with a ZERO initial state, as a direct-form FIR:
//   out[k] = sat16( rshift_round( sub_sat32( in[k] << 12, sum_j B[j] * in[k-1-j] ), 12 ) ),  in[<0] = 0
// (the reference's delay-line update is a plain shift, and its SMLABB accumulation wraps, so the
// tap order is irrelevant).  Wave-parallel over k; caller must wv_sync() afterwards.
SX_HD void sx_lpc_analysis_filter_zero_state(const i16* in, const i16* B, i16* out, int len, int order) {
    SX_PAR(k, len) {
        i32 acc = 0;
        for (int j = 0; j < order; j++) {
            int t = k - 1 - j;
            if (t >= 0) acc = sx_smlabb(acc, in[t], B[j]);
        }
        i32 o = sx_sub_sat32(sx_shl((i32)in[k], 12), acc);
        out[k] = (i16)sx_sat16(sx_rshift_round(o, 12));
    }
}

// SKP_Silk_sigm_Q15, SKP_Silk_sigm_Q15.c:54
SX_HD i32 sx_sigm_Q15(i32 in_Q5) {
    if (in_Q5 < 0) {
        in_Q5 = -in_Q5;
        if (in_Q5 >= 6 * 32) return 0;
        int ind = in_Q5 >> 5;
        return T_sigm_neg_Q15[ind] - sx_smulbb(T_sigm_slope_Q10[ind], in_Q5 & 0x1F);
    }
    if (in_Q5 >= 6 * 32) return 32767;
    int ind = in_Q5 >> 5;
    return T_sigm_pos_Q15[ind] + sx_smulbb(T_sigm_slope_Q10[ind], in_Q5 & 0x1F);
}

#ifdef SX_LANE_STREAM
// ---- one vector per 16-lane row (gfx950 build): lane j of a row holds element j in a register; the reference's inner loops over the
// element index disappear, reversals / broadcasts are lane gathers inside the row, neighbour recursions are DPP row shifts.  (Lane
// exchanges stay outside conditionals: every lane must take part.)
#define SX_ROWB(v, jj) __shfl((v), (SX_LANE & 48) | (jj), 64)                 // element jj of the own row, in all of its lanes
#define SX_ROWG(v, idx) __shfl((v), (SX_LANE & 48) | ((idx) & 15), 64)          // element idx (per lane) of the own row
#define SX_ROW_NEXT(v) SX_DPP_((v), 0x101)                                      // element j + 1 (row_shl:1; 0 beyond the row)

// SKP_Silk_NLSF_VQ_weights_laroia (sx_nlsf_weights_laroia above) with one division per lane: weight k is the sum of the inverses of the two
// intervals next to coefficient k, lane k <= D <= 15 of a row divides for the interval below coefficient k (the last one for the interval up to 2^15)
SX_HD void sx_row_nlsf_weights_laroia(i32* pW_Q6, const i32* pNLSF_Q15, int D, bool row_active) {
    const int k = SX_LANE & 15;
    i32 inv = 0;
    if (row_active && k <= D) {
        const i32 lo = k > 0 ? pNLSF_Q15[k - 1] : 0, hi = k < D ? pNLSF_Q15[k] : (1 << 15);
        inv = (1 << 21) / sx_max(hi - lo, 3);
    }
    const i32 nx = SX_ROW_NEXT(inv);
    if (row_active && k < D) pW_Q6[k] = sx_min(inv + nx, 32767);
}

// SKP_Silk_LPC_inverse_pred_gain_Q24 (SKP_Silk_LPC_inv_pred_gain.c:134 -> :43): a = coefficient j in Q16; returns invGain_Q30 as
// the reference leaves it (also when it bails out on an unstable filter)
template <int ORDER>
SX_HD i32 sx_row_inv_pred_gain_Q16_n(i32 a, bool* unstable) {
    const i32 A_LIMIT = 65520;
    const int j = SX_LANE & 15;
    i32 inv = 1 << 30;
    bool done = false;
    for (int k = ORDER - 1; k > 0; k--) {
        const i32 ak = SX_ROWB(a, k);
        done = done || ak > A_LIMIT || ak < -A_LIMIT;
        const i32 rc_Q31 = sx_neg(sx_shl(ak, 31 - 16));
        const i32 m1 = (SX_I32_MAX >> 1) - sx_smmul(rc_Q31, rc_Q31);
        i32 m2 = sx_inverse32_varQ_pos(m1, 46);                  // (m1 >= 2^19 while the filter is stable: |ak| <= A_LIMIT)
        const i32 inv_n = sx_shl(sx_smmul(inv, m1), 2);
        const int headrm = sx_clz32(m2) - 1;
        m2 = sx_shl(m2, headrm);
        const i32 ar = SX_ROWG(a, k - 1 - j);
        const i32 tmp = sx_sub(a, sx_shl(sx_smmul(ar, rc_Q31), 1));
        const i32 an = sx_shl(sx_smmul(tmp, m2), 16 - headrm);
        if (!done) {
            inv = inv_n;
            if (j < k) a = an;
        }
    }
    const i32 a0 = SX_ROWB(a, 0);
    done = done || a0 > A_LIMIT || a0 < -A_LIMIT;
    if (!done) {
        const i32 rc_Q31 = sx_neg(sx_shl(a0, 31 - 16));
        const i32 m1 = (SX_I32_MAX >> 1) - sx_smmul(rc_Q31, rc_Q31);
        inv = sx_shl(sx_smmul(inv, m1), 2);
    }
    *unstable = done;
    return inv;
}

// SKP_Silk_NLSF2A_stable (SKP_Silk_NLSF2A_stable.c:31, SKP_Silk_NLSF2A.c:59) for one vector per 16-lane row, the common case only:
// lane j of the row ends up with coefficient j.  The two polynomials are built side by side -- P in lanes 0 .. dd, Q in lanes 8 ..
// 8 + dd of the row (dd + 1 <= 8) -- one step of find_poly per k for all n at once (the reference walks n downwards, so every update
// reads values of the previous step).  Returns false where the reference would start correcting -- a coefficient beyond int16
// (NLSF2A.c:93) or an unstable filter (NLSF2A_stable.c:44) -- the caller then runs the serial restatement for that call.
// one polynomial of NLSF2A_find_poly in the lanes n = 0 .. dd of `lane n`'s row: h = 0: P (even cosines), 1: Q (odd); `n` is the lane's
// coefficient index, lanes past dd compute zeros
template <int DD>
SX_HD i32 sx_row_find_poly(i32 cosv, int n, int h) {
    const i32 c0 = SX_ROWG(cosv, h);                          // (lane exchanges stay outside conditionals: every lane must take part)
    i32 out = n == 0 ? (1 << 20) : (n == 1 ? sx_neg(c0) : 0);
#pragma unroll
    for (int k = 1; k < DD; k++) {
        const i32 ftmp = SX_ROWG(cosv, 2 * k + h);
        const i32 o1 = SX_DPP_(out, 0x111), o2 = SX_DPP_(out, 0x112);          // out[n - 1], out[n - 2] (row_shr:1, :2)
        const i32 R = (i32)sx_rshift_round64(sx_smull(ftmp, o1), 20);
        i32 nv = out;
        if (n >= 2 && n <= k) nv = sx_add(out, sx_sub(o2, R));
        if (n == k + 1) nv = sx_sub(sx_shl(o2, 1), R);
        if (n == 1) nv = sx_sub(out, ftmp);
        out = nv;
    }
    return out;
}
template <int ORDER>
SX_HD bool sx_row_nlsf2a_stable_n(i16* pAR_Q12, const i32* pNLSF) {
    static_assert(ORDER % 2 == 0 && ORDER <= 16, "one vector per 16-lane row");
    constexpr int dd = ORDER / 2;
    constexpr bool SIDE_BY_SIDE = dd + 1 <= 8;                // both polynomials in one pass: P in lanes 0 .. dd, Q in lanes 8 .. 8 + dd
    const int j = SX_LANE & 15;
    i32 cosv = 0;
    if (j < ORDER) {
        const i32 v = pNLSF[j];
        const i32 f_int = v >> 8, f_frac = v - (f_int << 8);
        const i32 cos_val = T_lsf_cos_Q12[f_int];
        cosv = sx_add(sx_shl(cos_val, 8), sx_mul(T_lsf_cos_Q12[f_int + 1] - cos_val, f_frac));
    }
    // a32[k] = -rshift_round(Ptmp + Qtmp, 9), a32[d - 1 - k] = rshift_round(Qtmp - Ptmp, 9), Ptmp = P[k + 1] + P[k], Qtmp = Q[k + 1] - Q[k]
    const int kk = j < dd ? j : ORDER - 1 - j;
    i32 Ptmp, Qtmp;
    if constexpr (SIDE_BY_SIDE) {
        const i32 out = sx_row_find_poly<dd>(cosv, j & 7, j >> 3);
        Ptmp = sx_add(SX_ROWG(out, kk + 1), SX_ROWG(out, kk));
        Qtmp = sx_sub(SX_ROWG(out, 8 + kk + 1), SX_ROWG(out, 8 + kk));
    } else {                                                  // (order 16: nine coefficients each, one polynomial after the other)
        const i32 outP = sx_row_find_poly<dd>(cosv, j, 0), outQ = sx_row_find_poly<dd>(cosv, j, 1);
        Ptmp = sx_add(SX_ROWG(outP, kk + 1), SX_ROWG(outP, kk));
        Qtmp = sx_sub(SX_ROWG(outQ, kk + 1), SX_ROWG(outQ, kk));
    }
    i32 a32 = j < dd ? sx_neg(sx_rshift_round(sx_add(Ptmp, Qtmp), 9)) : sx_rshift_round(sx_sub(Qtmp, Ptmp), 9);
    if (j >= ORDER) a32 = 0;
    i32 maxabs = sx_abs(a32);
    SX_ROW_REDUCE(maxabs, (t_ > maxabs ? t_ : maxabs))
    bool unstable;
    (void)sx_row_inv_pred_gain_Q16_n<ORDER>(sx_shl((i32)(i16)a32, 4), &unstable);
    if (j < ORDER) pAR_Q12[j] = (i16)a32;
    return maxabs <= 32767 && !unstable;
}
#define SX_HAVE_ROW_NLSF2A 1
SX_HD bool sx_row_nlsf2a_stable(i16* pAR_Q12, const i32* pNLSF) { return sx_row_nlsf2a_stable_n<SX_LPC>(pAR_Q12, pNLSF); }
#endif
