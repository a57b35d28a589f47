// solo_enc_analysis.h -- noise-shape analysis, prefilter, LTP/LPC prediction analysis, NLSF MSVQ,
// gain processing.  Rows E5d-E5g of SURVEY.md section 8(a).  Reference (JC1_SDK_SRC_ARM/src/libSATECodec/):
//   SKP_Silk_noise_shape_analysis_FIX.c:33-531, SKP_Silk_warped_autocorrelation_FIX.c:36, SKP_Silk_schur64.c:42,
//   SKP_Silk_k2a_Q16.c:40, SKP_Silk_prefilter_FIX.c:43-224, SKP_Silk_find_pred_coefs_FIX.c:31,
//   SKP_Silk_find_LTP_FIX.c:39-243, SKP_Silk_corrMatrix_FIX.c:35-152, SKP_Silk_solve_LS_FIX.c:71-241,
//   SKP_Silk_regularize_correlations_FIX.c:31, SKP_Silk_residual_energy16_FIX.c:31, SKP_Silk_quant_LTP_gains_FIX.c:30,
//   SKP_Silk_VQ_nearest_neighbor_FIX.c:31, SKP_Silk_LTP_scale_ctrl_FIX.c:39, SKP_Silk_LTP_analysis_filter_FIX.c:30,
//   SKP_Silk_find_LPC_FIX.c:32, SKP_Silk_burg_modified.c:49, SKP_Silk_A2NLSF.c:46-287, SKP_Silk_process_NLSFs_FIX.c:31,
//   SKP_Silk_NLSF_MSVQ_encode_FIX.c:33, SKP_Silk_NLSF_VQ_rate_distortion_FIX.c:31, SKP_Silk_NLSF_VQ_sum_error_FIX.c:33,
//   SKP_Silk_residual_energy_FIX.c:32, SKP_Silk_process_gains_FIX.c:32, SKP_Silk_gain_quant.c:42
#pragma once
#include "solo_enc_front.h"
#include "solo_dec.h"   // sx_nlsf_msvq_decode (shared with the decoder)

// ---------------------------------------------------------------------------------------------------
// noise shape analysis
// ---------------------------------------------------------------------------------------------------
// SKP_Silk_warped_autocorrelation_FIX, SKP_Silk_warped_autocorrelation_FIX.c:36 (order 16, serial chain)
SX_FN void sx_warped_autocorr(i32* corr, i32* scale, const i16* input, i32 warping_Q16, int length) {
    const int order = SX_SHAPE_ORDER;
    i32 state[SX_SHAPE_ORDER + 1];
    i64 corr_QC[SX_SHAPE_ORDER + 1];
    for (int i = 0; i <= order; i++) { state[i] = 0; corr_QC[i] = 0; }
    for (int n = 0; n < length; n++) {
        i32 tmp1 = sx_shl((i32)input[n], 14), tmp2;
        for (int i = 0; i < order; i += 2) {
            tmp2 = sx_smlawb(state[i], state[i + 1] - tmp1, warping_Q16);
            state[i] = tmp1;
            corr_QC[i] += sx_smull(tmp1, state[0]) >> 18;
            tmp1 = sx_smlawb(state[i + 1], state[i + 2] - tmp2, warping_Q16);
            state[i + 1] = tmp2;
            corr_QC[i + 1] += sx_smull(tmp2, state[0]) >> 18;
        }
        state[order] = tmp1;
        corr_QC[order] += sx_smull(tmp1, state[0]) >> 18;
    }
    int lsh = sx_clz64(corr_QC[0]) - 35;
    lsh = sx_limit(lsh, -12 - 10, 30 - 10);
    *scale = -(10 + lsh);
    if (lsh >= 0) {
        for (int i = 0; i <= order; i++) corr[i] = (i32)(corr_QC[i] << lsh);
    } else {
        for (int i = 0; i <= order; i++) corr[i] = (i32)(corr_QC[i] >> (-lsh));
    }
}

// SKP_Silk_schur64, SKP_Silk_schur64.c:42
SX_HD i32 sx_schur64(i32* rc_Q16, const i32* c, int order, i32 (*C)[2]) {
    if (c[0] <= 0) {
        for (int k = 0; k < order; k++) rc_Q16[k] = 0;
        return 0;
    }
    for (int k = 0; k < order + 1; k++) C[k][0] = C[k][1] = c[k];
    for (int k = 0; k < order; k++) {
        i32 rc_Q31 = sx_div32_varQ(sx_neg(C[k + 1][0]), C[0][1], 31);
        rc_Q16[k] = sx_rshift_round(rc_Q31, 15);
        for (int n = 0; n < order - k; n++) {
            i32 t1 = C[n + k + 1][0], t2 = C[n][1];
            C[n + k + 1][0] = sx_add(t1, sx_smmul(sx_shl(t2, 1), rc_Q31));
            C[n][1] = sx_add(t2, sx_smmul(sx_shl(t1, 1), rc_Q31));
        }
    }
    return C[0][1];
}

// SKP_Silk_k2a_Q16, SKP_Silk_k2a_Q16.c:40
SX_HD void sx_k2a_Q16(i32* A_Q24, const i32* rc_Q16, int order, i32* Atmp) {
    for (int k = 0; k < order; k++) {
        for (int n = 0; n < k; n++) Atmp[n] = A_Q24[n];
        for (int n = 0; n < k; n++) A_Q24[n] = sx_smlaww(A_Q24[n], Atmp[k - n - 1], rc_Q16[k]);
        A_Q24[k] = sx_neg(sx_shl(rc_Q16[k], 8));
    }
}

// warped_gain, noise_shape_analysis_FIX.c:33
SX_HD i32 sx_warped_gain(const i32* coefs_Q24, i32 lambda_Q16, int order) {
    lambda_Q16 = -lambda_Q16;
    i32 gain_Q24 = coefs_Q24[order - 1];
    for (int i = order - 2; i >= 0; i--) gain_Q24 = sx_smlawb(coefs_Q24[i], gain_Q24, lambda_Q16);
    gain_Q24 = sx_smlawb(K_1p0_Q24, gain_Q24, -lambda_Q16);
    return sx_inverse32_varQ(gain_Q24, 40);
}

// limit_warped_coefs, noise_shape_analysis_FIX.c:52
SX_FN1 void sx_limit_warped_coefs(i32* syn, i32* ana, i32 lambda_Q16, i32 limit_Q24, int order) {
    int ind = 0;
    i32 nom_Q16, den_Q24, gain_syn_Q16, gain_ana_Q16;
    lambda_Q16 = -lambda_Q16;
    for (int i = order - 1; i > 0; i--) {
        syn[i - 1] = sx_smlawb(syn[i - 1], syn[i], lambda_Q16);
        ana[i - 1] = sx_smlawb(ana[i - 1], ana[i], lambda_Q16);
    }
    lambda_Q16 = -lambda_Q16;
    nom_Q16 = sx_smlawb(K_1p0_Q16, -lambda_Q16, lambda_Q16);
    den_Q24 = sx_smlawb(K_1p0_Q24, syn[0], lambda_Q16);
    gain_syn_Q16 = sx_div32_varQ(nom_Q16, den_Q24, 24);
    den_Q24 = sx_smlawb(K_1p0_Q24, ana[0], lambda_Q16);
    gain_ana_Q16 = sx_div32_varQ(nom_Q16, den_Q24, 24);
    for (int i = 0; i < order; i++) {
        syn[i] = sx_smulww(gain_syn_Q16, syn[i]);
        ana[i] = sx_smulww(gain_ana_Q16, ana[i]);
    }
    for (int iter = 0; iter < 10; iter++) {
        i32 maxabs_Q24 = -1;
        for (int i = 0; i < order; i++) {
            i32 a = (syn[i] ^ (syn[i] >> 31)) - (syn[i] >> 31), b = (ana[i] ^ (ana[i] >> 31)) - (ana[i] >> 31);
            i32 tmp = sx_max(a, b);
            if (tmp > maxabs_Q24) { maxabs_Q24 = tmp; ind = i; }
        }
        if (maxabs_Q24 <= limit_Q24) return;
        for (int i = 1; i < order; i++) {
            syn[i - 1] = sx_smlawb(syn[i - 1], syn[i], lambda_Q16);
            ana[i - 1] = sx_smlawb(ana[i - 1], ana[i], lambda_Q16);
        }
        gain_syn_Q16 = sx_inverse32_varQ(gain_syn_Q16, 32);
        gain_ana_Q16 = sx_inverse32_varQ(gain_ana_Q16, 32);
        for (int i = 0; i < order; i++) {
            syn[i] = sx_smulww(gain_syn_Q16, syn[i]);
            ana[i] = sx_smulww(gain_ana_Q16, ana[i]);
        }
        i32 chirp_Q16 = K_0p99_Q16 - sx_div32_varQ(sx_smulwb(maxabs_Q24 - limit_Q24, sx_smlabb(K_0p8_Q10, K_0p1_Q10, iter)),
                                                  sx_mul(maxabs_Q24, ind + 1), 22);
        sx_bwexpander_32(syn, order, chirp_Q16);
        sx_bwexpander_32(ana, order, chirp_Q16);
        lambda_Q16 = -lambda_Q16;
        for (int i = order - 1; i > 0; i--) {
            syn[i - 1] = sx_smlawb(syn[i - 1], syn[i], lambda_Q16);
            ana[i - 1] = sx_smlawb(ana[i - 1], ana[i], lambda_Q16);
        }
        lambda_Q16 = -lambda_Q16;
        nom_Q16 = sx_smlawb(K_1p0_Q16, -lambda_Q16, lambda_Q16);
        den_Q24 = sx_smlawb(K_1p0_Q24, syn[0], lambda_Q16);
        gain_syn_Q16 = sx_div32_varQ(nom_Q16, den_Q24, 24);
        den_Q24 = sx_smlawb(K_1p0_Q24, ana[0], lambda_Q16);
        gain_ana_Q16 = sx_div32_varQ(nom_Q16, den_Q24, 24);
        for (int i = 0; i < order; i++) {
            syn[i] = sx_smulww(gain_syn_Q16, syn[i]);
            ana[i] = sx_smulww(gain_ana_Q16, ana[i]);
        }
    }
}

// LDS scratch of the noise-shape analysis: the four subframes are analysed side by side
struct SxShapeWork {
    i16 xw[SX_NB_SUBFR][SX_SHAPE_WIN];                    // windowed input of the four subframes
    i32 win[2][SX_LA_SHAPE];                              // rising / falling sine-window gains (Q16)
    i32 o[2][SX_NB_SUBFR][SX_SHAPE_ORDER];                // all-pass section outputs in flight (double buffered)
    i64 cq[SX_NB_SUBFR][SX_SHAPE_ORDER + 1];
    i32 corr[SX_NB_SUBFR][SX_SHAPE_ORDER + 1];
    i32 scale[SX_NB_SUBFR];
    i32 C[SX_NB_SUBFR][SX_SHAPE_ORDER + 1][2];
    i32 rc[SX_NB_SUBFR][SX_SHAPE_ORDER];
    i32 A2[SX_NB_SUBFR][SX_SHAPE_ORDER], A1[SX_NB_SUBFR][SX_SHAPE_ORDER], At[SX_NB_SUBFR][SX_SHAPE_ORDER];
    i32 inv[SX_NB_SUBFR][2][SX_MAX_LPC];
};

#if SX_NLANES == 1
#define SX_NP64 64
#define SX_PL(l) (l)
#else
#define SX_NP64 1
#define SX_PL(l) 0
#endif

// SKP_Silk_warped_autocorrelation_FIX (SKP_Silk_warped_autocorrelation_FIX.c:36) for the four subframe windows at once.
// The reference walks 16 first-order all-pass sections per sample; section j at sample n only needs section j-1 at
// samples n, n-1 and itself at n-1, so lane (k, j) runs section j of subframe k skewed by j samples: 120 + 15 steps
// instead of 4 x 120 x 16.  Each lane accumulates correlation j of its subframe in 64 bits; section outputs move to the
// neighbour lane through a double-buffered LDS row.
SX_HD void sx_warped_autocorr4(SxShapeWork* sw, i32 warping_Q16) {
#ifdef SX_LANE_STREAM
    // row k = subframe, lane j of the row = all-pass section j, skewed by j samples; section outputs move to the next lane with
    // a DPP row shift (registers only), every lane prefetches its own x(n) a step ahead
    i64 acc_l = 0, acc16_l = 0;
    {
        const int k = SX_LANE >> 4, j = SX_LANE & 15;
        const i32 lam = sx_pre16(warping_Q16);
        i32 pin = 0, pout = 0, out = 0;
        i32 xn = j == 0 ? (i32)sw->xw[k][0] : 0;
        // steps 15 .. SX_SHAPE_WIN - 1: every section holds a sample of the window; section 0's input comes in through the DPP move's
        // `old` operand, every lane accumulates the last correlation (only section 15's is kept).  The first and last 15 steps are the
        // same body with the updates under the section's range check (loads outside the window: never used)
#define SX_WA_BODY(ACTIVE, LAST)                                                                                \
        {   const i32 xcur = xn;                                                                                \
            xn = (i32)xp[t];                                                                                    \
            const i32 x0 = sx_shl(xcur, 14);                                                                    \
            const i32 in = __builtin_amdgcn_update_dpp(x0, out, 0x111, 0xF, 0xF, false);                         \
            if (ACTIVE) {                                                                                       \
                const i32 o = sx_smlaw_pre(pin, pout - in, lam);                                                \
                acc_l += sx_smull(in, x0) >> 18;                                                                \
                if (LAST) acc16_l += sx_smull(o, x0) >> 18;                                                     \
                pin = in; pout = o; out = o;                                                                    \
            }                                                                                                   \
        }
        {
            const i16* xp = &sw->xw[k][0] - j + 1;                      // xp[t] = x(n + 1) of step t
            for (int t = 0; t < SX_SHAPE_ORDER - 1; t++) SX_WA_BODY(j <= t, false)                      // (section 15 joins at step 15)
#pragma unroll 5
            for (int t = SX_SHAPE_ORDER - 1; t < SX_SHAPE_WIN; t++) SX_WA_BODY(true, true)
            for (int t = SX_SHAPE_WIN; t < SX_SHAPE_WIN + SX_SHAPE_ORDER - 1; t++) SX_WA_BODY(j > t - SX_SHAPE_WIN, true)
        }
#undef SX_WA_BODY
    }
    {
        const int k = SX_LANE >> 4, j = SX_LANE & 15;
        sw->cq[k][j] = acc_l;
        if (j == SX_SHAPE_ORDER - 1) sw->cq[k][SX_SHAPE_ORDER] = acc16_l;
    }
#else
    i32 pin[SX_NP64], pout[SX_NP64];
    i64 acc[SX_NP64], acc16[SX_NP64];
    for (int a = 0; a < SX_NP64; a++) { pin[a] = 0; pout[a] = 0; acc[a] = 0; acc16[a] = 0; }
    for (int t = 0; t < SX_SHAPE_WIN + SX_SHAPE_ORDER - 1; t++) {
        SX_PAR(l, 64) {
            const int k = l >> 4, j = l & 15, n = t - j, pl = SX_PL(l);
            if (n >= 0 && n < SX_SHAPE_WIN) {
                const i32 x0 = sx_shl((i32)sw->xw[k][n], 14);
                const i32 in = j == 0 ? x0 : sw->o[(t - 1) & 1][k][j - 1];
                const i32 out = sx_smlawb(pin[pl], pout[pl] - in, warping_Q16);
                acc[pl] += sx_smull(in, x0) >> 18;
                if (j == SX_SHAPE_ORDER - 1) acc16[pl] += sx_smull(out, x0) >> 18;
                pin[pl] = in;
                pout[pl] = out;
                sw->o[t & 1][k][j] = out;
            }
        }
        wv_sync();
    }
    SX_PAR(l, 64) {
        const int k = l >> 4, j = l & 15, pl = SX_PL(l);
        sw->cq[k][j] = acc[pl];
        if (j == SX_SHAPE_ORDER - 1) sw->cq[k][SX_SHAPE_ORDER] = acc16[pl];
    }
#endif
    wv_sync();
    SX_PAR(l, SX_NB_SUBFR * (SX_SHAPE_ORDER + 1)) {
        const int k = l / (SX_SHAPE_ORDER + 1), j = l - k * (SX_SHAPE_ORDER + 1);
        int lsh = sx_clz64(sw->cq[k][0]) - 35;
        lsh = sx_limit(lsh, -12 - 10, 30 - 10);
        if (j == 0) sw->scale[k] = -(10 + lsh);
        sw->corr[k][j] = lsh >= 0 ? (i32)(sw->cq[k][j] << lsh) : (i32)(sw->cq[k][j] >> (-lsh));
    }
    wv_sync();
}

// SKP_Silk_LPC_inverse_pred_gain_Q24 (SKP_Silk_LPC_inv_pred_gain.c:134) with caller-provided step-down rows
SX_HD int sx_lpc_inv_pred_gain_Q24_ws(i32* invGain_Q30, const i32* A_Q24, int order, i32 (*A)[SX_MAX_LPC]) {
    for (int k = 0; k < order; k++) A[order & 1][k] = sx_rshift_round(A_Q24[k], 8);
    return sx_lpc_inv_pred_gain_QA(invGain_Q30, A, order);
}

// float island of the "fixed point" encoder (noise_shape_analysis_FIX.c:407): IEEE single division
SX_HD float sx_fdiv(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fdiv_rn(a, b);
#else
    volatile float r = a / b;
    return r;
#endif
}

#ifdef SX_LANE_STREAM
// ---- order-16 recursions of the noise-shape analysis, one subframe per 16-lane row -------------------------------------------
// Lane j of row s holds element j of subframe s's coefficient vector in a register; the reference's inner loops over the
// element index disappear (one step per outer iteration), reversals / broadcasts are lane gathers inside the row, neighbour
// recursions are DPP row shifts.  Values that the reference holds in scalars are computed redundantly by all lanes of the row.

SX_HD i32 sx_row_inv_pred_gain_Q16(i32 a) {
    bool unstable;
    return sx_row_inv_pred_gain_Q16_n<SX_SHAPE_ORDER>(a, &unstable);
}


// x[i-1] = smlawb(x[i-1], x[i], lambda) for i = 15 .. 1 (every element sees its already updated upper neighbour), two vectors
#define SX_ROW_SUFFIX2(xa, xb, lam_)                                                                                    \
    for (int step_ = 0; step_ < SX_SHAPE_ORDER - 1; step_++) {                                                           \
        const i32 ta_ = SX_ROW_NEXT(xa), tb_ = SX_ROW_NEXT(xb);                                                          \
        if (j == SX_SHAPE_ORDER - 2 - step_) { xa = sx_smlawb(xa, ta_, lam_); xb = sx_smlawb(xb, tb_, lam_); }           \
    }
// SKP_Silk_bwexpander_32 with chirp (uniform in the row): element i is scaled by the i-th iterate of tmp <- smulww(chirp, tmp)
#define SX_ROW_BWE2(xa, xb, chirp_)                                                                                     \
    {                                                                                                                    \
        i32 t_ = (chirp_), m_ = t_;                                                                                      \
        for (int i_ = 1; i_ < SX_SHAPE_ORDER; i_++) { t_ = sx_smulww((chirp_), t_); if (j == i_) m_ = t_; }             \
        xa = sx_smulww(xa, m_); xb = sx_smulww(xb, m_);                                                                  \
    }

// Schur recursion .. coefficient limiting of noise_shape_analysis_FIX.c:339-399 for the four subframes at once
SX_FN1 void sx_shape_rows(SxShapeWork* sw, SxEncCtrl* c, i32 warping_Q16, i32 BWExp1_Q16, i32 BWExp2_Q16) {
    SX_IN_LDS(sw); SX_IN_LDS(c);
    const int s = SX_LANE >> 4, j = SX_LANE & 15;
    const i32* auto_corr = sw->corr[s];
    i32 c0 = auto_corr[0];
    c0 = sx_add(c0, sx_max(sx_smulwb(c0 >> 4, K_SHAPE_WHITE_NOISE_FRACTION_Q20), 1));
    // ---- SKP_Silk_schur64: C1 = C[j][1]; D = C[j + k + 1][0] at step k (the row is shifted down by one element per step)
    const bool valid = c0 > 0;
    i32 C1 = j == 0 ? c0 : auto_corr[j];
    i32 D = auto_corr[j + 1];
    i32 rcv = 0;                                     // rc_Q16[j]
    for (int k = 0; k < SX_SHAPE_ORDER; k++) {
        const i32 b0 = SX_ROWB(D, 0), c01 = SX_ROWB(C1, 0);
        const i32 rc_Q31 = sx_div32_varQ(sx_neg(b0), c01, 31);
        if (j == k) rcv = sx_rshift_round(rc_Q31, 15);
        const i32 t1 = D, t2 = C1;
        const bool in = j < SX_SHAPE_ORDER - k;
        const i32 Dn = in ? sx_add(t1, sx_smmul(sx_shl(t2, 1), rc_Q31)) : t1;
        if (in) C1 = sx_add(t2, sx_smmul(sx_shl(t1, 1), rc_Q31));
        D = SX_ROW_NEXT(Dn);
    }
    i32 nrg = SX_ROWB(C1, 0);
    if (!valid) { rcv = 0; nrg = 0; }
    // ---- SKP_Silk_k2a_Q16
    i32 A2 = 0;
    for (int k = 0; k < SX_SHAPE_ORDER; k++) {
        const i32 rck = SX_ROWB(rcv, k);
        const i32 g = SX_ROWG(A2, k - 1 - j);
        if (j < k) A2 = sx_smlaww(A2, g, rck);
        if (j == k) A2 = sx_neg(sx_shl(rck, 8));
    }
    // ---- gain
    int Qnrg = -sw->scale[s];
    if (Qnrg & 1) { Qnrg -= 1; nrg >>= 1; }
    const i32 sq = sx_sqrt_approx(nrg);
    Qnrg >>= 1;
    i32 g = sx_lshift_sat32(sq, 16 - Qnrg);
    {   // warped_gain (noise_shape_analysis_FIX.c:33): Horner from the top coefficient down
        const i32 lam = -warping_Q16;
        i32 G = A2;
        for (int step = 0; step < SX_SHAPE_ORDER - 1; step++) {
            const i32 t = SX_ROW_NEXT(G);
            if (j == SX_SHAPE_ORDER - 2 - step) G = sx_smlawb(A2, t, lam);
        }
        i32 gain_Q24 = SX_ROWB(G, 0);
        gain_Q24 = sx_smlawb(K_1p0_Q24, gain_Q24, -lam);
        g = sx_smulww(g, sx_inverse32_varQ(gain_Q24, 40));
        if (g < 0) g = SX_I32_MAX;
    }
    if (j == 0) c->Gains_Q16[s] = g;
    // ---- bandwidth expansion: AR2 by BWExp2, AR1 = AR2 further by BWExp1
    i32 A1;
    {
        i32 dummy = 0;
        SX_ROW_BWE2(A2, dummy, BWExp2_Q16)
        A1 = A2;
        SX_ROW_BWE2(A1, dummy, BWExp1_Q16)
    }
    // ---- pre-gain from the two inverse prediction gains
    {
        i32 pre_nrg_Q30 = sx_row_inv_pred_gain_Q16(sx_rshift_round(A2, 8));
        const i32 nrg1 = sx_row_inv_pred_gain_Q16(sx_rshift_round(A1, 8));
        pre_nrg_Q30 = sx_shl(sx_smulwb(pre_nrg_Q30, K_0p7_Q15), 1);
        if (j == 0) c->GainsPre_Q14[s] = K_0p3_Q14 + sx_div32_varQ(pre_nrg_Q30, nrg1, 14);
    }
    // ---- limit_warped_coefs (noise_shape_analysis_FIX.c:52): syn = A2, ana = A1
    {
        const i32 limit_Q24 = K_3p999_Q24;
        i32 lam = -warping_Q16;
        SX_ROW_SUFFIX2(A2, A1, lam)
        lam = -lam;
        i32 nom_Q16 = sx_smlawb(K_1p0_Q16, -lam, lam);
        i32 gs = sx_div32_varQ(nom_Q16, sx_smlawb(K_1p0_Q24, SX_ROWB(A2, 0), lam), 24);
        i32 ga = sx_div32_varQ(nom_Q16, sx_smlawb(K_1p0_Q24, SX_ROWB(A1, 0), lam), 24);
        A2 = sx_smulww(gs, A2);
        A1 = sx_smulww(ga, A1);
        bool act = true;                             // this row still iterates
        for (int iter = 0; iter < 10; iter++) {
            // largest |coefficient| of the row and its first position
            const i32 aa = (A2 ^ (A2 >> 31)) - (A2 >> 31), ab = (A1 ^ (A1 >> 31)) - (A1 >> 31);
            i32 bv = sx_max(aa, ab), bi = j;
            SX_ARG_STEP(0xB1, >) SX_ARG_STEP(0x4E, >) SX_ARG_STEP(0x141, >) SX_ARG_STEP(0x140, >)
            const i32 maxabs_Q24 = bv;
            const int ind = bi;
            act = act && maxabs_Q24 > limit_Q24;
            if (__builtin_amdgcn_ballot_w64(act) == 0) break;
            i32 n2 = A2, n1 = A1;
            {   // x[i-1] = smlawb(x[i-1], x[i], lambda) for i = 1 .. 15: every element sees its OLD upper neighbour
                const i32 t2 = SX_ROW_NEXT(n2), t1 = SX_ROW_NEXT(n1);
                if (j < SX_SHAPE_ORDER - 1) { n2 = sx_smlawb(n2, t2, lam); n1 = sx_smlawb(n1, t1, lam); }
            }
            gs = sx_inverse32_varQ(gs, 32);
            ga = sx_inverse32_varQ(ga, 32);
            n2 = sx_smulww(gs, n2);
            n1 = sx_smulww(ga, n1);
            const i32 chirp_Q16 = K_0p99_Q16 - sx_div32_varQ(sx_smulwb(maxabs_Q24 - limit_Q24, sx_smlabb(K_0p8_Q10, K_0p1_Q10, iter)),
                                                            sx_mul(maxabs_Q24, ind + 1), 22);
            SX_ROW_BWE2(n2, n1, chirp_Q16)
            lam = -lam;
            SX_ROW_SUFFIX2(n2, n1, lam)
            lam = -lam;
            nom_Q16 = sx_smlawb(K_1p0_Q16, -lam, lam);
            const i32 gs_n = sx_div32_varQ(nom_Q16, sx_smlawb(K_1p0_Q24, SX_ROWB(n2, 0), lam), 24);
            const i32 ga_n = sx_div32_varQ(nom_Q16, sx_smlawb(K_1p0_Q24, SX_ROWB(n1, 0), lam), 24);
            n2 = sx_smulww(gs_n, n2);
            n1 = sx_smulww(ga_n, n1);
            if (act) { A2 = n2; A1 = n1; gs = gs_n; ga = ga_n; }
        }
    }
    c->AR1_Q13[s * SX_SHAPE_ORDER + j] = (i16)sx_sat16(sx_rshift_round(A1, 11));
    c->AR2_Q13[s * SX_SHAPE_ORDER + j] = (i16)sx_sat16(sx_rshift_round(A2, 11));
}
#endif

// SKP_Silk_noise_shape_analysis_FIX, noise_shape_analysis_FIX.c:137.
// pitch_res = res_pitch + frame_length; x = x_buf + frame_length; x_windowed: 120-sample scratch (LDS)
SX_FN1 void sx_noise_shape_analysis(SxEncState* st, SxEncCtrl* c, const i16* pitch_res, const i16* x, SxShapeWork* sw) {
    SX_IN_LDS(st); SX_IN_LDS(c); SX_IN_LDS(pitch_res); SX_IN_LDS(x); SX_IN_LDS(sw);
    i32 tmp32;
    const i16* x_ptr = x - SX_LA_SHAPE;
    c->current_SNR_dB_Q7 = st->SNR_dB_Q7;                 // DISABLE_BUF_RD (SKP_Silk_define.h:53)
    c->current_SNRPerMD_dB_Q7 = st->SNRPerMD_dB_Q7;
    // (inBandFEC_SNR_comp_Q8 == 0: LBRR disabled)
    c->input_quality_Q14 = (c->input_quality_bands_Q15[0] + c->input_quality_bands_Q15[1]) >> 2;
    c->coding_quality_Q14 = sx_sigm_Q15(sx_rshift_round(c->current_SNR_dB_Q7 - K_18p0_Q7, 4)) >> 1;
    i32 b_Q8 = K_1p0_Q8 - st->speech_activity_Q8;
    b_Q8 = sx_smulwb(sx_shl(b_Q8, 8), b_Q8);
    i32 SNR_adj_dB_Q7 = sx_smlawb(c->current_SNR_dB_Q7, sx_smulbb(K_mBG_SNR_DECR_dB_Q7 >> 5, b_Q8),
                                  sx_smulwb(K_1p0_Q14 + c->input_quality_Q14, c->coding_quality_Q14));
    if (c->sigtype == 0) {
        SNR_adj_dB_Q7 = sx_smlawb(SNR_adj_dB_Q7, K_HARM_SNR_INCR_dB_Q8, st->LTPCorr_Q15);
    } else {
        SNR_adj_dB_Q7 = sx_smlawb(SNR_adj_dB_Q7, sx_smlawb(K_6p0_Q9, -K_0p4_Q18, c->current_SNR_dB_Q7), K_1p0_Q14 - c->input_quality_Q14);
    }
    i32 md_input_quality_Q14 = sx_sigm_Q15(sx_rshift_round(c->current_SNRPerMD_dB_Q7 - K_18p0_Q7, 4)) >> 1;
    i32 md_SNR_adj_dB_Q7 = sx_smlawb(c->current_SNRPerMD_dB_Q7, sx_smulbb(K_mBG_SNR_DECR_dB_Q7 >> 5, b_Q8),
                                     sx_smulwb(K_1p0_Q14 + md_input_quality_Q14, c->coding_quality_Q14));
    if (c->sigtype == 0) {
        md_SNR_adj_dB_Q7 = sx_smlawb(md_SNR_adj_dB_Q7, K_HARM_SNR_INCR_dB_Q8, st->LTPCorr_Q15);
    } else {
        md_SNR_adj_dB_Q7 = sx_smlawb(md_SNR_adj_dB_Q7, sx_smlawb(K_6p0_Q9, -K_0p4_Q18, c->current_SNRPerMD_dB_Q7),
                                     K_1p0_Q14 - c->input_quality_Q14);
    }
    // sparseness
    if (c->sigtype == 0) {
        c->QuantOffsetType = 0;
        c->sparseness_Q8 = 0;
    } else {
        const int nSamples = 2 * SX_FS_KHZ;      // energy per 2 ms (noise_shape_analysis_FIX.c:253)
        i32 energy_variation_Q7 = 0, log_energy_prev_Q7 = 0;
        // (the ten segment energies side by side, one lane each; the sum of their log differences afterwards)
        i32* lg = sw->scale;                     // ten words of the shape work area, free until the warped autocorrelation
        static_assert(sizeof(sw->scale) + sizeof(sw->C) >= 10 * sizeof(i32) && offsetof(SxShapeWork, C) == offsetof(SxShapeWork, scale) + sizeof(sw->scale),
                      "scratch of the sparseness measure");
        SX_PAR(k, 10) {
            i32 e, sh;
            sx_sum_sqr_shift_n<2 * SX_FS_KHZ>(&e, &sh, pitch_res + k * nSamples, 0);
            e += nSamples >> sh;
            lg[k] = sx_lin2log(e);
        }
        wv_sync();
        for (int k = 0; k < 10; k++) {
            const i32 log_energy_Q7 = lg[k];
            if (k > 0) energy_variation_Q7 += sx_abs(log_energy_Q7 - log_energy_prev_Q7);
            log_energy_prev_Q7 = log_energy_Q7;
        }
        c->sparseness_Q8 = sx_sigm_Q15(sx_smulwb(energy_variation_Q7 - K_5p0_Q7, K_0p1_Q16)) >> 7;
        c->QuantOffsetType = c->sparseness_Q8 > K_SPARSENESS_THRESHOLD_QNT_OFFSET_Q8 ? 0 : 1;
        SNR_adj_dB_Q7 = sx_smlawb(SNR_adj_dB_Q7, K_SPARSE_SNR_INCR_dB_Q15, c->sparseness_Q8 - K_0p5_Q8);
        md_SNR_adj_dB_Q7 = sx_smlawb(md_SNR_adj_dB_Q7, K_SPARSE_SNR_INCR_dB_Q15, c->sparseness_Q8 - K_0p5_Q8);
    }
    SX_S(37)
    // bandwidth expansion
    i32 strength_Q16 = sx_smulwb(c->predGain_Q16, K_FIND_PITCH_WHITE_NOISE_FRACTION_Q16);
    i32 BWExp1_Q16, BWExp2_Q16;
    BWExp1_Q16 = BWExp2_Q16 = sx_div32_varQ(K_BANDWIDTH_EXPANSION_Q16, sx_smlaww(K_1p0_Q16, strength_Q16, strength_Q16), 16);
    i32 delta_Q16 = sx_smulwb(K_1p0_Q16 - sx_smulbb(3, c->coding_quality_Q14), K_LOW_RATE_BANDWIDTH_EXPANSION_DELTA_Q16);
    BWExp1_Q16 = sx_sub(BWExp1_Q16, delta_Q16);
    BWExp2_Q16 = sx_add(BWExp2_Q16, delta_Q16);
    BWExp1_Q16 = sx_shl(BWExp1_Q16, 14) / (BWExp2_Q16 >> 2);
    i32 warping_Q16 = sx_smlawb(SX_WARPING_Q16, c->coding_quality_Q14, K_0p01_Q18);
    // per-subframe shaping filters: windows and warped autocorrelations of the four subframes side by side
    {
        // sine-window gains (SKP_Silk_apply_sine_window.c:39, length 40): a data-independent recursion, evaluated once
        const int length = SX_LA_SHAPE;
        const i32 f_Q16 = T_sine_win_freq_Q16[(length >> 2) - 4];
        const i32 c_Q16 = sx_smulwb(f_Q16, -f_Q16);
        for (int wt = 0; wt < 2; wt++) {
            i32 S0 = wt == 0 ? 0 : (1 << 16);
            i32 S1 = wt == 0 ? f_Q16 + (length >> 3) : (1 << 16) + (c_Q16 >> 1) + (length >> 4);
            for (int k = 0; k < length; k += 4) {
                sw->win[wt][k] = (S0 + S1) >> 1;
                sw->win[wt][k + 1] = S1;
                S0 = sx_smulwb(S1, c_Q16) + sx_shl(S1, 1) - S0 + 1;
                S0 = sx_min(S0, 1 << 16);
                sw->win[wt][k + 2] = (S0 + S1) >> 1;
                sw->win[wt][k + 3] = S0;
                S1 = sx_smulwb(S0, c_Q16) + sx_shl(S0, 1) - S1;
                S1 = sx_min(S1, 1 << 16);
            }
        }
        wv_sync();
        SX_PAR(t, SX_NB_SUBFR * SX_SHAPE_WIN) {
            const int k = t / SX_SHAPE_WIN, i = t - k * SX_SHAPE_WIN;
            const i16 v = x_ptr[k * SX_SUBFR + i];
            // slope_part = la_shape samples of sine window either side of a flat 5 ms (noise_shape_analysis_FIX.c:312-323)
            sw->xw[k][i] = i < SX_LA_SHAPE ? (i16)sx_smulwb(sw->win[0][i], v)
                                           : (i < SX_LA_SHAPE + 5 * SX_FS_KHZ ? v : (i16)sx_smulwb(sw->win[1][i - (SX_LA_SHAPE + 5 * SX_FS_KHZ)], v));
        }
        wv_sync();
        SX_S(38)
        sx_warped_autocorr4(sw, (i16)warping_Q16);
        SX_S(39)
    }
    SX_T_BEGIN
#ifdef SX_LANE_STREAM
    sx_shape_rows(sw, c, warping_Q16, BWExp1_Q16, BWExp2_Q16);
#else
    // Schur recursion, warped gain, bandwidth expansion, pre-gains and coefficient limiting: subframe k on lane k
    SX_PAR(k, SX_NB_SUBFR) {
        i32* auto_corr = sw->corr[k];
        i32* AR2_Q24 = sw->A2[k];
        i32* AR1_Q24 = sw->A1[k];
        auto_corr[0] = sx_add(auto_corr[0], sx_max(sx_smulwb(auto_corr[0] >> 4, K_SHAPE_WHITE_NOISE_FRACTION_Q20), 1));
        i32 nrg = sx_schur64(sw->rc[k], auto_corr, SX_SHAPE_ORDER, sw->C[k]);
        sx_k2a_Q16(AR2_Q24, sw->rc[k], SX_SHAPE_ORDER, sw->At[k]);
        int Qnrg = -sw->scale[k];
        if (Qnrg & 1) { Qnrg -= 1; nrg >>= 1; }
        const i32 sq = sx_sqrt_approx(nrg);
        Qnrg >>= 1;
        i32 g = sx_lshift_sat32(sq, 16 - Qnrg);
        {
            i32 gain_mult_Q16 = sx_warped_gain(AR2_Q24, warping_Q16, SX_SHAPE_ORDER);
            g = sx_smulww(g, gain_mult_Q16);
            if (g < 0) g = SX_I32_MAX;
        }
        c->Gains_Q16[k] = g;
        sx_bwexpander_32(AR2_Q24, SX_SHAPE_ORDER, BWExp2_Q16);
        for (int i = 0; i < SX_SHAPE_ORDER; i++) AR1_Q24[i] = AR2_Q24[i];
        sx_bwexpander_32(AR1_Q24, SX_SHAPE_ORDER, BWExp1_Q16);
        i32 pre_nrg_Q30, nrg1;
        sx_lpc_inv_pred_gain_Q24_ws(&pre_nrg_Q30, AR2_Q24, SX_SHAPE_ORDER, sw->inv[k]);
        sx_lpc_inv_pred_gain_Q24_ws(&nrg1, AR1_Q24, SX_SHAPE_ORDER, sw->inv[k]);
        pre_nrg_Q30 = sx_shl(sx_smulwb(pre_nrg_Q30, K_0p7_Q15), 1);
        c->GainsPre_Q14[k] = K_0p3_Q14 + sx_div32_varQ(pre_nrg_Q30, nrg1, 14);
        sx_limit_warped_coefs(AR2_Q24, AR1_Q24, warping_Q16, K_3p999_Q24, SX_SHAPE_ORDER);
        for (int i = 0; i < SX_SHAPE_ORDER; i++) {
            c->AR1_Q13[k * SX_SHAPE_ORDER + i] = (i16)sx_sat16(sx_rshift_round(AR1_Q24[i], 11));
            c->AR2_Q13[k * SX_SHAPE_ORDER + i] = (i16)sx_sat16(sx_rshift_round(AR2_Q24[i], 11));
        }
    }
#endif
    wv_sync();
    SX_T(26)
    // gain tweaking
    i32 md_gain_mult_Q16 = sx_log2lin(sx_neg(sx_smlawb(-K_16p0_Q7, md_SNR_adj_dB_Q7, K_0p16_Q16)));
    i32 gain_mult_Q16 = sx_log2lin(sx_neg(sx_smlawb(-K_16p0_Q7, SNR_adj_dB_Q7, K_0p16_Q16)));
    c->md_delta_gain_par = sx_fdiv((float)gain_mult_Q16, (float)md_gain_mult_Q16);
    i32 gain_add_Q16 = sx_log2lin(sx_smlawb(K_16p0_Q7, K_NOISE_FLOOR_dB_Q7, K_0p16_Q16));
    tmp32 = sx_log2lin(sx_smlawb(K_16p0_Q7, K_RELATIVE_MIN_GAIN_dB_Q7, K_0p16_Q16));
    tmp32 = sx_smulww(st->avgGain_Q16, tmp32);
    gain_add_Q16 = sx_add_sat32(gain_add_Q16, tmp32);
    for (int k = 0; k < SX_NB_SUBFR; k++) {
        c->Gains_Q16[k] = sx_smulww(c->Gains_Q16[k], gain_mult_Q16);
        if (c->Gains_Q16[k] < 0) c->Gains_Q16[k] = SX_I32_MAX;
    }
    for (int k = 0; k < SX_NB_SUBFR; k++) {
        c->Gains_Q16[k] = sx_add_pos_sat32(c->Gains_Q16[k], gain_add_Q16);
        st->avgGain_Q16 = sx_add_sat32(st->avgGain_Q16, sx_smulwb(c->Gains_Q16[k] - st->avgGain_Q16,
                                       sx_rshift_round(sx_smulbb(st->speech_activity_Q8, K_GAIN_SMOOTHING_COEF_Q10), 2)));
    }
    // de-essing (noise_shape_analysis_FIX.c:435-454): only at fs 16 / 24 kHz
    gain_mult_Q16 = K_1p0_Q16 + sx_rshift_round(sx_add(K_INPUT_TILT_Q26, sx_mul(c->coding_quality_Q14, K_HIGH_RATE_INPUT_TILT_Q12)), 10);
#if SX_FS_KHZ == 16
    if (c->input_tilt_Q15 <= 0 && c->sigtype == 1) {
        const i32 essStrength_Q15 = sx_smulww(-c->input_tilt_Q15, sx_smulbb(st->speech_activity_Q8, K_1p0_Q8 - c->sparseness_Q8));
        tmp32 = sx_log2lin(K_16p0_Q7 - sx_smulwb(essStrength_Q15, sx_smulwb(K_DE_ESSER_COEF_WB_dB_Q7, K_0p16_Q17)));
        gain_mult_Q16 = sx_smulww(gain_mult_Q16, tmp32);
    }
#endif
    for (int k = 0; k < SX_NB_SUBFR; k++) c->GainsPre_Q14[k] = sx_smulwb(gain_mult_Q16, c->GainsPre_Q14[k]);
    // low-frequency shaping and tilt
    strength_Q16 = sx_mul(K_LOW_FREQ_SHAPING_Q0, K_1p0_Q16 + sx_smulbb(K_LOW_QUALITY_LOW_FREQ_SHAPING_DECR_Q1,
                                                                        c->input_quality_bands_Q15[0] - K_1p0_Q15));
    i32 Tilt_Q16;
    if (c->sigtype == 0) {
        i32 fs_kHz_inv = K_0p2_Q14 / SX_FS_KHZ;
        SX_PAR(k, SX_NB_SUBFR) {                     // (one division per subframe: side by side)
            i32 b_Q14 = fs_kHz_inv + K_3p0_Q14 / c->pitchL[k];
            c->LF_shp_Q14[k] = sx_shl(K_1p0_Q14 - b_Q14 - sx_smulwb(strength_Q16, b_Q14), 16) | (i32)(u16)(b_Q14 - K_1p0_Q14);
        }
        Tilt_Q16 = -K_HP_NOISE_COEF_Q16 - sx_smulwb(K_1p0_Q16 - K_HP_NOISE_COEF_Q16, sx_smulwb(K_HARM_HP_NOISE_COEF_Q24, st->speech_activity_Q8));
    } else {
        i32 b_Q14 = 21299 / SX_FS_KHZ;
        c->LF_shp_Q14[0] = sx_shl(K_1p0_Q14 - b_Q14 - sx_smulwb(strength_Q16, sx_smulwb(K_0p6_Q16, b_Q14)), 16);
        c->LF_shp_Q14[0] |= (i32)(u16)(b_Q14 - K_1p0_Q14);
        for (int k = 1; k < SX_NB_SUBFR; k++) c->LF_shp_Q14[k] = c->LF_shp_Q14[0];
        Tilt_Q16 = -K_HP_NOISE_COEF_Q16;
    }
    // harmonic shaping
    i32 HarmBoost_Q16 = sx_smulwb(sx_smulwb(K_1p0_Q17 - sx_shl(c->coding_quality_Q14, 3), st->LTPCorr_Q15), K_LOW_RATE_HARMONIC_BOOST_Q16);
    HarmBoost_Q16 = sx_smlawb(HarmBoost_Q16, K_1p0_Q16 - sx_shl(c->input_quality_Q14, 2), K_LOW_INPUT_QUALITY_HARMONIC_BOOST_Q16);
    i32 HarmShapeGain_Q16;
    if (c->sigtype == 0) {
        HarmShapeGain_Q16 = sx_smlawb(K_HARMONIC_SHAPING_Q16,
                                      K_1p0_Q16 - sx_smulwb(K_1p0_Q18 - sx_shl(c->coding_quality_Q14, 4), c->input_quality_Q14),
                                      K_HIGH_RATE_OR_LOW_QUALITY_HARMONIC_SHAPING_Q16);
        HarmShapeGain_Q16 = sx_smulwb(sx_shl(HarmShapeGain_Q16, 1), sx_sqrt_approx(sx_shl(st->LTPCorr_Q15, 15)));
    } else {
        HarmShapeGain_Q16 = 0;
    }
    for (int k = 0; k < SX_NB_SUBFR; k++) {
        st->HarmBoost_smth_Q16 = sx_smlawb(st->HarmBoost_smth_Q16, HarmBoost_Q16 - st->HarmBoost_smth_Q16, K_SUBFR_SMTH_COEF_Q16);
        st->HarmShapeGain_smth_Q16 = sx_smlawb(st->HarmShapeGain_smth_Q16, HarmShapeGain_Q16 - st->HarmShapeGain_smth_Q16, K_SUBFR_SMTH_COEF_Q16);
        st->Tilt_smth_Q16 = sx_smlawb(st->Tilt_smth_Q16, Tilt_Q16 - st->Tilt_smth_Q16, K_SUBFR_SMTH_COEF_Q16);
        c->HarmBoost_Q14[k] = sx_rshift_round(st->HarmBoost_smth_Q16, 2);
        c->HarmShapeGain_Q14[k] = sx_rshift_round(st->HarmShapeGain_smth_Q16, 2);
        c->Tilt_Q14[k] = sx_rshift_round(st->Tilt_smth_Q16, 2);
    }
}

// ---------------------------------------------------------------------------------------------------
// prefilter
// ---------------------------------------------------------------------------------------------------
// SKP_Silk_prefilter_FIX + warped_LPC_analysis_filter_FIX + prefilt_FIX, SKP_Silk_prefilter_FIX.c:43-224
struct SxPrefWork {                  // LDS scratch of the prefilter
    i16 ring[SX_LTP_BUF];            // staged harmonic-shaping ring (pf_sLTP_shp)
    i32 o[2][SX_SHAPE_ORDER][2];     // (section output, running sum) in flight between neighbouring lanes, double buffered
    i16 st_res[SX_FRAME + 1];        // short-term residual of the frame; [0] = last sample of the previous frame
    alignas(16) i32 x_filt_Q12[SX_FRAME];
    i32 vend[SX_SHAPE_ORDER + 1];
    i32 lf_end[2];                   // final (sLF_AR, sLF_MA) of the low-frequency recursion (GPU build: it runs on one lane)
};

SX_FN1 void sx_prefilter(SxEncState* st, const SxEncCtrl* c, i16* xw, const i16* x, SxPrefWork* pw) {
    SX_IN_LDS(st); SX_IN_LDS(c); SX_IN_LDS(xw); SX_IN_LDS(x); SX_IN_LDS(pw);
    i16* pf_sLTP_shp = pw->ring;
    const i32 lambda_Q16 = (i16)SX_WARPING_Q16;
#ifdef SX_LANE_STREAM
    // ---- warped_LPC_analysis_filter_FIX (SKP_Silk_prefilter_FIX.c:43) for the whole frame, skewed over the 16 lanes of a row ----
    // v_0(n) = x(n) << 14;  v_1(n) = v_0(n-1) + lambda * v_1(n-1);  v_{j+1}(n) = v_j(n-1) + lambda * (v_{j+1}(n-1) - v_j(n));
    // acc(n) = sum_j coef_k(n)[j-1] * v_j(n).  Lane l owns section j = l + 1 and works on sample n = t - j at step t; (v_j(n),
    // partial acc) move to lane l + 1 with a DPP row shift, so the recursion never leaves the registers.  (All four rows run
    // the same computation; their stores coincide.)
    {
        const int l = SX_LANE & 15;
        i32 pv = st->pf_sAR_shp[l + 1], pin = st->pf_sAR_shp[l];
        const i32 a0 = sx_pre16(c->AR1_Q13[l]), a1 = sx_pre16(c->AR1_Q13[SX_SHAPE_ORDER + l]),
                  a2 = sx_pre16(c->AR1_Q13[2 * SX_SHAPE_ORDER + l]), a3 = sx_pre16(c->AR1_Q13[3 * SX_SHAPE_ORDER + l]);
        const i32 lam = sx_pre16(lambda_Q16);
        pw->st_res[0] = (i16)st->pf_sHarmHP;
        i32 out = 0, acc = 0;
        i32 xn = (0 - l >= 0) ? (i32)x[0] : 0;                        // x[n] of the coming step (n = t - 1 - l), fetched a step ahead
        // Steps 16 .. 160: every lane holds a sample of the frame.  Lane 0's input comes in through the DPP move's `old` operand, its
        // partial sum through the move's zero fill; the coefficient set changes at sample bnd (lane by lane as the skew passes it);
        // lane 15 stores its sum through a walking pointer, the other lanes through a pointer that stays on a dump word.  The first and last
        // 15 steps are the same body with the state update (and the store) under the lane's range check; their loads may fall outside
        // the frame (the neighbouring samples of the buffer: never used)
#define SX_PF_BODY(ACTIVE, COEF, STORE)                                                                                               \
        {   const i32 xcur = xn;                                                                                                      \
            xn = (i32)xp[t];                                                                                                          \
            const i32 in = __builtin_amdgcn_update_dpp(sx_shl(xcur, 14), out, 0x111, 0xF, 0xF, false);                                 \
            const i32 acc_prev = SX_DPP_(acc, 0x111);                                                                                 \
            if (ACTIVE) {                                                                                                             \
                const i32 o = sx_smlaw_pre(pin, pv - (in & not_first), lam);                                                          \
                acc = sx_smlaw_pre(acc_prev, o, (COEF));                                                                              \
                pin = in; pv = o; out = o;                                                                                            \
                if (STORE) { *op = acc; op += ostride; }                                                                               \
            }                                                                                                                         \
        }
#define SX_PF_STEADY(T0, T1, CA, CB, BND)                                                                                             \
        _Pragma("unroll 4")                                                                                                           \
        for (int t = (T0); t <= (T1); t++) SX_PF_BODY(true, ((BND) == 0 || t - 1 - l >= (BND)) ? (CA) : (CB), true)
        static_assert(SX_SHAPE_ORDER == 16 && SX_SUBFR > SX_SHAPE_ORDER && SX_FRAME == 4 * SX_SUBFR, "the steady-state split of the warped filter");
        {
            const i32 not_first = l == 0 ? 0 : -1;
            const i16* xp = x - l;                                                      // xp[t] = x[n + 1] of step t
            // (section 15's sums go to x_filt_Q12 -- free until the next step fills it -- as they are; the residual is formed from
            // them afterwards, all samples side by side, instead of five more instructions in every step of the recursion)
            const int ostride = l == SX_SHAPE_ORDER - 1 ? 1 : 0;
            i32* op = l == SX_SHAPE_ORDER - 1 ? &pw->x_filt_Q12[0] : &pw->o[0][l][0];
            for (int t = 1; t < SX_SHAPE_ORDER; t++) SX_PF_BODY(l < t, a0, false)                        // lanes join one by one (lane 15 at step 16)
            SX_PF_STEADY(SX_SHAPE_ORDER, SX_SUBFR, a0, a0, 0)
            SX_PF_STEADY(SX_SUBFR + 1, 2 * SX_SUBFR, a1, a0, SX_SUBFR)
            SX_PF_STEADY(2 * SX_SUBFR + 1, 3 * SX_SUBFR, a2, a1, 2 * SX_SUBFR)
            SX_PF_STEADY(3 * SX_SUBFR + 1, 4 * SX_SUBFR, a3, a2, 3 * SX_SUBFR)
            for (int t = SX_FRAME + 1; t < SX_FRAME + SX_SHAPE_ORDER; t++) SX_PF_BODY(l >= t - SX_FRAME, a3, true)   // ... and leave one by one
        }
#undef SX_PF_BODY
#undef SX_PF_STEADY
        const i32 vend = pv, vend0 = pin;                               // (a lane's last update was its sample SX_FRAME - 1)
        if (SX_LANE < SX_SHAPE_ORDER) {
            st->pf_sAR_shp[l + 1] = vend;
            if (l == 0) st->pf_sAR_shp[0] = vend0;
        }
        wv_sync();
        SX_PAR(n, SX_FRAME) pw->st_res[1 + n] = (i16)sx_sat16((i32)x[n] - sx_rshift_round(pw->x_filt_Q12[n], 11));
        wv_sync();
    }
#else
    // ---- warped_LPC_analysis_filter_FIX (SKP_Silk_prefilter_FIX.c:43) for the whole frame, skewed over 16 lanes ----
    // v_0(n) = x(n) << 14;  v_1(n) = v_0(n-1) + lambda * v_1(n-1);  v_{j+1}(n) = v_j(n-1) + lambda * (v_{j+1}(n-1) - v_j(n));
    // acc(n) = sum_j coef_k(n)[j-1] * v_j(n).  Lane l owns section j = l + 1 and works on sample n = t - j at step t, passing
    // (v_j(n), partial acc) to lane l + 1 through LDS.
    {
        i32 pv[SX_NP64], pin[SX_NP64];
        SX_PAR(l, SX_SHAPE_ORDER) {
            pv[SX_PL(l)] = st->pf_sAR_shp[l + 1];       // v_j(-1)
            pin[SX_PL(l)] = st->pf_sAR_shp[l];          // v_{j-1}(-1)
        }
        pw->st_res[0] = (i16)st->pf_sHarmHP;
        for (int t = 1; t < SX_FRAME + SX_SHAPE_ORDER; t++) {
            SX_PAR(l, SX_SHAPE_ORDER) {
                const int j = l + 1, n = t - j, pl = SX_PL(l);
                if (n >= 0 && n < SX_FRAME) {
                    i32 in, acc;
                    if (l == 0) { in = sx_shl((i32)x[n], 14); acc = 0; }
                    else { in = pw->o[(t - 1) & 1][l - 1][0]; acc = pw->o[(t - 1) & 1][l - 1][1]; }
                    const i32 out = l == 0 ? sx_smlawb(pin[pl], pv[pl], lambda_Q16) : sx_smlawb(pin[pl], pv[pl] - in, lambda_Q16);
                    acc = sx_smlawb(acc, out, c->AR1_Q13[(n / SX_SUBFR) * SX_SHAPE_ORDER + l]);
                    pin[pl] = in;
                    pv[pl] = out;
                    pw->o[t & 1][l][0] = out;
                    pw->o[t & 1][l][1] = acc;
                    if (l == SX_SHAPE_ORDER - 1) pw->st_res[1 + n] = (i16)sx_sat16((i32)x[n] - sx_rshift_round(acc, 11));
                    if (n == SX_FRAME - 1) { pw->vend[j] = out; if (l == 0) pw->vend[0] = in; }
                }
            }
            wv_sync();
        }
        SX_PAR(l, SX_SHAPE_ORDER + 1) st->pf_sAR_shp[l] = pw->vend[l];
        wv_sync();
    }
#endif
    SX_S(40)
    // ---- per-subframe FIR on the residual, then prefilt_FIX (SKP_Silk_prefilter_FIX.c:174): serial shaping recursion ----
    SX_PAR(n, SX_FRAME) {
        const int k = n / SX_SUBFR;
        const i32 HarmShapeGain_Q12 = sx_smulwb(c->HarmShapeGain_Q14[k], 16384 - c->HarmBoost_Q14[k]);
        const i32 B_lo = sx_rshift_round(c->GainsPre_Q14[k], 2);
        i32 tmp_32 = sx_smlabb(K_INPUT_TILT_Q26, c->HarmBoost_Q14[k], HarmShapeGain_Q12);
        tmp_32 = sx_smlabb(tmp_32, c->coding_quality_Q14, K_HIGH_RATE_INPUT_TILT_Q12);
        tmp_32 = sx_smulwb(tmp_32, -c->GainsPre_Q14[k]);
        tmp_32 = sx_rshift_round(tmp_32, 12);
        const i32 B_hi = sx_sat16(tmp_32);
        pw->x_filt_Q12[n] = sx_add(sx_smulbb(pw->st_res[1 + n], B_lo), sx_smulbb(pw->st_res[n], B_hi));
    }
    wv_sync();
#if defined(SX_LANE_STREAM) && SX_FS_KHZ == 8
    // (8 kHz only: at 16 kHz the span lag + frame = 288 + 2 + 320 exceeds the 512-entry ring, the writes of a frame would run
    // over entries its first samples still have to read; the wide-band build takes the sample-serial form below)
    // prefilt_FIX's low-frequency recursion (sLF_AR, sLF_MA) only consumes x_filt; the harmonic term is a FIR over the ring of
    // past sLF_MA values that never feeds back.  So: the scalar recursion streams its 160 samples through lane registers,
    // then the ring update and the harmonic FIR + output run lane-parallel.  (Sample n reads ring entries written by samples
    // <= n - lag + 1 < n or by earlier frames; every entry is written once per frame and the 307-entry span cannot wrap.)
    i32 sLF_AR = SX_UNI(st->pf_sLF_AR_shp_Q12), sLF_MA = SX_UNI(st->pf_sLF_MA_shp_Q12);
    const int buf_idx0 = SX_UNI(st->pf_sLTP_shp_buf_idx);
    i32 ma[3] = {0, 0, 0};
    {
        // the recursion runs on the vector unit of one lane (SX_VEC, see sx_allpass2_lanes): three high-word multiplies and five
        // adds / shifts per sample, x_filt read four samples at a time and replaced in place by sLF_MA
        if (SX_LANE == 0) {
            i32 ar = sLF_AR, mq = sLF_MA;
            SX_VEC(ar); SX_VEC(mq);
            SxV4i* xq = (SxV4i*)pw->x_filt_Q12;
#pragma unroll
            for (int k = 0; k < SX_NB_SUBFR; k++) {
                i32 tilt = sx_pre16(c->Tilt_Q14[k]), lfb = sx_pre16(c->LF_shp_Q14[k]), lft = (i32)((u32)c->LF_shp_Q14[k] & 0xFFFF0000u);
                SX_VEC(tilt); SX_VEC(lfb); SX_VEC(lft);
                for (int n4 = k * (SX_SUBFR / 4); n4 < (k + 1) * (SX_SUBFR / 4); n4++) {
                    SxV4i v = xq[n4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const i32 n_Tilt_Q10 = sx_smulw_pre(ar, tilt);
                        const i32 n_LF_Q10 = sx_add(sx_smulw_pre(ar, lft), sx_smulw_pre(mq, lfb));
                        ar = sx_sub(v.v[u], sx_shl(n_Tilt_Q10, 2));
                        mq = sx_sub(ar, sx_shl(n_LF_Q10, 2));
                        v.v[u] = mq;
                    }
                    xq[n4] = v;
                }
            }
            pw->lf_end[0] = ar;
            pw->lf_end[1] = mq;
        }
        wv_sync();
        sLF_AR = SX_UNI(pw->lf_end[0]);
        sLF_MA = SX_UNI(pw->lf_end[1]);
#pragma unroll
        for (int j = 0; j < 3; j++) { const int i = SX_LANE + 64 * j; ma[j] = i < SX_FRAME ? pw->x_filt_Q12[i] : 0; }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int n = SX_LANE + 64 * j;
        if (n < SX_FRAME) pf_sLTP_shp[(buf_idx0 - 1 - n) & SX_LTP_MASK] = (i16)sx_sat16(sx_rshift_round(ma[j], 12));
    }
    wv_sync();
    const int buf_idx = (buf_idx0 - SX_FRAME) & SX_LTP_MASK;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int n = SX_LANE + 64 * j;
        if (n < SX_FRAME) {
            const int k = n / SX_SUBFR;
            const int lag = c->sigtype == 0 ? c->pitchL[k] : st->pf_lagPrev;
            const i32 HarmShapeGain_Q12 = sx_smulwb(c->HarmShapeGain_Q14[k], 16384 - c->HarmBoost_Q14[k]);
            i32 HarmShapeFIRPacked_Q12 = HarmShapeGain_Q12 >> 2;
            HarmShapeFIRPacked_Q12 |= sx_shl(HarmShapeGain_Q12 >> 1, 16);
            i32 n_LTP_Q12 = 0;
            if (lag > 0) {
                const int idx = lag + buf_idx0 - n;
                n_LTP_Q12 = sx_smulbb(pf_sLTP_shp[(idx - 2) & SX_LTP_MASK], HarmShapeFIRPacked_Q12);
                n_LTP_Q12 = sx_add(n_LTP_Q12, sx_smulbt(pf_sLTP_shp[(idx - 1) & SX_LTP_MASK], HarmShapeFIRPacked_Q12));
                n_LTP_Q12 = sx_smlabb(n_LTP_Q12, pf_sLTP_shp[idx & SX_LTP_MASK], HarmShapeFIRPacked_Q12);
            }
            xw[n] = (i16)sx_sat16(sx_rshift_round(sx_sub(ma[j], n_LTP_Q12), 12));
        }
    }
#else
    int lag = st->pf_lagPrev;
    i32 sLF_AR = st->pf_sLF_AR_shp_Q12, sLF_MA = st->pf_sLF_MA_shp_Q12;
    int buf_idx = st->pf_sLTP_shp_buf_idx;
    for (int k = 0; k < SX_NB_SUBFR; k++) {
        if (c->sigtype == 0) lag = c->pitchL[k];
        const i32 HarmShapeGain_Q12 = sx_smulwb(c->HarmShapeGain_Q14[k], 16384 - c->HarmBoost_Q14[k]);
        i32 HarmShapeFIRPacked_Q12 = HarmShapeGain_Q12 >> 2;
        HarmShapeFIRPacked_Q12 |= sx_shl(HarmShapeGain_Q12 >> 1, 16);
        const i32 Tilt_Q14 = c->Tilt_Q14[k], LF_shp_Q14 = c->LF_shp_Q14[k];
        for (int i = 0; i < SX_SUBFR; i++) {
            i32 n_LTP_Q12 = 0;
            if (lag > 0) {
                int idx = lag + buf_idx;
                n_LTP_Q12 = sx_smulbb(pf_sLTP_shp[(idx - 2) & SX_LTP_MASK], HarmShapeFIRPacked_Q12);
                n_LTP_Q12 = sx_add(n_LTP_Q12, sx_smulbt(pf_sLTP_shp[(idx - 1) & SX_LTP_MASK], HarmShapeFIRPacked_Q12));
                n_LTP_Q12 = sx_smlabb(n_LTP_Q12, pf_sLTP_shp[idx & SX_LTP_MASK], HarmShapeFIRPacked_Q12);
            }
            i32 n_Tilt_Q10 = sx_smulwb(sLF_AR, Tilt_Q14);
            i32 n_LF_Q10 = sx_smlawb(sx_smulwt(sLF_AR, LF_shp_Q14), sLF_MA, LF_shp_Q14);
            sLF_AR = sx_sub(pw->x_filt_Q12[k * SX_SUBFR + i], sx_shl(n_Tilt_Q10, 2));
            sLF_MA = sx_sub(sLF_AR, sx_shl(n_LF_Q10, 2));
            buf_idx = (buf_idx - 1) & SX_LTP_MASK;
            pf_sLTP_shp[buf_idx] = (i16)sx_sat16(sx_rshift_round(sLF_MA, 12));
            xw[k * SX_SUBFR + i] = (i16)sx_sat16(sx_rshift_round(sx_sub(sLF_MA, n_LTP_Q12), 12));
        }
    }
#endif
    wv_sync();
    st->pf_sLF_AR_shp_Q12 = sLF_AR;
    st->pf_sLF_MA_shp_Q12 = sLF_MA;
    st->pf_sLTP_shp_buf_idx = buf_idx;
    st->pf_sHarmHP = pw->st_res[SX_FRAME];
    st->pf_lagPrev = c->pitchL[SX_NB_SUBFR - 1];
    wv_sync();
}

// ---------------------------------------------------------------------------------------------------
// LTP analysis
// ---------------------------------------------------------------------------------------------------
// SKP_Silk_corrMatrix_FIX + corrVector_FIX, SKP_Silk_corrMatrix_FIX.c:35-152 (order 5, L = 40)
// The inner products over the L samples -- the first column of the matrix and the whole vector: 9 per subframe, 4 x 9 of the
// frame -- are taken side by side (one lane each) between the energy part and the recurrences along the diagonals, which stay with
// the subframe's lane.
// part 1: energy of x, the right shift of all correlations of the subframe, the main diagonal
SX_HD int sx_corr_matrix_diag(const i16* x, int L, int order, int head_room, i32* XX, int rshifts_in, int x_odd) {
    i32 energy, rshifts_local;
    sx_sum_sqr_shift_n<SX_SUBFR + SX_LTP_ORDER - 1>(&energy, &rshifts_local, x, x_odd);      // (L + order - 1 of the one caller, sx_find_LTP)
    int head_room_rshifts = sx_max(head_room - sx_clz32(energy), 0);
    energy = energy >> head_room_rshifts;
    rshifts_local += head_room_rshifts;
    for (int i = 0; i < order - 1; i++) energy -= sx_smulbb(x[i], x[i]) >> rshifts_local;
    if (rshifts_local < rshifts_in) {
        energy = energy >> (rshifts_in - rshifts_local);
        rshifts_local = rshifts_in;
    }
    XX[0] = energy;
    const i16* ptr1 = &x[order - 1];
    for (int j = 1; j < order; j++) {
        energy = sx_sub(energy, sx_smulbb(ptr1[L - j], ptr1[L - j]) >> rshifts_local);
        energy = sx_add(energy, sx_smulbb(ptr1[-j], ptr1[-j]) >> rshifts_local);
        XX[j * order + j] = energy;
    }
    return rshifts_local;
}
// inner product m of a subframe: m < order - 1: x[order-1 ..] . x[order-1-lag ..], lag = m + 1 (matrix, corrMatrix_FIX.c:118);
// m >= order - 1: x[order-1-lag ..] . t, lag = m - (order - 1) (vector, corrMatrix_FIX.c:35).  Every term is shifted before it
// is added (no shift: the reference's wrapping multiply-accumulate)
SX_HD i32 sx_corr_inner(const i16* x, const i16* t, int L, int order, int m, int rshifts) {
    const i16* pa = m < order - 1 ? &x[order - 1] : &x[order - 1 - (m - (order - 1))];
    const i16* pb = m < order - 1 ? &x[order - 2 - m] : t;
    i32 acc = 0;
    for (int i = 0; i < L; i++) acc = sx_add(acc, sx_smulbb(pa[i], pb[i]) >> rshifts);
    return acc;
}
// part 2: the off-diagonals from their first elements ip[lag - 1]
SX_HD void sx_corr_matrix_off(const i16* x, int L, int order, i32* XX, int rshifts_local, const i32* ip) {
    const i16* ptr1 = &x[order - 1];
    const i16* ptr2 = &x[order - 2];
    for (int lag = 1; lag < order; lag++) {
        i32 energy = ip[lag - 1];
        XX[lag * order] = energy;
        XX[lag] = energy;
        for (int j = 1; j < order - lag; j++) {
            energy = sx_sub(energy, sx_smulbb(ptr1[L - j], ptr2[L - j]) >> rshifts_local);
            energy = sx_add(energy, sx_smulbb(ptr1[-j], ptr2[-j]) >> rshifts_local);
            XX[(lag + j) * order + j] = energy;
            XX[j * order + lag + j] = energy;
        }
        ptr2--;
    }
}

// SKP_Silk_solve_LDL_FIX and helpers, SKP_Silk_solve_LS_FIX.c:71-241 (M = 5)
SX_FN1 void sx_solve_LDL(i32* A, int M, const i32* b, i32* x_Q16, i32* ws /* 50 words of LDS */) {
    SX_IN_LDS(A); SX_IN_LDS(b); SX_IN_LDS(x_Q16); SX_IN_LDS(ws);
    i32 *L_Q16 = ws, *Y = ws + 25, *inv_D_Q36 = ws + 30, *inv_D_Q48 = ws + 35, *v_Q0 = ws + 40, *D_Q0 = ws + 45;
    int status = 1;
    i32 diag_min_value = sx_max(sx_smmul(sx_add_sat32(A[0], A[M * M - 1]), K_FIND_LTP_COND_FAC_Q31), 1 << 9);
    for (int loop_count = 0; loop_count < M && status == 1; loop_count++) {
        status = 0;
        for (int j = 0; j < M; j++) {
            const i32* ptr1 = &L_Q16[j * M];
            i32 tmp_32 = 0;
            for (int i = 0; i < j; i++) {
                v_Q0[i] = sx_smulww(D_Q0[i], ptr1[i]);
                tmp_32 = sx_smlaww(tmp_32, v_Q0[i], ptr1[i]);
            }
            tmp_32 = sx_sub(A[j * M + j], tmp_32);
            if (tmp_32 < diag_min_value) {
                tmp_32 = sx_sub(sx_smulbb(loop_count + 1, diag_min_value), tmp_32);
                for (int i = 0; i < M; i++) A[i * M + i] = sx_add(A[i * M + i], tmp_32);
                status = 1;
                break;
            }
            D_Q0[j] = tmp_32;
            i32 one_div_diag_Q36 = sx_inverse32_varQ(tmp_32, 36);
            i32 one_div_diag_Q40 = sx_shl(one_div_diag_Q36, 4);
            i32 err = sx_sub(1 << 24, sx_smulww(tmp_32, one_div_diag_Q40));
            i32 one_div_diag_Q48 = sx_smulww(err, one_div_diag_Q40);
            inv_D_Q36[j] = one_div_diag_Q36;
            inv_D_Q48[j] = one_div_diag_Q48;
            L_Q16[j * M + j] = 65536;
            ptr1 = &A[j * M];
            const i32* ptr2 = &L_Q16[(j + 1) * M];
            for (int i = j + 1; i < M; i++) {
                tmp_32 = 0;
                for (int k = 0; k < j; k++) tmp_32 = sx_smlaww(tmp_32, v_Q0[k], ptr2[k]);
                tmp_32 = sx_sub(ptr1[i], tmp_32);
                L_Q16[i * M + j] = sx_add(sx_smmul(tmp_32, one_div_diag_Q48), sx_smulww(tmp_32, one_div_diag_Q36) >> 4);
                ptr2 += M;
            }
        }
    }
    for (int i = 0; i < M; i++) {   // SolveFirst
        i32 tmp_32 = 0;
        for (int j = 0; j < i; j++) tmp_32 = sx_smlaww(tmp_32, L_Q16[i * M + j], Y[j]);
        Y[i] = sx_sub(b[i], tmp_32);
    }
    for (int i = 0; i < M; i++) {   // divide
        i32 t = Y[i];
        Y[i] = sx_add(sx_smmul(t, inv_D_Q48[i]), sx_smulww(t, inv_D_Q36[i]) >> 4);
    }
    for (int i = M - 1; i >= 0; i--) {   // SolveLast
        i32 tmp_32 = 0;
        for (int j = M - 1; j > i; j--) tmp_32 = sx_smlaww(tmp_32, L_Q16[j * M + i], x_Q16[j]);
        x_Q16[i] = sx_sub(Y[i], tmp_32);
    }
}

// SKP_Silk_residual_energy16_covar_FIX, SKP_Silk_residual_energy16_FIX.c:31 (cQ = 14, D = 5)
SX_HD i32 sx_residual_energy16_covar(const i16* cvec, const i32* wXX, const i32* wXx, i32 wxx, int D, int cQ) {
    i32 cn[SX_MAX_LPC];
    int lshifts = 16 - cQ, Qxtra = lshifts;
    i32 c_max = 0;
    for (int i = 0; i < D; i++) c_max = sx_max(c_max, sx_abs((i32)cvec[i]));
    Qxtra = sx_min(Qxtra, sx_clz32(c_max) - 17);
    i32 w_max = sx_max(wXX[0], wXX[D * D - 1]);
    Qxtra = sx_min(Qxtra, sx_clz32(sx_mul(D, sx_smulwb(w_max, c_max) >> 4)) - 5);
    Qxtra = sx_max(Qxtra, 0);
    for (int i = 0; i < D; i++) cn[i] = sx_shl((i32)cvec[i], Qxtra);
    lshifts -= Qxtra;
    i32 tmp = 0;
    for (int i = 0; i < D; i++) tmp = sx_smlawb(tmp, wXx[i], cn[i]);
    i32 nrg = (wxx >> (1 + lshifts)) - tmp;
    i32 tmp2 = 0;
    for (int i = 0; i < D; i++) {
        tmp = 0;
        const i32* pRow = &wXX[i * D];
        for (int j = i + 1; j < D; j++) tmp = sx_smlawb(tmp, pRow[j], cn[j]);
        tmp = sx_smlawb(tmp, pRow[i] >> 1, cn[i]);
        tmp2 = sx_smlawb(tmp2, tmp, cn[i]);
    }
    nrg = sx_add(nrg, sx_shl(tmp2, lshifts));
    if (nrg < 1) nrg = 1;
    else if (nrg > (SX_I32_MAX >> (lshifts + 2))) nrg = SX_I32_MAX >> 1;
    else nrg = sx_shl(nrg, lshifts + 1);
    return nrg;
}

struct SxLtpWork {                   // LDS scratch of the LTP analysis: subframe k is analysed by lane k
    i32 b_Q16[SX_NB_SUBFR][SX_LTP_ORDER], Rr[SX_NB_SUBFR][SX_LTP_ORDER], delta_b_Q14[SX_NB_SUBFR][SX_LTP_ORDER];
    i32 rr[SX_NB_SUBFR], nrg[SX_NB_SUBFR], w[SX_NB_SUBFR], corr_rshifts[SX_NB_SUBFR], d_Q14[SX_NB_SUBFR];
    i32 ldl[SX_NB_SUBFR][50];
    i32 ip[SX_NB_SUBFR][2 * SX_LTP_ORDER - 1], rr_shifts[SX_NB_SUBFR];
};

// SKP_Silk_find_LTP_FIX, SKP_Silk_find_LTP_FIX.c:39.  res_pitch: LPC residual buffer (336 samples)
SX_FN1 void sx_find_LTP(i16* b_Q14, i32* WLTP, i32* LTPredCodGain_Q7, const i16* res_pitch, const i32* lag, const i32* Wght_Q15,
                       SxLtpWork* lw) {
    SX_IN_LDS(b_Q14); SX_IN_LDS(WLTP); SX_IN_LDS(LTPredCodGain_Q7); SX_IN_LDS(res_pitch); SX_IN_LDS(lag); SX_IN_LDS(Wght_Q15); SX_IN_LDS(lw);
    const int subfr_length = SX_SUBFR, mem_offset = SX_FRAME, HEAD = 2;
    i32* d_Q14 = lw->d_Q14;
    i32* nrg = lw->nrg;
    i32* w = lw->w;
    i32* rr = lw->rr;
    i32* corr_rshifts = lw->corr_rshifts;
    SX_PAR(k, SX_NB_SUBFR) {
        i32* WLTP_ptr = WLTP + k * 25;
        const i16* r_ptr = res_pitch + mem_offset + k * subfr_length;     // r_first / r_last of the reference address one timeline
        const i16* lag_ptr = r_ptr - (lag[k] + SX_LTP_ORDER / 2);
        i32 rr_shifts, rrk;
        sx_sum_sqr_shift_n<SX_SUBFR>(&rrk, &rr_shifts, r_ptr, 0);
        int LZs = sx_clz32(rrk);
        if (LZs < HEAD) {
            rrk = sx_rshift_round(rrk, HEAD - LZs);
            rr_shifts += HEAD - LZs;
        }
        corr_rshifts[k] = sx_corr_matrix_diag(lag_ptr, subfr_length, SX_LTP_ORDER, HEAD, WLTP_ptr, rr_shifts, lag[k] & 1);
        rr[k] = rrk;
        lw->rr_shifts[k] = rr_shifts;
    }
    wv_sync();
    SX_PAR(t, SX_NB_SUBFR * (2 * SX_LTP_ORDER - 1)) {
        const int k = t / (2 * SX_LTP_ORDER - 1), m = t - k * (2 * SX_LTP_ORDER - 1);
        const i16* r_ptr = res_pitch + mem_offset + k * subfr_length;
        lw->ip[k][m] = sx_corr_inner(r_ptr - (lag[k] + SX_LTP_ORDER / 2), r_ptr, subfr_length, SX_LTP_ORDER, m, corr_rshifts[k]);
    }
    wv_sync();
    SX_PAR(k, SX_NB_SUBFR) {
        i16* b_Q14_ptr = b_Q14 + k * SX_LTP_ORDER;
        i32* WLTP_ptr = WLTP + k * 25;
        i32* Rr = lw->Rr[k];
        i32* b_Q16 = lw->b_Q16[k];
        const i16* r_ptr = res_pitch + mem_offset + k * subfr_length;
        const i16* lag_ptr = r_ptr - (lag[k] + SX_LTP_ORDER / 2);
        const i32 rr_shifts = lw->rr_shifts[k], crs = corr_rshifts[k];
        i32 rrk = rr[k];
        sx_corr_matrix_off(lag_ptr, subfr_length, SX_LTP_ORDER, WLTP_ptr, crs, lw->ip[k]);
        for (int i = 0; i < SX_LTP_ORDER; i++) Rr[i] = lw->ip[k][SX_LTP_ORDER - 1 + i];
        if (crs > rr_shifts) rrk = rrk >> (crs - rr_shifts);
        i32 regu = 1;
        regu = sx_smlawb(regu, rrk, K_LTP_DAMPING_DIV3_Q16);
        regu = sx_smlawb(regu, WLTP_ptr[0], K_LTP_DAMPING_DIV3_Q16);
        regu = sx_smlawb(regu, WLTP_ptr[24], K_LTP_DAMPING_DIV3_Q16);
        for (int i = 0; i < 5; i++) WLTP_ptr[i * 5 + i] = sx_add(WLTP_ptr[i * 5 + i], regu);
        rrk += regu;
        sx_solve_LDL(WLTP_ptr, SX_LTP_ORDER, Rr, b_Q16, lw->ldl[k]);
        for (int i = 0; i < 5; i++) b_Q14_ptr[i] = (i16)sx_sat16(sx_rshift_round(b_Q16[i], 2));
        const i32 nrgk = sx_residual_energy16_covar(b_Q14_ptr, WLTP_ptr, Rr, rrk, SX_LTP_ORDER, 14);
        int extra_shifts = sx_min(crs, HEAD);
        i32 denom32 = sx_add(sx_lshift_sat32(sx_smulwb(nrgk, Wght_Q15[k]), 1 + extra_shifts),
                             sx_smulwb(subfr_length, 655) >> (crs - extra_shifts));
        denom32 = sx_max(denom32, 1);
        i32 temp32 = sx_shl(Wght_Q15[k], 16) / denom32;
        temp32 = temp32 >> (31 + crs - extra_shifts - 26);
        i32 WLTP_max = 0;
        for (int i = 0; i < 25; i++) WLTP_max = sx_max(WLTP_ptr[i], WLTP_max);
        int lshift = sx_clz32(WLTP_max) - 1 - 3;
        if (26 - 18 + lshift < 31) temp32 = sx_min(temp32, sx_shl(1, 26 - 18 + lshift));
        for (int i = 0; i < 25; i++) WLTP_ptr[i] = (i32)(sx_smull(WLTP_ptr[i], temp32) >> 8);
        w[k] = WLTP_ptr[2 * 5 + 2];
        rr[k] = rrk;
        nrg[k] = nrgk;
        corr_rshifts[k] = crs;
    }
    wv_sync();
    int maxRshifts = 0;
    for (int k = 0; k < 4; k++) maxRshifts = sx_max(corr_rshifts[k], maxRshifts);
    {
        i32 LPC_LTP_res_nrg = 0, LPC_res_nrg = 0;
        for (int k = 0; k < 4; k++) {
            LPC_res_nrg = sx_add(LPC_res_nrg, sx_add(sx_smulwb(rr[k], Wght_Q15[k]), 1) >> (1 + (maxRshifts - corr_rshifts[k])));
            LPC_LTP_res_nrg = sx_add(LPC_LTP_res_nrg, sx_add(sx_smulwb(nrg[k], Wght_Q15[k]), 1) >> (1 + (maxRshifts - corr_rshifts[k])));
        }
        LPC_LTP_res_nrg = sx_max(LPC_LTP_res_nrg, 1);
        i32 div_Q16 = sx_div32_varQ(LPC_res_nrg, LPC_LTP_res_nrg, 16);
        *LTPredCodGain_Q7 = sx_smulbb(3, sx_lin2log(div_Q16) - (16 << 7));
    }
    SX_PAR(k, SX_NB_SUBFR) {
        i32 d = 0;
        for (int i = 0; i < 5; i++) d += b_Q14[k * 5 + i];
        d_Q14[k] = d;
    }
    wv_sync();
    i32 max_abs_d_Q14 = 0, max_w_bits = 0;
    for (int k = 0; k < 4; k++) {
        max_abs_d_Q14 = sx_max(max_abs_d_Q14, sx_abs(d_Q14[k]));
        max_w_bits = sx_max(max_w_bits, 32 - sx_clz32(w[k]) + corr_rshifts[k] - maxRshifts);
    }
    int extra_shifts = max_w_bits + 32 - sx_clz32(max_abs_d_Q14) - 14;
    extra_shifts -= (32 - 1 - 2 + maxRshifts);
    extra_shifts = sx_max(extra_shifts, 0);
    int maxRshifts_wxtra = maxRshifts + extra_shifts;
    i32 temp32 = (262 >> (maxRshifts + extra_shifts)) + 1;
    i32 wd = 0;
    for (int k = 0; k < 4; k++) {
        temp32 = sx_add(temp32, w[k] >> (maxRshifts_wxtra - corr_rshifts[k]));
        wd = sx_add(wd, sx_shl(sx_smulww(w[k] >> (maxRshifts_wxtra - corr_rshifts[k]), d_Q14[k]), 2));
    }
    i32 m_Q12 = sx_div32_varQ(wd, temp32, 12);
    SX_PAR(k, SX_NB_SUBFR) {
        i16* b_Q14_ptr = b_Q14 + k * 5;
        i32* delta_b_Q14 = lw->delta_b_Q14[k];
        i32 t32;
        if (2 - corr_rshifts[k] > 0) t32 = w[k] >> (2 - corr_rshifts[k]);
        else t32 = sx_lshift_sat32(w[k], corr_rshifts[k] - 2);
        i32 g_Q26 = sx_mul(K_LTP_SMOOTHING_Q26 / ((K_LTP_SMOOTHING_Q26 >> 10) + t32),
                           sx_lshift_sat32(sx_sub_sat32(m_Q12, d_Q14[k] >> 2), 4));
        t32 = 0;
        for (int i = 0; i < 5; i++) {
            delta_b_Q14[i] = sx_max(b_Q14_ptr[i], 1638);
            t32 += delta_b_Q14[i];
        }
        t32 = g_Q26 / t32;
        for (int i = 0; i < 5; i++)
            b_Q14_ptr[i] = (i16)sx_limit((i32)b_Q14_ptr[i] + sx_smulwb(sx_lshift_sat32(t32, 4), delta_b_Q14[i]), -16000, 28000);
    }
    wv_sync();
}

// SKP_Silk_VQ_WMat_EC_FIX, SKP_Silk_VQ_nearest_neighbor_FIX.c:31 (generic, non-packed form of the arithmetic)
SX_HD void sx_vq_wmat_ec(i32* ind, i32* rate_dist_Q14, const i16* in_Q14, const i32* W, const i16* cb_Q14, const i16* cl_Q6, i32 mu_Q8, int L) {
    *rate_dist_Q14 = SX_I32_MAX;
    for (int k = 0; k < L; k++) {
        const i16* row = &cb_Q14[k * 5];
        i32 d0 = (i16)(in_Q14[0] - row[0]), d1 = (i16)(in_Q14[1] - row[1]), d2 = (i16)(in_Q14[2] - row[2]),
            d3 = (i16)(in_Q14[3] - row[3]), d4 = (i16)(in_Q14[4] - row[4]);
        i32 sum1 = sx_smulbb(mu_Q8, cl_Q6[k]), sum2;
        sum2 = sx_smulwb(W[1], d1); sum2 = sx_smlawb(sum2, W[2], d2); sum2 = sx_smlawb(sum2, W[3], d3); sum2 = sx_smlawb(sum2, W[4], d4);
        sum2 = sx_shl(sum2, 1); sum2 = sx_smlawb(sum2, W[0], d0); sum1 = sx_smlawb(sum1, sum2, d0);
        sum2 = sx_smulwb(W[7], d2); sum2 = sx_smlawb(sum2, W[8], d3); sum2 = sx_smlawb(sum2, W[9], d4);
        sum2 = sx_shl(sum2, 1); sum2 = sx_smlawb(sum2, W[6], d1); sum1 = sx_smlawb(sum1, sum2, d1);
        sum2 = sx_smulwb(W[13], d3); sum2 = sx_smlawb(sum2, W[14], d4);
        sum2 = sx_shl(sum2, 1); sum2 = sx_smlawb(sum2, W[12], d2); sum1 = sx_smlawb(sum1, sum2, d2);
        sum2 = sx_smulwb(W[19], d4);
        sum2 = sx_shl(sum2, 1); sum2 = sx_smlawb(sum2, W[18], d3); sum1 = sx_smlawb(sum1, sum2, d3);
        sum2 = sx_smulwb(W[24], d4); sum1 = sx_smlawb(sum1, sum2, d4);
        if (sum1 < *rate_dist_Q14) { *rate_dist_Q14 = sum1; *ind = k; }
    }
}

// one codebook entry of SKP_Silk_VQ_WMat_EC_FIX (SKP_Silk_VQ_nearest_neighbor_FIX.c:31)
SX_HD i32 sx_vq_wmat_ec_one(const i16* in_Q14, const i32* W, const i16* row, i32 cl_Q6, i32 mu_Q8) {
    const i32 d0 = (i16)(in_Q14[0] - row[0]), d1 = (i16)(in_Q14[1] - row[1]), d2 = (i16)(in_Q14[2] - row[2]),
              d3 = (i16)(in_Q14[3] - row[3]), d4 = (i16)(in_Q14[4] - row[4]);
    i32 sum1 = sx_smulbb(mu_Q8, cl_Q6), sum2;
    sum2 = sx_smulwb(W[1], d1); sum2 = sx_smlawb(sum2, W[2], d2); sum2 = sx_smlawb(sum2, W[3], d3); sum2 = sx_smlawb(sum2, W[4], d4);
    sum2 = sx_shl(sum2, 1); sum2 = sx_smlawb(sum2, W[0], d0); sum1 = sx_smlawb(sum1, sum2, d0);
    sum2 = sx_smulwb(W[7], d2); sum2 = sx_smlawb(sum2, W[8], d3); sum2 = sx_smlawb(sum2, W[9], d4);
    sum2 = sx_shl(sum2, 1); sum2 = sx_smlawb(sum2, W[6], d1); sum1 = sx_smlawb(sum1, sum2, d1);
    sum2 = sx_smulwb(W[13], d3); sum2 = sx_smlawb(sum2, W[14], d4);
    sum2 = sx_shl(sum2, 1); sum2 = sx_smlawb(sum2, W[12], d2); sum1 = sx_smlawb(sum1, sum2, d2);
    sum2 = sx_smulwb(W[19], d4);
    sum2 = sx_shl(sum2, 1); sum2 = sx_smlawb(sum2, W[18], d3); sum1 = sx_smlawb(sum1, sum2, d3);
    sum2 = sx_smulwb(W[24], d4); sum1 = sx_smlawb(sum1, sum2, d4);
    return sum1;
}

// SKP_Silk_quant_LTP_gains_FIX, SKP_Silk_quant_LTP_gains_FIX.c:30 (lowComplexity = 0): all (codebook, subframe, entry)
// rate-distortions at once (3 x 4 x up to 40 = 280 lanes' worth), then one lane per (codebook, subframe) picks the first minimum
SX_FN1 void sx_quant_LTP_gains(i16* B_Q14, i32* cbk_index, i32* periodicity_index, const i32* W_Q18, i32 mu_Q8, i32* rd /* [3][4][40] LDS */,
                              i32* best /* [3][4][2] LDS */) {
    SX_IN_LDS(B_Q14); SX_IN_LDS(cbk_index); SX_IN_LDS(periodicity_index); SX_IN_LDS(W_Q18); SX_IN_LDS(rd); SX_IN_LDS(best);
    // (the three codebooks hold 10 + 20 + 40 = 70 entries: the 4 x 70 evaluations are numbered densely, five rounds of the wave)
    SX_PAR(t, 4 * 70) {
        const int j = t / 70, r = t - j * 70;
        const int k = r < 10 ? 0 : (r < 30 ? 1 : 2), e = r - (r < 10 ? 0 : (r < 30 ? 10 : 30));
        const i16* cl = k == 0 ? T_bits_ltp_gain0_Q6 : (k == 1 ? T_bits_ltp_gain1_Q6 : T_bits_ltp_gain2_Q6);
        const i16* cbk = k == 0 ? T_ltp_vq0_Q14 : (k == 1 ? T_ltp_vq1_Q14 : T_ltp_vq2_Q14);
        rd[k * 160 + j * 40 + e] = sx_vq_wmat_ec_one(&B_Q14[j * 5], &W_Q18[j * 25], &cbk[e * 5], cl[e], mu_Q8);
    }
    wv_sync();
#ifdef SX_LANE_STREAM
    {   // a (codebook, subframe) pair on the four lanes of a quad, ten entries each; the quad's first minimum by two DPP steps
        const int kj = SX_LANE >> 2, q = SX_LANE & 3;
        i32 bv = SX_I32_MAX, bi = 0;
        if (kj < 12) {
            const int L = T_ltp_vq_sizes[kj >> 2];
            for (int e = 10 * q; e < 10 * q + 10; e++) {
                const i32 v = e < L ? rd[kj * 40 + e] : SX_I32_MAX;
                if (v < bv) { bv = v; bi = e; }
            }
        }
        SX_ARG_STEP(0xB1, <) SX_ARG_STEP(0x4E, <)
        if (kj < 12 && q == 0) { best[kj * 2] = bv; best[kj * 2 + 1] = bi; }
    }
#else
    SX_PAR(kj, 12) {
        const int k = kj >> 2;
        const int L = T_ltp_vq_sizes[k];
        i32 bv = SX_I32_MAX, bi = 0;
        for (int e = 0; e < L; e++) {
            const i32 v = rd[kj * 40 + e];
            if (v < bv) { bv = v; bi = e; }
        }
        best[kj * 2] = bv;
        best[kj * 2 + 1] = bi;
    }
#endif
    wv_sync();
    i32 min_rate_dist = SX_I32_MAX;
    int per = 0;
    for (int k = 0; k < 3; k++) {
        i32 rate_dist = 0;
        for (int j = 0; j < 4; j++) rate_dist = sx_add_pos_sat32(rate_dist, best[(k * 4 + j) * 2]);
        rate_dist = sx_min(SX_I32_MAX - 1, rate_dist);
        if (rate_dist < min_rate_dist) { min_rate_dist = rate_dist; per = k; }
    }
    wv_sync();
    *periodicity_index = per;
    const i16* cbk = per == 0 ? T_ltp_vq0_Q14 : (per == 1 ? T_ltp_vq1_Q14 : T_ltp_vq2_Q14);
    SX_PAR(t, 20) {
        const int j = t / 5, i = t - j * 5;
        const int ix = best[(per * 4 + j) * 2 + 1];
        if (i == 0) cbk_index[j] = ix;
        B_Q14[t] = cbk[ix * 5 + i];
    }
    wv_sync();
}

// SKP_Silk_LTP_scale_ctrl_FIX, SKP_Silk_LTP_scale_ctrl_FIX.c:39 (PacketLoss_perc = 0, PacketSize_ms = 20 x frames per packet)
SX_HD void sx_LTP_scale_ctrl(SxEncState* st, SxEncCtrl* c) {
    st->HPLTPredCodGain_Q7 = sx_max(c->LTPredCodGain_Q7 - st->prevLTPredCodGain_Q7, 0) + sx_rshift_round(st->HPLTPredCodGain_Q7, 1);
    st->prevLTPredCodGain_Q7 = c->LTPredCodGain_Q7;
    i32 g_out_Q5 = sx_rshift_round((c->LTPredCodGain_Q7 >> 1) + (st->HPLTPredCodGain_Q7 >> 1), 3);
    i32 g_limit_Q15 = sx_sigm_Q15(g_out_Q5 - (3 << 5));
    c->LTP_scaleIndex = 0;
    if (st->nFramesInPayloadBuf == 0) {
        int round_loss = 0 + st->fpp - 1;
        i32 thrld1 = T_ltp_scale_thresholds_Q15[sx_min(round_loss, 10)];
        i32 thrld2 = T_ltp_scale_thresholds_Q15[sx_min(round_loss + 1, 10)];
        if (g_limit_Q15 > thrld1) c->LTP_scaleIndex = 2;
        else if (g_limit_Q15 > thrld2) c->LTP_scaleIndex = 1;
    }
    c->LTP_scale_Q14 = T_ltp_scales_Q14[c->LTP_scaleIndex];
}

// SKP_Silk_LTP_analysis_filter_FIX, SKP_Silk_LTP_analysis_filter_FIX.c:30 (wave-parallel over the 4 x 50 outputs)
SX_HD void sx_LTP_analysis_filter(i16* LTP_res, const i16* x, const i16* LTPCoef_Q14, const i32* pitchL, const i32* invGains_Q16) {
    const int n = SX_SUBFR + SX_LPC;
    SX_PAR(t, 4 * n) {
        int k = t / n, i = t - k * n;
        const i16* x_ptr = x + k * SX_SUBFR;
        const i16* x_lag_ptr = x_ptr - pitchL[k] + i;
        const i16* B = &LTPCoef_Q14[k * 5];
        i32 est = sx_smulbb(x_lag_ptr[2], B[0]);
        for (int j = 1; j < 5; j++) est = sx_smlabb(est, x_lag_ptr[2 - j], B[j]);
        est = sx_rshift_round(est, 14);
        i32 r = sx_sat16((i32)x_ptr[i] - est);
        LTP_res[k * n + i] = (i16)sx_smulwb(invGains_Q16[k], r);
    }
}

// ---------------------------------------------------------------------------------------------------
// LPC analysis: Burg, A2NLSF, interpolation search
// ---------------------------------------------------------------------------------------------------
// LDS scratch of the LPC analysis (Burg, A2NLSF, interpolation search): wave-uniform arrays live here, never in
// per-lane scratch memory
struct SxBurgWork {
    i32 Cf[SX_MAX_LPC], Cl[SX_MAX_LPC], Af[SX_MAX_LPC], CAf[SX_MAX_LPC + 1], CAb[SX_MAX_LPC + 1];
    i32 T1[4], T2[4];
    i32 p1[64], p2[64];              // per (subframe, k) partial products / per-k reduction terms
    i32 q3[SX_MAX_LPC], q4[SX_MAX_LPC];
};
struct SxA2nlsfGrid {                // the two polynomials on the 129-point cosine grid (LDS), plus the grid itself
    i32 x[129 + 1], yP[129 + 1], yQ[129 + 1];
};

struct SxLpcWork {
    SxBurgWork burg;
    i32 a_Q16[SX_MAX_LPC], a_tmp_Q16[SX_MAX_LPC], NLSF0_Q15[SX_MAX_LPC];
    i16 a_tmp_Q12[SX_MAX_LPC];
    i32 P[SX_MAX_LPC / 2 + 1], Q[SX_MAX_LPC / 2 + 1];
    union {
        SxA2nlsfGrid grid;
        struct {                     // interpolation search: the four candidates side by side
            i32 NLSF0[4][SX_MAX_LPC];
            i16 a_Q12[4][SX_MAX_LPC];
            i32 ws[4][SX_NLSF2A_WS];
            i16 res[4][2 * (SX_SUBFR + SX_LPC)];
            i32 nrg[4][2], rsh[4][2];
        } it;
    } u;
};

// SKP_Silk_burg_modified, SKP_Silk_burg_modified.c:49.  The reference's scalar loops over the coefficient index k are
// spread over lanes (all its accumulations are wrapping int32 sums of independent terms, so lane order is immaterial):
// lane (s, k) builds the per-subframe prediction terms, lane k owns row / correlation element k.
// (SITE: one instance per call site -- a stage function that several sites call with different work areas, lengths and orders is compiled
// for none of them; an instance with ONE caller gets that caller's constants: LDS addresses as immediates, loop bounds, the filter order)
template <int SITE = 0>
SX_FN void sx_burg_modified(i32* res_nrg, i32* res_nrg_Q, i32* A_Q16, const i16* x, int subfr_length, int nb_subfr, i32 WhiteNoiseFrac_Q32,
                            int D, SxBurgWork* bw) {
    SX_IN_LDS(x); SX_IN_LDS(bw); SX_IN_LDS(A_Q16);
    const int QA = 25, MAX_RSHIFTS = 32 - 25, MIN_RSHIFTS = -16;
    // (arguments of a real call arrive in vector registers: tell the compiler they are wave-uniform, so that the control flow below
    // becomes scalar branches instead of exec-mask regions that run both sides)
    subfr_length = SX_UNI(subfr_length); nb_subfr = SX_UNI(nb_subfr); D = SX_UNI(D); WhiteNoiseFrac_Q32 = SX_UNI(WhiteNoiseFrac_Q32);
    const int L = subfr_length;
    i32 C0, rshifts;
    SX_T_BEGIN
    // (pairs of samples per lane at this call site: the frame's 4 x (SX_SUBFR + SX_LPC) samples, the high band's 4 x SX_HB_LPCBLK)
    constexpr int PAIRS_PER_LANE = SITE == 2 ? (4 * (10 * SX_FS_KHZ + SX_HB_LPC) / 2 + 63) / 64 : (4 * (SX_SUBFR + SX_MAX_LPC) / 2 + 63) / 64;
    sx_sum_sqr_shift_wv_inl<PAIRS_PER_LANE>(&C0, &rshifts, x, nb_subfr * subfr_length, 0);
    C0 = SX_UNI(C0); rshifts = SX_UNI(rshifts);
    SX_T(23)
    if (rshifts > MAX_RSHIFTS) {
        C0 = sx_shl(C0, rshifts - MAX_RSHIFTS);
        rshifts = MAX_RSHIFTS;
    } else {
        int lz = sx_clz32(C0) - 1;
        int rshifts_extra = 2 - lz;
        if (rshifts_extra > 0) {
            rshifts_extra = sx_min(rshifts_extra, MAX_RSHIFTS - rshifts);
            C0 = C0 >> rshifts_extra;
        } else {
            rshifts_extra = sx_max(rshifts_extra, MIN_RSHIFTS - rshifts);
            C0 = sx_shl(C0, -rshifts_extra);
        }
        rshifts += rshifts_extra;
    }
    C0 = SX_UNI(C0); rshifts = SX_UNI(rshifts);
    // first row of the correlation matrix: lane (s, n) computes one inner product serially
    SX_PAR(sn, nb_subfr * 16) {
        const int s = sn >> 4, n = (sn & 15) + 1;
        i32 v = 0;
        if (n <= D) {
            const i16* xs = x + s * L;
            if (rshifts > 0) {
                i64 acc = 0;
                for (int i = 0; i < L - n; i++) acc += (i64)((i32)xs[i] * (i32)xs[i + n]);
                v = (i32)(acc >> rshifts);
            } else {
                i32 acc = 0;
                for (int i = 0; i < L - n; i++) acc = sx_smlabb(acc, xs[i], xs[i + n]);
                v = sx_shl(acc, -rshifts);
            }
        }
        bw->p1[sn] = v;
    }
    wv_sync();
    SX_PAR(k, SX_MAX_LPC) {
        i32 v = 0;
        for (int s = 0; s < nb_subfr; s++) v = sx_add(v, bw->p1[s * 16 + k]);
        bw->Cf[k] = v;
        bw->Cl[k] = v;
        bw->Af[k] = 0;
    }
    SX_T(24)
    const i32 CA0 = sx_add(sx_add(C0, sx_smmul(WhiteNoiseFrac_Q32, C0)), 1);
    bw->CAf[0] = CA0;
    bw->CAb[0] = CA0;
    wv_sync();
#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64 && SX_FS_KHZ == 8
    // (orders <= 15: element k of every vector sits in column k of a 16-lane row; the order-16 analysis of the wide-band build needs
    // 17 correlation entries and takes the LDS form below)
    // Register-resident recursion: lane (s, k) = (row, column) of the wave.  Column k of a row holds coefficient / correlation
    // element k (Af in every row; the correlation rows Cf / Cl in rows 0 - 1, CAf / CAb in rows 2 - 3), so the per-subframe terms
    // reduce inside a row (DPP) and index reversals n-1-k are one lane gather; nothing goes through LDS but the signal.
    i32 nrg, tmp1;
    {
        const int row = SX_LANE >> 4, k = SX_LANE & 15, rowbase = SX_LANE & 48;
        const int srow = row < nb_subfr ? row : 0;            // rows past nb_subfr compute a copy of subframe 0 and are ignored
        const i16* xr = x + srow * L;
        // The two correlation rows (first: Cf, last: Cl) are kept by rows 0 and 1 of the wave, the forward / backward correlations (CAf, CAb)
        // by rows 2 and 3 -- P_k = Cf | CAf, Q_k = Cl | CAb: the four per-subframe terms of step (b) then reach the rows that keep their
        // totals with three lane swaps (a transposing reduction) instead of two swaps each into every row.
        const bool upper = row >= 2;
        i32 Af_k = 0, P_k = upper ? (k == 0 ? CA0 : 0) : bw->Cf[k], Q_k = P_k;
        // element index of step (c)'s gather, gidx_a + (n & gidx_n): rows 0, 1: n - k - 1; row 2: n - k; row 3: k + 1
        const int gidx_a = row < 2 ? -k - 1 : (row == 2 ? -k : k + 1), gidx_n = row < 3 ? -1 : 0;
#define SX_GATHER(v, idx) __shfl((v), rowbase | ((idx) & 15), 64)
        // (the recursion exists twice, once per scaling regime of the input -- `hi` = rshifts > -2, the usual one -- so that the regime is
        // a compile-time constant inside the loop: a wave-uniform condition nested in the per-column conditions below is otherwise
        // carried as an execution mask through every step)
        auto recursion = [&](auto regime) {
        constexpr bool hi = decltype(regime)::value;
        for (int n = 0; n < D; n++) {
            // (a) forward / backward prediction errors at the two edges of subframe `row`
            i32 a = 0, b = 0;
            if (k < n) {
                if (hi) {
                    a = sx_smulwb(Af_k, xr[n - k - 1]);
                    b = sx_smulwb(Af_k, xr[L - n + k]);
                } else {
                    const i32 Atmp1 = sx_rshift_round(Af_k, QA - 17);
                    a = sx_mul(xr[n - k - 1], Atmp1);
                    b = sx_mul(xr[L - n + k], Atmp1);
                }
            }
            a = wv_row_sum(a);
            b = wv_row_sum(b);
            i32 t1 = sx_add(sx_shl((i32)xr[n], hi ? QA - 16 : 17), a), t2 = sx_add(sx_shl((i32)xr[L - n - 1], hi ? QA - 16 : 17), b);
            t1 = sx_neg(t1); t2 = sx_neg(t2);
            if (hi) { t1 = sx_shl(t1, 32 - QA - rshifts); t2 = sx_shl(t2, 32 - QA - rshifts); }
            // (b) column k: correlation rows and forward / backward correlations.  Lane (row, k) adds the terms of ITS subframe (the
            // row's own prediction errors t1 / t2 are already in its lanes), the four rows are then summed column by column
            // (wrapping adds: the order is free), which leaves the same totals in every row
            {
                i32 dcf = 0, dcl = 0, dcaf = 0, dcab = 0;
                if (row < nb_subfr) {
                    const i16* xs = xr;
                    const int kk = sx_min(k, n);                       // columns past n only feed lanes that are never read
                    if (hi) {
                        const i32 x1 = sx_neg(sx_shl((i32)xs[n], 16 - rshifts)), x2 = sx_neg(sx_shl((i32)xs[L - n - 1], 16 - rshifts));
                        if (k < n) {
                            dcf = sx_smulwb(x1, xs[n - k - 1]);
                            dcl = sx_smulwb(x2, xs[L - n + k]);
                        }
                        dcaf = sx_smulwb(t1, xs[n - kk]);
                        dcab = sx_smulwb(t2, xs[L - n + kk - 1]);
                    } else {
                        const i32 x1 = sx_neg(sx_shl((i32)xs[n], -rshifts)), x2 = sx_neg(sx_shl((i32)xs[L - n - 1], -rshifts));
                        if (k < n) {
                            dcf = sx_mul(x1, xs[n - k - 1]);
                            dcl = sx_mul(x2, xs[L - n + k]);
                        }
                        dcaf = sx_smulww(t1, sx_shl((i32)xs[n - kk], -rshifts - 1));
                        dcab = sx_smulww(t2, sx_shl((i32)xs[L - n + kk - 1], -rshifts - 1));
                    }
                }
                {   // v_permlane32_swap(X, Y): {X.lower | Y.lower}, {X.upper | Y.upper}: their sum holds X's half-sums in rows 0, 1 and Y's in
                    // rows 2, 3; v_permlane16_swap of that with itself: the even row of each half in both of its rows, and the odd one
                    auto p_ = __builtin_amdgcn_permlane32_swap(dcf, dcaf, false, false);
                    auto q_ = __builtin_amdgcn_permlane32_swap(dcl, dcab, false, false);
                    const i32 u_ = sx_add((i32)p_[0], (i32)p_[1]), v_ = sx_add((i32)q_[0], (i32)q_[1]);
                    auto pu_ = __builtin_amdgcn_permlane16_swap(u_, u_, false, false);
                    auto qv_ = __builtin_amdgcn_permlane16_swap(v_, v_, false, false);
                    const i32 dP = sx_add((i32)pu_[0], (i32)pu_[1]), dQ = sx_add((i32)qv_[0], (i32)qv_[1]);
                    if (upper ? k <= n : k < n) { P_k = sx_add(P_k, dP); Q_k = sx_add(Q_k, dQ); }
                }
            }
            // (c) reflection coefficient: numerator / denominator terms of column k < n.  The four sums over k (against the reversed
            // Cl, Cf, CAb and the shifted CAb + CAf) are taken by the four rows, one each: the rows hold identical copies of the vectors
            i32 s1, s2, num, den;
            {
                const i32 src = row < 2 ? (row == 0 ? Q_k : P_k) : (row == 2 ? Q_k : sx_add(Q_k, P_k));
                const i32 g = SX_GATHER(src, gidx_a + (n & gidx_n));
                i32 pq = 0;
                if (k < n) {
                    int lz = sx_clz32(sx_abs(Af_k)) - 1;
                    lz = sx_min(32 - QA, lz);
                    pq = sx_shl(sx_smmul(g, sx_shl(Af_k, lz)), 32 - QA - lz);
                }
                const i32 tot = wv_row_sum(pq);
                s1 = sx_add(__builtin_amdgcn_readlane(tot, 0), __builtin_amdgcn_readlane(P_k, n));
                s2 = sx_add(__builtin_amdgcn_readlane(tot, 16), __builtin_amdgcn_readlane(Q_k, n));
                num = __builtin_amdgcn_readlane(tot, 32);
                den = sx_add(__builtin_amdgcn_readlane(tot, 48), sx_add(__builtin_amdgcn_readlane(Q_k, 32), __builtin_amdgcn_readlane(P_k, 32)));
            }
            if (upper && k == n + 1) { P_k = s1; Q_k = s2; }
            num = sx_add(num, s2);
            num = sx_shl(sx_neg(num), 1);
            if (!(sx_abs(num) < den)) {
                if (k >= n) Af_k = 0;
                break;
            }
            const i32 rc_Q31 = sx_div32_varQ(num, den, 31);
            // (d) symmetric coefficient update and correlation update
            {
                const i32 Af_r = SX_GATHER(Af_k, n - k - 1);
                const i32 CAb_r = SX_GATHER(Q_k, n + 1 - k), CAf_r = SX_GATHER(P_k, n + 1 - k);
                if (k < n) Af_k = sx_add(Af_k, sx_shl(sx_smmul(Af_r, rc_Q31), 1));
                if (upper && k <= n + 1) {
                    P_k = sx_add(P_k, sx_shl(sx_smmul(CAb_r, rc_Q31), 1));
                    Q_k = sx_add(Q_k, sx_shl(sx_smmul(CAf_r, rc_Q31), 1));
                }
                if (k == n) Af_k = rc_Q31 >> (31 - QA);
            }
        }
        };
        if (rshifts > -2) recursion(SxConst<bool, true>{}); else recursion(SxConst<bool, false>{});
#undef SX_GATHER
        SX_T(25)
        // residual energy and output
        const i32 At = k < D ? sx_rshift_round(Af_k, QA - 16) : 0;
        const i32 CAf_next = __shfl(P_k, rowbase | ((k + 1) & 15), 64);
        nrg = sx_add(__builtin_amdgcn_readlane(P_k, 32), __builtin_amdgcn_readlane(wv_row_sum(k < D ? sx_smulww(CAf_next, At) : 0), 32));
        tmp1 = sx_add(1 << 16, SX_UNI(wv_row_sum(sx_smulww(At, At))));
        if (SX_LANE < D) A_Q16[SX_LANE] = sx_neg(At);
    }
    *res_nrg = sx_smlaww(nrg, sx_smmul(WhiteNoiseFrac_Q32, C0), sx_neg(tmp1));
    *res_nrg_Q = -rshifts;
    wv_sync();
}
#else
    for (int n = 0; n < D; n++) {
        // (a) per (subframe, k): terms of the forward / backward prediction errors at the two subframe edges
        SX_PAR(sk, nb_subfr * 16) {
            const int s = sk >> 4, k = sk & 15;
            i32 a = 0, b = 0;
            if (k < n) {
                const i16* xs = x + s * L;
                if (rshifts > -2) {
                    const i32 Atmp_QA = bw->Af[k];
                    a = sx_smulwb(Atmp_QA, xs[n - k - 1]);
                    b = sx_smulwb(Atmp_QA, xs[L - n + k]);
                } else {
                    const i32 Atmp1 = sx_rshift_round(bw->Af[k], QA - 17);
                    a = sx_mul(xs[n - k - 1], Atmp1);
                    b = sx_mul(xs[L - n + k], Atmp1);
                }
            }
            bw->p1[sk] = a;
            bw->p2[sk] = b;
        }
        wv_sync();
        SX_PAR(s, nb_subfr) {
            const i16* xs = x + s * L;
            i32 tmp1, tmp2;
            if (rshifts > -2) {
                tmp1 = sx_shl((i32)xs[n], QA - 16);
                tmp2 = sx_shl((i32)xs[L - n - 1], QA - 16);
            } else {
                tmp1 = sx_shl((i32)xs[n], 17);
                tmp2 = sx_shl((i32)xs[L - n - 1], 17);
            }
            for (int k = 0; k < n; k++) {
                tmp1 = sx_add(tmp1, bw->p1[s * 16 + k]);
                tmp2 = sx_add(tmp2, bw->p2[s * 16 + k]);
            }
            if (rshifts > -2) {
                bw->T1[s] = sx_shl(sx_neg(tmp1), 32 - QA - rshifts);
                bw->T2[s] = sx_shl(sx_neg(tmp2), 32 - QA - rshifts);
            } else {
                bw->T1[s] = sx_neg(tmp1);
                bw->T2[s] = sx_neg(tmp2);
            }
        }
        wv_sync();
        // (b) lane k: update correlation rows and the forward / backward correlations with all subframes
        SX_PAR(k, n + 1) {
            i32 cf = 0, cl = 0, caf = bw->CAf[k], cab = bw->CAb[k];
            if (k < n) { cf = bw->Cf[k]; cl = bw->Cl[k]; }
            for (int s = 0; s < nb_subfr; s++) {
                const i16* xs = x + s * L;
                if (rshifts > -2) {
                    const i32 x1 = sx_neg(sx_shl((i32)xs[n], 16 - rshifts)), x2 = sx_neg(sx_shl((i32)xs[L - n - 1], 16 - rshifts));
                    if (k < n) {
                        cf = sx_smlawb(cf, x1, xs[n - k - 1]);
                        cl = sx_smlawb(cl, x2, xs[L - n + k]);
                    }
                    caf = sx_smlawb(caf, bw->T1[s], xs[n - k]);
                    cab = sx_smlawb(cab, bw->T2[s], xs[L - n + k - 1]);
                } else {
                    const i32 x1 = sx_neg(sx_shl((i32)xs[n], -rshifts)), x2 = sx_neg(sx_shl((i32)xs[L - n - 1], -rshifts));
                    if (k < n) {
                        cf = sx_add(cf, sx_mul(x1, xs[n - k - 1]));
                        cl = sx_add(cl, sx_mul(x2, xs[L - n + k]));
                    }
                    caf = sx_smlaww(caf, bw->T1[s], sx_shl((i32)xs[n - k], -rshifts - 1));
                    cab = sx_smlaww(cab, bw->T2[s], sx_shl((i32)xs[L - n + k - 1], -rshifts - 1));
                }
            }
            if (k < n) { bw->Cf[k] = cf; bw->Cl[k] = cl; }
            bw->CAf[k] = caf;
            bw->CAb[k] = cab;
        }
        wv_sync();
        // (c) lane k: its terms of the reflection-coefficient numerator / denominator
        SX_PAR(k, n) {
            const i32 Atmp_QA = bw->Af[k];
            int lz = sx_clz32(sx_abs(Atmp_QA)) - 1;
            lz = sx_min(32 - QA, lz);
            const i32 Atmp1 = sx_shl(Atmp_QA, lz);
            const int sh = 32 - QA - lz;
            bw->p1[k] = sx_shl(sx_smmul(bw->Cl[n - k - 1], Atmp1), sh);
            bw->p2[k] = sx_shl(sx_smmul(bw->Cf[n - k - 1], Atmp1), sh);
            bw->q3[k] = sx_shl(sx_smmul(bw->CAb[n - k], Atmp1), sh);
            bw->q4[k] = sx_shl(sx_smmul(sx_add(bw->CAb[k + 1], bw->CAf[k + 1]), Atmp1), sh);
        }
        wv_sync();
        i32 tmp1 = bw->Cf[n], tmp2 = bw->Cl[n];
        i32 num = 0;
        i32 nrg = sx_add(bw->CAb[0], bw->CAf[0]);
        for (int k = 0; k < n; k++) {
            tmp1 = sx_add(tmp1, bw->p1[k]);
            tmp2 = sx_add(tmp2, bw->p2[k]);
            num = sx_add(num, bw->q3[k]);
            nrg = sx_add(nrg, bw->q4[k]);
        }
        wv_sync();
        bw->CAf[n + 1] = tmp1;
        bw->CAb[n + 1] = tmp2;
        num = sx_add(num, tmp2);
        num = sx_shl(sx_neg(num), 1);
        i32 rc_Q31;
        if (sx_abs(num) < nrg) {
            rc_Q31 = sx_div32_varQ(num, nrg, 31);
        } else {
            wv_sync();
            SX_PAR(k, D - n) bw->Af[n + k] = 0;
            wv_sync();
            break;
        }
        wv_sync();
        // (d) lane k: symmetric coefficient update and correlation update
        SX_PAR(k, (n + 1) >> 1) {
            const i32 t1 = bw->Af[k], t2 = bw->Af[n - k - 1];
            bw->Af[k] = sx_add(t1, sx_shl(sx_smmul(t2, rc_Q31), 1));
            bw->Af[n - k - 1] = sx_add(t2, sx_shl(sx_smmul(t1, rc_Q31), 1));
        }
        SX_PAR(k, n + 2) {
            const i32 t1 = bw->CAf[k], t2 = bw->CAb[n - k + 1];
            bw->CAf[k] = sx_add(t1, sx_shl(sx_smmul(t2, rc_Q31), 1));
            bw->CAb[n - k + 1] = sx_add(t2, sx_shl(sx_smmul(t1, rc_Q31), 1));
        }
        wv_sync();
        bw->Af[n] = rc_Q31 >> (31 - QA);
        wv_sync();
    }
    SX_T(25)
    i32 nrg = bw->CAf[0];
    i32 tmp1 = 1 << 16;
    for (int k = 0; k < D; k++) {
        const i32 Atmp1 = sx_rshift_round(bw->Af[k], QA - 16);
        nrg = sx_smlaww(nrg, bw->CAf[k + 1], Atmp1);
        tmp1 = sx_smlaww(tmp1, Atmp1, Atmp1);
    }
    SX_PAR(k, D) A_Q16[k] = sx_neg(sx_rshift_round(bw->Af[k], QA - 16));
    *res_nrg = sx_smlaww(nrg, sx_smmul(WhiteNoiseFrac_Q32, C0), sx_neg(tmp1));
    *res_nrg_Q = -rshifts;
    wv_sync();
}
#endif

// A2NLSF helpers, SKP_Silk_A2NLSF.c:46-123
SX_HD void sx_a2nlsf_trans_poly(i32* p, int dd) {
    for (int k = 2; k <= dd; k++) {
        for (int n = dd; n > k; n--) p[n - 2] -= p[n];
        p[k - 2] -= sx_shl(p[k], 1);
    }
}
SX_HD i32 sx_a2nlsf_eval_poly(const i32* p, i32 x, int dd) {
    i32 y32 = p[dd], x_Q16 = sx_shl(x, 4);
    for (int n = dd - 1; n >= 0; n--) y32 = sx_smlaww(p[n], y32, x_Q16);
    return y32;
}
SX_HD void sx_a2nlsf_init(const i32* a_Q16, i32* P, i32* Q, int dd) {
    P[dd] = 1 << 16;
    Q[dd] = 1 << 16;
    for (int k = 0; k < dd; k++) {
        P[k] = sx_sub(sx_neg(a_Q16[dd - k - 1]), a_Q16[dd + k]);
        Q[k] = sx_add(sx_neg(a_Q16[dd - k - 1]), a_Q16[dd + k]);
    }
    for (int k = dd; k > 0; k--) {
        P[k - 1] -= P[k];
        Q[k - 1] += Q[k];
    }
    sx_a2nlsf_trans_poly(P, dd);
    sx_a2nlsf_trans_poly(Q, dd);
}
// SKP_Silk_A2NLSF, SKP_Silk_A2NLSF.c:127
// SKP_Silk_A2NLSF, SKP_Silk_A2NLSF.c:127.  The reference walks the cosine grid evaluating one polynomial per step; here
// both polynomials are evaluated on the whole grid up front, lane-parallel, and the (inherently serial) root scan reads
// those values from LDS -- only the three bisection points per root are evaluated on the spot.
template <int SITE = 0>
SX_FN void sx_a2nlsf(i32* NLSF, i32* a_Q16, int d, i32* P, i32* Q, SxA2nlsfGrid* g) {
    SX_IN_LDS(NLSF); SX_IN_LDS(a_Q16); SX_IN_LDS(P); SX_IN_LDS(Q); SX_IN_LDS(g);
    d = SX_UNI(d);                           // (an argument of a real call arrives in a vector register: the root scan below is scalar code)
    const int dd = d >> 1;
    int i = 0;
    for (;;) {                               // one pass per bandwidth-expansion retry (A2NLSF.c:259-281)
        sx_a2nlsf_init(a_Q16, P, Q, dd);
        wv_sync();
        SX_PAR(k, 129) {
            const i32 x = T_lsf_cos_Q12[k];
            g->x[k] = x;
            g->yP[k] = sx_a2nlsf_eval_poly(P, x, dd);
            g->yQ[k] = sx_a2nlsf_eval_poly(Q, x, dd);
        }
        wv_sync();
#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64
        // The reference walks the grid serially; where it stops depends only on sign patterns, so the stops are found with
        // bit scans over wave ballots (crossing / sign masks of both polynomials over the 128 intervals), and the refinement of
        // the d roots (3 bisections + interpolation each) runs one root per lane.
        bool retry = false;
        {
            typedef unsigned long long u64m;
            // per polynomial (P, Q) and half h: bit (k - 1 - 64 h) <-> grid interval k = 1..128
            u64m crP0, crP1, crQ0, crQ1, leP0, leP1, leQ0, leQ1, geP0, geP1, geQ0, geQ1;
            {
                const int k = SX_LANE + 1;
                const i32 p0 = g->yP[k - 1], p1 = g->yP[k], q0 = g->yQ[k - 1], q1 = g->yQ[k];
                crP0 = __builtin_amdgcn_ballot_w64((p0 <= 0 && p1 >= 0) || (p0 >= 0 && p1 <= 0));
                crQ0 = __builtin_amdgcn_ballot_w64((q0 <= 0 && q1 >= 0) || (q0 >= 0 && q1 <= 0));
                leP0 = __builtin_amdgcn_ballot_w64(p1 <= 0); geP0 = __builtin_amdgcn_ballot_w64(p1 >= 0);
                leQ0 = __builtin_amdgcn_ballot_w64(q1 <= 0); geQ0 = __builtin_amdgcn_ballot_w64(q1 >= 0);
            }
            {
                const int k = SX_LANE + 65;
                const i32 p0 = g->yP[k - 1], p1 = g->yP[k], q0 = g->yQ[k - 1], q1 = g->yQ[k];
                crP1 = __builtin_amdgcn_ballot_w64((p0 <= 0 && p1 >= 0) || (p0 >= 0 && p1 <= 0));
                crQ1 = __builtin_amdgcn_ballot_w64((q0 <= 0 && q1 >= 0) || (q0 >= 0 && q1 <= 0));
                leP1 = __builtin_amdgcn_ballot_w64(p1 <= 0); geP1 = __builtin_amdgcn_ballot_w64(p1 >= 0);
                leQ1 = __builtin_amdgcn_ballot_w64(q1 <= 0); geQ1 = __builtin_amdgcn_ballot_w64(q1 >= 0);
            }
            // Root r is a root of P (r even) or Q (r odd); after the first one the reference restarts each search from an artificial
            // left value of +4096 (r & 2 == 0) or -4096, so the first test of root r looks at "y <= 0" resp. "y >= 0" of its interval.
            // Which masks a step uses therefore depends on r & 3 only: the scan is written four roots per round with the masks
            // named in the source.  "The first crossing at or after interval k" is tabulated beforehand, interval k by lane k - 1
            // (k <= 64) / k - 65, so that a step fetches it with a v_readlane instead of shifting and scanning two 64-bit masks.
            i32 nxP0, nxP1, nxQ0, nxQ1;
            {
                const int l = SX_LANE;
                const u64m tp0 = crP0 >> l, tp1 = crP1 >> l, tq0 = crQ0 >> l, tq1 = crQ1 >> l;
                nxP1 = tp1 ? l + 65 + (int)__builtin_ctzll(tp1) : 129;
                nxQ1 = tq1 ? l + 65 + (int)__builtin_ctzll(tq1) : 129;
                nxP0 = tp0 ? l + 1 + (int)__builtin_ctzll(tp0) : (crP1 ? 65 + (int)__builtin_ctzll(crP1) : 129);
                nxQ0 = tq0 ? l + 1 + (int)__builtin_ctzll(tq0) : (crQ1 ? 65 + (int)__builtin_ctzll(crQ1) : 129);
            }
            int k = 1;
            const int first_root = SX_UNI(g->yP[0]) < 0 ? 1 : 0;
            int my_k = 0;
            i32 my_art = 0;
            bool fail = false, have_art = false;
#define SX_ROOT_STEP(R, SGN0, SGN1, NX0, NX1, ART)                                                                      \
            if (!fail && (R) >= first_root && (R) < d) {                                                                  \
                bool hit = false;                                                                                       \
                if (have_art) hit = ((k > 64 ? (SGN1) : (SGN0)) >> ((k - 1) & 63)) & 1;   /* only the sign of y[k] matters */ \
                const int k0 = have_art ? k + 1 : k;         /* real crossings from here on: 1 .. 129 */                \
                const int n0_ = __builtin_amdgcn_readlane((NX0), (k0 - 1) & 63), n1_ = __builtin_amdgcn_readlane((NX1), (k0 - 65) & 63); \
                const int kk = hit ? k : (k0 <= 64 ? n0_ : (k0 <= 128 ? n1_ : 129));                                     \
                if (kk > 128) fail = true;                                                                              \
                else { if (SX_LANE == (R)) { my_k = kk; my_art = hit ? (ART) : 0; } k = kk; have_art = true; }           \
            }
            for (int r4 = 0; r4 < d; r4 += 4) {
                SX_ROOT_STEP(r4, leP0, leP1, nxP0, nxP1, 4096)
                SX_ROOT_STEP(r4 + 1, leQ0, leQ1, nxQ0, nxQ1, 4096)
                SX_ROOT_STEP(r4 + 2, geP0, geP1, nxP0, nxP1, -4096)
                SX_ROOT_STEP(r4 + 3, geQ0, geQ1, nxQ0, nxQ1, -4096)
            }
#undef SX_ROOT_STEP
            if (!fail) {
                if (first_root && SX_LANE == 0) NLSF[0] = 0;
                if (SX_LANE >= first_root && SX_LANE < d) {
                    const int kq = my_k;
                    const i32* pp = (SX_LANE & 1) ? Q : P;
                    const i32* yp = (SX_LANE & 1) ? g->yQ : g->yP;
                    i32 xlo = g->x[kq - 1], ylo = my_art != 0 ? my_art : yp[kq - 1], xhi = g->x[kq], yhi = yp[kq];
                    i32 ffrac = -256;
                    for (int m = 0; m < 3; m++) {
                        i32 xmid = sx_rshift_round(xlo + xhi, 1);
                        i32 ymid = sx_a2nlsf_eval_poly(pp, xmid, dd);
                        if ((ylo <= 0 && ymid >= 0) || (ylo >= 0 && ymid <= 0)) {
                            xhi = xmid;
                            yhi = ymid;
                        } else {
                            xlo = xmid;
                            ylo = ymid;
                            ffrac = ffrac + (128 >> m);
                        }
                    }
                    if (sx_abs(ylo) < 65536) {
                        i32 den = ylo - yhi;
                        i32 nom = sx_shl(ylo, 8 - 3) + (den >> 1);
                        if (den != 0) ffrac += nom / den;
                    } else {
                        ffrac += ylo / ((ylo - yhi) >> (8 - 3));
                    }
                    NLSF[SX_LANE] = sx_min(sx_shl(kq, 8) + ffrac, 32767);
                }
            } else {
                i++;
                if (i > 30) {
                    wv_sync();
                    NLSF[0] = (1 << 15) / (d + 1);
                    for (int kf = 1; kf < d; kf++) NLSF[kf] = sx_smulbb(kf + 1, (1 << 15) / (d + 1));
                    wv_sync();
                    return;
                }
                wv_sync();
                sx_bwexpander_32(a_Q16, d, 65536 - sx_smulbb(10 + i, i));
                retry = true;
            }
        }
#else
        const i32* p = P;
        const i32* yp = g->yP;
        i32 xlo = g->x[0], ylo = yp[0], xhi, yhi;
        int root_ix;
        if (ylo < 0) {
            NLSF[0] = 0;
            p = Q;
            yp = g->yQ;
            ylo = yp[0];
            root_ix = 1;
        } else {
            root_ix = 0;
        }
        int k = 1;
        bool retry = false;
        for (;;) {
            xhi = g->x[k];
            yhi = yp[k];
            if ((ylo <= 0 && yhi >= 0) || (ylo >= 0 && yhi <= 0)) {
                i32 ffrac = -256;
                for (int m = 0; m < 3; m++) {
                    i32 xmid = sx_rshift_round(xlo + xhi, 1);
                    i32 ymid = sx_a2nlsf_eval_poly(p, xmid, dd);
                    if ((ylo <= 0 && ymid >= 0) || (ylo >= 0 && ymid <= 0)) {
                        xhi = xmid;
                        yhi = ymid;
                    } else {
                        xlo = xmid;
                        ylo = ymid;
                        ffrac = ffrac + (128 >> m);
                    }
                }
                if (sx_abs(ylo) < 65536) {
                    i32 den = ylo - yhi;
                    i32 nom = sx_shl(ylo, 8 - 3) + (den >> 1);
                    if (den != 0) ffrac += nom / den;
                } else {
                    ffrac += ylo / ((ylo - yhi) >> (8 - 3));
                }
                NLSF[root_ix] = sx_min(sx_shl(k, 8) + ffrac, 32767);
                root_ix++;
                if (root_ix >= d) break;
                if (root_ix & 1) { p = Q; yp = g->yQ; } else { p = P; yp = g->yP; }
                xlo = g->x[k - 1];
                ylo = sx_shl(1 - (root_ix & 2), 12);
            } else {
                k++;
                xlo = xhi;
                ylo = yhi;
                if (k > 128) {
                    i++;
                    if (i > 30) {
                        NLSF[0] = (1 << 15) / (d + 1);
                        for (k = 1; k < d; k++) NLSF[k] = sx_smulbb(k + 1, NLSF[0]);
                        wv_sync();
                        return;
                    }
                    wv_sync();
                    sx_bwexpander_32(a_Q16, d, 65536 - sx_smulbb(10 + i, i));
                    retry = true;
                    break;
                }
            }
        }
#endif
        wv_sync();
        if (!retry) return;
    }
}

// SKP_Silk_find_LPC_FIX, SKP_Silk_find_LPC_FIX.c:32.  x: nb_subfr blocks of subfr_length samples (incl. `order` pre-samples)
// LPC_res: scratch of 2*subfr_length samples (LDS)
SX_FN1 void sx_find_LPC(i32* NLSF_Q15, i32* interpIndex, const i32* prev_NLSFq_Q15, int useInterp, int order, const i16* x,
                       int subfr_length, i16* LPC_res, SxLpcWork* lw) {
    SX_IN_LDS(NLSF_Q15); SX_IN_LDS(interpIndex); SX_IN_LDS(prev_NLSFq_Q15); SX_IN_LDS(x); SX_IN_LDS(LPC_res); SX_IN_LDS(lw);
    i32* a_Q16 = lw->a_Q16;
    i32* a_tmp_Q16 = lw->a_tmp_Q16;
    i32 res_nrg, res_tmp_nrg, res_nrg_Q, res_tmp_nrg_Q;
    order = SX_LPC; subfr_length = SX_SUBFR + SX_LPC;       // (what the one caller passes: loop bounds and index arithmetic become constants)
    *interpIndex = 4;
    sx_burg_modified(&res_nrg, &res_nrg_Q, a_Q16, x, subfr_length, 4, K_FIND_LPC_COND_FAC_Q32, order, &lw->burg);
    sx_bwexpander_32(a_Q16, order, K_FIND_LPC_CHIRP_Q16);
    wv_sync();
    SX_T_BEGIN
    if (useInterp == 1) {
        sx_burg_modified<1>(&res_tmp_nrg, &res_tmp_nrg_Q, a_tmp_Q16, x + 2 * subfr_length, subfr_length, 2, K_FIND_LPC_COND_FAC_Q32, order, &lw->burg);
        sx_bwexpander_32(a_tmp_Q16, order, K_FIND_LPC_CHIRP_Q16);
        wv_sync();
        SX_T(22)
        int shift = res_tmp_nrg_Q - res_nrg_Q;
        if (shift >= 0) {
            if (shift < 32) res_nrg = res_nrg - (res_tmp_nrg >> shift);
        } else {
            res_nrg = (res_nrg >> (-shift)) - res_tmp_nrg;
            res_nrg_Q = res_tmp_nrg_Q;
        }
        sx_a2nlsf<1>(NLSF_Q15, a_tmp_Q16, order, lw->P, lw->Q, &lw->u.grid);
        wv_sync();
        SX_T(21)
        // the four interpolation candidates (SKP_Silk_find_LPC_FIX.c:81-140) are evaluated side by side: candidate k on lane k
        // for the NLSF -> LPC conversion, lane-strided for the whitening filter, lane (k, half) for the residual energies
#if defined(SX_LANE_STREAM) && defined(SX_HAVE_ROW_NLSF2A)
        bool all_ok;
        {   // candidate k on row k (sx_row_nlsf2a_stable); the serial form below only when a row needs the reference's corrections
            const int k = SX_LANE >> 4, i = SX_LANE & 15;
            if (i < SX_LPC) lw->u.it.NLSF0[k][i] = prev_NLSFq_Q15[i] + (sx_mul(NLSF_Q15[i] - prev_NLSFq_Q15[i], k) >> 2);
            wv_sync();
            all_ok = __builtin_amdgcn_ballot_w64(!sx_row_nlsf2a_stable(lw->u.it.a_Q12[k], lw->u.it.NLSF0[k])) == 0;
            wv_sync();
        }
        if (!all_ok)
#endif
        {
            SX_S(55)
            SX_PAR(k, 4) {
                for (int i = 0; i < order; i++) lw->u.it.NLSF0[k][i] = prev_NLSFq_Q15[i] + (sx_mul(NLSF_Q15[i] - prev_NLSFq_Q15[i], k) >> 2);
                sx_nlsf2a_stable_ws(lw->u.it.a_Q12[k], lw->u.it.NLSF0[k], order, lw->u.it.ws[k]);
            }
            wv_sync();
        }
        SX_S(53)
        {
            // (the one caller passes SX_SUBFR + SX_LPC: t / len by a constant instead of a run-time division per lane and round)
            const int len = 2 * (SX_SUBFR + SX_LPC);
            SX_PAR(t, 4 * len) {
                const int k = t / len, n = t - k * len;
                const i16* B = lw->u.it.a_Q12[k];
                i32 acc = 0;
                for (int j = 0; j < order; j++) {
                    const int u = n - 1 - j;
                    if (u >= 0) acc = sx_smlabb(acc, x[u], B[j]);
                }
                const i32 o = sx_sub_sat32(sx_shl((i32)x[n], 12), acc);
                lw->u.it.res[k][n] = (i16)sx_sat16(sx_rshift_round(o, 12));
            }
        }
        wv_sync();
        SX_S(54)
        SX_PAR(kh, 8) {
            const int k = kh >> 1, h = kh & 1;
            i32 e, sh;
            sx_sum_sqr_shift_n<SX_SUBFR>(&e, &sh, lw->u.it.res[k] + order + h * subfr_length, 0);      // (subfr_length - order = SX_SUBFR: the one caller)
            lw->u.it.nrg[k][h] = e;
            lw->u.it.rsh[k][h] = sh;
        }
        wv_sync();
        for (int k = 3; k >= 0; k--) {
            i32 res_nrg0 = lw->u.it.nrg[k][0], res_nrg1 = lw->u.it.nrg[k][1], res_nrg_interp, res_nrg_interp_Q;
            const i32 rshift0 = lw->u.it.rsh[k][0], rshift1 = lw->u.it.rsh[k][1];
            shift = rshift0 - rshift1;
            if (shift >= 0) {
                res_nrg1 = res_nrg1 >> shift;
                res_nrg_interp_Q = -rshift0;
            } else {
                res_nrg0 = res_nrg0 >> (-shift);
                res_nrg_interp_Q = -rshift1;
            }
            res_nrg_interp = sx_add(res_nrg0, res_nrg1);
            shift = res_nrg_interp_Q - res_nrg_Q;
            int isInterpLower;
            if (shift >= 0) {
                isInterpLower = (res_nrg_interp >> shift) < res_nrg;
            } else {
                isInterpLower = (-shift < 32) ? (res_nrg_interp < (res_nrg >> (-shift))) : 0;
            }
            if (isInterpLower) {
                res_nrg = res_nrg_interp;
                res_nrg_Q = res_nrg_interp_Q;
                *interpIndex = k;
            }
        }
        wv_sync();
    }
    SX_T(18)
    if (*interpIndex == 4) sx_a2nlsf(NLSF_Q15, a_Q16, order, lw->P, lw->Q, &lw->u.grid);
    wv_sync();
}

// ---------------------------------------------------------------------------------------------------
// NLSF MSVQ
// ---------------------------------------------------------------------------------------------------
// 16 kHz internal rate: the codebooks are 7.3 KB (216 vectors of 16) and would be most of the analysis kernel's LDS; they are read
// from the tables where they lie instead (the same few KB for every stream: cache hits), which takes the workgroup from 16.0 to
// 12 KB = two resident rounds of a compute unit's sixteen streams beside the quantiser instead of three
#ifndef SX_MSVQ_CB_IN_LDS
#define SX_MSVQ_CB_IN_LDS (SX_FS_KHZ == 8)
#endif
struct SxMsvqWork {                   // LDS scratch (16 survivors x up to 16 vectors per later stage; 64 in stage 0)
#if SX_NLANES == 1
    i32 RateDist_Q18[256];
    u8 taken[256];
#else
    i32 RateDist_Q18[16];
#endif
    i32 Sorted_Q18[16];
#if SX_MSVQ_CB_IN_LDS
    i16 cb[SX_NLSF_CB_MAXVEC * SX_LPC], rates[SX_NLSF_CB_MAXVEC];   // the signal type's codebook and rate table, staged from HBM once per frame
#endif
    i32 ndelta[SX_LPC + 2], nvec[SX_NLSF_STAGES];
    i32 W_Q6[SX_MAX_LPC];             // NLSF weights (read by every lane of the rate-distortion search)
    i32 NLSF0[SX_MAX_LPC], W0_Q6[SX_MAX_LPC];
    i32 ws[2][SX_NLSF2A_WS];
    i32 Res_Q15[16 * SX_LPC];
};
// ... and the part of it that lives OUTSIDE the prediction analysis' union: in the residual / window buffers of the front end
// (SxFrontWork::res_pitch, ::Wsig), which nothing reads between the long-term prediction analysis and the next frame.  (The
// analysis kernel's LDS decides how many of its workgroups fit on a compute unit beside the quantiser's: 16 x 8.6 KB do, 16 x
// 9.6 KB did not -- the last two of a CU's 16 streams then ran as a second round)
struct SxMsvqAux {
    i32 Res_new_Q15[16 * SX_LPC];
    i32 Rate_Q5[16], Rate_new_Q5[16];
    i32 TempIndices[16];
    u8 Path[16 * SX_NLSF_STAGES], Path_new[16 * SX_NLSF_STAGES];
};

#if SX_NLANES != 1
// compare-exchange of two (value, index) keys kept as signed 64-bit words (value in the high half): ascending
#define SX_KEY_CX(a, b) { const i64 lo_ = (a) < (b) ? (a) : (b), hi_ = (a) < (b) ? (b) : (a); (a) = lo_; (b) = hi_; }
// minimum of a 64-bit key over the wave (same DPP ladder as wv_min), result uniform
SX_HD i64 wv_min_key(i64 k) {
    i32 lo = (i32)(u32)k, hi = (i32)((u64)k >> 32);
#define SX_KEY_STEP(CTRL) { const i32 tl = SX_DPP_(lo, CTRL), th = SX_DPP_(hi, CTRL);                      \
        const bool take_ = (th < hi) | ((th == hi) & ((u32)tl < (u32)lo)); lo = take_ ? tl : lo; hi = take_ ? th : hi; }
    SX_KEY_STEP(0xB1) SX_KEY_STEP(0x4E) SX_KEY_STEP(0x141) SX_KEY_STEP(0x140)
#undef SX_KEY_STEP
    i32 rl = __builtin_amdgcn_readlane(lo, 0), rh = __builtin_amdgcn_readlane(hi, 0);
#pragma unroll
    for (int row = 1; row < 4; row++) {
        const i32 tl = __builtin_amdgcn_readlane(lo, row * 16), th = __builtin_amdgcn_readlane(hi, row * 16);
        const bool take_ = (th < rh) | ((th == rh) & ((u32)tl < (u32)rl));
        rl = take_ ? tl : rl; rh = take_ ? th : rh;
    }
    return (i64)(((u64)(u32)rh << 32) | (u32)rl);
}
#endif

// SKP_Silk_NLSF_MSVQ_encode_FIX, SKP_Silk_NLSF_MSVQ_encode_FIX.c:33 (16 survivors, 6 stages, order 10)
SX_FN1 void sx_nlsf_msvq_encode(i32* NLSFIndices, i32* pNLSF_Q15, int sigtype, const i32* prev_q_Q15, const i32* pW_Q6,
                               i32 mu_Q15, i32 mu_fluc_red_Q16, int deactivate_fluc_red, SxMsvqWork* w, SxMsvqAux* x) {
    SX_IN_LDS(w); SX_IN_LDS(x); SX_IN_LDS(NLSFIndices); SX_IN_LDS(pNLSF_Q15); SX_IN_LDS(prev_q_Q15); SX_IN_LDS(pW_Q6);   // pW_Q6 = w->W_Q6
    const i32 nvec0[SX_NLSF_STAGES] = T_NLSF_CB0_NVEC, nvec1[SX_NLSF_STAGES] = T_NLSF_CB1_NVEC;
    const i32* nvec = w->nvec;
    const int nStages = SX_NLSF_STAGES, S = SX_MSVQ_SURVIVORS;
    {   // stage the codebook of this signal type in LDS (32-bit words, coalesced)
        const i32* gnd = sigtype == 0 ? T_nlsf_cb0_ndelta_min_Q15 : T_nlsf_cb1_ndelta_min_Q15;
#if SX_MSVQ_CB_IN_LDS
        const u32* gcb = (const u32*)(sigtype == 0 ? T_nlsf_cb0_Q15 : T_nlsf_cb1_Q15);
        const u32* grt = (const u32*)(sigtype == 0 ? T_nlsf_cb0_rates_Q5 : T_nlsf_cb1_rates_Q5);
        const int nv = sigtype == 0 ? SX_NLSF_CB0_NVEC_TOTAL : SX_NLSF_CB1_NVEC_TOTAL;
        u32* lcb = (u32*)w->cb;
        u32* lrt = (u32*)w->rates;
        SX_PAR(i, nv * SX_LPC / 2) lcb[i] = gcb[i];
        SX_PAR(i, nv / 2) lrt[i] = grt[i];
#endif
        SX_PAR(i, SX_LPC + 1) w->ndelta[i] = gnd[i];
        SX_PAR(i, SX_NLSF_STAGES) w->nvec[i] = sigtype == 0 ? nvec0[i] : nvec1[i];
        wv_sync();
    }
#if SX_MSVQ_CB_IN_LDS
    const i16* cb = w->cb;
    const i16* rates = w->rates;
#else
    const i16* cb = (const i16*)(sigtype == 0 ? T_nlsf_cb0_Q15 : T_nlsf_cb1_Q15);
    const i16* rates = (const i16*)(sigtype == 0 ? T_nlsf_cb0_rates_Q5 : T_nlsf_cb1_rates_Q5);
#endif
    // The survivors' residuals, rates and paths of consecutive stages alternate between two sets of arrays (the reference copies the
    // new set over the old one after every stage: three passes and a round trip through LDS per stage)
    i32 *res_cur = w->Res_Q15, *res_new = x->Res_new_Q15, *rate_cur = x->Rate_Q5, *rate_new = x->Rate_new_Q5;
    u8 *path_cur = x->Path, *path_new = x->Path_new;
    SX_PAR(i, S) rate_cur[i] = 0;
    SX_PAR(i, SX_LPC) res_cur[i] = pNLSF_Q15[i];
    wv_sync();
    int prev_survivors = 1, cur_survivors = 0;
    const int min_survivors = S / 2;
    int cb_base = 0;
    for (int s = 0; s < nStages; s++) {
        const int K = nvec[s];
        const int lgK = 31 - sx_clz32(K);          // every stage size is a power of two (static_assert below the function): t / K is a shift
        const i16* cbs = cb + cb_base * SX_LPC;
        const i16* rts = rates + cb_base;
        cur_survivors = sx_min(S, sx_smulbb(prev_survivors, K));
        const int total = prev_survivors * K;
#if SX_NLANES == 1
        // rate-distortion of every (survivor, codebook vector) pair
        SX_PAR(t, total) {
            int n = t >> lgK, i = t - (n << lgK);
            const i32* in = &res_cur[n * SX_LPC];
            const i16* cv = &cbs[i * SX_LPC];
            i32 sum_error = 0;
            for (int m = 0; m < SX_LPC; m++) {
                i32 diff = in[m] - cv[m];
                sum_error = sx_smlawb(sum_error, sx_smulbb(diff, diff), pW_Q6[m]);
            }
            w->RateDist_Q18[t] = sx_smlabb(sum_error, rate_cur[n] + rts[i], mu_Q15);
            w->taken[t] = 0;
        }
        // SKP_Silk_insertion_sort_increasing (the cur_survivors best of `total`, value ascending, first index wins ties)
        for (int r = 0; r < cur_survivors; r++) {
            i32 bv = SX_I32_MAX, bi = SX_I32_MAX;
            for (int t = 0; t < total; t++) {
                const i32 v = w->RateDist_Q18[t];
                if (!w->taken[t] && (v < bv || (v == bv && t < bi))) { bv = v; bi = t; }
            }
            w->Sorted_Q18[r] = bv;
            x->TempIndices[r] = bi;
            w->taken[bi] = 1;
        }
        for (int r = 0; r < cur_survivors; r++) w->RateDist_Q18[r] = w->Sorted_Q18[r];
#else
        // rate-distortion of every (survivor, codebook vector) pair: a lane owns up to four pairs and keeps them in registers as
        // sorted (value, pair index) keys; the reference's insertion sort (cur_survivors best, value ascending, first index wins
        // ties) is then cur_survivors wave minima over the lanes' heads
        SX_S(47)
        const i64 KEY_MAX = 0x7FFFFFFFFFFFFFFFLL;
        i64 k0 = KEY_MAX, k1 = KEY_MAX, k2 = KEY_MAX, k3 = KEY_MAX;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int t = SX_LANE + 64 * j;
            if (j * 64 < total && t < total) {
                int n = t >> lgK, i = t - (n << lgK);
                const i32* in = &res_cur[n * SX_LPC];
                const i16* cv = &cbs[i * SX_LPC];
                i32 sum_error = 0;
#pragma unroll
                for (int m = 0; m < SX_LPC; m++) {
                    i32 diff = in[m] - cv[m];
                    sum_error = sx_smlawb(sum_error, sx_smulbb(diff, diff), pW_Q6[m]);
                }
                const i32 rd = sx_smlabb(sum_error, rate_cur[n] + rts[i], mu_Q15);
                const i64 key = (i64)(((u64)(u32)rd << 32) | (u32)t);
                if (j == 0) k0 = key; else if (j == 1) k1 = key; else if (j == 2) k2 = key; else k3 = key;
            }
        }
        SX_S(52)
#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64 && !defined(SX_MSVQ_SERIAL_SELECT)
        // The cur_survivors smallest keys in ascending order = the head of the SORTED key set (keys are distinct: the pair index is part of
        // them).  Instead of cur_survivors wave-minimum rounds -- each a chain of six dependent lane exchanges, a read-back, a vote and a
        // queue shift, 400 cycles of a lone wave -- every register of keys (one key per lane) is sorted along its 16-lane rows with a
        // bitonic network of ten compare-exchange stages (partner by DPP: i ^ 1, flip in 4, in 8, in 16, i ^ 2, i ^ 4), the sorted
        // lists are merged pairwise keeping the lower sixteen (min against the mirrored partner list + four half-cleaner stages):
        // first the registers of a lane, then the rows of a half (v_permlane16_swap), then the halves (v_permlane32_swap).  The
        // registers' networks are independent instruction streams: the scheduler interleaves them.  (Network checked lane by lane
        // against a sort in tools/debug/msvq_bitonic_model.py.)
        {
            // lanes that keep the SMALLER key of a pair, as wave masks: (lane & 1) == 0, & 2, & 4, & 8.  "take the partner's key" = (partner <
            // mine) in these lanes, the opposite in the others: one s_xnor of the compare's mask, handed back to the select as a lane condition
            const unsigned long long km1 = 0x5555555555555555ull, km2 = 0x3333333333333333ull, km4 = 0x0F0F0F0F0F0F0F0Full, km8 = 0x00FF00FF00FF00FFull;
#define SXK_D(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
#define SXK_X4(v) __builtin_amdgcn_update_dpp(__builtin_amdgcn_update_dpp((v), (v), 0x104, 0xF, 0x5, false), (v), 0x114, 0xF, 0xA, false)
#define SXK_CX_(K, PL, PH, KM) { const i32 pl_ = (PL), ph_ = (PH); const i64 pk_ = (i64)(((u64)(u32)ph_ << 32) | (u32)pl_);                     \
            const bool take_ = __builtin_amdgcn_inverse_ballot_w64(~(__builtin_amdgcn_ballot_w64(pk_ < (K)) ^ (KM))); (K) = take_ ? pk_ : (K); }
#define SXK_CXD(K, ctrl, KM) SXK_CX_(K, SXK_D((i32)(u32)(K), ctrl), SXK_D((i32)((u64)(K) >> 32), ctrl), KM)
#define SXK_CX4(K) SXK_CX_(K, SXK_X4((i32)(u32)(K)), SXK_X4((i32)((u64)(K) >> 32)), km4)
            // (one stage for several registers in a row: their chains are independent, the stages alternate in program order)
#define SXK_ST1(A, OP) OP(A)
#define SXK_ST2(A, B, OP) OP(A) OP(B)
#define SXK_ST4(A, B, C, D, OP) OP(A) OP(B) OP(C) OP(D)
#define SXK_S0(K) SXK_CXD(K, 0xB1, km1)
#define SXK_S1(K) SXK_CXD(K, 0x1B, km2)
#define SXK_S2(K) SXK_CXD(K, 0x141, km4)
#define SXK_S3(K) SXK_CXD(K, 0x4E, km2)
#define SXK_S4(K) SXK_CXD(K, 0x140, km8)
#define SXK_S5(K) SXK_CX4(K)
#define SXK_S6(K) SXK_CXD(K, 0x128, km8)
#define SXK_SORT16_(ST) ST(SXK_S0) ST(SXK_S1) ST(SXK_S0) ST(SXK_S2) ST(SXK_S3) ST(SXK_S0) ST(SXK_S4) ST(SXK_S5) ST(SXK_S3) ST(SXK_S0)
#define SXK_CLEAN16_(ST) ST(SXK_S6) ST(SXK_S5) ST(SXK_S3) ST(SXK_S0)
            // A <- the lower sixteen of two sorted lists (B mirrored), as a bitonic sequence (the half-cleaners follow)
#define SXK_MIN_MIRROR(A, BL, BH) { const i32 pl_ = SXK_D((BL), 0x140), ph_ = SXK_D((BH), 0x140); const i64 pk_ = (i64)(((u64)(u32)ph_ << 32) | (u32)pl_); \
            (A) = pk_ < (A) ? pk_ : (A); }
#define SXK_MERGE(A, BL, BH) { SXK_MIN_MIRROR(A, BL, BH) SXK_CLEAN16_(SXK_Z1) }
#define SXK_Z1(OP) SXK_ST1(z, OP)
#define SXK_Z2(OP) SXK_ST2(z, k1, OP)
#define SXK_Z4(OP) SXK_ST4(z, k1, k2, k3, OP)
#define SXK_ZK2(OP) SXK_ST2(z, k2, OP)
            i64 z = k0;
            if (total > 128) {
                SXK_SORT16_(SXK_Z4)
                SXK_MIN_MIRROR(z, (i32)(u32)k1, (i32)((u64)k1 >> 32)) SXK_MIN_MIRROR(k2, (i32)(u32)k3, (i32)((u64)k3 >> 32))
                SXK_CLEAN16_(SXK_ZK2)
                SXK_MERGE(z, (i32)(u32)k2, (i32)((u64)k2 >> 32))
            } else if (total > 64) {
                SXK_SORT16_(SXK_Z2)
                SXK_MERGE(z, (i32)(u32)k1, (i32)((u64)k1 >> 32))
            } else {
                SXK_SORT16_(SXK_Z1)
            }
            {   // the rows of a half: element [0] = the even row's list in both rows, [1] = the odd row's
                auto sl_ = __builtin_amdgcn_permlane16_swap((u32)z, (u32)z, false, false);
                auto sh_ = __builtin_amdgcn_permlane16_swap((u32)((u64)z >> 32), (u32)((u64)z >> 32), false, false);
                z = (i64)(((u64)(u32)sh_[0] << 32) | (u32)sl_[0]);
                SXK_MERGE(z, (i32)sl_[1], (i32)sh_[1])
            }
            if (total > 32) {   // the halves ([0] = the lower half's list, [1] = the upper's); 32 pairs sit in the lower half alone
                auto sl_ = __builtin_amdgcn_permlane32_swap((u32)z, (u32)z, false, false);
                auto sh_ = __builtin_amdgcn_permlane32_swap((u32)((u64)z >> 32), (u32)((u64)z >> 32), false, false);
                z = (i64)(((u64)(u32)sh_[0] << 32) | (u32)sl_[0]);
                SXK_MERGE(z, (i32)sl_[1], (i32)sh_[1])
            }
            if (SX_LANE < cur_survivors) {
                w->RateDist_Q18[SX_LANE] = (i32)((u64)z >> 32);
                x->TempIndices[SX_LANE] = (i32)(u32)z;
            }
#undef SXK_D
#undef SXK_X4
#undef SXK_CX_
#undef SXK_CXD
#undef SXK_CX4
#undef SXK_MERGE
#undef SXK_MIN_MIRROR
#undef SXK_ST1
#undef SXK_ST2
#undef SXK_ST4
#undef SXK_S0
#undef SXK_S1
#undef SXK_S2
#undef SXK_S3
#undef SXK_S4
#undef SXK_S5
#undef SXK_S6
#undef SXK_SORT16_
#undef SXK_CLEAN16_
#undef SXK_Z1
#undef SXK_Z2
#undef SXK_Z4
#undef SXK_ZK2
        }
#else
        if (total > 64) { SX_KEY_CX(k0, k1) SX_KEY_CX(k2, k3) SX_KEY_CX(k0, k2) SX_KEY_CX(k1, k3) SX_KEY_CX(k1, k2) }
        // heads as (value, index) register pairs; one selection round = wave minimum of the values, then wave minimum of the
        // indices among the lanes that hold that value (two 32-bit DPP ladders instead of one 64-bit compare ladder)
        i32 v0 = (i32)((u64)k0 >> 32), v1 = (i32)((u64)k1 >> 32), v2 = (i32)((u64)k2 >> 32), v3 = (i32)((u64)k3 >> 32);
        i32 i0 = (i32)(u32)k0 & 0x7FFFFFFF, i1 = (i32)(u32)k1 & 0x7FFFFFFF, i2 = (i32)(u32)k2 & 0x7FFFFFFF, i3 = (i32)(u32)k3 & 0x7FFFFFFF;
        i32 mine_v = SX_I32_MAX, mine_i = 0;
        for (int r = 0; r < cur_survivors; r++) {
            const i32 vmin = wv_min(v0);
#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64
            // (equal rate-distortion values in two lanes are rare: one lane holds the minimum -> its index is read straight out of it)
            const unsigned long long tie_ = __builtin_amdgcn_ballot_w64(v0 == vmin);
            i32 imin;
            if (__builtin_popcountll(tie_) == 1) imin = __builtin_amdgcn_readlane(i0, __builtin_ctzll(tie_));
            else imin = wv_min(v0 == vmin ? i0 : SX_I32_MAX);
#else
            const i32 imin = wv_min(v0 == vmin ? i0 : SX_I32_MAX);
#endif
            if (v0 == vmin && i0 == imin) { v0 = v1; i0 = i1; v1 = v2; i1 = i2; v2 = v3; i2 = i3; v3 = SX_I32_MAX; i3 = SX_I32_MAX; }
            if (SX_LANE == r) { mine_v = vmin; mine_i = imin; }
        }
        if (SX_LANE < cur_survivors) {
            w->RateDist_Q18[SX_LANE] = mine_v;
            x->TempIndices[SX_LANE] = mine_i;
        }
#endif
#endif
        SX_S(48)
        wv_sync();
        if (w->RateDist_Q18[0] < SX_I32_MAX / 16) {
            i32 thr = sx_smlawb(w->RateDist_Q18[0], sx_mul(S, w->RateDist_Q18[0]), K_NLSF_MSVQ_SURV_MAX_REL_RD_Q16);
            while (w->RateDist_Q18[cur_survivors - 1] > thr && cur_survivors > min_survivors) cur_survivors--;
        }
        SX_S(49)
        // new residuals, rates and paths of the survivors: lane (k, i)
        // (a row of SX_MSVQ_ROW lanes per survivor: SX_LPC residual entries, the rate, up to nStages - 1 inherited path entries, the new one)
        SX_PAR(ki, cur_survivors * SX_MSVQ_ROW) {
            const int k = ki / SX_MSVQ_ROW, i = ki % SX_MSVQ_ROW;
            int input_index = 0, cb_index = x->TempIndices[k];
            if (s > 0) {
                input_index = cb_index >> lgK;
                cb_index = cb_index - (input_index << lgK);
            }
            if (i < SX_LPC) res_new[k * SX_LPC + i] = res_cur[input_index * SX_LPC + i] - (i32)cbs[cb_index * SX_LPC + i];
            if (i == SX_LPC) rate_new[k] = rate_cur[input_index] + rts[cb_index];
            if (i > SX_LPC && i - SX_LPC - 1 < s) path_new[k * nStages + (i - SX_LPC - 1)] = path_cur[input_index * nStages + (i - SX_LPC - 1)];
            if (i == SX_MSVQ_ROW - 1) path_new[k * nStages + s] = (u8)cb_index;
        }
        wv_sync();
        if (s < nStages - 1) {       // (after the last stage path_new stays the survivors' paths)
            { i32* t_ = res_cur; res_cur = res_new; res_new = t_; }
            { i32* t_ = rate_cur; rate_cur = rate_new; rate_new = t_; }
            { u8* t_ = path_cur; path_cur = path_new; path_new = t_; }
        }
        prev_survivors = cur_survivors;
        cb_base += K;
    }
    SX_S(51)
    i32 bestRateDist_Q20 = SX_I32_MAX, bestIndex = 0;
    if (deactivate_fluc_red != 1) {
        // every survivor is decoded (and stabilised) by its own lane into its own row of Res_Q15
        i32 bv = SX_I32_MAX, bi = SX_I32_MAX;
        SX_PAR(sv, cur_survivors) {
            i32* out = &w->Res_Q15[sv * SX_LPC];
            sx_nlsf_msvq_decode_cb(out, &path_new[sv * nStages], cb, nvec, w->ndelta);
            i32 wsse_Q20 = 0;
            for (int i = 0; i < SX_LPC; i++) {
                i32 se = out[i] - prev_q_Q15[i];
                wsse_Q20 = sx_smlawb(wsse_Q20, sx_smulbb(se, se), pW_Q6[i]);
            }
            wsse_Q20 = sx_add_pos_sat32(w->RateDist_Q18[sv], sx_smulwb(wsse_Q20, mu_fluc_red_Q16));
            if (wsse_Q20 < bv) { bv = wsse_Q20; bi = sv; }
        }
        wv_argmin(&bv, &bi);
        if (bv < bestRateDist_Q20) { bestRateDist_Q20 = bv; bestIndex = bi; }
        wv_sync();
    }
    SX_PAR(i, nStages) NLSFIndices[i] = path_new[bestIndex * nStages + i];
    wv_sync();
    sx_nlsf_msvq_decode_cb(pNLSF_Q15, NLSFIndices, cb, nvec, w->ndelta);
    wv_sync();
}

constexpr bool sx_all_pow2(const int* v, int n) { for (int i = 0; i < n; i++) if (v[i] <= 0 || (v[i] & (v[i] - 1))) return false; return true; }
constexpr int sx_nvec0_[SX_NLSF_STAGES] = T_NLSF_CB0_NVEC, sx_nvec1_[SX_NLSF_STAGES] = T_NLSF_CB1_NVEC;
static_assert(sx_all_pow2(sx_nvec0_, SX_NLSF_STAGES) && sx_all_pow2(sx_nvec1_, SX_NLSF_STAGES), "the survivor search divides by the stage sizes with shifts");

// SKP_Silk_process_NLSFs_FIX, SKP_Silk_process_NLSFs_FIX.c:31
SX_FN1 void sx_process_NLSFs(SxEncState* st, SxEncCtrl* c, i32* pNLSF_Q15, SxMsvqWork* w, SxMsvqAux* aux) {
    SX_IN_LDS(st); SX_IN_LDS(c); SX_IN_LDS(w); SX_IN_LDS(pNLSF_Q15); SX_IN_LDS(aux);
    i32 NLSF_mu_Q15, NLSF_mu_fluc_red_Q16;
    i32* pNLSFW_Q6 = w->W_Q6;
    if (c->sigtype == 0) {
        NLSF_mu_Q15 = sx_smlawb(66, -8388, st->speech_activity_Q8);
        NLSF_mu_fluc_red_Q16 = sx_smlawb(6554, -838848, st->speech_activity_Q8);
    } else {
        NLSF_mu_Q15 = sx_smlawb(164, -33554, st->speech_activity_Q8);
        NLSF_mu_fluc_red_Q16 = sx_smlawb(13107, -1677696, st->speech_activity_Q8 + c->sparseness_Q8);
    }
    NLSF_mu_Q15 = sx_max(NLSF_mu_Q15, 1);
    const int doInterpolate = c->NLSFInterpCoef_Q2 < 4;     // useInterpolatedNLSFs == 1
    const i32 interp_Q2 = c->NLSFInterpCoef_Q2;
    // Laroia weights of the target and (if interpolating) of the interpolated vector: one vector per lane, both lanes in step
    // (the same instructions on per-lane pointers; written as two branches the two would run one after the other, as scalar code)
    if (doInterpolate) {
        SX_PAR(i, SX_LPC) w->NLSF0[i] = st->prev_NLSFq_Q15[i] + (sx_mul(pNLSF_Q15[i] - st->prev_NLSFq_Q15[i], interp_Q2) >> 2);
        wv_sync();
    }
#if defined(SX_LANE_STREAM) && SX_LPC <= 15
    {   // (vector v on row v, one division per lane)
        const int v = SX_LANE >> 4;
        sx_row_nlsf_weights_laroia(v ? w->W0_Q6 : pNLSFW_Q6, v ? w->NLSF0 : pNLSF_Q15, SX_LPC, v < (doInterpolate ? 2 : 1));
    }
#else
    SX_PAR(v, doInterpolate ? 2 : 1) sx_nlsf_weights_laroia(v ? w->W0_Q6 : pNLSFW_Q6, v ? w->NLSF0 : pNLSF_Q15, SX_LPC);
#endif
    wv_sync();
    if (doInterpolate) {
        const i32 i_sqr_Q15 = sx_shl(sx_smulbb(interp_Q2, interp_Q2), 11);
        SX_PAR(i, SX_LPC) pNLSFW_Q6[i] = sx_smlawb(pNLSFW_Q6[i] >> 1, w->W0_Q6[i], i_sqr_Q15);
    }
    wv_sync();
    SX_S(46)
    sx_nlsf_msvq_encode(c->NLSFIndices, pNLSF_Q15, c->sigtype, st->prev_NLSFq_Q15, pNLSFW_Q6, NLSF_mu_Q15, NLSF_mu_fluc_red_Q16,
                        st->first_frame_after_reset, w, aux);
    SX_S(50)
    // quantised NLSFs -> LPC for the two frame halves: half v on lane v, both lanes in step
    if (doInterpolate) {
        SX_PAR(i, SX_LPC) w->NLSF0[i] = st->prev_NLSFq_Q15[i] + (sx_mul(pNLSF_Q15[i] - st->prev_NLSFq_Q15[i], interp_Q2) >> 2);
        wv_sync();
    }
#if defined(SX_LANE_STREAM) && defined(SX_HAVE_ROW_NLSF2A)
    bool both_ok;
    {   // half v on row v (rows 2, 3 repeat them and store the same values)
        const int v = (SX_LANE >> 4) & 1;
        bool ok = true;
        if (doInterpolate) ok = sx_row_nlsf2a_stable(c->PredCoef_Q12[v], v ? pNLSF_Q15 : w->NLSF0);
        else ok = sx_row_nlsf2a_stable(c->PredCoef_Q12[1], pNLSF_Q15);
        both_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
        wv_sync();
    }
    if (!both_ok)
#endif
    {
        SX_S(56)
        SX_PAR(v, 2) {
            if (v == 1 || doInterpolate) sx_nlsf2a_stable_ws(c->PredCoef_Q12[v], v ? pNLSF_Q15 : w->NLSF0, SX_LPC, w->ws[v]);
        }
        wv_sync();
    }
    if (!doInterpolate) {
        SX_PAR(i, SX_LPC) c->PredCoef_Q12[0][i] = c->PredCoef_Q12[1][i];
        wv_sync();
    }
}

// SKP_Silk_residual_energy_FIX, SKP_Silk_residual_energy_FIX.c:32.  LPC_res: 100-sample scratch
SX_FN1 void sx_residual_energy(i32* nrgs, i32* nrgsQ, const i16* x, i16 a_Q12[2][SX_MAX_LPC], const i32* gains, i16* LPC_res) {
    // (no SX_IN_LDS marks here: with them clang 22 (ROCm 7.2) crashes in correlated-propagation on this function, depending on what else the
    // translation unit holds)
    const int offset = SX_LPC + SX_SUBFR;
    const i16* x_ptr = x;
    for (int i = 0; i < 2; i++) {
        sx_lpc_analysis_filter_zero_state(x_ptr, a_Q12[i], LPC_res, 2 * offset, SX_LPC);
        wv_sync();
        for (int j = 0; j < 2; j++) {
            i32 rshift;
            static_assert(SX_SUBFR <= 128, "one pair of samples per lane");
            sx_sum_sqr_shift_wv_inl<1>(&nrgs[i * 2 + j], &rshift, LPC_res + SX_LPC + j * offset, SX_SUBFR, 0);
            nrgsQ[i * 2 + j] = -rshift;
        }
        x_ptr += 2 * offset;
        wv_sync();
    }
    for (int i = 0; i < 4; i++) {
        int lz1 = sx_clz32(nrgs[i]) - 1, lz2 = sx_clz32(gains[i]) - 1;
        i32 tmp32 = sx_shl(gains[i], lz2);
        tmp32 = sx_smmul(tmp32, tmp32);
        nrgs[i] = sx_smmul(tmp32, sx_shl(nrgs[i], lz1));
        nrgsQ[i] += lz1 + 2 * lz2 - 32 - 32;
    }
}

struct SxPredWork {                   // LDS scratch of find_pred_coefs
    i32 WLTP[4 * 25];
    i16 LPC_in_pre[4 * (SX_SUBFR + SX_LPC)];
    i16 LPC_res[2 * (SX_SUBFR + SX_LPC)];
    i32 NLSF_Q15[SX_MAX_LPC];
    i32 invGains_Q16[SX_NB_SUBFR], local_gains[SX_NB_SUBFR], Wght_Q15[SX_NB_SUBFR];
    union {
        SxLtpWork ltp;
        struct { i32 rd[3 * 4 * 40]; i32 best[3 * 4 * 2]; } vq;
        SxLpcWork lpc;
        SxMsvqWork msvq;
    } u;
};

// SKP_Silk_find_pred_coefs_FIX, SKP_Silk_find_pred_coefs_FIX.c:31
// res_pitch: the LPC residual of the pitch analysis (read by the long-term prediction analysis), followed by SX_PITCH_LPC_WIN more
// samples of dead scratch (SxFrontWork::Wsig): the NLSF quantiser's survivor tables are kept there afterwards (SxMsvqAux)
SX_FN1 void sx_find_pred_coefs(SxEncState* st, SxEncCtrl* c, const i16* x_buf, i16* res_pitch, SxPredWork* w) {
    SX_IN_LDS(st); SX_IN_LDS(c); SX_IN_LDS(x_buf); SX_IN_LDS(res_pitch); SX_IN_LDS(w);
    i32 *invGains_Q16 = w->invGains_Q16, *local_gains = w->local_gains, *Wght_Q15 = w->Wght_Q15;
    i32* NLSF_Q15 = w->NLSF_Q15;
    SX_T_BEGIN
    SX_PAR(i, 4) {                               // (the four subframes side by side: two divisions each)
        i32 min_gain_Q16 = SX_I32_MAX >> 6;
        for (int k = 0; k < 4; k++) min_gain_Q16 = sx_min(min_gain_Q16, c->Gains_Q16[k]);
        const i32 inv = sx_max(sx_div32_varQ(min_gain_Q16, c->Gains_Q16[i], 16 - 2), 363);
        invGains_Q16[i] = inv;
        Wght_Q15[i] = sx_smulwb(inv, inv) >> 1;
        local_gains[i] = (1 << 16) / inv;
    }
    wv_sync();
    SX_STRETCH_DENSE();
    if (c->sigtype == 0) {
        sx_find_LTP(c->LTPCoef_Q14, w->WLTP, &c->LTPredCodGain_Q7, res_pitch, c->pitchL, Wght_Q15, &w->u.ltp);
        SX_S(42)
        sx_quant_LTP_gains(c->LTPCoef_Q14, c->LTPIndex, &c->PERIndex, w->WLTP, K_MU_LTP_QUANT_Q8, w->u.vq.rd, w->u.vq.best);
        SX_S(43)
        sx_LTP_scale_ctrl(st, c);
        sx_LTP_analysis_filter(w->LPC_in_pre, x_buf + SX_FRAME - SX_LPC, c->LTPCoef_Q14, c->pitchL, invGains_Q16);
        wv_sync();
    } else {
        const int n = SX_SUBFR + SX_LPC;
        const i16* x_ptr = x_buf + SX_FRAME - SX_LPC;
        SX_PAR(t, 4 * n) {
            int k = t / n, i = t - k * n;
            w->LPC_in_pre[t] = (i16)sx_smulwb(invGains_Q16[k], x_ptr[k * SX_SUBFR + i]);
        }
        wv_sync();
        for (int i = 0; i < 20; i++) c->LTPCoef_Q14[i] = 0;
        c->LTPredCodGain_Q7 = 0;
    }
    SX_T(16)
    sx_find_LPC(NLSF_Q15, &c->NLSFInterpCoef_Q2, st->prev_NLSFq_Q15, 1 - st->first_frame_after_reset, SX_LPC, w->LPC_in_pre,
                SX_SUBFR + SX_LPC, w->LPC_res, &w->u.lpc);
    SX_T(17)
    static_assert(sizeof(SxMsvqAux) <= sizeof(i16) * (2 * SX_FRAME + SX_LA_PITCH + SX_PITCH_LPC_WIN), "survivor tables exceed res_pitch + Wsig");
    wv_sync();                                   // (the last readers of res_pitch are done)
    SX_STRETCH_LATENCY();
    sx_process_NLSFs(st, c, NLSF_Q15, &w->u.msvq, (SxMsvqAux*)(void*)res_pitch);
    SX_T(19)
    SX_STRETCH_DENSE();
    sx_residual_energy(c->ResNrg, c->ResNrgQ, w->LPC_in_pre, c->PredCoef_Q12, local_gains, w->LPC_res);
    SX_STRETCH_LATENCY();
    SX_T(20)
    for (int i = 0; i < SX_LPC; i++) st->prev_NLSFq_Q15[i] = NLSF_Q15[i];
}

// ---------------------------------------------------------------------------------------------------
// gains
// ---------------------------------------------------------------------------------------------------
// SKP_Silk_gains_quant, SKP_Silk_gain_quant.c:42 (md_enable = 1)
SX_HD void sx_gains_quant(i32* ind, i32* gain_Q16, i32* prev_ind, int conditional, i32* ind2, i32* DeltaGains_Q16) {
    const i32 OFFSET = (6 * 128) / 6 + 16 * 128;
    const i32 SCALE_Q16 = (65536 * (64 - 1)) / (((86 - 6) * 128) / 6);
    const i32 INV_SCALE_Q16 = (65536 * (((86 - 6) * 128) / 6)) / (64 - 1);
    const i32 AlphaDis_Q16 = 32768 / 8;
    i32 inv_gain_Q16 = sx_inverse32_varQ(sx_max(*DeltaGains_Q16, 1), 32);
    inv_gain_Q16 -= 32767;
    *ind2 = 0;
    for (int k = 0; k < 8; k++) {
        if (inv_gain_Q16 > k * AlphaDis_Q16 && inv_gain_Q16 <= (k + 1) * AlphaDis_Q16) {
            *ind2 = k;
            inv_gain_Q16 = (k + 1) * AlphaDis_Q16;
        }
    }
    inv_gain_Q16 += 32767;
    *DeltaGains_Q16 = sx_inverse32_varQ(sx_max(inv_gain_Q16, 1), 32);
    // the logarithms of the four gains and the four quantised gains side by side; the index chain between them is the serial part
    SX_PAR(k, SX_NB_SUBFR) ind[k] = sx_smulwb(SCALE_Q16, sx_lin2log(gain_Q16[k]) - OFFSET);
    wv_sync();
    for (int k = 0; k < SX_NB_SUBFR; k++) {
        if (ind[k] < *prev_ind) ind[k]++;
        if (k == 0 && conditional == 0) {
            ind[k] = sx_limit(ind[k], 0, 63);
            ind[k] = sx_max(ind[k], *prev_ind + (-4));
            *prev_ind = ind[k];
        } else {
            ind[k] = sx_limit(ind[k] - *prev_ind, -4, 40);
            *prev_ind += ind[k];
            ind[k] -= -4;
        }
        gain_Q16[k] = *prev_ind;
    }
    wv_sync();
    SX_PAR(k, SX_NB_SUBFR) gain_Q16[k] = sx_log2lin(sx_min(sx_smulwb(INV_SCALE_Q16, gain_Q16[k]) + OFFSET, 3967));
    wv_sync();
}

// SKP_Silk_process_gains_FIX, SKP_Silk_process_gains_FIX.c:32
SX_FN1 void sx_process_gains(SxEncState* st, SxEncCtrl* c) {
    SX_IN_LDS(st); SX_IN_LDS(c);
    st = SX_VPTR(st); c = SX_VPTR(c);            // wave-uniform scalar stage: on the vector unit (SX_VPTR)
    const bool voiced = c->sigtype == 0;
    const i32 s_Q16 = voiced ? -sx_sigm_Q15(sx_rshift_round(c->LTPredCodGain_Q7 - K_12p0_Q7, 4)) : 0;
    i32 InvMaxSqrVal_Q16 = sx_log2lin(sx_smulwb(K_70p0_Q7 - c->current_SNR_dB_Q7, K_0p33_Q16)) / SX_SUBFR;
    SX_PAR(k, 4) {                               // (the four subframes side by side)
        if (voiced) c->Gains_Q16[k] = sx_smlawb(c->Gains_Q16[k], c->Gains_Q16[k], s_Q16);
        i32 ResNrg = c->ResNrg[k];
        i32 ResNrgPart = sx_smulww(ResNrg, InvMaxSqrVal_Q16);
        if (c->ResNrgQ[k] > 0) {
            if (c->ResNrgQ[k] < 32) ResNrgPart = sx_rshift_round(ResNrgPart, c->ResNrgQ[k]);
            else ResNrgPart = 0;
        } else if (c->ResNrgQ[k] != 0) {
            if (ResNrgPart > (SX_I32_MAX >> (-c->ResNrgQ[k]))) ResNrgPart = SX_I32_MAX;
            else ResNrgPart = sx_shl(ResNrgPart, -c->ResNrgQ[k]);
        }
        i32 gain = c->Gains_Q16[k];
        i32 gain_squared = sx_add_sat32(ResNrgPart, sx_smmul(gain, gain));
        if (gain_squared < 32767) {
            gain_squared = sx_smlaww(sx_shl(ResNrgPart, 16), gain, gain);
            gain = sx_sqrt_approx(gain_squared);
            c->Gains_Q16[k] = sx_lshift_sat32(gain, 8);
        } else {
            gain = sx_sqrt_approx(gain_squared);
            c->Gains_Q16[k] = sx_lshift_sat32(gain, 16);
        }
    }
    wv_sync();
    // MD delta gain: float / double island (process_gains_FIX.c:90-92)
    float tmp_float = sx_fdiv(1.0f, c->md_delta_gain_par);
    tmp_float = tmp_float * 65536.0f;
    tmp_float = tmp_float > 131072.0f ? 131072.0f : (tmp_float < -131072.0f ? -131072.0f : tmp_float);
    double dd = (double)tmp_float - (0.05 * 65536.0f);
    i32 Delta_Gains_Q16 = (i32)(dd > 0 ? dd + 0.5 : dd - 0.5);
    sx_gains_quant(c->GainsIndices, c->Gains_Q16, &st->LastGainIndex, st->nFramesInPayloadBuf, &c->DeltaGainsIndices, &Delta_Gains_Q16);
    c->DeltaGains_Q16 = Delta_Gains_Q16;
    if (c->sigtype == 0) {
        if (c->LTPredCodGain_Q7 + (c->input_tilt_Q15 >> 8) > K_1p0_Q7) c->QuantOffsetType = 0;
        else c->QuantOffsetType = 1;
    }
    i32 quant_offset_Q10 = T_quant_offsets_Q10[c->sigtype * 2 + c->QuantOffsetType];
    c->Lambda_Q10 = K_LAMBDA_OFFSET_Q10 + sx_smulbb(K_LAMBDA_DELAYED_DECISIONS_Q10, SX_DD_STATES) +
                    sx_smulwb(K_LAMBDA_SPEECH_ACT_Q18, st->speech_activity_Q8) + sx_smulwb(K_LAMBDA_INPUT_QUALITY_Q12, c->input_quality_Q14) +
                    sx_smulwb(K_LAMBDA_CODING_QUALITY_Q12, c->coding_quality_Q14) + sx_smulwb(K_LAMBDA_QUANT_OFFSET_Q16, quant_offset_Q10);
}
