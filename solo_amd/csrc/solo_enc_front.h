// solo_enc_front.h -- encoder front end: QMF band split, VAD, adaptive high-pass, pitch analysis.
// Rows E2, E5a, E5b, E5c of SURVEY.md section 8(a).  Reference (JC1_SDK_SRC_ARM/src/...):
//   libBWE/AGR_BWE_qmf.c:38-80, libSATECodec/SKP_Silk_VAD.c:75-318, SKP_Silk_ana_filt_bank_1.c:45,
//   SKP_Silk_HP_variable_cutoff_FIX.c:37, SKP_Silk_biquad_alt.c:38, SKP_Silk_find_pitch_lags_FIX.c:32,
//   SKP_Silk_pitch_analysis_core.c:65-706, SKP_Silk_apply_sine_window.c:47, SKP_Silk_autocorr.c:40,
//   SKP_Silk_schur.c:40, SKP_Silk_k2a.c:40, SKP_Silk_resampler_down2.c:41, SKP_Silk_sort.c:80
#pragma once
#include "solo_enc_state.h"

// AGR_Sate_qmf_decomp (libBWE/AGR_BWE_qmf.c:38) in direct form, wave-parallel over the 320 output pairs.
//   x[t] = in[t] >> 1 on a continuous timeline (63 samples of history), taps reversed (a[j] = aa[63-j]):
//   y1[k] = sat( pshr15( sum_{j<32} a[j] * (int16)( x[2k+j-63] + x[2k-j] ) ) )
//   y2[k] = sat( pshr15( sum_{j<32} (j even ? -1 : +1) * a[j] * ( x[2k+j-63] - x[2k-j] ) ) )
// `tl` is a 63+640 sample scratch timeline (LDS).
// n: samples of the packet (SX_PACKET, or half of it with framesize_ms = 20)
SX_FN1 void sx_qmf_decomp(SxEncHist* hist, const i16* pcm, i16* tl, i16* lo, i16* hi, int n = SX_PACKET) {
    SX_IN_LDS(tl);
    n = SX_UNI(n);
    SX_PAR(i, 63) tl[i] = hist->qmf_hist[i];
    SX_PAR(i, n) tl[63 + i] = (i16)(pcm[i] >> 1);
    wv_sync();
    SX_PAR(k, n >> 1) {
        i32 y1 = 0, y2 = 0;
        const i16* xa = tl + 2 * k;          // x[2k + j - 63]  -> tl[63 + 2k + j - 63]
        const i16* xb = tl + 63 + 2 * k;     // x[2k - j]
        for (int j = 0; j < 32; j++) {
            i32 a = T_qmf_taps[63 - j];
            i32 p = xa[j], q = xb[-j];
            y1 = sx_add(y1, sx_mul(a, (i32)(i16)(p + q)));
            i32 d = sx_mul(a, (i32)(i16)(p - q));
            y2 = (j & 1) ? sx_add(y2, d) : sx_sub(y2, d);
        }
        lo[k] = (i16)sx_saturate(sx_pshr32(y1, 15), 32767);
        hi[k] = (i16)sx_saturate(sx_pshr32(y2, 15), 32767);
    }
    wv_sync();
    SX_PAR(i, 63) hist->qmf_hist[i] = tl[n + i];
    wv_sync();
}

struct alignas(16) SxV4i { i32 v[4]; };        // 16-byte LDS moves
#ifdef SX_LANE_STREAM
// Serial recursions cost the wave one instruction slot per operation whichever unit executes them, but the scalar unit issues
// only one instruction per four cycles PER SIMD for all of its waves together (the vector unit: two): wave-uniform sample
// recursions left to the scalar unit are the dearest instructions of the analysis kernel (tools/debug/mb_issue.hip,
// DESIGN.md section 4).  SX_VEC(x) pins a value to a vector register and hides its uniformity from the compiler, so the
// arithmetic that depends on it is emitted for the vector unit.
// Two first-order all-pass chains side by side -- the structure of SKP_Silk_ana_filt_bank_1 (ana_filt_bank_1.c:45) and of
// SKP_Silk_resampler_down2 (resampler_down2.c:41): lane 0 runs the chain of the EVEN input samples, X = Y + (Y * cA >> 16),
// lane 1 that of the ODD ones, X = Y * cB >> 16; per pair of samples the two chain outputs are exchanged inside the lane pair
// (DPP) and lane 0 stores the low band sat16(rshift_round(o1 + o0, 11)), lane 1 the high band sat16(rshift_round(o1 - o0, 11))
// (a caller that has no use for it passes a dump area).  S: the two chain states (null = zero state, not written back).  `in` may alias outL (in-place decimation):
// every block of four pairs is read before any of its outputs is stored, and outputs trail the inputs.
// npairs must be a multiple of 4 and >= 4 (blocks of four pairs with the next block's inputs prefetched): the callers pass
// SX_FRAME / 2, / 4, / 8 pairs (VAD filter banks) and 320 / 160 pairs (pitch analysis decimators at the 16 kHz internal rate)
static_assert((SX_FRAME / 8) % 4 == 0 && SX_FRAME / 8 >= 4, "all-pass pairs are processed in blocks of four");
SX_HD void sx_allpass2_lanes(const i16* in, int npairs, i32* S, i32 cA, i32 cB, i16* outL, i16* outH) {
    if (SX_LANE < 2) {
        const int l = SX_LANE;
        i32 s = S ? S[l] : 0;
        i32 cpre = sx_pre16(l ? cB : cA);
        SX_VEC(cpre);
        const i32 mask = l ? 0 : -1;
        const i16* ip = in + l;                       // the lane's samples: in[2k + l]
        i16* op = l ? outH : outL;
        i32 x[4], xn[4];
#pragma unroll
        for (int u = 0; u < 4; u++) xn[u] = ip[2 * u];
        for (int k = 0; k < npairs; k += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) x[u] = xn[u];
            if (k + 4 < npairs) {
#pragma unroll
                for (int u = 0; u < 4; u++) xn[u] = ip[2 * (k + 4 + u)];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const i32 in32 = sx_shl(x[u], 10);
                const i32 Y = sx_sub(in32, s);
                const i32 X = sx_add(sx_smulw_pre(Y, cpre), Y & mask);
                const i32 o = sx_add(s, X);
                s = sx_add(in32, X);
                // lane 0 stores o1 + o0, lane 1 o1 - o0: each lane hands the other what that one has to ADD (quad_perm [1,0,3,2]).
                // |in32| <= 2^25 and the sections are all-pass: |o| < 2^28, so the rounding cannot overflow (sx_rshift_round_small)
                const i32 p = SX_DPP_(l ? o : sx_neg(o), 0xB1);
                const i32 v = sx_add(o, p);
                op[k + u] = (i16)sx_sat16(sx_rshift_round_small(v, 11));
            }
        }
        if (S) S[l] = s;
    }
    wv_sync();
}
#endif

#ifdef SX_LANE_STREAM
// The same pair of chains with the band sums left as they are: lane 0 writes o1 + o0, lane 1 o1 - o0, 32 bits each, to rawL / rawH
// (rawH null: nowhere); the caller rounds and saturates them afterwards, all samples side by side (sx_allpass2_finish) -- three
// instructions less in every step of the recursion.  `in` must not overlap the raw buffers.
SX_HD void sx_allpass2_lanes_raw(const i16* in, int npairs, i32* S, i32 cA, i32 cB, i32* rawL, i32* rawH, i32* dump4) {
    if (SX_LANE < 2) {
        const int l = SX_LANE;
        i32 s = S ? S[l] : 0;
        i32 cpre = sx_pre16(l ? cB : cA);
        SX_VEC(cpre);
        const i32 mask = l ? 0 : -1;
        const i16* ip = in + l;                       // the lane's samples: in[2k + l]
        SxV4i* op = (SxV4i*)(l ? (rawH ? rawH : dump4) : rawL);
        const int ostep = (l && !rawH) ? 0 : 1;
        i32 x[4], xn[4];
#pragma unroll
        for (int u = 0; u < 4; u++) xn[u] = ip[2 * u];
        for (int k = 0; k < npairs; k += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) x[u] = xn[u];
            if (k + 4 < npairs) {
#pragma unroll
                for (int u = 0; u < 4; u++) xn[u] = ip[2 * (k + 4 + u)];
            }
            SxV4i r;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const i32 in32 = sx_shl(x[u], 10);
                const i32 Y = sx_sub(in32, s);
                const i32 X = sx_add(sx_smulw_pre(Y, cpre), Y & mask);
                const i32 o = sx_add(s, X);
                s = sx_add(in32, X);
                const i32 p = SX_DPP_(l ? o : sx_neg(o), 0xB1);
                r.v[u] = sx_add(o, p);
            }
            *op = r;
            op += ostep;
        }
        if (S) S[l] = s;
    }
    wv_sync();
}
// out[i] = sat16(RSHIFT_ROUND(raw[i], 11)) (|raw| < 2^29: sx_rshift_round_small)
SX_HD void sx_allpass2_finish(i16* out, const i32* raw, int n) {
    SX_PAR(i, n) out[i] = (i16)sx_sat16(sx_rshift_round_small(raw[i], 11));
}
#endif

// SKP_Silk_ana_filt_bank_1, SKP_Silk_ana_filt_bank_1.c:45 (serial first-order all-pass pair)
// raw: N words of LDS scratch (gfx950 build: the band sums before rounding)
SX_HD void sx_ana_filt_bank_1(const i16* in, i32* S, i16* outL, i16* outH, int N, i32* raw) {
    const i32 A20 = (i16)(5394 << 1), A21 = (i16)(20623 << 1);   // the int16 wrap of A_fb1_21 is intentional
#if defined(SX_LANE_STREAM) && !defined(SX_NO_RAW_VAD)
    sx_allpass2_lanes_raw(in, N >> 1, S, A21, A20, raw, raw + (N >> 1), (i32*)0);
    sx_allpass2_finish(outL, raw, N >> 1);
    sx_allpass2_finish(outH, raw + (N >> 1), N >> 1);
    wv_sync();
#elif defined(SX_LANE_STREAM)
    sx_allpass2_lanes(in, N >> 1, S, A21, A20, outL, outH);
#elif 0
    // N <= 64 * SX_FCH samples in (a frame), half as many out per band (in may alias outL: everything is read before anything is stored)
    i32 r[SX_FCH], oL[(SX_FCH + 1) / 2], oH[(SX_FCH + 1) / 2];
#pragma unroll
    for (int j = 0; j < (SX_FCH + 1) / 2; j++) { oL[j] = 0; oH[j] = 0; }
#pragma unroll
    for (int j = 0; j < SX_FCH; j++) { const int i = SX_LANE + 64 * j; r[j] = i < N ? (i32)in[i] : 0; }
    i32 s0 = SX_UNI(S[0]), s1 = SX_UNI(S[1]);
#pragma unroll
    for (int c = 0; c < SX_FCH; c++) {
        const int kend = sx_min(N >> 1, 32 * (c + 1));
        for (int k = 32 * c; k < kend; k++) {
            i32 in32 = sx_shl(SX_RDLANE(r[c], (2 * k) & 63), 10);
            i32 Y = sx_sub(in32, s0);
            i32 X = sx_smlawb(Y, Y, A21);
            i32 out_1 = sx_add(s0, X);
            s0 = sx_add(in32, X);
            in32 = sx_shl(SX_RDLANE(r[c], (2 * k + 1) & 63), 10);
            Y = sx_sub(in32, s1);
            X = sx_smulwb(Y, A20);
            i32 out_2 = sx_add(s1, X);
            s1 = sx_add(in32, X);
            SX_WRLANE(oL[c >> 1], k & 63, sx_sat16(sx_rshift_round(sx_add(out_2, out_1), 11)));
            SX_WRLANE(oH[c >> 1], k & 63, sx_sat16(sx_rshift_round(sx_sub(out_2, out_1), 11)));
        }
    }
#pragma unroll
    for (int j = 0; j < (SX_FCH + 1) / 2; j++) {
        const int i = SX_LANE + 64 * j;
        if (i < (N >> 1)) { outL[i] = (i16)oL[j]; outH[i] = (i16)oH[j]; }
    }
    S[0] = s0;
    S[1] = s1;
    wv_sync();
#else
    i32 s0 = S[0], s1 = S[1];
    for (int k = 0; k < (N >> 1); k++) {
        i32 in32 = sx_shl((i32)in[2 * k], 10);
        i32 Y = sx_sub(in32, s0);
        i32 X = sx_smlawb(Y, Y, A21);
        i32 out_1 = sx_add(s0, X);
        s0 = sx_add(in32, X);
        in32 = sx_shl((i32)in[2 * k + 1], 10);
        Y = sx_sub(in32, s1);
        X = sx_smulwb(Y, A20);
        i32 out_2 = sx_add(s1, X);
        s1 = sx_add(in32, X);
        outL[k] = (i16)sx_sat16(sx_rshift_round(sx_add(out_2, out_1), 11));
        outH[k] = (i16)sx_sat16(sx_rshift_round(sx_sub(out_2, out_1), 11));
    }
    S[0] = s0;
    S[1] = s1;
#endif
}

// SKP_Silk_VAD_GetSA_Q8 (+ GetNoiseLevels), SKP_Silk_VAD.c:75-318.  X is a 4 x 80 int16 scratch (LDS).
// raw: SX_FRAME words of LDS scratch (see sx_ana_filt_bank_1)
SX_FN1 void sx_vad(SxEncState* st, SxEncCtrl* c, const i16* pIn, i16* X, i32* pSNR_dB_Q7, i32* raw) {
    SX_IN_LDS(st); SX_IN_LDS(c); SX_IN_LDS(X); SX_IN_LDS(raw);
    SxVAD* v = &st->vad;
    i16* X0 = X, *X1 = X + 80, *X2 = X + 160, *X3 = X + 240;
    sx_ana_filt_bank_1(pIn, v->AnaState, X0, X3, SX_FRAME, raw);
    sx_ana_filt_bank_1(X0, v->AnaState1, X0, X2, SX_FRAME >> 1, raw);
    sx_ana_filt_bank_1(X0, v->AnaState2, X0, X1, SX_FRAME >> 2, raw);
    // HP filter on lowest band (differentiator): h[i] = X0[i] >> 1, X0[i] = h[i] - h[i - 1] (h[-1] = the state), state = h[last]
    const int dfl = SX_FRAME >> 3;
    static_assert((SX_FRAME >> 3) % 2 == 0 && (SX_FRAME >> 3) + 2 * 16 <= 80, "the tail of X0 (its band is SX_FRAME / 8 long by now) is the scratch of the band statistics");
    i32* part = (i32*)(X + dfl);
#if SX_NLANES == 1
    {
        X0[dfl - 1] = (i16)(X0[dfl - 1] >> 1);
        const i16 HPstateTmp = X0[dfl - 1];
        for (int i = dfl - 1; i > 0; i--) {
            X0[i - 1] = (i16)(X0[i - 1] >> 1);
            X0[i] = (i16)(X0[i] - X0[i - 1]);
        }
        X0[0] = (i16)(X0[0] - (i16)v->HPstate);
        v->HPstate = HPstateTmp;
    }
#else
    {
        static_assert((SX_FRAME >> 3) <= 64, "one sample of the lowest band per lane");
        const int i = SX_LANE < dfl ? SX_LANE : 0;
        const i16 h = (i16)(X0[i] >> 1), hm = (i16)(i > 0 ? X0[i - 1] >> 1 : (i16)v->HPstate);
        wv_sync();
        if (SX_LANE < dfl) X0[i] = (i16)(h - hm);
        if (SX_LANE == dfl - 1) v->HPstate = h;
        wv_sync();
    }
#endif
    // band energies: lane (b, s) sums the squares of quarter s of band b (a wrapping 32-bit sum: the order is free) ...
    SX_PAR(t, 16) {
        const int b = t >> 2, q = t & 3;
        const int sub_len = (SX_FRAME >> sx_min(4 - b, 3)) >> 2;
        const i16* Xb = X + 80 * b + q * sub_len;
        i32 sum = 0;
        for (int i = 0; i < sub_len; i++) {
            const i32 x_tmp = Xb[i] >> 3;
            sum = sx_smlabb(sum, x_tmp, x_tmp);
        }
        part[t] = sum;
    }
    wv_sync();
    // ... and lane b folds its band's four with the reference's saturating adds, then runs the band's noise-level tracker
    // (SKP_Silk_VAD_GetNoiseLevels, VAD.c:260) and signal-to-noise terms; the sums over the bands are wrapping adds of per-band terms
    const i32 min_coef = v->counter < 1000 ? 32767 / ((v->counter >> 4) + 1) : 0;
    SX_PAR(b, 4) {
        i32 e = v->XnrgSubfr[b];
        for (int q = 0; q < 3; q++) e = sx_add_pos_sat32(e, part[4 * b + q]);
        const i32 last = part[4 * b + 3];
        e = sx_add_pos_sat32(e, last >> 1);
        v->XnrgSubfr[b] = last;
        i32 nl = v->NL[b];
        {
            const i32 nrg = sx_add_pos_sat32(e, v->NoiseLevelBias[b]);
            const i32 inv_nrg = SX_I32_MAX / nrg;
            i32 coef;
            if (nrg > sx_shl(nl, 3)) coef = 1024 >> 3;
            else if (nrg < nl) coef = 1024;
            else coef = sx_smulwb(sx_smulww(inv_nrg, nl), 1024 << 1);
            coef = sx_max(coef, min_coef);
            v->inv_NL[b] = sx_smlawb(v->inv_NL[b], inv_nrg - v->inv_NL[b], coef);
            nl = SX_I32_MAX / v->inv_NL[b];
            nl = sx_min(nl, 0x00FFFFFF);
            v->NL[b] = nl;
        }
        i32 ratio = 256, sq = 0, tilt = 0;
        const i32 speech_nrg_b = e - nl;
        if (speech_nrg_b > 0) {
            if ((e & 0xFF800000) == 0) ratio = sx_shl(e, 8) / (nl + 1);
            else ratio = e / ((nl >> 8) + 1);
            i32 SNR_Q7 = sx_lin2log(ratio) - 8 * 128;
            sq = sx_smulbb(SNR_Q7, SNR_Q7);
            if (speech_nrg_b < (1 << 20)) SNR_Q7 = sx_smulwb(sx_shl(sx_sqrt_approx(speech_nrg_b), 6), SNR_Q7);
            tilt = sx_smulwb(T_vad_tilt_weights[b], SNR_Q7);
        }
        part[4 * b] = ratio; part[4 * b + 1] = sq; part[4 * b + 2] = tilt; part[4 * b + 3] = (b + 1) * (speech_nrg_b >> 4);
    }
    v->counter++;
    wv_sync();
    i32 sumSquared = 0, input_tilt = 0, speech_nrg = 0;
    for (int b = 0; b < 4; b++) {
        sumSquared = sx_add(sumSquared, part[4 * b + 1]);
        input_tilt = sx_add(input_tilt, part[4 * b + 2]);
        speech_nrg = sx_add(speech_nrg, part[4 * b + 3]);
    }
    sumSquared = sumSquared / 4;
    *pSNR_dB_Q7 = (i16)(3 * sx_sqrt_approx(sumSquared));
    i32 SA_Q15 = sx_sigm_Q15(sx_smulwb(45000, *pSNR_dB_Q7) - 128);
    c->input_tilt_Q15 = sx_shl(sx_sigm_Q15(input_tilt) - 16384, 1);
    if (speech_nrg <= 0) {
        SA_Q15 = SA_Q15 >> 1;
    } else if (speech_nrg < 32768) {
        speech_nrg = sx_sqrt_approx(sx_shl(speech_nrg, 15));
        SA_Q15 = sx_smulwb(32768 + speech_nrg, SA_Q15);
    }
    st->speech_activity_Q8 = sx_min(SA_Q15 >> 7, 255);
    i32 smooth_coef_Q16 = (i16)sx_smulwb(4096, sx_smulwb(SA_Q15, SA_Q15));
    SX_PAR(b, 4) {
        v->NrgRatioSmth_Q8[b] = sx_smlawb(v->NrgRatioSmth_Q8[b], part[4 * b] - v->NrgRatioSmth_Q8[b], smooth_coef_Q16);
        i32 SNR_Q7 = 3 * (sx_lin2log(v->NrgRatioSmth_Q8[b]) - 8 * 128);
        c->input_quality_bands_Q15[b] = sx_sigm_Q15((SNR_Q7 - 16 * 128) >> 4);
    }
    wv_sync();
}

// SKP_Silk_HP_variable_cutoff_FIX (HP_variable_cutoff_FIX.c:37) + SKP_Silk_biquad_alt (biquad_alt.c:38)
// scratch: SX_HP_SCRATCH_WORDS words of LDS (GPU build: the input-only terms of the biquad for the whole frame)
#define SX_HP_SCRATCH_WORDS (4 * (SX_FRAME + 1))     // (+1: the recursion requests entry k + 1 while it works on entry k)
SX_FN1 void sx_hp_variable_cutoff(SxEncState* st, SxEncCtrl* c, i16* out, const i16* in, i32* scratch) {
    SX_IN_LDS(st); SX_IN_LDS(c); SX_IN_LDS(out); SX_IN_LDS(scratch);
    if (st->prev_sigtype == 0) {
        i32 pitch_freq_Hz_Q16 = sx_shl(SX_FS_KHZ * 1000, 16) / st->prevLag;
        i32 pitch_freq_log_Q7 = sx_lin2log(pitch_freq_Hz_Q16) - (16 << 7);
        i32 quality_Q15 = c->input_quality_bands_Q15[0];
        pitch_freq_log_Q7 = sx_sub(pitch_freq_log_Q7, sx_smulwb(sx_smulwb(sx_shl(quality_Q15, 2), quality_Q15), pitch_freq_log_Q7 - 809));
        pitch_freq_log_Q7 = sx_add(pitch_freq_log_Q7, (K_0p6_Q15 - quality_Q15) >> 9);
        i32 delta_freq_Q7 = pitch_freq_log_Q7 - (st->variable_HP_smth1_Q15 >> 8);
        if (delta_freq_Q7 < 0) delta_freq_Q7 = sx_mul(delta_freq_Q7, 3);
        delta_freq_Q7 = sx_limit(delta_freq_Q7, -K_VARIABLE_HP_MAX_DELTA_FREQ_Q7, K_VARIABLE_HP_MAX_DELTA_FREQ_Q7);
        st->variable_HP_smth1_Q15 = sx_smlawb(st->variable_HP_smth1_Q15,
                                              sx_mul(sx_shl(st->speech_activity_Q8, 1), delta_freq_Q7), K_VARIABLE_HP_SMTH_COEF1_Q16);
    }
    st->variable_HP_smth2_Q15 = sx_smlawb(st->variable_HP_smth2_Q15, st->variable_HP_smth1_Q15 - st->variable_HP_smth2_Q15,
                                          K_VARIABLE_HP_SMTH_COEF2_Q16);
    c->pitch_freq_low_Hz = sx_log2lin(st->variable_HP_smth2_Q15 >> 8);
    c->pitch_freq_low_Hz = sx_limit(c->pitch_freq_low_Hz, K_VARIABLE_HP_MIN_FREQ_Q0, K_VARIABLE_HP_MAX_FREQ_Q0);
    i32 Fc_Q19 = sx_smulbb(1482, c->pitch_freq_low_Hz) / SX_FS_KHZ;
    i32 r_Q28 = K_1p0_Q28 - sx_mul(K_0p92_Q9, Fc_Q19);
    i32 B0 = r_Q28, B1 = sx_shl(sx_neg(r_Q28), 1), B2 = r_Q28;
    i32 r_Q22 = r_Q28 >> 6;
    i32 A0 = sx_smulww(r_Q22, sx_smulww(Fc_Q19, Fc_Q19) - K_2p0_Q22);
    i32 A1 = sx_smulww(r_Q22, r_Q22);
    // biquad_alt (direct form II transposed), serial
    i32 A0_L = sx_neg(A0) & 0x3FFF, A0_U = sx_neg(A0) >> 14;
    i32 A1_L = sx_neg(A1) & 0x3FFF, A1_U = sx_neg(A1) >> 14;
    A0_L = SX_UNI(A0_L); A0_U = SX_UNI(A0_U); A1_L = SX_UNI(A1_L); A1_U = SX_UNI(A1_U);
    B0 = SX_UNI(B0); B1 = SX_UNI(B1); B2 = SX_UNI(B2);
    i32 S0 = SX_UNI(st->In_HP_State[0]), S1 = SX_UNI(st->In_HP_State[1]);
#ifdef SX_LANE_STREAM
    // The three products with the input sample feed nothing back: they are formed for the whole frame at once, lane-parallel.
    // What remains of the recursion per sample -- the two products with the new output and their rounding -- runs on the vector
    // unit (SX_VEC: see sx_allpass2_lanes), reading its input terms 16 bytes at a time and storing the output sample.
    {
        SxV4i* bx = (SxV4i*)scratch;
        const i32 b0p = B0, b1p = B1, b2p = B2;
        SX_PAR(k, SX_FRAME) {
            const i32 xs = sx_pre16(in[k]);              // smulwb(B, x) = (B * (x << 16)) >> 32
            SxV4i v;
            v.v[0] = sx_smulw_pre(b0p, xs); v.v[1] = sx_smulw_pre(b1p, xs); v.v[2] = sx_smulw_pre(b2p, xs); v.v[3] = 0;
            bx[k] = v;
        }
        wv_sync();
        {
            // the four products with the new output are taken by the four lanes of a quad (every quad of the wave runs the same
            // recursion; their stores coincide): lane q multiplies by its coefficient (A0_L, A0_U, A1_L, A1_U), the _L lanes round
            // their product by 14 bits ((m + 2^13) >> 14 = RSHIFT_ROUND(m, 14): |m| < 2^29, the coefficient has 14 bits), lane pairs
            // add up, and both sums go back to all four lanes (quad_perm broadcasts), which keep the two states replicated
            const int q = SX_LANE & 3;
            i32 s0 = S0, s1 = S1;
            i32 coef = sx_pre16(q == 0 ? A0_L : (q == 1 ? A0_U : (q == 2 ? A1_L : A1_U)));
            i32 radd = (q & 1) ? 0 : (1 << 13), rsh = (q & 1) ? 0 : 14;
            SX_VEC(s0); SX_VEC(s1); SX_VEC(coef); SX_VEC(radd); SX_VEC(rsh);
            SxV4i cur = bx[0];
#pragma unroll 4
            for (int k = 0; k < SX_FRAME; k++) {
                const SxV4i t = cur;
                cur = bx[k + 1];
                const i32 out32_Q14 = sx_shl(sx_add(s0, t.v[0]), 2);
                // (SX_VEC on the high word: the compiler would otherwise fuse the rounding shift into a 64-bit product -- low word,
                // unsigned high word and sign corrections -- three times the instructions of v_mul_hi_i32 + shift)
                i32 p = sx_smulw_pre(out32_Q14, coef);
                SX_VEC(p);
                const i32 r = sx_add(p, radd) >> rsh;
                const i32 w = sx_add(r, SX_DPP_(r, 0xB1));                                   // lanes 0, 1: round(m0) + u0; lanes 2, 3: round(m1) + u1
                const i32 n0 = sx_add(sx_add(s1, t.v[1]), SX_DPP_(w, 0x00));
                const i32 n1 = sx_add(t.v[2], SX_DPP_(w, 0xAA));
                s0 = n0; s1 = n1;
                bx[k].v[3] = out32_Q14;                  // (the record's spare word; rounded and saturated below, all samples side by side)
            }
            if (SX_LANE == 0) {
                st->In_HP_State[0] = s0;
                st->In_HP_State[1] = s1;
            }
        }
        wv_sync();
        SX_PAR(k, SX_FRAME) out[k] = (i16)sx_sat16(sx_add(bx[k].v[3], (1 << 14) - 1) >> 14);
        wv_sync();
        return;
    }
#elif 0
    i32 r[SX_FCH], o[SX_FCH];
#pragma unroll
    for (int j = 0; j < SX_FCH; j++) { const int i = SX_LANE + 64 * j; r[j] = i < SX_FRAME ? (i32)in[i] : 0; o[j] = 0; }
#pragma unroll
    for (int c = 0; c < SX_FCH; c++) {
        const int kend = sx_min(SX_FRAME, 64 * (c + 1));
        for (int k = 64 * c; k < kend; k++) {
            i32 inval = SX_RDLANE(r[c], k & 63);
            i32 out32_Q14 = sx_shl(sx_smlawb(S0, B0, inval), 2);
            S0 = sx_add(S1, sx_rshift_round(sx_smulwb(out32_Q14, A0_L), 14));
            S0 = sx_smlawb(S0, out32_Q14, A0_U);
            S0 = sx_smlawb(S0, B1, inval);
            S1 = sx_rshift_round(sx_smulwb(out32_Q14, A1_L), 14);
            S1 = sx_smlawb(S1, out32_Q14, A1_U);
            S1 = sx_smlawb(S1, B2, inval);
            SX_WRLANE(o[c], k & 63, sx_sat16(sx_add(out32_Q14, (1 << 14) - 1) >> 14));
        }
    }
#pragma unroll
    for (int j = 0; j < SX_FCH; j++) { const int i = SX_LANE + 64 * j; if (i < SX_FRAME) out[i] = (i16)o[j]; }
#else
    for (int k = 0; k < SX_FRAME; k++) {
        i32 inval = in[k];
        i32 out32_Q14 = sx_shl(sx_smlawb(S0, B0, inval), 2);
        S0 = sx_add(S1, sx_rshift_round(sx_smulwb(out32_Q14, A0_L), 14));
        S0 = sx_smlawb(S0, out32_Q14, A0_U);
        S0 = sx_smlawb(S0, B1, inval);
        S1 = sx_rshift_round(sx_smulwb(out32_Q14, A1_L), 14);
        S1 = sx_smlawb(S1, out32_Q14, A1_U);
        S1 = sx_smlawb(S1, B2, inval);
        out[k] = (i16)sx_sat16(sx_add(out32_Q14, (1 << 14) - 1) >> 14);
    }
#endif
    st->In_HP_State[0] = S0;
    st->In_HP_State[1] = S1;
}

// SKP_Silk_apply_sine_window, SKP_Silk_apply_sine_window.c:47 (serial recursion, 4 samples per step)
SX_HD void sx_apply_sine_window(i16* px_win, const i16* px, int win_type, int length) {
    int k = (length >> 2) - 4;
    i32 f_Q16 = T_sine_win_freq_Q16[k];
    i32 c_Q16 = sx_smulwb(f_Q16, -f_Q16);
    i32 S0, S1;
    if (win_type == 1) {
        S0 = 0;
        S1 = f_Q16 + (length >> 3);
    } else {
        S0 = 1 << 16;
        S1 = (1 << 16) + (c_Q16 >> 1) + (length >> 4);
    }
    for (k = 0; k < length; k += 4) {
        px_win[k] = (i16)sx_smulwb((S0 + S1) >> 1, px[k]);
        px_win[k + 1] = (i16)sx_smulwb(S1, px[k + 1]);
        S0 = sx_smulwb(S1, c_Q16) + sx_shl(S1, 1) - S0 + 1;
        S0 = sx_min(S0, 1 << 16);
        px_win[k + 2] = (i16)sx_smulwb((S0 + S1) >> 1, px[k + 2]);
        px_win[k + 3] = (i16)sx_smulwb(S0, px[k + 3]);
        S1 = sx_smulwb(S0, c_Q16) + sx_shl(S0, 1) - S1;
        S1 = sx_min(S1, 1 << 16);
    }
}

// int64 inner product of two int16 vectors, wave-parallel + reduction (inner_prod16_aligned_64)
SX_HD i64 sx_inner_prod64(const i16* a, const i16* b, int len) {
    i64 s = 0;
    SX_PAR(i, len) s += (i64)((i32)a[i] * (i32)b[i]);
    return wv_sum64(s);
}
// wrapping int32 inner product (inner_prod_aligned: SMLABB chain), wave-parallel + reduction
SX_HD i32 sx_inner_prod32(const i16* a, const i16* b, int len) {
    i32 s = 0;
    SX_PAR(i, len) s = sx_smlabb(s, a[i], b[i]);
    return wv_sum(s);
}
SX_HD i32 sx_clz64(i64 x) {
    i32 hi = (i32)(x >> 32);
    return hi == 0 ? 32 + sx_clz32((i32)x) : sx_clz32(hi);
}

// SKP_Silk_autocorr, SKP_Silk_autocorr.c:40
SX_HD void sx_autocorr(i32* results, i32* scale, const i16* x, int n, int count) {
    int corrCount = sx_min(n, count);
    i64 corr64 = sx_inner_prod64(x, x, n) + 1;
    int lz = sx_clz64(corr64);
    int nRightShifts = 35 - lz;
    *scale = nRightShifts;
    if (nRightShifts <= 0) {
        results[0] = sx_shl((i32)corr64, -nRightShifts);
        for (int i = 1; i < corrCount; i++) results[i] = sx_shl(sx_inner_prod32(x, x + i, n - i), -nRightShifts);
    } else {
        results[0] = (i32)(corr64 >> nRightShifts);
        for (int i = 1; i < corrCount; i++) results[i] = (i32)(sx_inner_prod64(x, x + i, n - i) >> nRightShifts);
    }
}

// SKP_Silk_schur, SKP_Silk_schur.c:40
SX_HD i32 sx_schur(i16* rc_Q15, const i32* c, int order) {
    i32 C[SX_MAX_LPC + 1][2];
    int lz = sx_clz32(c[0]);
    if (lz < 2) {
        for (int k = 0; k < order + 1; k++) C[k][0] = C[k][1] = c[k] >> 1;
    } else if (lz > 2) {
        lz -= 2;
        for (int k = 0; k < order + 1; k++) C[k][0] = C[k][1] = sx_shl(c[k], lz);
    } else {
        for (int k = 0; k < order + 1; k++) C[k][0] = C[k][1] = c[k];
    }
    for (int k = 0; k < order; k++) {
        i32 rc = sx_neg(C[k + 1][0] / sx_max(C[0][1] >> 15, 1));
        rc = sx_sat16(rc);
        rc_Q15[k] = (i16)rc;
        for (int n = 0; n < order - k; n++) {
            i32 t1 = C[n + k + 1][0], t2 = C[n][1];
            C[n + k + 1][0] = sx_smlawb(t1, sx_shl(t2, 1), rc);
            C[n][1] = sx_smlawb(t2, sx_shl(t1, 1), rc);
        }
    }
    return C[0][1];
}

// SKP_Silk_k2a, SKP_Silk_k2a.c:40
SX_HD void sx_k2a(i32* A_Q24, const i16* rc_Q15, int order) {
    i32 Atmp[SX_MAX_LPC];
    for (int k = 0; k < order; k++) {
        for (int n = 0; n < k; n++) Atmp[n] = A_Q24[n];
        for (int n = 0; n < k; n++) A_Q24[n] = sx_smlawb(A_Q24[n], sx_shl(Atmp[k - n - 1], 1), rc_Q15[k]);
        A_Q24[k] = sx_neg(sx_shl((i32)rc_Q15[k], 9));
    }
}

// SKP_Silk_resampler_down2, SKP_Silk_resampler_down2.c:41 (zero initial state, serial)
#define SX_DOWN2_MAXIN (40 * SX_FS_KHZ)      // the longest input: the pitch analysis buffer, 40 ms at the internal rate
// dump: inLen / 2 samples of LDS that may be overwritten (GPU build: the unused high band of the all-pass pair lands there)
SX_HD void sx_down2_zero_state(i16* out, const i16* in, int inLen, i16* dump) {
#ifdef SX_LANE_STREAM
    sx_allpass2_lanes(in, inLen >> 1, (i32*)0, T_down2_c1[0], T_down2_c0[0], out, dump);
    return;
#endif
    i32 S0 = 0, S1 = 0;
    const i32 c0 = SX_UNI(T_down2_c0[0]), c1 = SX_UNI(T_down2_c1[0]);
#if 0
    // inLen <= SX_DOWN2_MAXIN samples in (lane registers), half as many out
    i32 r[SX_DOWN2_MAXIN / 64], o[(SX_DOWN2_MAXIN / 2 + 63) / 64];
#pragma unroll
    for (int j = 0; j < (SX_DOWN2_MAXIN / 2 + 63) / 64; j++) o[j] = 0;
#pragma unroll
    for (int j = 0; j < SX_DOWN2_MAXIN / 64; j++) { const int i = SX_LANE + 64 * j; r[j] = i < inLen ? (i32)in[i] : 0; }
#pragma unroll
    for (int c = 0; c < SX_DOWN2_MAXIN / 64; c++) {
        const int kend = sx_min(inLen >> 1, 32 * (c + 1));
        for (int k = 32 * c; k < kend; k++) {
            i32 in32 = sx_shl(SX_RDLANE(r[c], (2 * k) & 63), 10);
            i32 Y = sx_sub(in32, S0);
            i32 X = sx_smlawb(Y, Y, c1);
            i32 out32 = sx_add(S0, X);
            S0 = sx_add(in32, X);
            in32 = sx_shl(SX_RDLANE(r[c], (2 * k + 1) & 63), 10);
            Y = sx_sub(in32, S1);
            X = sx_smulwb(Y, c0);
            out32 = sx_add(out32, S1);
            out32 = sx_add(out32, X);
            S1 = sx_add(in32, X);
            SX_WRLANE(o[c >> 1], k & 63, sx_sat16(sx_rshift_round(out32, 11)));
        }
    }
#pragma unroll
    for (int j = 0; j < (SX_DOWN2_MAXIN / 2 + 63) / 64; j++) { const int i = SX_LANE + 64 * j; if (i < (inLen >> 1)) out[i] = (i16)o[j]; }
#else
    for (int k = 0; k < (inLen >> 1); k++) {
        i32 in32 = sx_shl((i32)in[2 * k], 10);
        i32 Y = sx_sub(in32, S0);
        i32 X = sx_smlawb(Y, Y, c1);
        i32 out32 = sx_add(S0, X);
        S0 = sx_add(in32, X);
        in32 = sx_shl((i32)in[2 * k + 1], 10);
        Y = sx_sub(in32, S1);
        X = sx_smulwb(Y, c0);
        out32 = sx_add(out32, S1);
        out32 = sx_add(out32, X);
        S1 = sx_add(in32, X);
        out[k] = (i16)sx_sat16(sx_rshift_round(out32, 11));
    }
#endif
}

// SKP_Silk_int16_array_maxabs (array_maxabs.c:41) -> SKP_FIX_P_Ana_find_scaling (pitch_analysis_core.c:681)
SX_HD i32 sx_pitch_find_scaling(const i16* sig, int len, int sum_sqr_len) {
    // the reference scans for the largest SQUARE and returns |x| there, clamped to 32767: same as max |x| clamped
    i32 m = 0;
    SX_PAR(i, len) { i32 v = sig[i] < 0 ? -(i32)sig[i] : (i32)sig[i]; m = v > m ? v : m; }
    m = wv_max(m);
    i32 x_max = sx_min(m, 32767);
    i32 nbits;
    if (x_max < 32767) nbits = 32 - sx_clz32(sx_smulbb(x_max, x_max));
    else nbits = 30;
    nbits += 17 - sx_clz16((i16)sum_sqr_len);
    return nbits < 31 ? 0 : nbits - 30;
}

struct SxPitchWork {                 // LDS scratch of the pitch analysis
    alignas(16) i32 tmp32[160];      // (first, 16-byte aligned: also the raw band sums of the 8 -> 4 kHz decimator, stored 16 bytes at a time)
    alignas(16) i32 d_srch[24];
    i16 sig8[320];
    i16 sig4[160];
    i16 C[4][221];
    i16 d_comp[221];
#if SX_FS_KHZ == 16
    i16 sig16[640];                  // third stage: the (scaled) 16 kHz input
    i32 corr3[4][24], nrg3[4][24];   // per subframe: cross-correlation / basis energy over its stage-3 lag range (SCRATCH_SIZE = 22)
#endif
};

// SKP_Silk_pitch_analysis_core, SKP_Silk_pitch_analysis_core.c:65, Fs = 8 kHz, complexity 2 (so the
// third stage is skipped and the extended 11-entry stage-2 codebook is used).  Returns sigtype.
SX_FN1 int sx_pitch_analysis_core(const i16* signal, i32* pitch_out, i32* lagIndex, i32* contourIndex, i32* LTPCorr_Q15,
                                 i32 prevLag, i32 search_thres1_Q16, i32 search_thres2_Q15, SxPitchWork* w) {
    SX_IN_LDS(signal); SX_IN_LDS(pitch_out); SX_IN_LDS(lagIndex); SX_IN_LDS(contourIndex); SX_IN_LDS(LTPCorr_Q15); SX_IN_LDS(w);
    const int min_lag_4 = 8, max_lag_4 = 72, min_lag_8 = 16, max_lag_8 = 144, sf8 = 40;
    SX_PAR(i, 4 * 221) (&w->C[0][0])[i] = 0;
#if SX_FS_KHZ == 8
    SX_PAR(i, 320) w->sig8[i] = signal[i];
    wv_sync();
#else
    sx_down2_zero_state(w->sig8, signal, 640, (i16*)w->tmp32);      // 16 -> 8 kHz (pitch_analysis_core.c:127-129)
    wv_sync();
#endif
#if defined(SX_LANE_STREAM) && !defined(SX_NO_RAW_DECIM)
    // (the 160 low-band sums land in tmp32 as they are, the unused high band on d_srch, which is free until the first stage; rounded below)
    static_assert(sizeof(w->tmp32) >= 160 * sizeof(i32) && sizeof(w->d_srch) >= 16, "raw band sums of the 8 -> 4 kHz decimator");
    sx_allpass2_lanes_raw(w->sig8, 160, (i32*)0, T_down2_c1[0], T_down2_c0[0], w->tmp32, (i32*)0, w->d_srch);
    sx_allpass2_finish(w->sig4, w->tmp32, 160);
    wv_sync();
#else
    sx_down2_zero_state(w->sig4, w->sig8, 320, (i16*)w->tmp32);
#endif
#if SX_NLANES == 1
    for (int i = 159; i > 0; i--) w->sig4[i] = (i16)sx_sat16((i32)w->sig4[i] + (i32)w->sig4[i - 1]);
#else
    {   // y[i] = sat16(x[i] + x[i-1]) of the UNMODIFIED neighbours (the reference walks downwards): read all, then write
        i32 a[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = SX_LANE + 64 * j;
            a[j] = (i > 0 && i < 160) ? sx_sat16((i32)w->sig4[i] + (i32)w->sig4[i - 1]) : 0;
        }
        wv_sync();
#pragma unroll
        for (int j = 0; j < 3; j++) { const int i = SX_LANE + 64 * j; if (i > 0 && i < 160) w->sig4[i] = (i16)a[j]; }
        wv_sync();
    }
#endif
    i32 shift = sx_pitch_find_scaling(w->sig4, 160, sx_max(sf8, 80));
    if (shift > 0) {
        SX_PAR(i, 160) w->sig4[i] = (i16)(w->sig4[i] >> shift);
        wv_sync();
    }
    SX_S(34)
    // ---- first stage (4 kHz): normalised correlation of two 10 ms targets against lags 8..72 ----
    {
        // energies of the basis window at every lag: E(d) = sum_{n<40} b_d[n]^2, b_d = target - d
        // reference: N(8) = add_sat32(E(8), 40*4000), N(d) = N(d-1) + b_d[0]^2 - b_d[40]^2 (wrapping) = N(8) + E(d) - E(8)
        i32 E8k[2], N8k[2];
        for (int k = 0; k < 2; k++) {
            const i16* b = &w->sig4[80 + k * sf8] - min_lag_4;
            i32 E8 = 0;
            SX_PAR(n, sf8) E8 = sx_smlabb(E8, b[n], b[n]);
            E8 = wv_sum(E8);
            E8k[k] = E8;
            N8k[k] = sx_add_sat32(E8, sx_smulbb(sf8, 4000));
        }
        // (target, lag) pairs side by side: 2 x 65 of them are three rounds of the wave (a round per target would be four)
        const int nlag = max_lag_4 - min_lag_4 + 1;
        SX_PAR(t, 2 * nlag) {
            const int k = t / nlag, d = min_lag_4 + (t - k * nlag);
            const i16* target = &w->sig4[80 + k * sf8];
            const i16* b = target - d;
            i32 cc = 0, E = 0;
            for (int n = 0; n < sf8; n++) { cc = sx_smlabb(cc, target[n], b[n]); E = sx_smlabb(E, b[n], b[n]); }
            i32 normalizer = sx_add(k ? N8k[1] : N8k[0], sx_sub(E, k ? E8k[1] : E8k[0]));
            i32 tq = cc / (sx_sqrt_approx(normalizer) + 1);
            w->C[k][d] = (i16)sx_sat16(tq);
        }
        wv_sync();
    }
    SX_PAR(i, max_lag_4 + 1) {
        if (i >= min_lag_4) {
            i32 sum = ((i32)w->C[0][i] + (i32)w->C[1][i]) >> 1;
            sum = sx_smlawb(sum, sum, sx_shl(-i, 4));
            w->tmp32[i] = sum;
        }
    }
    wv_sync();
    // insertion_sort_decreasing_int16 of C[0][8..72], top-8 (value desc, index asc): rank by counting
    int length_d_srch = 4 + 2 * 2;
    const int L = max_lag_4 - min_lag_4 + 1;
#if SX_NLANES == 64 && defined(__HIP_DEVICE_COMPILE__)
    {
        // (L = 65: lane i ranks element i against all of them; the last element, which every other one precedes, is ranked with a ballot)
        static_assert(72 - 8 + 1 == 65, "one element more than lanes");
        const int i = SX_LANE;
        const i32 v = (i16)w->tmp32[min_lag_4 + i], vl = (i16)w->tmp32[min_lag_4 + L - 1];
        int rank = 0;
        for (int j = 0; j < L; j++) {
            i32 u = (i16)w->tmp32[min_lag_4 + j];
            rank += (u > v || (u == v && j < i)) ? 1 : 0;
        }
        const int rank_last = __builtin_popcountll(__builtin_amdgcn_ballot_w64(v >= vl));
        wv_sync();
        if (rank < length_d_srch) { w->C[0][min_lag_4 + rank] = (i16)v; w->d_srch[rank] = i; }
        if (SX_LANE == 0 && rank_last < length_d_srch) { w->C[0][min_lag_4 + rank_last] = (i16)vl; w->d_srch[rank_last] = L - 1; }
    }
#else
    SX_PAR(i, L) {
        i32 v = (i16)w->tmp32[min_lag_4 + i];
        int rank = 0;
        for (int j = 0; j < L; j++) {
            i32 u = (i16)w->tmp32[min_lag_4 + j];
            rank += (u > v || (u == v && j < i)) ? 1 : 0;
        }
        if (rank < length_d_srch) { w->C[0][min_lag_4 + rank] = (i16)v; w->d_srch[rank] = i; }
    }
#endif
    wv_sync();
    const i16* target = &w->sig4[80];
    i32 energy = 0;
    SX_PAR(n, 80) energy = sx_smlabb(energy, target[n], target[n]);
    energy = wv_sum(energy);
    energy = sx_add_pos_sat32(energy, 1000);
    i32 Cmax = w->C[0][min_lag_4];
    i32 threshold = sx_smulbb(Cmax, Cmax);
    if ((energy >> 6) > threshold) {
        for (int k = 0; k < 4; k++) pitch_out[k] = 0;
        *LTPCorr_Q15 = 0; *lagIndex = 0; *contourIndex = 0;
        return 1;
    }
    threshold = sx_smulwb(search_thres1_Q16, Cmax);
    for (int i = 0; i < length_d_srch; i++) {
        if (w->C[0][min_lag_4 + i] > threshold) w->d_srch[i] = (w->d_srch[i] + min_lag_4) << 1;
        else { length_d_srch = i; break; }
    }
#if SX_NLANES == 1
    for (int i = min_lag_8 - 5; i < max_lag_8 + 5; i++) w->d_comp[i] = 0;
    for (int i = 0; i < length_d_srch; i++) w->d_comp[w->d_srch[i]] = 1;
    for (int i = max_lag_8 + 3; i >= min_lag_8; i--) w->d_comp[i] = (i16)(w->d_comp[i] + w->d_comp[i - 1] + w->d_comp[i - 2]);
    length_d_srch = 0;
    for (int i = min_lag_8; i < max_lag_8 + 1; i++) {
        if (w->d_comp[i + 1] > 0) { w->d_srch[length_d_srch] = i; length_d_srch++; }
    }
    for (int i = max_lag_8 + 3; i >= min_lag_8; i--)
        w->d_comp[i] = (i16)(w->d_comp[i] + w->d_comp[i - 1] + w->d_comp[i - 2] + w->d_comp[i - 3]);
    int length_d_comp = 0;
    for (int i = min_lag_8; i < max_lag_8 + 4; i++) {
        if (w->d_comp[i] > 0) { w->d_comp[length_d_comp] = (i16)(i - 2); length_d_comp++; }
    }
#else
    int length_d_comp = 0;
    {
        // the reference's downward in-place smears read unmodified lower neighbours: y[i] = x[i] + x[i-1] + x[i-2] (+ x[i-3]);
        // its ordered scans become ballot compactions.  Lane l covers lags l, l + 64, l + 128 (< 160)
        const unsigned long long below = (1ull << SX_LANE) - 1ull;
        i32 m[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = SX_LANE + 64 * j;
            i32 v = 0;
            for (int q = 0; q < length_d_srch; q++) v |= (w->d_srch[q] == i) ? 1 : 0;
            m[j] = v;
        }
        wv_sync();
#pragma unroll
        for (int j = 0; j < 3; j++) { const int i = SX_LANE + 64 * j; if (i >= min_lag_8 - 5 && i < max_lag_8 + 5) w->d_comp[i] = (i16)m[j]; }
        wv_sync();
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = SX_LANE + 64 * j;
            m[j] = (i >= min_lag_8 && i <= max_lag_8 + 3) ? (i32)w->d_comp[i] + w->d_comp[i - 1] + w->d_comp[i - 2] : ((i < 160 && i >= min_lag_8 - 5) ? (i32)w->d_comp[i] : 0);
        }
        wv_sync();
#pragma unroll
        for (int j = 0; j < 3; j++) { const int i = SX_LANE + 64 * j; if (i >= min_lag_8 && i <= max_lag_8 + 3) w->d_comp[i] = (i16)m[j]; }
        wv_sync();
        // d_srch = { i in [min_lag_8, max_lag_8] : d_comp[i + 1] > 0 } in ascending order
        int n_srch = 0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = SX_LANE + 64 * j;
            const bool pr = i >= min_lag_8 && i <= max_lag_8 && w->d_comp[i + 1] > 0;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(pr);
            if (pr) w->d_srch[n_srch + __builtin_popcountll(bal & below)] = i;
            n_srch += __builtin_popcountll(bal);
        }
        length_d_srch = n_srch;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = SX_LANE + 64 * j;
            m[j] = (i >= min_lag_8 && i <= max_lag_8 + 3) ? (i32)w->d_comp[i] + w->d_comp[i - 1] + w->d_comp[i - 2] + w->d_comp[i - 3] : 0;
        }
        wv_sync();
        // d_comp[0 .. n) = { i - 2 : i in [min_lag_8, max_lag_8 + 3], smeared d_comp[i] > 0 } in ascending order
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = SX_LANE + 64 * j;
            const bool pr = i >= min_lag_8 && i <= max_lag_8 + 3 && m[j] > 0;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(pr);
            if (pr) w->d_comp[length_d_comp + __builtin_popcountll(bal & below)] = (i16)(i - 2);
            length_d_comp += __builtin_popcountll(bal);
        }
        wv_sync();
    }
#endif
    SX_S(35)
    // ---- second stage (8 kHz) ----
    shift = sx_pitch_find_scaling(w->sig8, 320, sf8);
    if (shift > 0) {
        SX_PAR(i, 320) w->sig8[i] = (i16)(w->sig8[i] >> shift);
    }
    SX_PAR(i, 4 * 221) (&w->C[0][0])[i] = 0;
    wv_sync();
    SX_PAR(t, 4 * length_d_comp) {
        int k = t / length_d_comp, j = t - k * length_d_comp;
        const i16* tp = &w->sig8[160 + k * sf8];
        int d = w->d_comp[j];
        const i16* bp = tp - d;
        i32 cross = 0, eb = 0, et = 0;
        for (int n = 0; n < sf8; n++) {
            cross = sx_smlabb(cross, tp[n], bp[n]);
            eb = sx_smlabb(eb, bp[n], bp[n]);
            et = sx_smlabb(et, tp[n], tp[n]);
        }
        i32 r = 0;
        if (cross > 0) {
            i32 en = sx_max(et, eb);
            i32 lz = sx_clz32(cross);
            i32 lshift = sx_limit(lz - 1, 0, 15);
            i32 t32 = sx_shl(cross, lshift) / ((en >> (15 - lshift)) + 1);
            t32 = sx_smulwb(cross, t32);
            t32 = sx_add_sat32(t32, t32);
            lz = sx_clz32(t32);
            lshift = sx_limit(lz - 1, 0, 15);
            en = sx_min(et, eb);
            r = sx_shl(t32, lshift) / ((en >> (15 - lshift)) + 1);
        }
        w->C[k][d] = (i16)r;
    }
    wv_sync();
    i32 CCmax = SX_I32_MIN, CCmax_b = SX_I32_MIN;
    int CBimax = 0, lag = -1;
#if SX_FS_KHZ == 16
    if (prevLag > 0) prevLag = prevLag >> 1;            // the previous lag on the 8 kHz grid (pitch_analysis_core.c:362-368)
#endif
    i32 prevLag_log2_Q7 = prevLag > 0 ? sx_lin2log(prevLag) : 0;
    i32 corr_thres_Q15 = sx_smulbb(search_thres2_Q15, search_thres2_Q15) >> 13;
    SX_S(36)
    const int nb_cbks = SX_FS_KHZ == 8 ? 11 : 3;        // PITCH_EST_NB_CBKS_STAGE2_EXT when 8 kHz is the last stage, else _STAGE2
#if SX_NLANES == 1
    for (int k = 0; k < length_d_srch; k++) {
        int d = w->d_srch[k];
        i32 CCmax_new = SX_I32_MIN;
        int CBimax_new = 0;
        for (int j = 0; j < nb_cbks; j++) {
            i32 cc = 0;
            for (int i = 0; i < 4; i++) cc += (i32)w->C[i][d + T_pitch_cb_stage2[i * 11 + j]];
            if (cc > CCmax_new) { CCmax_new = cc; CBimax_new = j; }
        }
        i32 lag_log2_Q7 = sx_lin2log(d);
        i32 CCmax_new_b = CCmax_new - (sx_smulbb(4 * 6554, lag_log2_Q7) >> 7);
        if (prevLag > 0) {
            i32 dl = lag_log2_Q7 - prevLag_log2_Q7;
            dl = sx_smulbb(dl, dl) >> 7;
            i32 bias = sx_smulbb(4 * 6554, *LTPCorr_Q15) >> 15;
            bias = sx_mul(bias, dl) / (dl + (1 << 6));
            CCmax_new_b -= bias;
        }
        if (CCmax_new_b > CCmax_b && CCmax_new > corr_thres_Q15 && T_pitch_cb_stage2[CBimax_new] <= min_lag_8) {
            CCmax_b = CCmax_new_b;
            CCmax = CCmax_new;
            lag = d;
            CBimax = CBimax_new;
        }
    }
#else
    {   // one candidate lag per lane; the reference keeps the FIRST lag with the largest biased correlation among those that pass
        // its tests (strict '>' in lag order) = arg-max with the lower index winning ties
        i32 best_b = SX_I32_MIN, best_k = SX_I32_MAX, my_cc = 0, my_cb = 0;
        const int k = SX_LANE;
        if (k < length_d_srch) {
            const int d = w->d_srch[k];
            i32 CCmax_new = SX_I32_MIN;
            int CBimax_new = 0;
            for (int j = 0; j < nb_cbks; j++) {
                i32 cc = 0;
                for (int i = 0; i < 4; i++) cc += (i32)w->C[i][d + T_pitch_cb_stage2[i * 11 + j]];
                if (cc > CCmax_new) { CCmax_new = cc; CBimax_new = j; }
            }
            i32 lag_log2_Q7 = sx_lin2log(d);
            i32 CCmax_new_b = CCmax_new - (sx_smulbb(4 * 6554, lag_log2_Q7) >> 7);
            if (prevLag > 0) {
                i32 dl = lag_log2_Q7 - prevLag_log2_Q7;
                dl = sx_smulbb(dl, dl) >> 7;
                i32 bias = sx_smulbb(4 * 6554, *LTPCorr_Q15) >> 15;
                bias = sx_mul(bias, dl) / (dl + (1 << 6));
                CCmax_new_b -= bias;
            }
            // (a candidate with CCmax_new_b == INT32_MIN can never be chosen by the reference's strict '>' either)
            if (CCmax_new_b > SX_I32_MIN && CCmax_new > corr_thres_Q15 && T_pitch_cb_stage2[CBimax_new] <= min_lag_8) {
                best_b = CCmax_new_b; best_k = k; my_cc = CCmax_new; my_cb = CBimax_new;
            }
        }
        wv_argmax(&best_b, &best_k);
        if (best_k != SX_I32_MAX && best_b > SX_I32_MIN) {
            CCmax = wv_bcast(my_cc, best_k);
            CBimax = wv_bcast(my_cb, best_k);
            lag = w->d_srch[best_k];
        }
    }
#endif
    if (lag == -1) {
        for (int k = 0; k < 4; k++) pitch_out[k] = 0;
        *LTPCorr_Q15 = 0; *lagIndex = 0; *contourIndex = 0;
        return 1;
    }
#if SX_FS_KHZ == 8
    CCmax = sx_max(CCmax, 0);
    *LTPCorr_Q15 = sx_sqrt_approx(sx_shl(CCmax, 13));
    for (int k = 0; k < 4; k++) pitch_out[k] = lag + T_pitch_cb_stage2[k * 11 + CBimax];
    *lagIndex = lag - min_lag_8;
    *contourIndex = CBimax;
    return 0;
#else
    // ---- third stage (16 kHz), pitch_analysis_core.c:445-548 with SKP_FIX_P_Ana_calc_corr_st3 / _calc_energy_st3 (complexity 2:
    // all 34 contours, five lags around twice the stage-2 lag).  The reference tabulates [subframe][contour][lag]; here the
    // per-subframe lag-range vectors are kept and indexed by contour offset + lag when the candidates are scored.
    const int min_lag = 2 * 16, max_lag = 18 * 16, sf = 80;
    shift = sx_pitch_find_scaling(signal, 640, sf);
    SX_PAR(i, 640) w->sig16[i] = (i16)(shift > 0 ? (signal[i] >> shift) : signal[i]);
    lag = sx_limit(lag << 1, min_lag, max_lag);
    const int start_lag = sx_max(lag - 2, min_lag), end_lag = sx_min(lag + 2, max_lag);
    int lag_new = lag;
    *LTPCorr_Q15 = sx_sqrt_approx(sx_shl(CCmax, 13));
    wv_sync();
    const i16* lr = &T_pitch_lag_range_stage3[2 * 4 * 2];       // [complexity 2][subframe][lo, hi]
    SX_PAR(t, 4 * 24) {
        const int k = t / 24, jj = t - k * 24, lo = lr[2 * k], hi = lr[2 * k + 1];
        if (jj <= hi - lo) {
            const i16* tp = &w->sig16[4 * sf + k * sf];
            const i16* bp = tp - (start_lag + lo + jj);
            i32 cc = 0;
            for (int n = 0; n < sf; n++) cc = sx_smlabb(cc, tp[n], bp[n]);
            w->corr3[k][jj] = cc;
        }
    }
    SX_PAR(k, 4) {      // basis energies: recursive over the lag range, with the reference's saturating add
        const int lo = lr[2 * k], hi = lr[2 * k + 1];
        const i16* bp = &w->sig16[4 * sf + k * sf] - (start_lag + lo);
        i32 e = 0;
        for (int n = 0; n < sf; n++) e = sx_smlabb(e, bp[n], bp[n]);
        w->nrg3[k][0] = e;
        for (int i = 1; i < hi - lo + 1; i++) {
            e -= sx_smulbb(bp[sf - i], bp[sf - i]);
            e = sx_add_sat32(e, sx_smulbb(bp[-i], bp[-i]));
            w->nrg3[k][i] = e;
        }
    }
    wv_sync();
    const i32 contour_bias = 52429 / lag;                       // PITCH_EST_FLATCONTOUR_BIAS_Q20 / lag
    const int cbk_size = T_pitch_cbk_sizes_stage3[2], cbk_offset = T_pitch_cbk_offsets_stage3[2];
    // candidates in the reference's order (lag-major, contour-minor); the FIRST largest score wins (strict '>')
    i32 best_v = SX_I32_MIN, best_t = SX_I32_MAX;
    const int ncand = (end_lag - start_lag + 1) * cbk_size;
    SX_PAR(t, ncand) {
        const int lc = t / cbk_size, j = cbk_offset + (t - lc * cbk_size), d = start_lag + lc;
        i32 cross = 0, energy = 0;
        for (int k = 0; k < 4; k++) {
            const int idx = T_pitch_cb_stage3[k * 34 + j] - lr[2 * k] + lc;
            energy += w->nrg3[k][idx] >> 2;
            cross += w->corr3[k][idx] >> 2;
        }
        i32 cn = 0;
        if (cross > 0) {
            const i32 lz = sx_clz32(cross);
            const i32 lshift = sx_limit(lz - 1, 0, 13);
            cn = sx_shl(cross, lshift) / ((energy >> (13 - lshift)) + 1);
            cn = sx_sat16(cn);
            cn = sx_smulwb(cross, cn);
            cn = cn > (SX_I32_MAX >> 3) ? SX_I32_MAX : sx_shl(cn, 3);
            i32 diff = j - (34 >> 1);
            diff = sx_mul(diff, diff);
            diff = 32767 - (sx_mul(contour_bias, diff) >> 5);
            cn = sx_shl(sx_smulwb(cn, diff), 1);
        }
        if (d + (int)T_pitch_cb_stage3[j] <= max_lag && (cn > best_v || (cn == best_v && t < best_t))) { best_v = cn; best_t = t; }
    }
    wv_argmax(&best_v, &best_t);
    CBimax = 0;
    if (best_t != SX_I32_MAX && best_v > SX_I32_MIN) {
        const int lc = best_t / cbk_size;
        lag_new = start_lag + lc;
        CBimax = cbk_offset + (best_t - lc * cbk_size);
    }
    for (int k = 0; k < 4; k++) pitch_out[k] = lag_new + T_pitch_cb_stage3[k * 34 + CBimax];
    *lagIndex = lag_new - min_lag;
    *contourIndex = CBimax;
    return 0;
#endif
}

#ifdef SX_LANE_STREAM
// steps K .. ORDER - 1 of SKP_Silk_k2a (SKP_Silk_k2a.c:40) on a coefficient vector held one element per lane of a 16-lane row
template <int K, int ORDER>
SX_HD void sx_row_k2a_steps(i32& A, i32 rcv, int j) {
    if constexpr (K < ORDER) {
        const i32 rck = SX_RDLANE(rcv, K);
        if constexpr (K > 0) {
            const i32 m = SX_DPP_(A, 0x140);                         // row_mirror: element 15 - j
            const i32 g = SX_DPP_(m, 0x100 + (16 - K));              // row_shl 16 - K: element K - 1 - j (lanes j < K)
            if (j < K) A = sx_smlawb(A, sx_shl(g, 1), rck);
        }
        if (j == K) A = sx_neg(sx_shl(rck, 9));
        sx_row_k2a_steps<K + 1, ORDER>(A, rcv, j);
    }
}
#endif

// SKP_Silk_find_pitch_lags_FIX, SKP_Silk_find_pitch_lags_FIX.c:32.  x = x_buf + frame_length.
// res: 336 samples (LDS).  Wsig: 192 samples scratch.
SX_FN1 void sx_find_pitch_lags(SxEncState* st, SxEncCtrl* c, const i16* x_buf, i16* res, i16* Wsig, SxPitchWork* pw) {
    SX_IN_LDS(st); SX_IN_LDS(c); SX_IN_LDS(x_buf); SX_IN_LDS(res); SX_IN_LDS(Wsig); SX_IN_LDS(pw);
    const int buf_len = SX_LA_PITCH + 2 * SX_FRAME;        // 336
    i32 auto_corr[SX_MAX_LPC + 1], A_Q24[SX_MAX_LPC], scale;
    i16 rc_Q15[SX_MAX_LPC], A_Q12[SX_MAX_LPC];
    const i16* x_ptr = x_buf + buf_len - SX_PITCH_LPC_WIN;
    sx_apply_sine_window(Wsig, x_ptr, 1, SX_LA_PITCH);
    SX_PAR(i, SX_PITCH_LPC_WIN - 2 * SX_LA_PITCH) Wsig[SX_LA_PITCH + i] = x_ptr[SX_LA_PITCH + i];
    sx_apply_sine_window(Wsig + SX_PITCH_LPC_WIN - SX_LA_PITCH, x_ptr + SX_PITCH_LPC_WIN - SX_LA_PITCH, 2, SX_LA_PITCH);
    wv_sync();
#if defined(SX_LANE_STREAM) && SX_PITCH_LPC_ORDER <= 15
    // Autocorrelation, Schur recursion, step-up and bandwidth expansion with one vector element per lane of a 16-lane row (the four rows
    // hold copies after the autocorrelation): lane (chunk, lag) sums a quarter of the window's products for its lag exactly, in 64 bits
    // -- the reference's wrapping 32-bit sums (autocorr.c:58) are the low words of the exact ones --, the order-n inner loops of
    // SKP_Silk_schur / SKP_Silk_k2a are one step each, the quotient of a Schur step is computed once per wave.
    {
        constexpr int ORD = SX_PITCH_LPC_ORDER, CH = SX_PITCH_LPC_WIN / 4;
        static_assert(SX_PITCH_LPC_WIN % 4 == 0 && CH > ORD && sizeof(pw->tmp32) >= 2 * 64 * sizeof(i32), "chunks of the pitch LPC window");
        i32 *part_lo = pw->tmp32, *part_hi = pw->tmp32 + 64;
        const int j = SX_LANE & 15, cch = SX_LANE >> 4;
        {
            i64 acc = 0;
            if (j <= ORD) {
                const i16 *xa = Wsig + cch * CH, *xb = xa + j;
                const int len = cch == 3 ? CH - j : CH;          // the lag's products end with the window
#pragma unroll 8
                for (int u = 0; u < CH - ORD; u++) acc += (i64)((i32)xa[u] * (i32)xb[u]);
                for (int u = CH - ORD; u < CH; u++) {            // (past the window: a factor of zero; the read stays inside the work area)
                    const i32 bv = (i32)xb[u];
                    acc += (i64)((i32)xa[u] * (u < len ? bv : 0));
                }
            }
            part_lo[SX_LANE] = (i32)acc;
            part_hi[SX_LANE] = (i32)(acc >> 32);
        }
        wv_sync();
        i64 S = 0;
        for (int r = 0; r < 4; r++) S += (i64)(((u64)(u32)part_hi[16 * r + j] << 32) | (u64)(u32)part_lo[16 * r + j]);
        wv_sync();
        const i64 corr64 = (i64)(((u64)(u32)SX_RDLANE((i32)(S >> 32), 0) << 32) | (u64)(u32)SX_RDLANE((i32)S, 0)) + 1;
        const int nRightShifts = 35 - sx_clz64(corr64);
        if (j == 0) S = corr64;
        i32 cj = nRightShifts <= 0 ? sx_shl((i32)S, -nRightShifts) : (i32)(S >> nRightShifts);
        if (j > ORD) cj = 0;
        if (j == 0) cj = sx_smlawb(cj, cj, K_FIND_PITCH_WHITE_NOISE_FRACTION_Q16);
        const i32 c0 = SX_RDLANE(cj, 0);
        // SKP_Silk_schur: lane n keeps C[n][1] and, at step k, C[n + k + 1][0]
        {
            const int lz = sx_clz32(c0);
            if (lz < 2) cj = cj >> 1;
            else if (lz > 2) cj = sx_shl(cj, lz - 2);
        }
        i32 C1 = cj, D = SX_ROW_NEXT(cj), rcv = 0;
#pragma unroll
        for (int k = 0; k < ORD; k++) {
            const i32 b0 = SX_RDLANE(D, 0), c01 = SX_RDLANE(C1, 0);
            const i32 rc = sx_sat16(sx_neg(b0 / sx_max(c01 >> 15, 1)));
            if (j == k) rcv = rc;
            const bool in = j < ORD - k;
            const i32 Dn = in ? sx_smlawb(D, sx_shl(C1, 1), rc) : D;
            if (in) C1 = sx_smlawb(C1, sx_shl(D, 1), rc);
            D = SX_ROW_NEXT(Dn);
        }
        const i32 res_nrg = SX_RDLANE(C1, 0);
        c->predGain_Q16 = sx_div32_varQ(c0, sx_max(res_nrg, 1), 16);
        // SKP_Silk_k2a: A[n] += (A[k - 1 - n] << 1) * rc[k] for n < k -- the reversed row is the mirrored row shifted by 16 - k lanes
        i32 A = 0;
        sx_row_k2a_steps<0, ORD>(A, rcv, j);
        // Q12, SKP_Silk_bwexpander: coefficient j is scaled by the chirp factor after j of its updates
        i32 a = sx_sat16(A >> 12);
        {
            i32 chirp_Q16 = K_FIND_PITCH_BANDWITH_EXPANSION_Q16, mine = chirp_Q16;
            const i32 chirp_minus_one_Q16 = chirp_Q16 - 65536;
            for (int i = 1; i < ORD; i++) {
                chirp_Q16 = sx_add(chirp_Q16, sx_rshift_round(sx_mul(chirp_Q16, chirp_minus_one_Q16), 16));
                if (j == i) mine = chirp_Q16;
            }
            a = (i16)sx_rshift_round(sx_mul(mine, a), 16);
        }
#pragma unroll
        for (int d = 0; d < ORD; d++) A_Q12[d] = (i16)SX_RDLANE(a, d);
        (void)auto_corr; (void)A_Q24; (void)rc_Q15; (void)scale;
    }
#else
    sx_autocorr(auto_corr, &scale, Wsig, SX_PITCH_LPC_WIN, SX_PITCH_LPC_ORDER + 1);
    auto_corr[0] = sx_smlawb(auto_corr[0], auto_corr[0], K_FIND_PITCH_WHITE_NOISE_FRACTION_Q16);
    i32 res_nrg = sx_schur(rc_Q15, auto_corr, SX_PITCH_LPC_ORDER);
    c->predGain_Q16 = sx_div32_varQ(auto_corr[0], sx_max(res_nrg, 1), 16);
    sx_k2a(A_Q24, rc_Q15, SX_PITCH_LPC_ORDER);
    for (int i = 0; i < SX_PITCH_LPC_ORDER; i++) A_Q12[i] = (i16)sx_sat16(A_Q24[i] >> 12);
    sx_bwexpander(A_Q12, SX_PITCH_LPC_ORDER, K_FIND_PITCH_BANDWITH_EXPANSION_Q16);
#endif
    // MA_Prediction with zero state over the whole buffer (SKP_Silk_MA.c:40), FIR form, then zero the first `order` outputs
    SX_PAR(k, buf_len) {
        i32 acc = 0;
        for (int d = 0; d < SX_PITCH_LPC_ORDER; d++) {
            int t = k - 1 - d;
            if (t >= 0) acc = sx_smlabb(acc, x_buf[t], A_Q12[d]);
        }
        i32 o = sx_rshift_round(sx_sub(sx_shl((i32)x_buf[k], 12), acc), 12);
        res[k] = k < SX_PITCH_LPC_ORDER ? (i16)0 : (i16)sx_sat16(o);
    }
    wv_sync();
    i32 thr = K_0p45_Q15;
    thr = sx_smlabb(thr, K_m0p004_Q15, SX_PITCH_LPC_ORDER);
    thr = sx_smlabb(thr, K_m0p1_Q7, st->speech_activity_Q8);
    thr = sx_smlabb(thr, K_0p15_Q15, st->prev_sigtype);
    thr = sx_smlawb(thr, K_m0p1_Q16, c->input_tilt_Q15);
    thr = sx_sat16(thr);
    SX_S(33)
    c->sigtype = sx_pitch_analysis_core(res, c->pitchL, &c->lagIndex, &c->contourIndex, &st->LTPCorr_Q15, st->prevLag,
                                        K_FIND_PITCH_CORRELATION_THRESHOLD_HC_MODE_Q16, (i16)thr, pw);
}
