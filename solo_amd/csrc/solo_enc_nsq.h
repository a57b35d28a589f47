// solo_enc_nsq.h -- the SOLO multiple-description noise-shaping quantiser: three coupled delayed-decision
// trellises (centre, MD1, MD2), 4 states each, quantising one 20 ms frame.  Row E6 of SURVEY.md section 8(a).
// Reference: JC1_SDK_SRC_ARM/src/libSATECodec/SKP_Silk_NSQ_del_dec.c:148-1694 and Agora_SILK_func.c:7-160.
//
// Mapping: lane tk = 4*track + state owns one (track, state) pair (12 active lanes) through the per-sample
// prediction / shaping / candidate phases; the survivor bookkeeping (JudgeWinner) is wave-uniform; state copies,
// scaling and re-whitening are lane-strided.  All trellis state lives in LDS (SxNsqWork); the persistent
// per-track NSQ state (SxNSQ) lives in the stream's HBM record.
#pragma once
#include "solo_enc_state.h"

#define SX_JOINT_LAMBDA 90000        // INTERNAL_JOINT_LAMBDA, SKP_Silk_define.h:48 (LARS_LAMBDA_AGR == 0)
#define SX_DD_MASK (SX_DD_DELAY - 1)
#define SX_LPC_RING 16
#define SX_LPC_MASK (SX_LPC_RING - 1)

struct SxDD {                        // NSQ_del_dec_struct, NSQ_del_dec.c:32 (live members only)
    i32 RandState[SX_DD_DELAY], Xq_Q10[SX_DD_DELAY], Pred_Q16[SX_DD_DELAY], Shape_Q10[SX_DD_DELAY];
    i32 sAR2_Q14[SX_SHAPE_ORDER];
    i32 sLPC_Q14[SX_LPC_RING];       // ring of the newest 16 quantised samples (the reference keeps 32 + 40; only 10 are ever read)
    i32 LF_AR_Q12, Seed, Seed2, SeedInit2, RD_Q10;
    i8 Q_Q0[SX_DD_DELAY];
};
#define SX_DD_WORDS ((int)(sizeof(SxDD) / 4))

struct SxSS {                        // NSQ_sample_struct, NSQ_del_dec.c:56 (live members only)
    i32 RD_Q10, Q_Q0, Q_Q10, Rd_ind_Q10;
    i32 xq_Q14, LF_AR_Q12, sLTP_shp_Q10, LPC_exc_Q16, exc_Q10;
};

struct SxNsqWork {
    SxDD dd[SX_N_TRACKS][SX_DD_STATES];
    SxSS ss[SX_N_TRACKS][SX_DD_STATES][2];
    i32 exc_Q10[SX_DD_STATES][SX_DD_DELAY];      // excitation ring of the CENTRE states (high-band gain reference)
    i32 Gain_ring[SX_DD_DELAY];
    i32 sLTP_Q16[SX_N_TRACKS][2 * SX_FRAME];
    i32 shp[SX_N_TRACKS][2 * SX_FRAME + 8];      // staged sLTP_shp_Q10 of the three tracks (+8: a side track with lag 0 reads one
                                                 // entry past the frame, always 0 in the reference)
    i32 x_sc_Q10[SX_SUBFR];
    i32 LTP_pred[12], LPC_pred[12], n_LTP[12], n_AR[12], n_LF[12], rD[12];
};

// Agora_Silk_RDCx1, NSQ_del_dec.c:559
SX_HD void sx_nsq_rdcx1(i32 RD_prev, SxSS* ss, i32 r_Q10, i32 r_p_Q10, i32 inv_of_delta_Q16, i32 Lambda_Q10, i32 offset_Q10) {
    i32 q1, q2, rd1, rd2, e;
    r_p_Q10 = sx_smulww(inv_of_delta_Q16, r_p_Q10);
    r_Q10 = sx_sub(r_Q10, offset_Q10);
    r_p_Q10 = sx_sub(r_p_Q10, offset_Q10);
    r_Q10 = sx_limit(r_Q10, -(64 << 10), 64 << 10);
    if (r_Q10 < -1536) {
        q1 = sx_shl(sx_rshift_round(r_Q10, 10), 10);
        e = sx_sub(r_p_Q10, q1);
        rd1 = sx_smlabb(sx_mul(sx_neg(sx_add(q1, offset_Q10)), Lambda_Q10), e, e) >> 10;
        q2 = sx_add(q1, 1024);
        e = sx_sub(r_p_Q10, q2);
        rd2 = sx_smlabb(sx_mul(sx_neg(sx_add(q2, offset_Q10)), Lambda_Q10), e, e) >> 10;
    } else if (r_Q10 > 512) {
        q1 = sx_shl(sx_rshift_round(r_Q10, 10), 10);
        e = sx_sub(r_p_Q10, q1);
        rd1 = sx_smlabb(sx_mul(sx_add(q1, offset_Q10), Lambda_Q10), e, e) >> 10;
        q2 = sx_sub(q1, 1024);
        e = sx_sub(r_p_Q10, q2);
        rd2 = sx_smlabb(sx_mul(sx_add(q2, offset_Q10), Lambda_Q10), e, e) >> 10;
    } else {
        q2 = 0;
        e = r_p_Q10;
        rd2 = sx_smlabb(sx_mul(sx_add(q2, offset_Q10), Lambda_Q10), e, e) >> 10;
        q1 = -1024;
        e = sx_sub(r_p_Q10, q1);
        rd1 = sx_smlabb(sx_mul(sx_neg(sx_add(q1, offset_Q10)), Lambda_Q10), e, e) >> 10;
    }
    const int first = rd1 < rd2 ? 0 : 1;        // slot of candidate 1
    SxSS* s1 = &ss[first];
    SxSS* s2 = &ss[1 - first];
    s1->RD_Q10 = sx_add(RD_prev, rd1);
    s2->RD_Q10 = sx_add(RD_prev, rd2);
    s1->Q_Q0 = (i8)(q1 >> 10);
    s2->Q_Q0 = (i8)(q2 >> 10);
    s1->Q_Q10 = sx_add(offset_Q10, q1);
    s2->Q_Q10 = sx_add(offset_Q10, q2);
    s1->Rd_ind_Q10 = rd1;
    s2->Rd_ind_Q10 = rd2;
}

SX_HD i32 sx_nsq_center_rd1(i32 q_Q10, i32 r_temp_Q10, i32 offset_Q10, i32 Lambda_Q10) {
    i32 e = sx_sub(r_temp_Q10, q_Q10);
    i32 a = sx_add(q_Q10, offset_Q10);
    if (q_Q10 < 0) a = sx_neg(a);
    return sx_smlabb(sx_mul(a, Lambda_Q10), e, e) >> 10;
}

// Agora_Silk_CenterRD, NSQ_del_dec.c:1152: choose the best two of the four side-candidate combinations and
// permute the side candidates so that slot s of every track belongs to combination w_s
SX_HD void sx_nsq_center_rd(i32 RD_prev, SxSS* sc, SxSS* s1, SxSS* s2, i32 res_Q10, i32 Lambda_Q10, i32 offset_Q10) {
    i32 qx[4], rdx[4];
    qx[0] = s1[0].Q_Q10 + s2[0].Q_Q10;
    qx[1] = s1[1].Q_Q10 + s2[1].Q_Q10;
    qx[2] = s1[0].Q_Q10 + s2[1].Q_Q10;
    qx[3] = s1[1].Q_Q10 + s2[0].Q_Q10;
    const i32 r_temp = sx_sub(res_Q10, offset_Q10);
    for (int s = 0; s < 4; s++) rdx[s] = sx_nsq_center_rd1(qx[s], r_temp, offset_Q10, Lambda_Q10);
    rdx[0] = sx_add(sx_add(rdx[0], sx_smulww(SX_JOINT_LAMBDA, s1[0].Rd_ind_Q10)), sx_smulww(SX_JOINT_LAMBDA, s2[0].Rd_ind_Q10));
    rdx[1] = sx_add(sx_add(rdx[1], sx_smulww(SX_JOINT_LAMBDA, s1[1].Rd_ind_Q10)), sx_smulww(SX_JOINT_LAMBDA, s2[1].Rd_ind_Q10));
    rdx[2] = sx_add(sx_add(rdx[2], sx_smulww(SX_JOINT_LAMBDA, s1[0].Rd_ind_Q10)), sx_smulww(SX_JOINT_LAMBDA, s2[1].Rd_ind_Q10));
    rdx[3] = sx_add(sx_add(rdx[3], sx_smulww(SX_JOINT_LAMBDA, s1[1].Rd_ind_Q10)), sx_smulww(SX_JOINT_LAMBDA, s2[0].Rd_ind_Q10));
    int w1 = 0;
    i32 m = rdx[0];
    for (int s = 1; s < 4; s++)
        if (rdx[s] < m) { m = rdx[s]; w1 = s; }
    int w2;
    if (w1 == 0) {
        m = rdx[1]; w2 = 1;
        for (int s = 2; s < 4; s++)
            if (rdx[s] < m) { m = rdx[s]; w2 = s; }
    } else {
        m = rdx[0]; w2 = 0;
        for (int s = 1; s < 4; s++)
            if (rdx[s] < m && s != w1) { m = rdx[s]; w2 = s; }
    }
    sc[0].RD_Q10 = sx_add(RD_prev, rdx[w1]);
    sc[1].RD_Q10 = sx_add(RD_prev, rdx[w2]);
    sc[0].Q_Q0 = qx[w1] >> 10;
    sc[1].Q_Q0 = qx[w2] >> 10;
    sc[0].Q_Q10 = qx[w1];
    sc[1].Q_Q10 = qx[w2];
    sc[0].Rd_ind_Q10 = rdx[w1];
    sc[1].Rd_ind_Q10 = rdx[w2];
    // the reference's 12-way memcpy case table (NSQ_del_dec.c:1266-1336) is this gather
    const int c1a = w1 & 1, c1b = w2 & 1;                             // MD1 member of combination w: {0,1,0,1}
    const int c2a = (w1 == 1 || w1 == 2), c2b = (w2 == 1 || w2 == 2); // MD2 member of combination w: {0,1,1,0}
    SxSS a0 = s1[c1a], a1 = s1[c1b], b0 = s2[c2a], b1 = s2[c2b];
    s1[0] = a0; s1[1] = a1; s2[0] = b0; s2[1] = b1;
}

// emit the decisionDelay-old sample of state `d` of track t (Agora_Silk_GetWinner{,_Side} / flush loops)
SX_HD void sx_nsq_emit(SxEncHist* hist, SxNsqWork* w, int t, int state, int ring_idx, int pos, i8* q, i32* r, int sLTP_idx, bool write_pred) {
    const SxDD* d = &w->dd[t][state];
    if (t == 0) r[pos] = w->exc_Q10[state][ring_idx];
    else q[(t - 1) * SX_FRAME + pos] = d->Q_Q0[ring_idx];
    hist->xq[t][SX_FRAME + pos] = (i16)sx_sat16(sx_rshift_round(sx_smulww(d->Xq_Q10[ring_idx], w->Gain_ring[ring_idx]), 10));
    w->shp[t][SX_FRAME + pos] = d->Shape_Q10[ring_idx];
    if (write_pred) w->sLTP_Q16[t][sLTP_idx] = d->Pred_Q16[ring_idx];
}

// SKP_Silk_NSQ_del_dec, NSQ_del_dec.c:931.  x: prefiltered input (160), q: [2][160] pulses of MD1 / MD2, r: centre excitation Q10 [160]
SX_FN void sx_nsq_del_dec(SxEncState* st, SxEncHist* hist, SxEncCtrl* c, const i16* x, i8* q, i32* r, SxNsqWork* w) {
    SX_IN_LDS(st); SX_IN_LDS(c); SX_IN_LDS(x); SX_IN_LDS(w);
    const int voiced = c->sigtype == 0;
    int lag_t[3] = {st->nsq[0].lagPrev, st->nsq[1].lagPrev, st->nsq[2].lagPrev};
    const i32 offset_Q10 = T_quant_offsets_Q10[c->sigtype * 2 + c->QuantOffsetType];
    int smpl_buf_idx = 0;
    int decisionDelay = sx_min(SX_DD_DELAY, SX_SUBFR);
    if (voiced) {
        for (int k = 0; k < SX_NB_SUBFR; k++) decisionDelay = sx_min(decisionDelay, c->pitchL[k] - SX_LTP_ORDER / 2 - 1);
    } else if (lag_t[0] > 0) {
        decisionDelay = sx_min(decisionDelay, lag_t[0] - SX_LTP_ORDER / 2 - 1);
    }
    const int LSF_interpolation_flag = c->NLSFInterpCoef_Q2 == 4 ? 0 : 1;
    const i32 Lambda_Q10 = c->Lambda_Q10;

    // Agora_Silk_Init_DelDecState (NSQ_del_dec.c:148): every track starts from the same seed
    {
        i32* p = (i32*)&w->dd[0][0];
        SX_PAR(i, 12 * SX_DD_WORDS) p[i] = 0;
        SX_PAR(i, SX_DD_STATES * SX_DD_DELAY) (&w->exc_Q10[0][0])[i] = 0;
        // stage the shaping history: after the previous frame's shift both halves of the reference's buffer hold the same values
        SX_PAR(ti, SX_N_TRACKS * (SX_FRAME + 8)) {
            const int t = ti / (SX_FRAME + 8), i = ti - t * (SX_FRAME + 8);
            const i32 v = i < SX_FRAME ? hist->sLTP_shp_Q10[t][i] : 0;
            if (i < SX_FRAME) w->shp[t][i] = v;
            w->shp[t][SX_FRAME + i] = v;
        }
        wv_sync();
        SX_PAR(tk, 12) {
            const int t = tk >> 2, k = tk & 3;
            SxDD* d = &w->dd[t][k];
            const SxNSQ* n = &st->nsq[t];
            d->Seed = d->Seed2 = d->SeedInit2 = (k + c->Seed) & 3;
            d->LF_AR_Q12 = n->sLF_AR_shp_Q12;
            d->Shape_Q10[0] = w->shp[t][SX_FRAME - 1];
            for (int i = 0; i < SX_LPC_RING; i++) d->sLPC_Q14[i] = n->sLPC_Q14[i];
            for (int i = 0; i < SX_SHAPE_ORDER; i++) d->sAR2_Q14[i] = n->sAR2_Q14[i];
        }
        wv_sync();
    }
    int sLTP_shp_buf_idx = SX_FRAME, sLTP_buf_idx = SX_FRAME;   // identical for all three tracks
    int lpc_pos = SX_LPC_RING;                                  // ring write position (identical for all states)
    int subfr = 0;

    // MD gain split (md_noise_shape_quantizer_del_dec, NSQ_del_dec.c:1401-1417)
    const i32 inv_gain_p1_Q16 = sx_inverse32_varQ(sx_max(c->DeltaGains_Q16, 1), 32);
    const i32 inv_gain_p2_Q16 = 65536 - inv_gain_p1_Q16;
    const i32 DeltaGains_p1_Q16 = sx_inverse32_varQ(sx_max(inv_gain_p1_Q16, 1), 32);
    const i32 DeltaGains_p2_Q16 = sx_inverse32_varQ(sx_max(inv_gain_p2_Q16, 1), 32);
    const i32 inv_of_delta_p1_Q16 = sx_inverse32_varQ(sx_max(DeltaGains_p1_Q16, 1), 32);   // recomputed inside RDCx1
    const i32 inv_of_delta_p2_Q16 = sx_inverse32_varQ(sx_max(DeltaGains_p2_Q16, 1), 32);
    const i32 offset_p1_Q10 = sx_smulww(inv_gain_p1_Q16, offset_Q10);       // _OFFSET_MD_ (SKP_Silk_define.h:41)
    const i32 offset_p2_Q10 = sx_smulww(inv_gain_p2_Q16, offset_Q10);

    for (int k = 0; k < SX_NB_SUBFR; k++) {
        const i16* A_Q12 = c->PredCoef_Q12[(k >> 1) | (1 - LSF_interpolation_flag)];
        const i16* B_Q14 = &c->LTPCoef_Q14[k * SX_LTP_ORDER];
        const i16* AR_shp_Q13 = &c->AR2_Q13[k * SX_SHAPE_ORDER];
        i32 HarmShapeFIRPacked_Q14 = c->HarmShapeGain_Q14[k] >> 2;
        HarmShapeFIRPacked_Q14 |= sx_shl(c->HarmShapeGain_Q14[k] >> 1, 16);
        const i32 Tilt_Q14 = c->Tilt_Q14[k], LF_shp_Q14 = c->LF_shp_Q14[k], Gain_Q16 = c->Gains_Q16[k];
        int rewhite = 0;
        i32 inv_gain_Q16 = sx_inverse32_varQ(sx_max(Gain_Q16, 1), 32);
        inv_gain_Q16 = sx_min(inv_gain_Q16, 32767);
        i32 inv_gain_Q32 = sx_shl(inv_gain_Q16, 16);                    // scale_states, NSQ_del_dec.c:1611-1616
        if (k == 0) inv_gain_Q32 = sx_shl(sx_smulwb(inv_gain_Q32, c->LTP_scale_Q14), 2);
        if (voiced) {
            lag_t[0] = lag_t[1] = lag_t[2] = c->pitchL[k];
            if ((k & (3 - sx_shl(LSF_interpolation_flag, 1))) == 0) {
                if (k == 2) {
                    subfr = 0;
                    // Agora_Silk_DelDec_Rewhitening{,_Side} (NSQ_del_dec.c:315, 400): flush the centre winner's lineage
                    int Winner_ind = 0;
                    i32 RDmin = w->dd[0][0].RD_Q10;
                    for (int i = 1; i < SX_DD_STATES; i++)
                        if (w->dd[0][i].RD_Q10 < RDmin) { RDmin = w->dd[0][i].RD_Q10; Winner_ind = i; }
                    wv_sync();
                    SX_PAR(tk, 12) {
                        const int t = tk >> 2, s = tk & 3;
                        if (s != Winner_ind) w->dd[t][s].RD_Q10 += SX_I32_MAX >> 4;
                    }
                    SX_PAR(ti, 3 * decisionDelay) {
                        const int t = ti / decisionDelay, i = ti - t * decisionDelay;
                        const int ring = (smpl_buf_idx + decisionDelay - 1 - i) & SX_DD_MASK;
                        sx_nsq_emit(hist, w, t, Winner_ind, ring, k * SX_SUBFR - decisionDelay + i, q, r, 0, false);
                    }
                    wv_sync();
                }
                // re-whiten the quantised signal with the new LPC (SKP_Silk_MA_Prediction from a zero state)
                const int lag = lag_t[0];
                const int start_idx = SX_FRAME - lag - SX_LPC - SX_LTP_ORDER / 2;
                const int len = SX_FRAME - start_idx;
                SX_PAR(tn, 3 * len) {
                    const int t = tn / len, n = tn - t * len;
                    const i16* in = &hist->xq[t][start_idx + k * SX_SUBFR];
                    i32 acc = 0;
                    for (int j = 0; j < SX_LPC; j++)
                        if (n - 1 - j >= 0) acc = sx_smlabb(acc, in[n - 1 - j], A_Q12[j]);
                    i32 o = sx_rshift_round(sx_sub(sx_shl((i32)in[n], 12), acc), 12);
                    // the re-whitened sample goes straight into the scaled LTP state (the reference stages it in sLTP[])
                    w->sLTP_Q16[t][start_idx + n] = sx_smulwb(inv_gain_Q32, sx_sat16(o));
                }
                sLTP_buf_idx = SX_FRAME;
                rewhite = 1;
                wv_sync();
            }
        }
        // Agora_Silk_DelDecScale + SKP_Silk_nsq_del_dec_scale_states (NSQ_del_dec.c:1593, 1668)
        SX_PAR(i, SX_SUBFR) w->x_sc_Q10[i] = sx_smulbb(x[k * SX_SUBFR + i], inv_gain_Q16) >> 6;
        {
            const int lag = c->pitchL[k];
            for (int t = 0; t < SX_N_TRACKS; t++) {
                SxNSQ* n = &st->nsq[t];
                if (inv_gain_Q16 != n->prev_inv_gain_Q16) {
                    const i32 gain_adj_Q16 = sx_div32_varQ(inv_gain_Q16, n->prev_inv_gain_Q16, 16);
                    SX_PAR(i, SX_FRAME) {
                        const int j = sLTP_shp_buf_idx - SX_FRAME + i;
                        w->shp[t][j] = sx_smulww(gain_adj_Q16, w->shp[t][j]);
                    }
                    if (!rewhite) {
                        const int m = lag + SX_LTP_ORDER / 2;
                        SX_PAR(i, m) {
                            const int j = sLTP_buf_idx - m + i;
                            w->sLTP_Q16[t][j] = sx_smulww(gain_adj_Q16, w->sLTP_Q16[t][j]);
                        }
                    }
                    // per state: LF_AR, sLPC ring, sAR2[0..16), Pred_Q16[0..32), Shape_Q10[0..32)
                    SX_PAR(si, SX_DD_STATES * 97) {
                        const int s = si / 97, i = si - s * 97;
                        SxDD* d = &w->dd[t][s];
                        i32* p = i < 16 ? &d->sLPC_Q14[i] : (i < 32 ? &d->sAR2_Q14[i - 16] : (i < 64 ? &d->Pred_Q16[i - 32] :
                                 (i < 96 ? &d->Shape_Q10[i - 64] : &d->LF_AR_Q12)));
                        *p = sx_smulww(gain_adj_Q16, *p);
                    }
                }
            }
            wv_sync();
            for (int t = 0; t < SX_N_TRACKS; t++) st->nsq[t].prev_inv_gain_Q16 = inv_gain_Q16;
        }

        // ---- the per-sample trellis (SKP_Silk_md_noise_shape_quantizer_del_dec, NSQ_del_dec.c:1341) ----
        const int odd = subfr & 1;
        const int shp_base = sLTP_shp_buf_idx, pred_base = sLTP_buf_idx;
        for (int i = 0; i < SX_SUBFR; i++) {
            // phase A: predictions, shaping, residual, dither -- one (track, state) per lane
            SX_PAR(tk, 12) {
                const int t = tk >> 2, s = tk & 3;
                SxDD* d = &w->dd[t][s];
                i32 LTP_pred_Q14 = 0;
                if (voiced) {
                    const i32* pl = &w->sLTP_Q16[t][pred_base - lag_t[t] + SX_LTP_ORDER / 2 + i];
                    for (int j = 0; j < SX_LTP_ORDER; j++) LTP_pred_Q14 = sx_smlawb(LTP_pred_Q14, pl[-j], B_Q14[j]);
                }
                i32 n_LTP_Q14 = 0;
                if (lag_t[0] > 0) {          // the reference tests the CENTRE lag for every track (NSQ_del_dec.c:1436-1446)
                    const i32* ps = &w->shp[t][shp_base - lag_t[t] + 1 + i];
                    n_LTP_Q14 = sx_smulwb(sx_add(ps[0], ps[-2]), HarmShapeFIRPacked_Q14);
                    n_LTP_Q14 = sx_smlawt(n_LTP_Q14, ps[-1], HarmShapeFIRPacked_Q14);
                    n_LTP_Q14 = sx_shl(n_LTP_Q14, 6);
                }
                i32 LPC_pred_Q10 = 0;
                for (int j = 0; j < SX_LPC; j++) LPC_pred_Q10 = sx_smlawb(LPC_pred_Q10, d->sLPC_Q14[(lpc_pos - 1 - j) & SX_LPC_MASK], A_Q12[j]);
                // Agora_Silk_STS (Agora_SILK_func.c:85): warped shaping filter, state updated in place
                const i32 warping_Q16 = SX_WARPING_Q16;
                i32 tmp2 = sx_smlawb(d->sLPC_Q14[(lpc_pos - 1) & SX_LPC_MASK], d->sAR2_Q14[0], warping_Q16);
                i32 tmp1 = sx_smlawb(d->sAR2_Q14[0], d->sAR2_Q14[1] - tmp2, warping_Q16);
                d->sAR2_Q14[0] = tmp2;
                i32 n_AR_Q10 = sx_smulwb(tmp2, AR_shp_Q13[0]);
                for (int j = 2; j < SX_SHAPE_ORDER; j += 2) {
                    tmp2 = sx_smlawb(d->sAR2_Q14[j - 1], d->sAR2_Q14[j] - tmp1, warping_Q16);
                    d->sAR2_Q14[j - 1] = tmp1;
                    n_AR_Q10 = sx_smlawb(n_AR_Q10, tmp1, AR_shp_Q13[j - 1]);
                    tmp1 = sx_smlawb(d->sAR2_Q14[j], d->sAR2_Q14[j + 1] - tmp2, warping_Q16);
                    d->sAR2_Q14[j] = tmp2;
                    n_AR_Q10 = sx_smlawb(n_AR_Q10, tmp2, AR_shp_Q13[j]);
                }
                d->sAR2_Q14[SX_SHAPE_ORDER - 1] = tmp1;
                n_AR_Q10 = sx_smlawb(n_AR_Q10, tmp1, AR_shp_Q13[SX_SHAPE_ORDER - 1]);
                n_AR_Q10 = n_AR_Q10 >> 1;
                n_AR_Q10 = sx_smlawb(n_AR_Q10, d->LF_AR_Q12, Tilt_Q14);
                i32 n_LF_Q10 = sx_shl(sx_smulwb(d->Shape_Q10[smpl_buf_idx], LF_shp_Q14), 2);
                n_LF_Q10 = sx_smlawt(n_LF_Q10, d->LF_AR_Q12, LF_shp_Q14);
                // Agora_Silk_DoPred_And_Shap (Agora_SILK_func.c:143)
                i32 tmp = sx_sub(LTP_pred_Q14, n_LTP_Q14) >> 4;
                tmp = sx_add(tmp, LPC_pred_Q10);
                tmp = sx_sub(tmp, n_AR_Q10);
                tmp = sx_sub(tmp, n_LF_Q10);
                i32 r_Q10 = sx_sub(w->x_sc_Q10[i], tmp);
                // Agora_Silk_Dither (NSQ_del_dec.c:520)
                d->Seed2 = sx_rand(d->Seed2);
                d->Seed = sx_rand(d->Seed);
                const i32 dither = d->Seed2 >> 31;
                r_Q10 = (r_Q10 ^ dither) - dither;
                w->LTP_pred[tk] = LTP_pred_Q14;
                w->LPC_pred[tk] = LPC_pred_Q10;
                w->n_LTP[tk] = n_LTP_Q14;
                w->n_AR[tk] = n_AR_Q10;
                w->n_LF[tk] = n_LF_Q10;
                w->rD[tk] = r_Q10;
            }
            wv_sync();
            // phase B: the two candidates of every side state
            SX_PAR(u, 8) {
                const int tk = 4 + u, t = tk >> 2, s = tk & 3;
                const int first = (t == 1) != (odd != 0);      // MD1 takes the p1 share on even subframes, MD2 on odd ones
                const i32 r_md_Q10 = sx_smulww(first ? inv_gain_p1_Q16 : inv_gain_p2_Q16, w->rD[s]);
                sx_nsq_rdcx1(w->dd[t][s].RD_Q10, w->ss[t][s], r_md_Q10, w->rD[tk], first ? inv_of_delta_p1_Q16 : inv_of_delta_p2_Q16,
                             Lambda_Q10, first ? offset_p1_Q10 : offset_p2_Q10);
            }
            wv_sync();
            // phase C: centre candidates = best two combinations of the side candidates
            SX_PAR(s, SX_DD_STATES) {
                sx_nsq_center_rd(w->dd[0][s].RD_Q10, w->ss[0][s], w->ss[1][s], w->ss[2][s], w->rD[s], Lambda_Q10, offset_p1_Q10 + offset_p2_Q10);
            }
            wv_sync();
            // phase D: undo dither, re-apply the side gains, simulate the decoder for both candidates
            SX_PAR(tk, 12) {
                const int t = tk >> 2, s = tk & 3;
                const i32 dither = w->dd[t][s].Seed2 >> 31;
                const int first = (t == 1) != (odd != 0);
                const i32 DG = first ? DeltaGains_p1_Q16 : DeltaGains_p2_Q16;
                for (int j = 0; j < 2; j++) {
                    SxSS* ss = &w->ss[t][s][j];
                    i32 Q = (ss->Q_Q10 ^ dither) - dither;
                    ss->exc_Q10 = Q;
                    if (t != 0) Q = sx_smulww(DG, Q);
                    ss->Q_Q10 = Q;
                    // Agora_Silk_UndoPred_And_Shap (NSQ_del_dec.c:482)
                    const i32 LPC_exc_Q10 = Q + sx_rshift_round(w->LTP_pred[tk], 4);
                    const i32 xq_Q10 = sx_add(LPC_exc_Q10, w->LPC_pred[tk]);
                    const i32 sLF_AR_shp_Q10 = sx_sub(xq_Q10, w->n_AR[tk]);
                    ss->sLTP_shp_Q10 = sx_sub(sLF_AR_shp_Q10, w->n_LF[tk]);
                    ss->LF_AR_Q12 = sx_shl(sLF_AR_shp_Q10, 2);
                    ss->xq_Q14 = sx_shl(xq_Q10, 4);
                    ss->LPC_exc_Q16 = sx_shl(LPC_exc_Q10, 6);
                }
            }
            wv_sync();
            smpl_buf_idx = (smpl_buf_idx - 1) & SX_DD_MASK;
            const int last_smple_idx = (smpl_buf_idx + decisionDelay) & SX_DD_MASK;
            // phase E: Agora_Silk_JudgeWinner (NSQ_del_dec.c:671), wave-uniform
            {
                int Winner_ind = 0;
                i32 RDmin = sx_add(sx_add(w->ss[0][0][0].RD_Q10, sx_smulww(w->ss[1][0][0].RD_Q10, SX_JOINT_LAMBDA)),
                                   sx_smulww(w->ss[2][0][0].RD_Q10, SX_JOINT_LAMBDA));
                for (int s = 1; s < SX_DD_STATES; s++) {
                    i32 j = sx_add(sx_add(w->ss[0][s][0].RD_Q10, sx_smulww(w->ss[1][s][0].RD_Q10, SX_JOINT_LAMBDA)),
                                   sx_smulww(w->ss[2][s][0].RD_Q10, SX_JOINT_LAMBDA));
                    if (j < RDmin) { RDmin = j; Winner_ind = s; }
                }
                const i32 wr0 = w->dd[0][Winner_ind].RandState[last_smple_idx], wr1 = w->dd[1][Winner_ind].RandState[last_smple_idx],
                          wr2 = w->dd[2][Winner_ind].RandState[last_smple_idx];
                int RandSyncCtl = 0;
                i32 rd0[SX_DD_STATES], rd1[SX_DD_STATES];
                for (int s = 0; s < SX_DD_STATES; s++) {
                    rd0[s] = w->ss[0][s][0].RD_Q10;
                    rd1[s] = w->ss[0][s][1].RD_Q10;
                    if (w->dd[0][s].RandState[last_smple_idx] != wr0 || w->dd[1][s].RandState[last_smple_idx] != wr1 ||
                        w->dd[2][s].RandState[last_smple_idx] != wr2) {
                        RandSyncCtl++;
                        rd0[s] = sx_add(rd0[s], SX_I32_MAX >> 4);
                        rd1[s] = sx_add(rd1[s], SX_I32_MAX >> 4);
                    }
                }
                wv_sync();
                for (int s = 0; s < SX_DD_STATES; s++) {
                    w->ss[0][s][0].RD_Q10 = rd0[s];
                    w->ss[0][s][1].RD_Q10 = rd1[s];
                }
                do {
                    i32 RDmax = rd0[0], RDmin2 = rd1[0];
                    int RDmax_ind = 0, RDmin_ind = 0;
                    for (int s = 1; s < SX_DD_STATES; s++) {
                        if (rd0[s] > RDmax) { RDmax = rd0[s]; RDmax_ind = s; }
                        if (rd1[s] < RDmin2) { RDmin2 = rd1[s]; RDmin_ind = s; }
                    }
                    if (RDmin2 < RDmax) {
                        // SKP_Silk_copy_del_dec_state (NSQ_del_dec.c:1668) for the three tracks + sample states
                        wv_sync();
                        if (RDmax_ind != RDmin_ind) {
                            SX_PAR(ti, 3 * SX_DD_WORDS + SX_DD_DELAY) {
                                if (ti < 3 * SX_DD_WORDS) {
                                    const int t = ti / SX_DD_WORDS, j = ti - t * SX_DD_WORDS;
                                    ((i32*)&w->dd[t][RDmax_ind])[j] = ((const i32*)&w->dd[t][RDmin_ind])[j];
                                } else {
                                    w->exc_Q10[RDmax_ind][ti - 3 * SX_DD_WORDS] = w->exc_Q10[RDmin_ind][ti - 3 * SX_DD_WORDS];
                                }
                            }
                        }
                        wv_sync();
                        for (int t = 0; t < SX_N_TRACKS; t++) {
                            SxSS tmp = w->ss[t][RDmin_ind][1];
                            w->ss[t][RDmax_ind][0] = tmp;
                        }
                        rd0[RDmax_ind] = rd1[RDmin_ind];
                        wv_sync();
                    }
                } while (--RandSyncCtl > 0);
            }
            // phase F: Agora_Silk_GetWinner{,_Side} (NSQ_del_dec.c:757, 820): emit the delayed sample
            {
                int Winner_ind = 0;
                i32 RDmin = sx_add(sx_add(w->ss[0][0][0].RD_Q10, sx_smulww(w->ss[1][0][0].RD_Q10, SX_JOINT_LAMBDA)),
                                   sx_smulww(w->ss[2][0][0].RD_Q10, SX_JOINT_LAMBDA));
                for (int s = 1; s < SX_DD_STATES; s++) {
                    i32 j = sx_add(sx_add(w->ss[0][s][0].RD_Q10, sx_smulww(w->ss[1][s][0].RD_Q10, SX_JOINT_LAMBDA)),
                                   sx_smulww(w->ss[2][s][0].RD_Q10, SX_JOINT_LAMBDA));
                    if (j < RDmin) { RDmin = j; Winner_ind = s; }
                }
                if (subfr > 0 || i >= decisionDelay) {
                    SX_PAR(t, SX_N_TRACKS) {
                        sx_nsq_emit(hist, w, t, Winner_ind, last_smple_idx, k * SX_SUBFR + i - decisionDelay, q, r,
                                    pred_base + i - decisionDelay, true);
                    }
                }
                wv_sync();
            }
            // phase G: Agora_Silk_Update_DelDecState (NSQ_del_dec.c:862)
            SX_PAR(tk, 12) {
                const int t = tk >> 2, s = tk & 3;
                SxDD* d = &w->dd[t][s];
                const SxSS* ss = &w->ss[t][s][0];
                d->LF_AR_Q12 = ss->LF_AR_Q12;
                d->sLPC_Q14[lpc_pos & SX_LPC_MASK] = ss->xq_Q14;
                d->Xq_Q10[smpl_buf_idx] = ss->xq_Q14 >> 4;
                d->Q_Q0[smpl_buf_idx] = (i8)ss->Q_Q0;
                d->Pred_Q16[smpl_buf_idx] = ss->LPC_exc_Q16;
                d->Shape_Q10[smpl_buf_idx] = ss->sLTP_shp_Q10;
                d->Seed = sx_add(d->Seed, ss->Q_Q0);
                d->RandState[smpl_buf_idx] = d->Seed;
                d->RD_Q10 = ss->RD_Q10;
                if (t == 0) w->exc_Q10[s][smpl_buf_idx] = ss->exc_Q10;
            }
            w->Gain_ring[smpl_buf_idx] = Gain_Q16;
            lpc_pos++;
            wv_sync();
        }
        sLTP_shp_buf_idx += SX_SUBFR;
        sLTP_buf_idx += SX_SUBFR;
        subfr++;
    }

    // Agora_Silk_DelDec_UpdateState_And_Output{,_Side} (NSQ_del_dec.c:175, 245)
    int Winner_ind = 0;
    {
        i32 RDmin = w->dd[0][0].RD_Q10;
        for (int s = 1; s < SX_DD_STATES; s++)
            if (w->dd[0][s].RD_Q10 < RDmin) { RDmin = w->dd[0][s].RD_Q10; Winner_ind = s; }
    }
    c->Seed = w->dd[0][Winner_ind].SeedInit2;
    SX_PAR(ti, 3 * decisionDelay) {
        const int t = ti / decisionDelay, i = ti - t * decisionDelay;
        const int ring = (smpl_buf_idx + decisionDelay - 1 - i) & SX_DD_MASK;
        sx_nsq_emit(hist, w, t, Winner_ind, ring, SX_FRAME - decisionDelay + i, q, r, 0, false);
    }
    wv_sync();
    for (int t = 0; t < SX_N_TRACKS; t++) {
        SxNSQ* n = &st->nsq[t];
        const SxDD* d = &w->dd[t][Winner_ind];
        for (int i = 0; i < SX_LPC_RING; i++) n->sLPC_Q14[i] = d->sLPC_Q14[(lpc_pos + i) & SX_LPC_MASK];
        for (int i = 0; i < SX_SHAPE_ORDER; i++) n->sAR2_Q14[i] = d->sAR2_Q14[i];
        n->sLF_AR_shp_Q12 = d->LF_AR_Q12;
        n->lagPrev = c->pitchL[SX_NB_SUBFR - 1];
    }
    wv_sync();
    // the current frame becomes the history of the next one
    SX_PAR(ti, 3 * SX_FRAME) {
        const int t = ti / SX_FRAME, i = ti - t * SX_FRAME;
        hist->sLTP_shp_Q10[t][i] = w->shp[t][SX_FRAME + i];
        hist->xq[t][i] = hist->xq[t][SX_FRAME + i];
    }
    wv_sync();
}
