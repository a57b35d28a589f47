// solo_enc_nsq.h -- the SOLO multiple-description noise-shaping quantiser: three coupled delayed-decision
// trellises (centre, MD1, MD2), 4 states each, quantising one 20 ms frame.  Row E6 of SURVEY.md section 8(a).
// Reference: JC1_SDK_SRC_ARM/src/libSATECodec/SKP_Silk_NSQ_del_dec.c:148-1694 and Agora_SILK_func.c:7-160.
//
// Mapping (MI355X): lane tk = 4*track + state owns one (track, state) pair -- 12 active lanes.
//  * The recursive per-state filter memories (16-tap warped shaping state, last 10 quantised samples, LF_AR, seeds,
//    cumulative RD) live in that lane's REGISTERS for the whole frame; candidate samples too.
//  * Lanes talk through wave shuffles (centre residual -> sides, side candidates -> centre, combination choice -> sides)
//    and v_readlane (survivor bookkeeping runs on wave-uniform values).
//  * The 32-deep decision-delay histories are NOT copied when a survivor replaces a state (the reference memcpy's ten
//    rings per track): every (time, slot) cell is stored once in LDS and each state carries a 64-bit "lineage" word
//    (2 bits per ring position = which slot holds its ancestor's sample).  A survivor copy is one 64-bit move plus a
//    register shuffle of the filter memories.
//  * LDS holds the shared rings, the re-whitened LTP state and the staged shaping history of the three tracks.
// The same source compiles for the host (tests/emu, SX_NLANES == 1): lane-private variables become arrays over the 12
// (track, state) pairs and shuffles become array reads.
#pragma once
#include "solo_enc_state.h"

#define SX_JOINT_LAMBDA 90000        // INTERNAL_JOINT_LAMBDA, SKP_Silk_define.h:48 (LARS_LAMBDA_AGR == 0)
#define SX_DD_MASK (SX_DD_DELAY - 1)

#if SX_NLANES == 1
#define SX_NSLOT 12
#define SX_LANES12(tk) for (int tk = 0; tk < 12; tk++)
#define SX_LANESALL(tk) for (int tk = 0; tk < 12; tk++)
#define SX_LI(tk) (tk)
#define SX_XL(arr, src) ((arr)[src])                       // value of a lane-private variable on lane `src`
#define SX_XL2(arr, j, src) ((arr)[src][j])
#define SX_RL(arr, src) ((arr)[src])                       // same, `src` wave-uniform (scalar result)
#define SX_RL2(arr, j, src) ((arr)[src][j])
#else
#define SX_NSLOT 1
#define SX_LANES12(tk) for (int tk = SX_LANE, once_ = 1; once_ && tk < 12; once_ = 0)
#define SX_LANESALL(tk) for (int tk = SX_LANE, once_ = 1; once_; once_ = 0)
#define SX_LI(tk) 0
#define SX_XL(arr, src) __shfl((arr)[0], (src), SX_NLANES)
#define SX_XL2(arr, j, src) __shfl((arr)[0][j], (src), SX_NLANES)
#if SX_NLANES == 64
#define SX_RL(arr, src) __builtin_amdgcn_readlane((arr)[0], (src))
#define SX_RL2(arr, j, src) __builtin_amdgcn_readlane((arr)[0][j], (src))
#else                                                        // several streams per wave: "uniform" = uniform in the 16-lane group
#define SX_RL(arr, src) __shfl((arr)[0], (src), SX_NLANES)
#define SX_RL2(arr, j, src) __shfl((arr)[0][j], (src), SX_NLANES)
#endif
#endif
// Cross-lane moves inside the 12-lane block of one stream.  The lanes of one track form a DPP quad and the three tracks sit
// 4 lanes apart in one 16-lane row, so on the GPU every exchange is ONE data-parallel-primitive VALU move (quad_perm /
// row_shl / row_shr) instead of a trip through the LDS crossbar.  Host emulation: plain array indexing.
//   SX_DN(arr, n, tk)  value of lane tk - n   (side lanes reading their centre lane: n = 4 or 8)
//   SX_UP(arr, n, tk)  value of lane tk + n   (centre lanes reading their side lanes)
//   SX_QX(arr, o, tk)  value of lane tk ^ o inside the quad (o = 1, 2)
//   SX_QB(arr, wv, tk) value of lane `wv` of the own quad (wv uniform in the group)
#if SX_NLANES == 1
#define SX_DN(arr, n, tk) ((tk) >= (n) ? (arr)[(tk) - (n)] : 0)
#define SX_UP(arr, n, tk) ((tk) + (n) < 12 ? (arr)[(tk) + (n)] : 0)
#define SX_QX(arr, o, tk) ((arr)[(tk) ^ (o)])
#define SX_QB(arr, wv, tk) ((arr)[((tk) & ~3) | (wv)])
#else
#define SX_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xf, 0xf, true)
#define SX_DN(arr, n, tk) SX_DPP((arr)[0], 0x110 | (n))          // row_shr:n
#define SX_UP(arr, n, tk) SX_DPP((arr)[0], 0x100 | (n))          // row_shl:n
#define SX_QX(arr, o, tk) ((o) == 1 ? SX_DPP((arr)[0], 0xB1) : SX_DPP((arr)[0], 0x4E))   // quad_perm [1,0,3,2] / [2,3,0,1]
#define SX_QB(arr, wv, tk) ((wv) == 0 ? SX_DPP((arr)[0], 0x00) : ((wv) == 1 ? SX_DPP((arr)[0], 0x55) : ((wv) == 2 ? SX_DPP((arr)[0], 0xAA) : SX_DPP((arr)[0], 0xFF))))
#endif
//   SX_QG(arr, sl, tk) value of lane `sl` of the own quad, `sl` any per-lane value (lane-indexed gather)
#if SX_NLANES == 1
#define SX_QG(arr, sl, tk) ((arr)[((tk) & ~3) | (sl)])
#else
#define SX_QG(arr, sl, tk) __shfl((arr)[0], (SX_LANE & ~3) | (sl), SX_NLANES)
#endif
// a value that all twelve lanes hold identically, as a (group-)uniform scalar for control flow
#if SX_NLANES == 1
#define SX_GRP(arr) ((arr)[0])
#elif SX_NLANES == 64
#define SX_GRP(arr) __builtin_amdgcn_readfirstlane((arr)[0])
#else
#define SX_GRP(arr) ((arr)[0])
#endif
// group-uniform copy of a value held by the four centre lanes (after a quad butterfly) to all twelve lanes
// (lanes 12..15 of a 16-lane group take part, so that values steering the group's control flow are defined in all its lanes)
#define SX_FROM_CENTRE(dst, arr, tk) { const i32 d4_ = SX_DN(arr, 4, tk), d8_ = SX_DN(arr, 8, tk), d12_ = SX_DN(arr, 12, tk); \
                                       dst = (tk) < 4 ? (arr)[SX_LI(tk)] : ((tk) < 8 ? d4_ : ((tk) < 12 ? d8_ : d12_)); }

#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64
#define SX_UNIFORM(v) __builtin_amdgcn_readfirstlane(v)      // wave-uniform value -> scalar register
#else
#define SX_UNIFORM(v) (v)
#endif

// phase timer of the quantiser (debug builds with -DSX_PROF): per-lane register accumulators, flushed once per frame
#if defined(SX_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define SX_TA_BEGIN unsigned long long ta_acc_[14] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long ta_last_ = __builtin_readcyclecounter();
#define SX_TA(id) { const unsigned long long t_ = __builtin_readcyclecounter(); ta_acc_[id] += t_ - ta_last_; ta_last_ = t_; }
#define SX_TA_END if (SX_LANE == 0) { for (int q_ = 0; q_ < 14; q_++) atomicAdd(&g_sx_prof[q_], ta_acc_[q_]); }
#define SX_TA_COUNT(id, n) ta_acc_[id] += (n);
#else
#define SX_TA_BEGIN
#define SX_TA(id)
#define SX_TA_COUNT(id, n)
#define SX_TA_END
#endif

struct SxRing {                      // decision-delay histories of one track: one cell per (ring position, state slot).
    i32 Rand[SX_DD_DELAY][SX_DD_STATES];         // (the histories that are only read at emission live in HBM: SxNsqRingG)
    i32 Shape_Q10[SX_DD_DELAY][SX_DD_STATES];
    i8 Q_Q0[SX_DD_DELAY][SX_DD_STATES];
};

struct SxNsqWork {
    SxRing ring[SX_N_TRACKS];
    i32 Gain_ring[SX_DD_DELAY];
    i16 x[SX_FRAME];                             // prefiltered input of the frame (staged from the hand-over record)
    i32 ebS[SX_N_TRACKS][SX_SUBFR], ebL[SX_N_TRACKS][SX_SUBFR];   // shaping / prediction samples emitted in the current subframe
};


SX_HD u64 sx_sel4u(u64 a0, u64 a1, u64 a2, u64 a3, int i) { return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3)); }
SX_HD i32 sx_sel4(i32 a0, i32 a1, i32 a2, i32 a3, int i) { return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3)); }

// Agora_Silk_RDCx1, NSQ_del_dec.c:559: the two quantisation candidates of one side state
SX_HD void sx_nsq_rdcx1(i32 RD_prev, i32 r_Q10, i32 r_p_Q10, i32 inv_of_delta_Q16, i32 Lambda_Q10, i32 offset_Q10,
                        i32* cRD, i32* cQ0, i32* cQ10, i32* cRdInd) {
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
    const bool first = rd1 < rd2;              // candidate 1 takes slot 0
    cRD[0] = sx_add(RD_prev, first ? rd1 : rd2);
    cRD[1] = sx_add(RD_prev, first ? rd2 : rd1);
    cQ0[0] = (i8)((first ? q1 : q2) >> 10);
    cQ0[1] = (i8)((first ? q2 : q1) >> 10);
    cQ10[0] = sx_add(offset_Q10, first ? q1 : q2);
    cQ10[1] = sx_add(offset_Q10, first ? q2 : q1);
    cRdInd[0] = first ? rd1 : rd2;
    cRdInd[1] = first ? rd2 : rd1;
}

SX_HD i32 sx_nsq_center_rd1(i32 q_Q10, i32 r_temp_Q10, i32 offset_Q10, i32 Lambda_Q10) {
    i32 e = sx_sub(r_temp_Q10, q_Q10);
    i32 a = sx_add(q_Q10, offset_Q10);
    if (q_Q10 < 0) a = sx_neg(a);
    return sx_smlabb(sx_mul(a, Lambda_Q10), e, e) >> 10;
}

// SKP_Silk_NSQ_del_dec, NSQ_del_dec.c:931.  x: prefiltered input (160), q: [2][160] pulses of MD1 / MD2, r: centre excitation Q10 [160]
SX_FN void sx_nsq_del_dec(SxNsqPersist* P, const SxNsqIn* c, SxNsqOut* out, SxNsqWork* w) {
    SX_IN_LDS(w);
    SxNsqGlobal* g = &P->g;
    SxNsqRingG* rgG = &P->rg;
    const i16* x = w->x;
    i8* q = &out->q[0][0];
    i32* r = out->r;
    SX_TA_BEGIN
    const int voiced = c->sigtype == 0;
    int lagC = P->nsq[0].lagPrev, lagP1 = P->nsq[1].lagPrev, lagP2 = P->nsq[2].lagPrev;
    const i32 offset_Q10 = T_quant_offsets_Q10[c->sigtype * 2 + c->QuantOffsetType];
    int smpl_buf_idx = 0;
    int decisionDelay = sx_min(SX_DD_DELAY, SX_SUBFR);
    if (voiced) {
        for (int k = 0; k < SX_NB_SUBFR; k++) decisionDelay = sx_min(decisionDelay, c->pitchL[k] - SX_LTP_ORDER / 2 - 1);
    } else if (lagC > 0) {
        decisionDelay = sx_min(decisionDelay, lagC - SX_LTP_ORDER / 2 - 1);
    }
    const int LSF_interpolation_flag = c->NLSFInterpCoef_Q2 == 4 ? 0 : 1;
    const i32 Lambda_Q10 = c->Lambda_Q10;

    // ---- lane-private state of the (track, state) pair (registers on the GPU) ----
    i32 sAR2[SX_NSLOT][SX_SHAPE_ORDER], sLPC[SX_NSLOT][SX_LPC];      // sLPC[0] = newest quantised sample (Q14)
    i32 LF_AR[SX_NSLOT], Seed[SX_NSLOT], Seed2[SX_NSLOT], SeedInit2[SX_NSLOT], RD[SX_NSLOT];
    i32 LTP_pred[SX_NSLOT], LPC_pred[SX_NSLOT], n_AR[SX_NSLOT], n_LF[SX_NSLOT], rD[SX_NSLOT];
    i32 cRD[SX_NSLOT][2], cQ0[SX_NSLOT][2], cQ10[SX_NSLOT][2], cRdInd[SX_NSLOT][2];
    i32 cXq14[SX_NSLOT][2], cLFAR[SX_NSLOT][2], cShp[SX_NSLOT][2], cExc16[SX_NSLOT][2], cExc10[SX_NSLOT][2];
    i32 W1[SX_NSLOT], W2[SX_NSLOT], myRand[SX_NSLOT], emitPred[SX_NSLOT];
    // own-slot cells (HBM rings) of the ring position that is emitted in the current / next sample, fetched a sample ahead
    i32 pfXq[SX_NSLOT], pfPred[SX_NSLOT], pfExc[SX_NSLOT], nxXq[SX_NSLOT], nxPred[SX_NSLOT], nxExc[SX_NSLOT], gXq[SX_NSLOT], gPred[SX_NSLOT], gExc[SX_NSLOT];
    i32 curL[SX_NSLOT][SX_LTP_ORDER], nxL[SX_NSLOT][SX_LTP_ORDER], curS[SX_NSLOT][3], nxS[SX_NSLOT][3];   // LTP / shaping taps, prefetched
    for (int a = 0; a < SX_NSLOT; a++) {       // lanes that own no state keep defined values
        for (int j = 0; j < SX_SHAPE_ORDER; j++) sAR2[a][j] = 0;
        for (int j = 0; j < SX_LPC; j++) sLPC[a][j] = 0;
        LF_AR[a] = Seed[a] = Seed2[a] = SeedInit2[a] = RD[a] = LTP_pred[a] = LPC_pred[a] = n_AR[a] = n_LF[a] = rD[a] = 0;
        W1[a] = W2[a] = myRand[a] = emitPred[a] = 0;
        pfXq[a] = pfPred[a] = pfExc[a] = nxXq[a] = nxPred[a] = nxExc[a] = gXq[a] = gPred[a] = gExc[a] = 0;
        for (int j = 0; j < SX_LTP_ORDER; j++) curL[a][j] = nxL[a][j] = 0;
        for (int j = 0; j < 3; j++) curS[a][j] = nxS[a][j] = 0;
        for (int j = 0; j < 2; j++) cRD[a][j] = cQ0[a][j] = cQ10[a][j] = cRdInd[a][j] = cXq14[a][j] = cLFAR[a][j] = cShp[a][j] = cExc16[a][j] = cExc10[a][j] = 0;
    }
    // lineage word of the lane's state (two 32-bit halves): ring position p lives in slot (lin >> 2p) & 3.  The three tracks
    // of one state index always hold the same word (they are copied together).
    i32 linLo[SX_NSLOT], linHi[SX_NSLOT];
    i32 jv[SX_NSLOT], ji[SX_NSLOT], tv[SX_NSLOT], ti[SX_NSLOT], mis[SX_NSLOT], xq0[SX_NSLOT], xq1[SX_NSLOT], xr0[SX_NSLOT], xr1[SX_NSLOT];
    for (int a = 0; a < SX_NSLOT; a++) { linLo[a] = linHi[a] = jv[a] = ji[a] = tv[a] = ti[a] = mis[a] = xq0[a] = xq1[a] = xr0[a] = xr1[a] = 0; }

    // Agora_Silk_Init_DelDecState (NSQ_del_dec.c:148): every track starts from the same seed
    {
        i32* p = (i32*)&w->ring[0];
        SX_PAR(i, (int)(sizeof(w->ring) / 4)) p[i] = 0;
        SX_PAR(i, (int)(sizeof(SxNsqRingG) / 4)) ((i32*)rgG)[i] = 0;
        SX_PAR(i, SX_FRAME) w->x[i] = c->xfw[i];
        wv_sync();
        SX_LANES12(tk) {
            const int t = tk >> 2, k = tk & 3, li = SX_LI(tk);
            const SxNSQ* n = &P->nsq[t];
            Seed[li] = Seed2[li] = SeedInit2[li] = (k + c->Seed) & 3;
            linLo[li] = linHi[li] = k * 0x55555555;                  // slot k at every ring position
            RD[li] = 0;
            LF_AR[li] = n->sLF_AR_shp_Q12;
            w->ring[t].Shape_Q10[0][k] = g->shp[t][SX_FRAME - 1];
            for (int i = 0; i < SX_LPC; i++) sLPC[li][i] = n->sLPC_Q14[SX_MAX_LPC - 1 - i];
            for (int i = 0; i < SX_SHAPE_ORDER; i++) sAR2[li][i] = n->sAR2_Q14[i];
        }
        wv_sync();
    }
    int sLTP_shp_buf_idx = SX_FRAME, sLTP_buf_idx = SX_FRAME;   // identical for all three tracks
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

#define SX_LIN_SLOT(lo_, hi_, pos_) ((int)((((pos_) < 16 ? (u32)(lo_) : (u32)(hi_)) >> (2 * ((pos_) & 15))) & 3u))
    // emit the decisionDelay-old sample of the lineage of state `win` (Agora_Silk_GetWinner{,_Side} / flush loops)
#define SX_NSQ_EMIT_V(t_, slot_, ring_idx_, pos_, sLTP_idx_, write_pred_, xq_v_, pred_v_, exc_v_)                          \
    {                                                                                                                        \
        const SxRing* rg_ = &w->ring[t_];                                                                                    \
        if ((t_) == 0) r[pos_] = (exc_v_);                                                                                   \
        else q[((t_)-1) * SX_FRAME + (pos_)] = rg_->Q_Q0[ring_idx_][slot_];                                                  \
        P->xq[t_][SX_FRAME + (pos_)] = (i16)sx_sat16(sx_rshift_round(sx_smulww((xq_v_), w->Gain_ring[ring_idx_]), 10));       \
        g->shp[t_][SX_FRAME + (pos_)] = rg_->Shape_Q10[ring_idx_][slot_];                                                    \
        if (write_pred_) { const i32 pv_ = (pred_v_); g->sLTP_Q16[t_][sLTP_idx_] = pv_; emitPred[SX_LI(4 * (t_))] = pv_;     \
                           w->ebL[t_][i] = pv_; w->ebS[t_][i] = rg_->Shape_Q10[ring_idx_][slot_]; }                          \
    }
    // flush form (lane-parallel over ring positions, cells read straight from HBM; callers wv_sync() first)
#define SX_NSQ_EMIT(t_, wlo_, whi_, ring_idx_, pos_, sLTP_idx_, write_pred_)                                                 \
    {                                                                                                                        \
        const int slotf_ = SX_LIN_SLOT(wlo_, whi_, ring_idx_);                                                               \
        SX_NSQ_EMIT_V(t_, slotf_, ring_idx_, pos_, sLTP_idx_, write_pred_, rgG->Xq_Q10[t_][ring_idx_][slotf_],               \
                      rgG->Pred_Q16[t_][ring_idx_][slotf_], rgG->exc_Q10[ring_idx_][slotf_])                                 \
    }

    for (int k = 0; k < SX_NB_SUBFR; k++) {
        const i16* A_Q12 = c->PredCoef_Q12[(k >> 1) | (1 - LSF_interpolation_flag)];
        const i16* B_Q14 = &c->LTPCoef_Q14[k * SX_LTP_ORDER];
        const i16* AR_shp_Q13 = &c->AR2_Q13[k * SX_SHAPE_ORDER];
        i32 HarmShapeFIRPacked_Q14 = c->HarmShapeGain_Q14[k] >> 2;
        HarmShapeFIRPacked_Q14 |= sx_shl(c->HarmShapeGain_Q14[k] >> 1, 16);
        const i32 Tilt_Q14 = c->Tilt_Q14[k], LF_shp_Q14 = c->LF_shp_Q14[k], Gain_Q16 = c->Gains_Q16[k];
        // filter coefficients of the subframe, pre-shifted for the one-instruction (a * (b << 16)) >> 32 form and held in
        // scalar registers for the 40 samples
        i32 Apre[SX_LPC], ARpre[SX_SHAPE_ORDER], Bpre[SX_LTP_ORDER];
        for (int j = 0; j < SX_LPC; j++) Apre[j] = SX_UNIFORM(sx_pre16(A_Q12[j]));
        for (int j = 0; j < SX_SHAPE_ORDER; j++) ARpre[j] = SX_UNIFORM(sx_pre16(AR_shp_Q13[j]));
        for (int j = 0; j < SX_LTP_ORDER; j++) Bpre[j] = SX_UNIFORM(sx_pre16(B_Q14[j]));
        const i32 warp_pre = sx_pre16(SX_WARPING_Q16), Tilt_pre = SX_UNIFORM(sx_pre16(Tilt_Q14));
        const i32 LFb_pre = SX_UNIFORM(sx_pre16(LF_shp_Q14)), LFt_pre = SX_UNIFORM((i32)((u32)LF_shp_Q14 & 0xFFFF0000u));
        const i32 Hb_pre = SX_UNIFORM(sx_pre16(HarmShapeFIRPacked_Q14)), Ht_pre = SX_UNIFORM((i32)((u32)HarmShapeFIRPacked_Q14 & 0xFFFF0000u));
        int rewhite = 0;
        i32 inv_gain_Q16 = sx_inverse32_varQ(sx_max(Gain_Q16, 1), 32);
        inv_gain_Q16 = sx_min(inv_gain_Q16, 32767);
        i32 inv_gain_Q32 = sx_shl(inv_gain_Q16, 16);                    // scale_states, NSQ_del_dec.c:1611-1616
        if (k == 0) inv_gain_Q32 = sx_shl(sx_smulwb(inv_gain_Q32, c->LTP_scale_Q14), 2);
        if (voiced) {
            lagC = lagP1 = lagP2 = c->pitchL[k];
            if ((k & (3 - sx_shl(LSF_interpolation_flag, 1))) == 0) {
                if (k == 2) {
                    subfr = 0;
                    // Agora_Silk_DelDec_Rewhitening{,_Side} (NSQ_del_dec.c:315, 400): flush the centre winner's lineage
                    int Winner_ind = 0;
                    i32 RDmin = SX_RL(RD, 0);
                    for (int i = 1; i < SX_DD_STATES; i++) {
                        const i32 v = SX_RL(RD, i);
                        if (v < RDmin) { RDmin = v; Winner_ind = i; }
                    }
                    SX_LANES12(tk) {
                        if ((tk & 3) != Winner_ind) RD[SX_LI(tk)] += SX_I32_MAX >> 4;
                    }
                    const i32 wlo = SX_RL(linLo, Winner_ind), whi = SX_RL(linHi, Winner_ind);
                    wv_sync();                      // the HBM ring cells of the last samples must have landed
                    SX_PAR(ti, 3 * decisionDelay) {
                        const int t = ti / decisionDelay, i = ti - t * decisionDelay;
                        const int ring = (smpl_buf_idx + decisionDelay - 1 - i) & SX_DD_MASK;
                        SX_NSQ_EMIT(t, wlo, whi, ring, k * SX_SUBFR - decisionDelay + i, 0, false)
                    }
                    wv_sync();
                }
                // re-whiten the quantised signal with the new LPC (SKP_Silk_MA_Prediction from a zero state)
                const int lag = lagC;
                const int start_idx = SX_FRAME - lag - SX_LPC - SX_LTP_ORDER / 2;
                const int len = SX_FRAME - start_idx;
                SX_PAR(tn, 3 * len) {
                    const int t = tn / len, n = tn - t * len;
                    const i16* in = &P->xq[t][start_idx + k * SX_SUBFR];
                    i32 acc = 0;
                    for (int j = 0; j < SX_LPC; j++)
                        if (n - 1 - j >= 0) acc = sx_smlabb(acc, in[n - 1 - j], A_Q12[j]);
                    i32 o = sx_rshift_round(sx_sub(sx_shl((i32)in[n], 12), acc), 12);
                    // the re-whitened sample goes straight into the scaled LTP state (the reference stages it in sLTP[])
                    g->sLTP_Q16[t][start_idx + n] = sx_smulwb(inv_gain_Q32, sx_sat16(o));
                }
                sLTP_buf_idx = SX_FRAME;
                rewhite = 1;
                wv_sync();
            }
        }
        // SKP_Silk_nsq_del_dec_scale_states (NSQ_del_dec.c:1593)
        {
            const int lag = c->pitchL[k];
            bool any = false;
            for (int t = 0; t < SX_N_TRACKS; t++) {
                SxNSQ* n = &P->nsq[t];
                if (inv_gain_Q16 != n->prev_inv_gain_Q16) {
                    any = true;
                    const i32 gain_adj_Q16 = sx_div32_varQ(inv_gain_Q16, n->prev_inv_gain_Q16, 16);
                    SX_PAR(i, SX_FRAME) {
                        const int j = sLTP_shp_buf_idx - SX_FRAME + i;
                        g->shp[t][j] = sx_smulww(gain_adj_Q16, g->shp[t][j]);
                    }
                    if (!rewhite) {
                        const int m = lag + SX_LTP_ORDER / 2;
                        SX_PAR(i, m) {
                            const int j = sLTP_buf_idx - m + i;
                            g->sLTP_Q16[t][j] = sx_smulww(gain_adj_Q16, g->sLTP_Q16[t][j]);
                        }
                    }
                    // every (position, slot) cell of the Pred / Shape histories is scaled once (the reference scales each
                    // state's private copy once)
                    SX_PAR(i, SX_DD_DELAY * SX_DD_STATES) {
                        i32* pp = &rgG->Pred_Q16[t][0][0] + i;
                        i32* ps = &w->ring[t].Shape_Q10[0][0] + i;
                        *pp = sx_smulww(gain_adj_Q16, *pp);
                        *ps = sx_smulww(gain_adj_Q16, *ps);
                    }
                }
            }
            if (any) {
                SX_LANES12(tk) {
                    const int t = tk >> 2, li = SX_LI(tk);
                    const i32 prev = P->nsq[t].prev_inv_gain_Q16;
                    if (inv_gain_Q16 != prev) {
                        const i32 gain_adj_Q16 = sx_div32_varQ(inv_gain_Q16, prev, 16);
                        LF_AR[li] = sx_smulww(gain_adj_Q16, LF_AR[li]);
                        for (int i = 0; i < SX_LPC; i++) sLPC[li][i] = sx_smulww(gain_adj_Q16, sLPC[li][i]);
                        for (int i = 0; i < SX_SHAPE_ORDER; i++) sAR2[li][i] = sx_smulww(gain_adj_Q16, sAR2[li][i]);
                    }
                }
            }
            wv_sync();
            for (int t = 0; t < SX_N_TRACKS; t++) P->nsq[t].prev_inv_gain_Q16 = inv_gain_Q16;
            wv_sync();
        }

        // ---- the per-sample trellis (SKP_Silk_md_noise_shape_quantizer_del_dec, NSQ_del_dec.c:1341) ----
        const int odd = subfr & 1;
        const int shp_base = sLTP_shp_buf_idx, pred_base = sLTP_buf_idx;
        // the long-term prediction / harmonic-shaping taps live in HBM; their addresses are known a sample ahead, so every
        // lane keeps the taps of the current sample in registers and fetches the next sample's while it works
        SX_LANES12(tk) {
            const int t = tk >> 2, li = SX_LI(tk);
            const int lag_me = t == 0 ? lagC : (t == 1 ? lagP1 : lagP2);
            if (voiced) {
                const i32* pl = &g->sLTP_Q16[t][pred_base - lag_me + SX_LTP_ORDER / 2];
                for (int j = 0; j < SX_LTP_ORDER; j++) curL[li][j] = pl[-j];
            }
            if (lagC > 0) {
                const i32* ps = &g->shp[t][shp_base - lag_me + 1];
                curS[li][0] = ps[0]; curS[li][1] = ps[-1]; curS[li][2] = ps[-2];
            }
            {   // ring cells that the first sample of this subframe emits
                const int s = tk & 3, pos = (smpl_buf_idx - 1 + decisionDelay) & SX_DD_MASK;
                pfXq[li] = rgG->Xq_Q10[t][pos][s];
                pfPred[li] = rgG->Pred_Q16[t][pos][s];
                pfExc[li] = rgG->exc_Q10[pos][s];
            }
        }
        // Inside the sample loop the lanes talk through LDS and shuffles only, unless the lag is so short that a tap read from
        // HBM can be an entry emitted earlier in this very subframe: only then must the emit stores be waited for.
        // A tap that was emitted earlier in this very subframe is taken from the LDS copy of the emitted samples (ebS / ebL),
        // every older one from HBM -- so no HBM store ever has to be waited for inside the subframe.
        const int firstS = shp_base - (subfr > 0 ? decisionDelay : 0), firstL = pred_base - (subfr > 0 ? decisionDelay : 0);
        SX_TA(1)
        for (int i = 0; i < SX_SUBFR; i++) {
            // phase A: predictions, shaping, residual, dither -- one (track, state) per lane
            SX_LANES12(tk) {
                const int t = tk >> 2, s = tk & 3, li = SX_LI(tk);
                const int lag_me = t == 0 ? lagC : (t == 1 ? lagP1 : lagP2);
                if (i + 1 < SX_SUBFR) {      // issue the next sample's tap loads now; they land while this sample is processed
                    if (voiced) {
                        const int a0 = pred_base - lag_me + SX_LTP_ORDER / 2 + i + 1;
                        for (int j = 0; j < SX_LTP_ORDER; j++) {
                            const int a = a0 - j, ip = a - (pred_base - decisionDelay);
                            nxL[li][j] = (a >= firstL && ip <= i - 1) ? w->ebL[t][ip] : g->sLTP_Q16[t][a];
                        }
                    }
                    if (lagC > 0) {
                        const int a0 = shp_base - lag_me + 1 + i + 1;
                        for (int j = 0; j < 3; j++) {
                            const int a = a0 - j, ip = a - (shp_base - decisionDelay);
                            nxS[li][j] = (a >= firstS && ip <= i - 1) ? w->ebS[t][ip] : g->shp[t][a];
                        }
                    }
                    {   // own-slot ring cells of the position the NEXT sample emits (written at least 12 samples ago)
                        const int pos = (smpl_buf_idx - 2 + decisionDelay) & SX_DD_MASK;
                        nxXq[li] = rgG->Xq_Q10[t][pos][s];
                        nxPred[li] = rgG->Pred_Q16[t][pos][s];
                        nxExc[li] = rgG->exc_Q10[pos][s];
                    }
                }
                i32 LTP_pred_Q14 = 0;
                if (voiced) {
                    for (int j = 0; j < SX_LTP_ORDER; j++) LTP_pred_Q14 = sx_smlaw_pre(LTP_pred_Q14, curL[li][j], Bpre[j]);
                }
                i32 n_LTP_Q14 = 0;
                if (lagC > 0) {              // the reference tests the CENTRE lag for every track (NSQ_del_dec.c:1436-1446)
                    n_LTP_Q14 = sx_smulw_pre(sx_add(curS[li][0], curS[li][2]), Hb_pre);
                    n_LTP_Q14 = sx_smlaw_pre(n_LTP_Q14, curS[li][1], Ht_pre);
                    n_LTP_Q14 = sx_shl(n_LTP_Q14, 6);
                }
                i32 LPC_pred_Q10 = 0;
                for (int j = 0; j < SX_LPC; j++) LPC_pred_Q10 = sx_smlaw_pre(LPC_pred_Q10, sLPC[li][j], Apre[j]);
                // Agora_Silk_STS (Agora_SILK_func.c:85): warped shaping filter, state updated in place
                i32 tmp2 = sx_smlaw_pre(sLPC[li][0], sAR2[li][0], warp_pre);
                i32 tmp1 = sx_smlaw_pre(sAR2[li][0], sAR2[li][1] - tmp2, warp_pre);
                sAR2[li][0] = tmp2;
                i32 n_AR_Q10 = sx_smulw_pre(tmp2, ARpre[0]);
#pragma unroll
                for (int j = 2; j < SX_SHAPE_ORDER; j += 2) {
                    tmp2 = sx_smlaw_pre(sAR2[li][j - 1], sAR2[li][j] - tmp1, warp_pre);
                    sAR2[li][j - 1] = tmp1;
                    n_AR_Q10 = sx_smlaw_pre(n_AR_Q10, tmp1, ARpre[j - 1]);
                    tmp1 = sx_smlaw_pre(sAR2[li][j], sAR2[li][j + 1] - tmp2, warp_pre);
                    sAR2[li][j] = tmp2;
                    n_AR_Q10 = sx_smlaw_pre(n_AR_Q10, tmp2, ARpre[j]);
                }
                sAR2[li][SX_SHAPE_ORDER - 1] = tmp1;
                n_AR_Q10 = sx_smlaw_pre(n_AR_Q10, tmp1, ARpre[SX_SHAPE_ORDER - 1]);
                n_AR_Q10 = n_AR_Q10 >> 1;
                n_AR_Q10 = sx_smlaw_pre(n_AR_Q10, LF_AR[li], Tilt_pre);
                // newest shaping sample of this state's lineage
                const int slot = SX_LIN_SLOT(linLo[li], linHi[li], smpl_buf_idx);
                i32 n_LF_Q10 = sx_shl(sx_smulw_pre(w->ring[t].Shape_Q10[smpl_buf_idx][slot], LFb_pre), 2);
                n_LF_Q10 = sx_smlaw_pre(n_LF_Q10, LF_AR[li], LFt_pre);
                // Agora_Silk_DelDecScale (NSQ_del_dec.c:1668) + Agora_Silk_DoPred_And_Shap (Agora_SILK_func.c:143)
                const i32 x_sc_Q10 = sx_smulbb(x[k * SX_SUBFR + i], inv_gain_Q16) >> 6;
                i32 tmp = sx_sub(LTP_pred_Q14, n_LTP_Q14) >> 4;
                tmp = sx_add(tmp, LPC_pred_Q10);
                tmp = sx_sub(tmp, n_AR_Q10);
                tmp = sx_sub(tmp, n_LF_Q10);
                i32 r_Q10 = sx_sub(x_sc_Q10, tmp);
                // Agora_Silk_Dither (NSQ_del_dec.c:520)
                Seed2[li] = sx_rand(Seed2[li]);
                Seed[li] = sx_rand(Seed[li]);
                const i32 dither = Seed2[li] >> 31;
                r_Q10 = (r_Q10 ^ dither) - dither;
                LTP_pred[li] = LTP_pred_Q14;
                LPC_pred[li] = LPC_pred_Q10;
                n_AR[li] = n_AR_Q10;
                n_LF[li] = n_LF_Q10;
                rD[li] = r_Q10;
            }
            SX_TA(2)
            // phase B: the two candidates of every side state (the centre residual comes over by shuffle)
            SX_LANES12(tk) {
                const int t = tk >> 2, s = tk & 3, li = SX_LI(tk);
                i32 rC;
                SX_FROM_CENTRE(rC, rD, tk)
                if (t != 0) {
                    const bool first = (t == 1) != (odd != 0);      // MD1 takes the p1 share on even subframes, MD2 on odd ones
                    const i32 r_md_Q10 = sx_smulww(first ? inv_gain_p1_Q16 : inv_gain_p2_Q16, rC);
                    sx_nsq_rdcx1(RD[li], r_md_Q10, rD[li], first ? inv_of_delta_p1_Q16 : inv_of_delta_p2_Q16, Lambda_Q10,
                                 first ? offset_p1_Q10 : offset_p2_Q10, cRD[li], cQ0[li], cQ10[li], cRdInd[li]);
                }
            }
            SX_TA(3)
            // phase C: Agora_Silk_CenterRD (NSQ_del_dec.c:1152): the centre takes the best two of the four combinations of side
            // candidates; the side candidates are then re-ordered so that slot s of every track belongs to combination w_s
            SX_LANES12(tk) {
                const int li = SX_LI(tk);
                xq0[li] = cQ10[li][0]; xq1[li] = cQ10[li][1]; xr0[li] = cRdInd[li][0]; xr1[li] = cRdInd[li][1];
            }
            SX_LANES12(tk) {
                const int t = tk >> 2, s = tk & 3, li = SX_LI(tk);
                const i32 p1q0 = SX_UP(xq0, 4, tk), p1q1 = SX_UP(xq1, 4, tk), p2q0 = SX_UP(xq0, 8, tk), p2q1 = SX_UP(xq1, 8, tk);
                const i32 p1r0 = SX_UP(xr0, 4, tk), p1r1 = SX_UP(xr1, 4, tk), p2r0 = SX_UP(xr0, 8, tk), p2r1 = SX_UP(xr1, 8, tk);
                if (t == 0) {
                    const i32 off = offset_p1_Q10 + offset_p2_Q10;
                    const i32 qx0 = p1q0 + p2q0, qx1 = p1q1 + p2q1, qx2 = p1q0 + p2q1, qx3 = p1q1 + p2q0;
                    const i32 r_temp = sx_sub(rD[li], off);
                    i32 rdx0 = sx_nsq_center_rd1(qx0, r_temp, off, Lambda_Q10), rdx1 = sx_nsq_center_rd1(qx1, r_temp, off, Lambda_Q10);
                    i32 rdx2 = sx_nsq_center_rd1(qx2, r_temp, off, Lambda_Q10), rdx3 = sx_nsq_center_rd1(qx3, r_temp, off, Lambda_Q10);
                    rdx0 = sx_add(sx_add(rdx0, sx_smulww(SX_JOINT_LAMBDA, p1r0)), sx_smulww(SX_JOINT_LAMBDA, p2r0));
                    rdx1 = sx_add(sx_add(rdx1, sx_smulww(SX_JOINT_LAMBDA, p1r1)), sx_smulww(SX_JOINT_LAMBDA, p2r1));
                    rdx2 = sx_add(sx_add(rdx2, sx_smulww(SX_JOINT_LAMBDA, p1r0)), sx_smulww(SX_JOINT_LAMBDA, p2r1));
                    rdx3 = sx_add(sx_add(rdx3, sx_smulww(SX_JOINT_LAMBDA, p1r1)), sx_smulww(SX_JOINT_LAMBDA, p2r0));
                    int w1 = 0;
                    i32 m = rdx0;
                    if (rdx1 < m) { m = rdx1; w1 = 1; }
                    if (rdx2 < m) { m = rdx2; w1 = 2; }
                    if (rdx3 < m) { m = rdx3; w1 = 3; }
                    int w2;
                    if (w1 == 0) {
                        m = rdx1; w2 = 1;
                        if (rdx2 < m) { m = rdx2; w2 = 2; }
                        if (rdx3 < m) { m = rdx3; w2 = 3; }
                    } else {
                        m = rdx0; w2 = 0;
                        if (rdx1 < m && w1 != 1) { m = rdx1; w2 = 1; }
                        if (rdx2 < m && w1 != 2) { m = rdx2; w2 = 2; }
                        if (rdx3 < m && w1 != 3) { m = rdx3; w2 = 3; }
                    }
                    const i32 q_w1 = sx_sel4(qx0, qx1, qx2, qx3, w1), q_w2 = sx_sel4(qx0, qx1, qx2, qx3, w2);
                    const i32 rd_w1 = sx_sel4(rdx0, rdx1, rdx2, rdx3, w1), rd_w2 = sx_sel4(rdx0, rdx1, rdx2, rdx3, w2);
                    cRD[li][0] = sx_add(RD[li], rd_w1);
                    cRD[li][1] = sx_add(RD[li], rd_w2);
                    cQ0[li][0] = q_w1 >> 10;
                    cQ0[li][1] = q_w2 >> 10;
                    cQ10[li][0] = q_w1;
                    cQ10[li][1] = q_w2;
                    W1[li] = w1;
                    W2[li] = w2;
                } else {
                    W1[li] = 0;
                    W2[li] = 1;
                }
            }
            SX_LANES12(tk) {
                const int t = tk >> 2, s = tk & 3, li = SX_LI(tk);
                int w1, w2;
                SX_FROM_CENTRE(w1, W1, tk)
                SX_FROM_CENTRE(w2, W2, tk)
                if (t != 0) {
                    // the reference's 12-way memcpy case table (NSQ_del_dec.c:1266-1336) is this gather;
                    // member of combination w: MD1 {0,1,0,1}, MD2 {0,1,1,0}
                    const bool ca = t == 1 ? (w1 & 1) != 0 : (w1 == 1 || w1 == 2), cb = t == 1 ? (w2 & 1) != 0 : (w2 == 1 || w2 == 2);
                    const i32 a0 = cRD[li][0], a1 = cRD[li][1], b0 = cQ0[li][0], b1 = cQ0[li][1], c0 = cQ10[li][0], c1 = cQ10[li][1];
                    cRD[li][0] = ca ? a1 : a0;  cRD[li][1] = cb ? a1 : a0;
                    cQ0[li][0] = ca ? b1 : b0;  cQ0[li][1] = cb ? b1 : b0;
                    cQ10[li][0] = ca ? c1 : c0; cQ10[li][1] = cb ? c1 : c0;
                }
                // phase D: undo dither, re-apply the side gains, simulate the decoder for both candidates
                const i32 dither = Seed2[li] >> 31;
                const bool first = (t == 1) != (odd != 0);
                const i32 DG = first ? DeltaGains_p1_Q16 : DeltaGains_p2_Q16;
                for (int j = 0; j < 2; j++) {
                    i32 Q = (cQ10[li][j] ^ dither) - dither;
                    cExc10[li][j] = Q;
                    if (t != 0) Q = sx_smulww(DG, Q);
                    // Agora_Silk_UndoPred_And_Shap (NSQ_del_dec.c:482)
                    const i32 LPC_exc_Q10 = Q + sx_rshift_round(LTP_pred[li], 4);
                    const i32 xq_Q10 = sx_add(LPC_exc_Q10, LPC_pred[li]);
                    const i32 sLF_AR_shp_Q10 = sx_sub(xq_Q10, n_AR[li]);
                    cShp[li][j] = sx_sub(sLF_AR_shp_Q10, n_LF[li]);
                    cLFAR[li][j] = sx_shl(sLF_AR_shp_Q10, 2);
                    cXq14[li][j] = sx_shl(xq_Q10, 4);
                    cExc16[li][j] = sx_shl(LPC_exc_Q10, 6);
                }
            }
            SX_TA(4)
            smpl_buf_idx = (smpl_buf_idx - 1) & SX_DD_MASK;
            const int last_smple_idx = (smpl_buf_idx + decisionDelay) & SX_DD_MASK;
            // phase E: Agora_Silk_JudgeWinner (NSQ_del_dec.c:671), lane-parallel: the centre lane of state s holds the joint cost
            // of s; winners / extremes are found by xor-butterflies inside the quad of centre lanes
#define SX_QUAD_ARG(CMP)                                                                                                     \
    for (int o_ = 1; o_ <= 2; o_ <<= 1) {                                                                                    \
        SX_LANES12(tk) { const int li = SX_LI(tk); tv[li] = SX_QX(jv, o_, tk); ti[li] = SX_QX(ji, o_, tk); }                  \
        SX_LANES12(tk) { const int li = SX_LI(tk); if (tv[li] CMP jv[li] || (tv[li] == jv[li] && ti[li] < ji[li])) { jv[li] = tv[li]; ji[li] = ti[li]; } } \
    }
            {
                const i32 PEN = SX_I32_MAX >> 4;
                // joint cost of candidate [0] of every state; the delayed random-state cell of every (track, state)
                SX_LANES12(tk) {
                    const int t = tk >> 2, s = tk & 3, li = SX_LI(tk);
                    xq0[li] = cRD[li][0];
                    myRand[li] = w->ring[t].Rand[last_smple_idx][SX_LIN_SLOT(linLo[li], linHi[li], last_smple_idx)];
                }
                SX_LANES12(tk) {
                    const int s = tk & 3, li = SX_LI(tk);
                    const i32 a = SX_UP(xq0, 4, tk), b = SX_UP(xq0, 8, tk);
                    jv[li] = sx_add(sx_add(xq0[li], sx_smulww(a, SX_JOINT_LAMBDA)), sx_smulww(b, SX_JOINT_LAMBDA));
                    ji[li] = s;
                }
                SX_QUAD_ARG(<)
                int Winner_ind = 0;
                SX_LANESALL(tk) { int wl; SX_FROM_CENTRE(wl, ji, tk) xr0[SX_LI(tk)] = wl; }
                Winner_ind = SX_GRP(xr0);
                // states whose decisionDelay-old ancestor differs from the winner's, in any track, are expired
                SX_LANES12(tk) {
                    const int li = SX_LI(tk);
                    const i32 wr = SX_QB(myRand, Winner_ind, tk);
                    mis[li] = myRand[li] != wr ? 1 : 0;
                }
                SX_LANES12(tk) {
                    const int li = SX_LI(tk);
                    const i32 m = mis[li] | SX_UP(mis, 4, tk) | SX_UP(mis, 8, tk);
                    jv[li] = m;
                    if (tk < 4 && m) { cRD[li][0] = sx_add(cRD[li][0], PEN); cRD[li][1] = sx_add(cRD[li][1], PEN); }
                }
                // number of expired states: quad sum on the centre lanes
                for (int o_ = 1; o_ <= 2; o_ <<= 1) {
                    SX_LANES12(tk) { tv[SX_LI(tk)] = SX_QX(jv, o_, tk); }
                    SX_LANES12(tk) { jv[SX_LI(tk)] += tv[SX_LI(tk)]; }
                }
                SX_LANESALL(tk) { int n_; SX_FROM_CENTRE(n_, jv, tk) xr0[SX_LI(tk)] = n_; }
                int RandSyncCtl = SX_GRP(xr0);
                SX_TA(5)
                SX_TA_COUNT(10, 1)
                do {
                    SX_TA_COUNT(11, 1)
                    // worst candidate [0] (first maximum) and best candidate [1] (first minimum) of the centre track
                    SX_LANES12(tk) { const int li = SX_LI(tk); jv[li] = cRD[li][0]; ji[li] = tk & 3; }
                    SX_QUAD_ARG(>)
                    SX_LANESALL(tk) { i32 a_, b_; SX_FROM_CENTRE(a_, jv, tk) SX_FROM_CENTRE(b_, ji, tk) xq0[SX_LI(tk)] = a_; xq1[SX_LI(tk)] = b_; }
                    const i32 RDmax = SX_GRP(xq0);
                    const int RDmax_ind = SX_GRP(xq1);
                    SX_LANES12(tk) { const int li = SX_LI(tk); jv[li] = cRD[li][1]; ji[li] = tk & 3; }
                    SX_QUAD_ARG(<)
                    SX_LANESALL(tk) { i32 a_, b_; SX_FROM_CENTRE(a_, jv, tk) SX_FROM_CENTRE(b_, ji, tk) xr0[SX_LI(tk)] = a_; xr1[SX_LI(tk)] = b_; }
                    const i32 RDmin2 = SX_GRP(xr0);
                    const int RDmin_ind = SX_GRP(xr1);
                    if (RDmin2 < RDmax) {
                        SX_TA_COUNT(12, 1)
                        if (RDmax_ind != RDmin_ind) { SX_TA_COUNT(13, 1) }
                        // SKP_Silk_copy_del_dec_state (NSQ_del_dec.c:1668) for the three tracks: lineage word + filter memories;
                        // then candidate [RDmax][0] <- candidate [RDmin][1]
#if SX_NLANES == 1
                        for (int t = 0; t < SX_N_TRACKS; t++) {
                            const int d = 4 * t + RDmax_ind, sL = 4 * t + RDmin_ind;
                            if (d != sL) {
                                for (int j = 0; j < SX_SHAPE_ORDER; j++) sAR2[d][j] = sAR2[sL][j];
                                for (int j = 0; j < SX_LPC; j++) sLPC[d][j] = sLPC[sL][j];
                                LF_AR[d] = LF_AR[sL]; Seed[d] = Seed[sL]; Seed2[d] = Seed2[sL]; SeedInit2[d] = SeedInit2[sL]; RD[d] = RD[sL];
                                linLo[d] = linLo[sL]; linHi[d] = linHi[sL];
                            }
                            cRD[d][0] = cRD[sL][1]; cQ0[d][0] = cQ0[sL][1]; cXq14[d][0] = cXq14[sL][1]; cLFAR[d][0] = cLFAR[sL][1];
                            cShp[d][0] = cShp[sL][1]; cExc16[d][0] = cExc16[sL][1]; cExc10[d][0] = cExc10[sL][1];
                        }
#else
                        {
                            // lane RDmax of every quad takes lane RDmin's registers: a lane-indexed permute, so the four streams
                            // of a wave (each with its own RDmax / RDmin) move in the same instructions
                            const int lane = SX_LANE;
                            const bool dst = lane < 12 && (lane & 3) == RDmax_ind;
                            const int src = dst ? ((lane & ~3) | RDmin_ind) : lane;
#define SX_MV(v) { const i32 t_ = __shfl((v), src, SX_NLANES); if (dst) (v) = t_; }
                            if (RDmax_ind != RDmin_ind) {
#pragma unroll
                                for (int j = 0; j < SX_SHAPE_ORDER; j++) SX_MV(sAR2[0][j])
#pragma unroll
                                for (int j = 0; j < SX_LPC; j++) SX_MV(sLPC[0][j])
                                SX_MV(LF_AR[0]) SX_MV(Seed[0]) SX_MV(Seed2[0]) SX_MV(SeedInit2[0]) SX_MV(RD[0]) SX_MV(linLo[0]) SX_MV(linHi[0])
                            }
#define SX_MV01(v) { const i32 t_ = __shfl((v)[0][1], src, SX_NLANES); if (dst) (v)[0][0] = t_; }
                            SX_MV01(cRD) SX_MV01(cQ0) SX_MV01(cXq14) SX_MV01(cLFAR) SX_MV01(cShp) SX_MV01(cExc16) SX_MV01(cExc10)
#undef SX_MV
#undef SX_MV01
                        }
#endif
                    }
                } while (--RandSyncCtl > 0);
                SX_TA(6)
                // phase F: Agora_Silk_GetWinner{,_Side} (NSQ_del_dec.c:757, 820): emit the delayed sample of the joint winner
                SX_LANES12(tk) { xq0[SX_LI(tk)] = cRD[SX_LI(tk)][0]; }
                SX_LANES12(tk) {
                    const int s = tk & 3, li = SX_LI(tk);
                    const i32 a = SX_UP(xq0, 4, tk), b = SX_UP(xq0, 8, tk);
                    jv[li] = sx_add(sx_add(xq0[li], sx_smulww(a, SX_JOINT_LAMBDA)), sx_smulww(b, SX_JOINT_LAMBDA));
                    ji[li] = s;
                }
                SX_QUAD_ARG(<)
                SX_LANESALL(tk) { int wl; SX_FROM_CENTRE(wl, ji, tk) xr0[SX_LI(tk)] = wl; }
                const int Win2 = SX_GRP(xr0);
                if (subfr > 0 || i >= decisionDelay) {
                    // the first lane of every track's quad emits that track; the lineage word of the winner comes by quad broadcast
                    SX_LANES12(tk) { const int li = SX_LI(tk); xq0[li] = SX_QB(linLo, Win2, tk); xq1[li] = SX_QB(linHi, Win2, tk); }
                    // the winner's cell of the emitted position sits in the prefetch registers of the lane that owns its slot
                    SX_LANES12(tk) { const int li = SX_LI(tk); tv[li] = SX_LIN_SLOT(xq0[li], xq1[li], last_smple_idx); }
                    SX_LANES12(tk) {
                        const int li = SX_LI(tk);
                        gXq[li] = SX_QG(pfXq, tv[li], tk); gPred[li] = SX_QG(pfPred, tv[li], tk); gExc[li] = SX_QG(pfExc, tv[li], tk);
                    }
                    SX_LANES12(tk) {
                        if ((tk & 3) == 0) {
                            const int t = tk >> 2, li = SX_LI(tk);
                            SX_NSQ_EMIT_V(t, tv[li], last_smple_idx, k * SX_SUBFR + i - decisionDelay, pred_base + i - decisionDelay, true,
                                          gXq[li], gPred[li], gExc[li])
                        }
                    }
                }
            }
#undef SX_QUAD_ARG
            wv_sync_lds();
            const bool emitted = subfr > 0 || i >= decisionDelay;
            SX_TA(7)
            // phase G: Agora_Silk_Update_DelDecState (NSQ_del_dec.c:862): every state pushes candidate [0] into its own cell
            SX_LANES12(tk) {
                const int t = tk >> 2, s = tk & 3, li = SX_LI(tk);
                SxRing* rg = &w->ring[t];
                LF_AR[li] = cLFAR[li][0];
                for (int j = SX_LPC - 1; j > 0; j--) sLPC[li][j] = sLPC[li][j - 1];
                sLPC[li][0] = cXq14[li][0];
                rgG->Xq_Q10[t][smpl_buf_idx][s] = cXq14[li][0] >> 4;
                rg->Q_Q0[smpl_buf_idx][s] = (i8)cQ0[li][0];
                rgG->Pred_Q16[t][smpl_buf_idx][s] = cExc16[li][0];
                rg->Shape_Q10[smpl_buf_idx][s] = cShp[li][0];
                Seed[li] = sx_add(Seed[li], cQ0[li][0]);
                rg->Rand[smpl_buf_idx][s] = Seed[li];
                RD[li] = cRD[li][0];
                if (t == 0) rgG->exc_Q10[smpl_buf_idx][s] = cExc10[li][0];
            }
            SX_LANES12(tk) {          // the state's own slot now holds its newest ring entry
                const int s = tk & 3, li = SX_LI(tk);
                const u32 m = 3u << (2 * (smpl_buf_idx & 15));
                if (smpl_buf_idx < 16) linLo[li] = (i32)(((u32)linLo[li] & ~m) | (((u32)s * 0x55555555u) & m));
                else linHi[li] = (i32)(((u32)linHi[li] & ~m) | (((u32)s * 0x55555555u) & m));
            }
            w->Gain_ring[smpl_buf_idx] = Gain_Q16;
            // next sample's taps become current; its newest LTP tap is the prediction sample emitted just now when the
            // decision delay is as long as the pitch lag allows (written after the prefetch was issued): forward it
            SX_LANES12(tk) {
                const int t = tk >> 2, li = SX_LI(tk);
                const int lag_me = t == 0 ? lagC : (t == 1 ? lagP1 : lagP2);
                const i32 fw = SX_QB(emitPred, 0, tk);
                for (int j = 0; j < SX_LTP_ORDER; j++) curL[li][j] = nxL[li][j];
                if (voiced && emitted && decisionDelay == lag_me - SX_LTP_ORDER / 2 - 1) curL[li][0] = fw;
                for (int j = 0; j < 3; j++) curS[li][j] = nxS[li][j];
                pfXq[li] = nxXq[li]; pfPred[li] = nxPred[li]; pfExc[li] = nxExc[li];
            }
            wv_sync_lds();
            SX_TA(8)
        }
        sLTP_shp_buf_idx += SX_SUBFR;
        sLTP_buf_idx += SX_SUBFR;
        subfr++;
    }

    SX_TA(1)
    // Agora_Silk_DelDec_UpdateState_And_Output{,_Side} (NSQ_del_dec.c:175, 245)
    int Winner_ind = 0;
    {
        i32 RDmin = SX_RL(RD, 0);
        for (int s = 1; s < SX_DD_STATES; s++) {
            const i32 v = SX_RL(RD, s);
            if (v < RDmin) { RDmin = v; Winner_ind = s; }
        }
    }
    out->Seed = SX_RL(SeedInit2, Winner_ind);
    {
        const i32 wlo = SX_RL(linLo, Winner_ind), whi = SX_RL(linHi, Winner_ind);
        wv_sync();                                  // the HBM ring cells of the last samples must have landed
        SX_PAR(ti, 3 * decisionDelay) {
            const int t = ti / decisionDelay, i = ti - t * decisionDelay;
            const int ring = (smpl_buf_idx + decisionDelay - 1 - i) & SX_DD_MASK;
            SX_NSQ_EMIT(t, wlo, whi, ring, SX_FRAME - decisionDelay + i, 0, false)
        }
    }
    wv_sync();
    SX_LANES12(tk) {
        const int t = tk >> 2, s = tk & 3, li = SX_LI(tk);
        if (s == Winner_ind) {
            SxNSQ* n = &P->nsq[t];
            for (int i = 0; i < SX_MAX_LPC; i++) n->sLPC_Q14[i] = (SX_MAX_LPC - 1 - i) < SX_LPC ? sLPC[li][SX_MAX_LPC - 1 - i] : 0;
            for (int i = 0; i < SX_SHAPE_ORDER; i++) n->sAR2_Q14[i] = sAR2[li][i];
            n->sLF_AR_shp_Q12 = LF_AR[li];
            n->lagPrev = c->pitchL[SX_NB_SUBFR - 1];
        }
    }
    wv_sync();
    // the current frame becomes the history of the next one
    SX_PAR(ti, 3 * SX_FRAME) {
        const int t = ti / SX_FRAME, i = ti - t * SX_FRAME;
        g->shp[t][i] = g->shp[t][SX_FRAME + i];      // (the upper half keeps its values: the reference's memcpy does the same)
        P->xq[t][i] = P->xq[t][SX_FRAME + i];
    }
    wv_sync();
    SX_TA(9)
    SX_TA_END
#undef SX_NSQ_EMIT
}
