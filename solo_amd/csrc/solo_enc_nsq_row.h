// solo_enc_nsq_row.h -- the SOLO multiple-description noise-shaping quantiser: three coupled delayed-decision
// trellises (centre, MD1, MD2), 4 states each, quantising one 20 ms frame.  Row E6 of SURVEY.md section 8(a).
// Reference: JC1_SDK_SRC_ARM/src/libSATECodec/SKP_Silk_NSQ_del_dec.c:148-1694 and Agora_SILK_func.c:7-160.
//
// Mapping (MI355X): ONE LANE = ONE DELAYED-DECISION STATE OF ONE TRACK.  A stream is one 16-lane DPP row: lane = 4 * track + state,
// tracks 0 (centre), 1 (MD1), 2 (MD2); the row's last quad is spare (it runs along on copies of the centre's data and never
// stores).  A 64-lane wavefront quantises FOUR streams; 4096 streams = 1024 wavefronts = one per SIMD of the chip.
//  * The per-sample dependent chain of a wavefront is the work of ONE track: the three tracks' prediction / shaping filters,
//    survivor moves, emissions and updates -- which the reference (and the round-3 kernel: one lane = one state of all three
//    tracks) executes one after the other -- are the same instructions on different lanes.
//  * What the tracks of a state tell each other per sample crosses the row with DPP row shifts (bank-masked, so one instruction
//    delivers a quad's value to another quad): the centre residual that the sides split (1 value), the side candidates the
//    centre combines (8), the centre's choice (1 packed word), the joint costs (sums over the row's quads by row rotations), the
//    expiry flags (or over the quads), the survivor plan (1 packed word).  No LDS exchange, no barrier.
//  * What the states of a track tell each other stays inside a quad: arg-min / arg-max butterflies (DPP quad_perm), quad
//    broadcasts + selects for single values on the decision's critical path, ds_bpermute for the bulk move of the survivors'
//    registers (37 per sample; latency hidden by their number).
//  * The reference's replace-worst-by-best loop (up to three rounds of struct copies, Agora_Silk_JudgeWinner) is first played on
//    three small index registers (parent state / candidate source / candidate number), then every filter register is moved once.
//  * The second candidate's decoder simulation (Agora_Silk_UndoPred_And_Shap) is evaluated after the decision, for the one
//    candidate a lane keeps, instead of for both candidates of every state before it.
//  * The 32-deep decision-delay histories are not copied when a survivor replaces a state: every (time, slot) cell is stored
//    once and each state carries a 64-bit "lineage" word (2 bits per ring position = which slot holds its ancestor's sample).
//    One 16-byte cell per lane and sample (quantised sample, pulse / centre excitation, prediction and shaping history, random
//    state) in an HBM ring laid out [position][lane]: ONE dwordx4 store and ONE dwordx4 prefetch per sample.
//  * The reference rescales all ring cells whenever the subframe gain changes; a cell crosses at most one such boundary before
//    it is emitted (decision delay <= 32 < subframe length), so the factor is applied to the one emitted cell instead.
// The same source compiles for the host (tests/emu, SX_NLANES == 1): lane-private variables become arrays over the twelve live
// lanes of a row and the exchanges become array reads.
#pragma once
#include <stddef.h>
#include "solo_enc_state.h"

#define SX_JOINT_LAMBDA 90000        // INTERNAL_JOINT_LAMBDA, SKP_Silk_define.h:48 (LARS_LAMBDA_AGR == 0)
#define SX_DD_MASK (SX_DD_DELAY - 1)

// ---- the lanes of a stream ---------------------------------------------------------------------------------------------------
// On the GPU one lane is one (track, state) pair (the row layout above); the host emulation walks twelve virtual lanes in loops.  The loop
// over "the lanes of a stream" (RW_FORK), per-(track, state) values in arrays of RW_NL, per-state values in arrays of RW_NS.  (Round 3's
// layout -- one lane = one state carrying its three tracks, sixteen streams per wavefront, a third of the instructions per stream but a
// three times longer chain per sample and ONE such wavefront per compute unit -- was measured slower and is gone: DESIGN_NOTES.md section 9.)
#if SX_NLANES == 1
#define RW_NL 12                                 // per (track, state) values
#define RW_NS 12                                 // per state values (the emulation keeps a copy per virtual lane)
#define RW_FORK(l) for (int l = 0; l < 12; l++)
#define RW_LI(l) (l)
#define RW_SI(l) (l)
#define RW_ONCE(l) 1                             // per-state work: every virtual lane maintains its copy
#define RW_IS_C(l) 1                             // centre-only / side-only work: done by every lane, used where it applies
#define RW_IS_S(l) 1
#else
#define RW_NL 1
#define RW_NS 1
#define RW_FORK(l) for (int l = SX_LANE, once_ = 1; once_; once_ = 0)
#define RW_LI(l) 0
#define RW_SI(l) 0
#define RW_ONCE(l) 1
#define RW_IS_C(l) 1
#define RW_IS_S(l) 1
#endif
#define RW_T(l) ((l) >> 2)                       // track of lane l (3: the spare quad of the GPU's row)
#define RW_K(l) ((l) & 3)                        // delayed-decision state
#define RW_TT(l) (RW_T(l) > 2 ? 0 : RW_T(l))     // track whose data the lane reads (the spare quad of a row shadows the centre)
#define RW_LIVE(l) (RW_T(l) < 3)
// a value every lane of the stream holds identically, as a scalar for the stream's control flow
#define RW_UNI(arr) ((arr)[0])

// Exchanges.  They stand OUTSIDE the RW_FORK loops (host: they walk the lanes themselves).
//  inside a quad (the four states of one track):
//   RWK_GATHER(dst, src, idx)    dst[l] = src[quad(l) + idx[l]]           per-track values, per-state index
//   RWK_PICK(dst, src, idx)      the same for an index that is the same in the four lanes of a quad (cheaper on the GPU)
//   RWS_PICK / RWS_ARGMIN / RWS_ARGMAX / RWS_SUM / RWS_PERM   the same exchanges for per-STATE values (identical in a state's three tracks)
//   RWK_ARGMIN / RWK_ARGMAX(val, mv, mi)   extreme of val over the quad and the LOWEST state index holding it, in every lane
//   RWK_SUM(val, out)
//   RWK_PERM(LV, idx)            LV(l) = LV(quad(l) + idx[l]) for an lvalue macro LV(lane)
//  across the quads of a row (the tracks of one state):
//   RWT_FROM(dst, src, T)        dst[l] = src[4 T + state(l)]                    (every live lane receives track T's value)
//   RWT_TO0(dst, src, T)         the same, but only the centre's lanes receive (the other lanes: undefined)
//   RWT_SUM(dst, src) / RWT_OR   dst[l] = sum / or over the three tracks of src[4 t + state(l)]
#if SX_NLANES == 1
#define RWK_GATHER(dst, src, idx) { i32 o_[12]; for (int q_ = 0; q_ < 12; q_++) o_[q_] = (src)[q_]; for (int q_ = 0; q_ < 12; q_++) (dst)[q_] = o_[(q_ & ~3) | (idx)[q_]]; }
#define RWK_ARG_(val, mv, mi, CMP) { for (int b_ = 0; b_ < 12; b_ += 4) { i32 bv_ = (val)[b_]; int bi_ = 0;                         \
        for (int q_ = 1; q_ < 4; q_++) if ((val)[b_ + q_] CMP bv_) { bv_ = (val)[b_ + q_]; bi_ = q_; }                                \
        for (int q_ = 0; q_ < 4; q_++) { (mv)[b_ + q_] = bv_; (mi)[b_ + q_] = bi_; } } }
#define RWK_ARGMIN(val, mv, mi) RWK_ARG_(val, mv, mi, <)
#define RWK_ARGMAX(val, mv, mi) RWK_ARG_(val, mv, mi, >)
#define RWK_SUM(val, out) { for (int b_ = 0; b_ < 12; b_ += 4) { const i32 s_ = sx_add(sx_add((val)[b_], (val)[b_ + 1]), sx_add((val)[b_ + 2], (val)[b_ + 3])); \
        for (int q_ = 0; q_ < 4; q_++) (out)[b_ + q_] = s_; } }
#define RWK_PERM(LV, idx) { i32 o_[12]; for (int q_ = 0; q_ < 12; q_++) o_[q_] = LV(q_); for (int q_ = 0; q_ < 12; q_++) LV(q_) = o_[(q_ & ~3) | (idx)[q_]]; }
#define RWK_PICK(dst, src, idx) RWK_GATHER(dst, src, idx)
#define RWS_PICK(dst, src, idx) RWK_GATHER(dst, src, idx)
#define RWS_ARGMIN(val, mv, mi) RWK_ARGMIN(val, mv, mi)
#define RWS_ARGMAX(val, mv, mi) RWK_ARGMAX(val, mv, mi)
#define RWS_SUM(val, out) RWK_SUM(val, out)
#define RWS_PERM(LV, idx) RWK_PERM(LV, idx)
#define RWT_FROM(dst, src, T) { i32 o_[4]; for (int q_ = 0; q_ < 4; q_++) o_[q_] = (src)[4 * (T) + q_]; for (int q_ = 0; q_ < 12; q_++) (dst)[q_] = o_[q_ & 3]; }
#define RWT_TO0(dst, src, T) { for (int q_ = 0; q_ < 4; q_++) (dst)[q_] = (src)[4 * (T) + q_]; }
#define RWT_SUM(dst, src) { i32 o_[4]; for (int q_ = 0; q_ < 4; q_++) o_[q_] = sx_add(sx_add((src)[q_], (src)[4 + q_]), (src)[8 + q_]); for (int q_ = 0; q_ < 12; q_++) (dst)[q_] = o_[q_ & 3]; }
#define RWT_OR(dst, src) { i32 o_[4]; for (int q_ = 0; q_ < 4; q_++) o_[q_] = (src)[q_] | (src)[4 + q_] | (src)[8 + q_]; for (int q_ = 0; q_ < 12; q_++) (dst)[q_] = o_[q_ & 3]; }
#else
#define RW_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xf, 0xf, true)
#define RW_SHR(n) (0x110 + (n))      // row_shr:n  lane i <- lane i - n
#define RW_SHL(n) (0x100 + (n))      // row_shl:n  lane i <- lane i + n
#define RW_ROR(n) (0x120 + (n))      // row_ror:n  lane i <- lane (i - n) mod 16
// value of state idx (0..3, may differ from lane to lane) of the own quad.  Two forms: four quad broadcasts + selects (seven VALU
// instructions, no LDS round trip: for the few values on the decision's critical path) and ds_bpermute (one LDS-crossbar
// instruction: for the bulk move of the survivors' registers, where latency is hidden by the number of them)
SX_HD i32 rwk_sel(i32 v, i32 idx) {
    const i32 b0 = RW_DPP(v, 0x00), b1 = RW_DPP(v, 0x55), b2 = RW_DPP(v, 0xAA), b3 = RW_DPP(v, 0xFF);
    // (two levels of two-way selects on the index bits: a chain of ?: on idx == 0 / 1 / 2 is compiled into exec-mask branches)
    const bool o_ = (idx & 1) != 0, h_ = (idx & 2) != 0;
    const i32 lo_ = o_ ? b1 : b0, hi_ = o_ ? b3 : b2;
    return h_ ? hi_ : lo_;
}
SX_HD i32 rwk_from(i32 v, i32 src) { return __builtin_amdgcn_ds_bpermute((int)((((threadIdx.x & ~3u) | (u32)src)) << 2), v); }
// arg-min / arg-max over the quad: the extreme by two DPP steps, then the FIRST state holding it from the wave's ballot of
// "my value is the extreme" (the quad's four bits, lowest set bit) -- the reference's serial scans keep the first extreme
SX_HD i32 rwk_first_(bool eq) {
    const unsigned long long b_ = __builtin_amdgcn_ballot_w64(eq);
    return (i32)__builtin_ctz((u32)(b_ >> (threadIdx.x & 60u)) & 15u);
}
SX_HD i32 rwk_min_(i32 v) { v = sx_min(v, RW_DPP(v, 0xB1)); return sx_min(v, RW_DPP(v, 0x4E)); }
SX_HD i32 rwk_max_(i32 v) { v = sx_max(v, RW_DPP(v, 0xB1)); return sx_max(v, RW_DPP(v, 0x4E)); }
SX_HD i32 rwk_or_(i32 v) { v |= RW_DPP(v, 0xB1); return v | RW_DPP(v, 0x4E); }
SX_HD i32 rwk_sum_(i32 v) { v = sx_add(v, RW_DPP(v, 0xB1)); return sx_add(v, RW_DPP(v, 0x4E)); }
// value of the state a quad-uniform index names: that lane's value or-ed through the quad
SX_HD i32 rwk_pick_(i32 v, i32 idx) { return rwk_or_((i32)(threadIdx.x & 3u) == idx ? v : 0); }
// per-STATE values (one copy per lane in both GPU layouts)
#define RWS_ARGMIN(val, mv, mi) { const i32 m_ = rwk_min_((val)[0]); (mv)[0] = m_; (mi)[0] = rwk_first_((val)[0] == m_); }
#define RWS_ARGMAX(val, mv, mi) { const i32 m_ = rwk_max_((val)[0]); (mv)[0] = m_; (mi)[0] = rwk_first_((val)[0] == m_); }
#define RWS_PICK(dst, src, idx) { (dst)[0] = rwk_pick_((src)[0], (idx)[0]); }
#define RWS_SUM(val, out) { (out)[0] = rwk_sum_((val)[0]); }
#define RWS_PERM(LV, idx) { LV(0) = rwk_from(LV(0), (idx)[0]); }
// ---- a 16-lane row per stream: one (track, state) per lane ----
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RW_CENTRE_SERIAL)
#define RW_ROW_C 1                    // the centre's four combinations: one per quad of the row (phase C of the sample step)
#endif
#define RWK_GATHER(dst, src, idx) { (dst)[0] = rwk_sel((src)[0], (idx)[0]); }
#define RWK_PICK(dst, src, idx) RWS_PICK(dst, src, idx)
#define RWK_PERM(LV, idx) RWS_PERM(LV, idx)
#define RWK_ARGMIN(val, mv, mi) RWS_ARGMIN(val, mv, mi)
#define RWK_ARGMAX(val, mv, mi) RWS_ARGMAX(val, mv, mi)
// (two independent row shifts + two selects on lane constants; three bank-masked moves into one register would each wait for the
// one before: a DPP source written by the previous instruction costs two idle issue slots.  The spare quad keeps its own value.)
SX_HD i32 rwt_from0(i32 v) { const u32 t_ = threadIdx.x & 12u; const i32 a_ = RW_DPP(v, RW_SHR(4)), b_ = RW_DPP(v, RW_SHR(8)); return t_ == 4u ? a_ : (t_ == 8u ? b_ : v); }
SX_HD i32 rwt_from1(i32 v) { const u32 t_ = threadIdx.x & 12u; const i32 a_ = RW_DPP(v, RW_SHL(4)), b_ = RW_DPP(v, RW_SHR(4)); return t_ == 0u ? a_ : (t_ == 8u ? b_ : v); }
SX_HD i32 rwt_from2(i32 v) { const u32 t_ = threadIdx.x & 12u; const i32 a_ = RW_DPP(v, RW_SHL(8)), b_ = RW_DPP(v, RW_SHL(4)); return t_ == 0u ? a_ : (t_ == 4u ? b_ : v); }
#define RWT_FROM(dst, src, T) { (dst)[0] = (T) == 0 ? rwt_from0((src)[0]) : ((T) == 1 ? rwt_from1((src)[0]) : rwt_from2((src)[0])); }
// (only the centre's lanes are written; what the other lanes hold afterwards is undefined)
#define RWT_TO0(dst, src, T) { (dst)[0] = __builtin_amdgcn_mov_dpp((src)[0], RW_SHL(4 * (T)), 0xf, 0x1, false); }
// (the spare quad contributes nothing; every quad of the row, the spare one included, receives the result)
#define RWT_SUM(dst, src) { const i32 v_ = (threadIdx.x & 12u) == 12u ? 0 : (src)[0]; \
        (dst)[0] = sx_add(sx_add(v_, RW_DPP(v_, RW_ROR(4))), sx_add(RW_DPP(v_, RW_ROR(8)), RW_DPP(v_, RW_ROR(12)))); }
#define RWT_OR(dst, src) { const i32 v_ = (threadIdx.x & 12u) == 12u ? 0 : (src)[0]; \
        (dst)[0] = (v_ | RW_DPP(v_, RW_ROR(4))) | (RW_DPP(v_, RW_ROR(8)) | RW_DPP(v_, RW_ROR(12))); }
#endif
#ifndef RW_ROW_C
#define RW_ROW_C 0
#endif

// One cell of the emission ring: what ONE state slot of ONE track wrote at ONE ring position.  16 bytes, written / prefetched as
// one dwordx4 per lane.  The quantised sample is already scaled and saturated with the gain of the subframe that wrote it
// (the reference keeps Xq_Q10 and a ring of gains and combines them when the sample is emitted).
//   w0  bits 0..15: quantised output sample (int16); bits 16..31: X bits 0..15
//   w1  bits 0..25: LPC excitation Q10 (only its low 26 bits reach the prediction history: the reference stores it << 6);
//       bits 26..31: X bits 16..21
//   w2  shaping history sample Q10
//   w3  random state of the slot after this sample
// X = what the track hands to the coder for the sample: the pulse (side tracks) / the excitation Q10 (centre: the high band's gain
// reference; |excitation| < 2^18), as a 22-bit signed number.
struct alignas(16) SxRowCell { i32 w0, w1, w2, w3; };
struct alignas(16) SxRowV4 { i32 x, y, z, w; };              // one 16-byte move
SX_HD i32 rw_cell_x(const SxRowCell& c) { return (i32)((u32)sx_shl(c.w1 >> 26, 16) | ((u32)c.w0 >> 16)); }
SX_HD i32 rw_cell_xq(const SxRowCell& c) { return (i32)(i16)c.w0; }
SX_HD i32 rw_cell_pred_Q16(const SxRowCell& c) { return sx_shl(c.w1, 6); }

static_assert(sizeof(((SxNsqOut*)0)->q[0]) >= SX_FRAME + 3 && offsetof(SxNsqOut, q) % 4 == 0 && sizeof(((SxNsqOut*)0)->q[0]) % 4 == 0,
              "a side track's pulses are emitted with 4-byte stores: three bytes of padding behind every row");
#define SX_TAPL_N (SX_SUBFR + SX_LTP_ORDER - 1)              // history entries the five prediction taps of one subframe can reach
#define SX_TAPS_N (SX_SUBFR + 2)                             // ... the three shaping taps
// layout of the coefficient block: A[SX_LPC] | AR[16] | B[5], warp, Tilt, LF (bottom, top), Harm (bottom, top), Lambda, offset sum, gain
#define RW_CA 0
#define RW_CAR (SX_LPC)
#define RW_CB (RW_CAR + SX_SHAPE_ORDER)
#define RW_CM (RW_CB + SX_LTP_ORDER)
#define RW_NCOEF ((RW_CM + 9 + 3) & ~3)
struct alignas(16) SxRowWorkBody {   // LDS, per stream
    // Tap windows of the current subframe, per track: the history entries the subframe's taps can reach, staged from HBM when the
    // subframe starts; a sample emitted during the subframe is also written to its place in the window.  Tap j of iteration i is
    // then ONE LDS read at a fixed place: tapL[i - j + 4] / tapS[i - j + 2].
    // (one more word per row: where an emission that no tap of this subframe can reach is dumped, so that the store needs no branch)
    // (the two windows of a track side by side: a lane addresses both from one base)
    struct { i32 tapL[SX_TAPL_N + 1];                     // long-term prediction history (sLTP_Q16)
             i32 tapS[SX_TAPS_N + 1]; } win[SX_N_TRACKS]; // shaping history (sLTP_shp_Q10)
    i32 xsc[SX_SUBFR];                                    // the subframe's input, scaled by its inverse gain (Q10)
    // Filter coefficients of the subframe, pre-shifted for the one-instruction (a * (b << 16)) >> 32 form, and the stream's scalars
    // of the sample step; the same for the three tracks and the four states of the stream.  They are READ FROM HERE by every sample
    // (16-byte LDS reads): as registers they would be 40 of every lane's budget, which has to stay at 128 for the analysis kernel's
    // sixteen waves to fit beside the quantiser's four.
    alignas(16) i32 coef[RW_NCOEF];
    alignas(16) i32 mdc[8];                               // the MD gain split of the frame: {inv_gain, inv_of_delta, offset, DeltaGains} x {p1, p2}
    // Gain-adjustment factors of the last eight subframe starts, per track: [0, 4) the previous frame's, [4, 8) this frame's
    // (65536 where the gain did not change), and this frame's pitch lags.  The reference rescales its history arrays at every
    // subframe start (SKP_Silk_nsq_del_dec_scale_states); here the histories in HBM are written ONCE, unscaled, and a history entry
    // receives the factors of the subframe starts that lie between its own subframe and the one that stages it when it is staged
    // into a tap window: no read-modify-write pass over the histories, no memory round trips for it in the subframe prologue.
    i32 gfac[SX_N_TRACKS][2 * SX_NB_SUBFR];
    i32 lagk[SX_NB_SUBFR];
#if SX_NLANES == 1
    SxRowCell ring_emu[SX_DD_DELAY * 12];                 // host emulation: the emission ring of the one stream
#endif
};
// The four streams of a wavefront read the same member of their own record in one LDS instruction (a quad the same word, the three
// tracks of a stream three rows of a window): conflict-free when the twelve words fall into twelve different banks.
constexpr bool rw_stride_ok(int stride, int row) {
    bool used[64] = {};
    for (int s = 0; s < 4; s++)
        for (int t = 0; t < 3; t++) {
            const int b = (s * stride + t * row) % 64;
            if (used[b]) return false;
            used[b] = true;
        }
    return true;
}
constexpr int rw_work_pad_words(int body_words) {
    int pad = 0;
    while ((body_words + pad) % 4 != 0 || !rw_stride_ok(body_words + pad, SX_TAPL_N + SX_TAPS_N + 2)) pad++;
    return pad;
}
#if SX_NLANES == 1
struct alignas(16) SxRowWork : SxRowWorkBody {};
#else
struct alignas(16) SxRowWork : SxRowWorkBody { i32 pad_[rw_work_pad_words((int)(sizeof(SxRowWorkBody) / 4))]; };
static_assert(rw_stride_ok((int)(sizeof(SxRowWork) / 4), SX_TAPL_N + SX_TAPS_N + 2) && sizeof(SxRowWork) % 16 == 0, "LDS stride of the per-stream records");
#endif

// SMULWW(x, INTERNAL_JOINT_LAMBDA) = (x * 90000) >> 16 with 90000 = 65536 + 24464: x + SMULWB(x, 24464), exactly (the first
// part of the product is a multiple of 65536) -- one high-word multiply instead of a 64-bit product
SX_HD i32 sx_mul_lambda(i32 x) { return sx_add(x, sx_smulw_pre(x, (i32)((u32)(SX_JOINT_LAMBDA - 65536) << 16))); }
static_assert(SX_JOINT_LAMBDA - 65536 > 0 && SX_JOINT_LAMBDA - 65536 < 32768, "lambda split");

// Agora_Silk_RDCx1, NSQ_del_dec.c:559: the two quantisation candidates of one side state.  The reference's three cases
// (r < -1.5, r > 0.5, in between) differ in the two levels and in the sign of the rate term; written with selects so that the
// lanes of a wavefront never diverge here.  Outputs: the two candidates' cost increments and quantised values (offset included),
// the cheaper one first.
SX_HD void sx_nsq_rdcx1(i32 r_Q10, i32 r_p_Q10, i32 inv_of_delta_Q16, i32 Lambda_Q10, i32 offset_Q10, i32* cRdInd, i32* cQ10) {
    r_p_Q10 = sx_smulww(inv_of_delta_Q16, r_p_Q10);
    r_Q10 = sx_sub(r_Q10, offset_Q10);
    r_p_Q10 = sx_sub(r_p_Q10, offset_Q10);
    r_Q10 = sx_limit(r_Q10, -(64 << 10), 64 << 10);
    const bool lo = r_Q10 < -1536, hi = r_Q10 > 512;
    const i32 rq = sx_shl(sx_rshift_round(r_Q10, 10), 10);
    const i32 q1 = (lo | hi) ? rq : -1024;
    const i32 q2 = lo ? sx_add(rq, 1024) : (hi ? sx_sub(rq, 1024) : 0);
    const i32 e1 = sx_sub(r_p_Q10, q1), e2 = sx_sub(r_p_Q10, q2);
    const i32 a1 = sx_add(q1, offset_Q10), a2 = sx_add(q2, offset_Q10);
    const i32 rd1 = sx_smlabb(sx_mul(hi ? a1 : sx_neg(a1), Lambda_Q10), e1, e1) >> 10;      // rate term negated unless r > 0.5
    const i32 rd2 = sx_smlabb(sx_mul(lo ? sx_neg(a2) : a2, Lambda_Q10), e2, e2) >> 10;      // rate term negated only if r < -1.5
    const bool first = rd1 < rd2;              // candidate 1 takes slot 0
    cQ10[0] = first ? a1 : a2;
    cQ10[1] = first ? a2 : a1;
    cRdInd[0] = first ? rd1 : rd2;
    cRdInd[1] = first ? rd2 : rd1;
}

SX_HD i32 sx_nsq_center_rd1(i32 q_Q10, i32 r_temp_Q10, i32 offset_Q10, i32 Lambda_Q10) {
    const i32 e = sx_sub(r_temp_Q10, q_Q10);
    const i32 a = sx_add(q_Q10, offset_Q10);
    return sx_smlabb(sx_mul(q_Q10 < 0 ? sx_neg(a) : a, Lambda_Q10), e, e) >> 10;
}

// SX_OPAQUE(x): hides how a value was computed from the optimiser (an empty asm that "modifies" the register).
// RW_MARK(name): with -DRW_MARKS, a scheduling fence + a comment in the assembly (tools/debug/nsq_row_mix.py counts the
// instructions between the marks: the static instruction mix of the sample step, phase by phase)
#if defined(__HIP_DEVICE_COMPILE__) && defined(RW_MARKS)
#define RW_MARK(name) { __builtin_amdgcn_sched_barrier(0); asm volatile("; RW_MARK " name); __builtin_amdgcn_sched_barrier(0); }
#else
#define RW_MARK(name)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SX_OPAQUE(x) asm volatile("" : "+v"(x))
#define SX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define SX_LAMBDA_INLINE __attribute__((always_inline))
#else
#define SX_OPAQUE(x)
#define SX_SCHED_FENCE()
#define SX_LAMBDA_INLINE
#endif

// (the quantiser kernel has exactly one call site: inlined there, so that no callee-saved registers go through scratch)
#if defined(__HIP_DEVICE_COMPILE__) && defined(SX_GROUP)
#define SX_NSQ_FN __device__ __forceinline__
#else
#define SX_NSQ_FN SX_FN
#endif
typedef i32 __attribute__((aligned(1))) rw_i32u;             // a 4-byte store to any byte address
// logical -> physical entry of the two circular histories (solo_enc_state.h, SxNsqTrack); L in [0, 2 SX_FRAME + SX_SUBFR)
SX_HD int rw_phys(int L, int base) { const int p_ = L + base; return p_ >= 4 * SX_FRAME ? p_ - 4 * SX_FRAME : (p_ >= 2 * SX_FRAME ? p_ - 2 * SX_FRAME : p_); }

// SKP_Silk_NSQ_del_dec, NSQ_del_dec.c:931.  c->xfw: prefiltered input; out->q: pulses of MD1 / MD2, out->r: centre excitation Q10.
// Addressing: the stores of the sample loop (ring cells, emitted samples) are written as  wave-uniform base + 32-bit lane offset,
// so that the base stays in scalar registers and no 64-bit per-lane pointers have to be kept alive across the loop:
//   Pu + pOff    the stream's SxNsqPersist,      Ou + oOff   its SxNsqOut of this frame,
//   ringu        the emission ring of the wavefront's streams, cell (position, lane of the row) at index position * rstride + rlane + lane
#define SX_AT(T, ubase, off) (*(T*)((char*)(ubase) + (size_t)(u32)(off)))
SX_NSQ_FN void sx_nsq_del_dec(char* Pu, u32 pOff, const SxNsqIn* c, char* Ou, u32 oOff, SxRowWork* w, SxRowCell* ringu, u32 rlane, int rstride) {
    SX_IN_LDS(w);
    SxNsqPersist* P = (SxNsqPersist*)(Pu + pOff);
    SxNsqOut* out = (SxNsqOut*)(Ou + oOff);
    const int voiced = c->sigtype == 0;
    const i32 offset_Q10 = T_quant_offsets_Q10[c->sigtype * 2 + c->QuantOffsetType];
    int lagC = P->trk[0].s.lagPrev;                            // the centre's lag: governs the decision delay and the shaping taps of all tracks
    const int hbase = P->trk[0].s.histBase;                    // circular histories: logical entry 0 sits at physical hbase
    const int cur0 = SX_FRAME - hbase;                         // ... so this frame's sample `pos` at physical cur0 + pos
    int smpl_buf_idx = 0;
    int decisionDelay = sx_min(SX_DD_DELAY, SX_SUBFR);
    if (voiced) {
        for (int k = 0; k < SX_NB_SUBFR; k++) decisionDelay = sx_min(decisionDelay, c->pitchL[k] - SX_LTP_ORDER / 2 - 1);
    } else if (lagC > 0) {
        decisionDelay = sx_min(decisionDelay, lagC - SX_LTP_ORDER / 2 - 1);
    }
    const int LSF_interpolation_flag = c->NLSFInterpCoef_Q2 == 4 ? 0 : 1;
#define RW_CELL(pos_, lane_) SX_AT(SxRowCell, ringu, ((u32)((pos_) * rstride) + rlane + (u32)(lane_)) * (u32)sizeof(SxRowCell))
    // the sample loop's own ring traffic (every cell is written once and read once, a decision delay later) with the non-temporal
    // cache policy: 32 MB of ring per 4096 streams otherwise sweep everything else -- the stream histories the subframe prologues
    // wait for -- out of the 32 MB of L2
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SX_RING_TEMPORAL)
    typedef int rw_v4i_ __attribute__((ext_vector_type(4)));
#define RW_CELL_LD(dst_, pos_, lane_) { const rw_v4i_ v_ = __builtin_nontemporal_load((const rw_v4i_*)&RW_CELL(pos_, lane_)); \
        (dst_).w0 = v_.x; (dst_).w1 = v_.y; (dst_).w2 = v_.z; (dst_).w3 = v_.w; }
#define RW_CELL_ST(pos_, lane_, src_) { rw_v4i_ v_; v_.x = (src_).w0; v_.y = (src_).w1; v_.z = (src_).w2; v_.w = (src_).w3; \
        __builtin_nontemporal_store(v_, (rw_v4i_*)&RW_CELL(pos_, lane_)); }
#else
#define RW_CELL_LD(dst_, pos_, lane_) (dst_) = RW_CELL(pos_, lane_);
#define RW_CELL_ST(pos_, lane_, src_) RW_CELL(pos_, lane_) = (src_);
#endif

    // ---- lane-private state: one delayed-decision state of one track (registers on the GPU) ----
    i32 sAR2[RW_NL][SX_SHAPE_ORDER], sLPC[RW_NL][SX_LPC];        // sLPC[0] = newest quantised sample (Q14)
    i32 LF_AR[RW_NL], Seed[RW_NL], RD[RW_NL], lastShp[RW_NL];
    i32 Seed2[RW_NS], SeedInit2[RW_NS], linLo[RW_NS], linHi[RW_NS];          // per state: identical in the three tracks' lanes
    i32 lagT[RW_NL], prevInv[RW_NL], gadj[RW_NL];                            // per track
    // byte offsets of the lane's track inside the stream's records: its SxNsqTrack; where its output row starts (centre: 4-byte
    // excitations, sides: 1-byte pulses)
    u32 pTrk[RW_NL], oX[RW_NL];
    // per sample
    i32 LTP_pred[RW_NL], LPC_pred[RW_NL], n_AR[RW_NL], n_LF[RW_NL], rD[RW_NL], rC[RW_NS], dith[RW_NS];
    i32 cInc[RW_NL][2], cQ10[RW_NL][2];                          // the two candidates: cost increment, quantised value
    i32 p1q0[RW_NS], p1q1[RW_NS], p1r0[RW_NS], p1r1[RW_NS], p2q0[RW_NS], p2q1[RW_NS], p2r0[RW_NS], p2r1[RW_NS];
    // The ring is a delay line in HBM; the cell a sample consumes is requested ONE sample before.  Two register sets take turns: an
    // even sample consumes set A (the own-slot cell of the ring position the sample emits) while set B is in flight for the odd
    // sample after it, and vice versa.  The sample loop is written two samples per iteration so that no set is ever copied into
    // another at the loop's back edge (such a copy would wait for the load just issued).
    SxRowCell qA[RW_NL], qB[RW_NL];
    // scratch of the joint decision
    i32 jv[RW_NS], mv[RW_NS], mi[RW_NS], mv2[RW_NS], mi2[RW_NS], tS[RW_NS], tS2[RW_NS], par[RW_NS], csrc[RW_NS], csel[RW_NS], c0[RW_NS], c1[RW_NS], nrep[RW_NS], gq[RW_NS];
    i32 tT[RW_NL], tT2[RW_NL], myRand[RW_NL];                    // per-track scratch
#pragma unroll
    for (int a = 0; a < RW_NS; a++) {
        Seed2[a] = SeedInit2[a] = linLo[a] = linHi[a] = dith[a] = rC[a] = 0;
        jv[a] = mv[a] = mi[a] = mv2[a] = mi2[a] = tS[a] = tS2[a] = par[a] = csrc[a] = csel[a] = c0[a] = c1[a] = nrep[a] = gq[a] = 0;
        p1q0[a] = p1q1[a] = p1r0[a] = p1r1[a] = p2q0[a] = p2q1[a] = p2r0[a] = p2r1[a] = 0;
    }
#pragma unroll
    for (int a = 0; a < RW_NL; a++) {
        lagT[a] = prevInv[a] = tT[a] = tT2[a] = myRand[a] = 0;
        gadj[a] = 65536;
        pTrk[a] = oX[a] = 0u;
#pragma unroll
        for (int j = 0; j < SX_SHAPE_ORDER; j++) sAR2[a][j] = 0;
#pragma unroll
        for (int j = 0; j < SX_LPC; j++) sLPC[a][j] = 0;
        LF_AR[a] = Seed[a] = RD[a] = lastShp[a] = LTP_pred[a] = LPC_pred[a] = n_AR[a] = n_LF[a] = rD[a] = 0;
#pragma unroll
        for (int j = 0; j < 2; j++) cInc[a][j] = cQ10[a][j] = 0;
        qA[a].w0 = qA[a].w1 = qA[a].w2 = qA[a].w3 = 0;
        qB[a] = qA[a];
    }

    // Agora_Silk_Init_DelDecState (NSQ_del_dec.c:148): every track starts from the same seed.  Only the random-state history is
    // ever read before it is written (the expiry test of the first decisionDelay samples of a frame): those reads give zero, see phase E.
    {
        SX_PAR(i, SX_N_TRACKS * SX_NB_SUBFR) {
            const int t = i / SX_NB_SUBFR, kq = i - t * SX_NB_SUBFR;
            w->gfac[t][kq] = P->trk[t].s.gadjPrev[kq];
            w->gfac[t][SX_NB_SUBFR + kq] = 65536;
        }
        SX_PAR(i, SX_NB_SUBFR) w->lagk[i] = c->pitchL[i];
        // MD gain split (md_noise_shape_quantizer_del_dec, NSQ_del_dec.c:1401-1417): frame constants, parked in LDS
        const i32 inv_gain_p1_Q16 = sx_inverse32_varQ(sx_max(c->DeltaGains_Q16, 1), 32);
        const i32 inv_gain_p2_Q16 = 65536 - inv_gain_p1_Q16;
        const i32 DeltaGains_p1_Q16 = sx_inverse32_varQ(sx_max(inv_gain_p1_Q16, 1), 32);
        const i32 DeltaGains_p2_Q16 = sx_inverse32_varQ(sx_max(inv_gain_p2_Q16, 1), 32);
        const i32 inv_of_delta_p1_Q16 = sx_inverse32_varQ(sx_max(DeltaGains_p1_Q16, 1), 32);   // recomputed inside RDCx1
        const i32 inv_of_delta_p2_Q16 = sx_inverse32_varQ(sx_max(DeltaGains_p2_Q16, 1), 32);
        const i32 offset_p1_Q10 = sx_smulww(inv_gain_p1_Q16, offset_Q10);       // _OFFSET_MD_ (SKP_Silk_define.h:41)
        const i32 offset_p2_Q10 = sx_smulww(inv_gain_p2_Q16, offset_Q10);
        RW_FORK(l) {
            if (l == 0) {
                w->mdc[0] = inv_gain_p1_Q16; w->mdc[1] = inv_of_delta_p1_Q16; w->mdc[2] = offset_p1_Q10; w->mdc[3] = DeltaGains_p1_Q16;
                w->mdc[4] = inv_gain_p2_Q16; w->mdc[5] = inv_of_delta_p2_Q16; w->mdc[6] = offset_p2_Q10; w->mdc[7] = DeltaGains_p2_Q16;
                w->coef[RW_CM + 6] = c->Lambda_Q10;
                w->coef[RW_CM + 7] = offset_p1_Q10 + offset_p2_Q10;
            }
        }
        RW_FORK(l) {
            const int li = RW_LI(l), si = RW_SI(l), tt = RW_TT(l), k = RW_K(l), t = RW_T(l);
            pTrk[li] = pOff + (u32)(offsetof(SxNsqPersist, trk) + (size_t)tt * sizeof(SxNsqTrack));
            // (the row's start moved back by this frame's place in the circular histories: the sample step addresses everything with
            // e4 = 4 x (cur0 + sample position))
            oX[li] = t == 0 ? oOff + (u32)offsetof(SxNsqOut, r) - (u32)(4 * cur0) : oOff + (u32)(offsetof(SxNsqOut, q) + (size_t)(tt - 1) * (SX_FRAME + 4)) - (u32)cur0;
            const SxNSQ* n = &SX_AT(SxNSQ, Pu, pTrk[li] + (u32)offsetof(SxNsqTrack, s));
            lagT[li] = n->lagPrev;
            prevInv[li] = n->prev_inv_gain_Q16;
            Seed2[si] = SeedInit2[si] = (k + c->Seed) & 3;
            linLo[si] = linHi[si] = k * 0x55555555;                  // slot k at every ring position
            Seed[li] = (k + c->Seed) & 3;
            RD[li] = 0;
            LF_AR[li] = n->sLF_AR_shp_Q12;
            // the reference seeds ring position 0 of every state with the newest shaping sample of the previous frame
            lastShp[li] = SX_AT(i32, Pu, pTrk[li] + (u32)(offsetof(SxNsqTrack, shp) + (size_t)rw_phys(SX_FRAME - 1, hbase) * 4));
#pragma unroll
            for (int i = 0; i < SX_LPC; i++) sLPC[li][i] = n->sLPC_Q14[SX_MAX_LPC - 1 - i];
#pragma unroll
            for (int i = 0; i < SX_SHAPE_ORDER; i++) sAR2[li][i] = n->sAR2_Q14[i];
        }
        wv_sync();
    }
    RW_FORK(l) {       // prime the ring's read queue: the cell sample 0 looks back at (not written by this frame; never used as such)
        const int l0 = (SX_DD_MASK + decisionDelay) & SX_DD_MASK;
        qA[RW_LI(l)] = RW_CELL(l0, l);
    }
    int sLTP_shp_buf_idx = SX_FRAME, sLTP_buf_idx = SX_FRAME;   // identical for all three tracks
    int subfr = 0;
    int rewhite_k = 0;                                          // the subframe whose start last re-whitened the prediction history

#define RW_LIN_SLOT(lo_, hi_, pos_) ((int)((((pos_) < 16 ? (u32)(lo_) : (u32)(hi_)) >> (2 * ((pos_) & 15))) & 3u))
    // outputs of one emitted sample of the lane's track (Agora_Silk_GetWinner{,_Side} / the flush loops); cell_ = the winner's ring
    // cell, e4_ = 4 x (cur0 + sample position): the coder's value (WIDE: with a 4-byte store whatever the track, see SxNsqOut -- only
    // where samples are emitted in time order, i.e. in the sample loop, not in the flushes), the quantised signal and the shaping
    // history at their place in the circular histories
#define RW_EMIT_OUT(l_, li_, cell_, e4_, WIDE)                                                                               \
    {                                                                                                                        \
        if (WIDE) SX_AT(rw_i32u, Ou, oX[li_] + (RW_T(l_) == 0 ? (u32)(e4_) : (u32)(e4_) >> 2)) = rw_cell_x(cell_);           \
        else if (RW_T(l_) == 0) SX_AT(i32, Ou, oX[li_] + (u32)(e4_)) = rw_cell_x(cell_);                                     \
        else SX_AT(i8, Ou, oX[li_] + ((u32)(e4_) >> 2)) = (i8)rw_cell_x(cell_);                                              \
        SX_AT(i16, Pu + offsetof(SxNsqTrack, xq), pTrk[li_] + ((u32)(e4_) >> 1)) = (i16)rw_cell_xq(cell_);                    \
        SX_AT(i32, Pu + offsetof(SxNsqTrack, shp), pTrk[li_] + (u32)(e4_)) = (cell_).w2;                                      \
    }
    // flush of the winner's lineage (wlo_, whi_), oldest sample at output position pos0_: every lane its own track, the four lanes
    // of a track every fourth sample.  WIDE stays false here: the four lanes of a track write interleaved positions, so a 4-byte
    // store would clobber pulses another lane has already written (the wide form relies on stores in strict time order; the
    // flushes follow the loop's wide stores behind a wv_sync and rewrite whatever those spilled into)
#define RW_FLUSH(wlo_, whi_, pos0_)                                                                                          \
    RW_FORK(l) {                                                                                                             \
        const int li = RW_LI(l);                                                                                             \
        if (RW_LIVE(l)) {                                                                                                    \
            for (int base = RW_K(l); base < decisionDelay; base += 16) {                                                     \
                SxRowCell cl[4];                                                                                             \
                _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                              \
                    const int i = base + 4 * u;                                                                              \
                    const int rp = (smpl_buf_idx + decisionDelay - 1 - i) & SX_DD_MASK;                                      \
                    if (i < decisionDelay) cl[u] = RW_CELL(rp, (l & ~3) | RW_LIN_SLOT(wlo_, whi_, rp));                      \
                }                                                                                                            \
                _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                              \
                    const int i = base + 4 * u;                                                                              \
                    if (i < decisionDelay) RW_EMIT_OUT(l, li, cl[u], 4 * (cur0 + (pos0_) + i), false)                                  \
                }                                                                                                            \
            }                                                                                                                \
        }                                                                                                                    \
    }

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
            lagC = c->pitchL[k];
            RW_FORK(l) { lagT[RW_LI(l)] = lagC; }
            if ((k & (3 - sx_shl(LSF_interpolation_flag, 1))) == 0) {
                if (k == 2) {
                    subfr = 0;
                    // Agora_Silk_DelDec_Rewhitening{,_Side} (NSQ_del_dec.c:315, 400): flush the centre winner's lineage
                    RWT_FROM(jv, RD, 0)
                    RWS_ARGMIN(jv, mv, mi)
                    RW_FORK(l) {
                        if (RW_K(l) != mi[RW_SI(l)]) RD[RW_LI(l)] += SX_I32_MAX >> 4;
                    }
                    RWS_PICK(tS, linLo, mi)
                    RWS_PICK(tS2, linHi, mi)
                    wv_sync();                      // the ring cells of the last samples must have landed
                    RW_FLUSH(tS[RW_SI(l)], tS2[RW_SI(l)], k * SX_SUBFR - decisionDelay)
                    wv_sync();
                }
                // re-whiten the quantised signal with the new LPC (SKP_Silk_MA_Prediction from a zero state)
                const int lag = lagC;
                const int start_idx = SX_FRAME - lag - SX_LPC - SX_LTP_ORDER / 2;
                const int len = SX_FRAME - start_idx;
                // every lane filters a contiguous run of outputs and keeps the last SX_LPC inputs in registers: one load per output
                const int per = (len + SX_NLANES - 1) / SX_NLANES;
                const int n0 = SX_LANE * per, n1 = sx_min(len, n0 + per);
                i32 Ac[SX_LPC];
#pragma unroll
                for (int j = 0; j < SX_LPC; j++) Ac[j] = A_Q12[j];
                for (int t = 0; t < SX_N_TRACKS; t++) {
                    const i16* xq = P->trk[t].xq;                      // circular: logical entry L at rw_phys(L, hbase)
                    const int L0 = start_idx + k * SX_SUBFR;           // logical entry of input 0
                    i32 h[SX_LPC];                                     // h[j] = in[n - 1 - j], zero before the start (zero initial state)
#pragma unroll
                    for (int j = 0; j < SX_LPC; j++) h[j] = (n0 - 1 - j >= 0 && n0 < n1) ? (i32)xq[rw_phys(L0 + n0 - 1 - j, hbase)] : 0;
                    int pp = rw_phys(L0 + n0, hbase);
#pragma unroll 2
                    for (int n = n0; n < n1; n++) {
                        i32 acc = 0;
#pragma unroll
                        for (int j = 0; j < SX_LPC; j++) acc = sx_smlabb(acc, h[j], Ac[j]);
                        const i32 xin = xq[pp];
                        pp = pp + 1 == 2 * SX_FRAME ? 0 : pp + 1;
                        i32 o = sx_rshift_round(sx_sub(sx_shl(xin, 12), acc), 12);
                        // the re-whitened sample goes straight into the scaled LTP state (the reference stages it in sLTP[]) -- and into
                        // this subframe's tap window, which covers [FRAME - lag - 2, FRAME) of it
                        const i32 rw = sx_smulwb(inv_gain_Q32, sx_sat16(o));
                        P->trk[t].sLTP_Q16[start_idx + n] = rw;
                        const int wi = n - SX_LPC;                            // = (start_idx + n) - (FRAME - lag - LTP_ORDER / 2)
                        if ((unsigned)wi < (unsigned)SX_TAPL_N) w->win[t].tapL[wi] = rw;
#pragma unroll
                        for (int j = SX_LPC - 1; j > 0; j--) h[j] = h[j - 1];
                        h[0] = xin;
                    }
                }
                sLTP_buf_idx = SX_FRAME;
                rewhite = 1;
                rewhite_k = k;
            }
        }
        // SKP_Silk_nsq_del_dec_scale_states (NSQ_del_dec.c:1593).  The ring cells are NOT rescaled here: gadj[] is applied to the
        // cells of the previous subframe when (and if) they are emitted.  The histories in HBM are not rescaled either: the factors
        // are recorded and applied when history entries are staged, below.
        RW_FORK(l) {
            const int li = RW_LI(l);
            gadj[li] = 65536;
            if (inv_gain_Q16 != prevInv[li]) {
                const i32 gain_adj_Q16 = sx_div32_varQ(inv_gain_Q16, prevInv[li], 16);
                gadj[li] = gain_adj_Q16;
                LF_AR[li] = sx_smulww(gain_adj_Q16, LF_AR[li]);
                lastShp[li] = sx_smulww(gain_adj_Q16, lastShp[li]);
#pragma unroll
                for (int i = 0; i < SX_LPC; i++) sLPC[li][i] = sx_smulww(gain_adj_Q16, sLPC[li][i]);
#pragma unroll
                for (int i = 0; i < SX_SHAPE_ORDER; i++) sAR2[li][i] = sx_smulww(gain_adj_Q16, sAR2[li][i]);
            }
            prevInv[li] = inv_gain_Q16;
            if (RW_K(l) == 0 && RW_LIVE(l)) w->gfac[RW_T(l)][SX_NB_SUBFR + k] = gadj[li];
        }
        SX_PAR(i, SX_SUBFR) w->xsc[i] = sx_smulbb(c->xfw[k * SX_SUBFR + i], inv_gain_Q16) >> 6;       // Agora_Silk_DelDecScale (NSQ_del_dec.c:1668)
        // filter coefficients of the subframe, pre-shifted for the one-instruction (a * (b << 16)) >> 32 form, into LDS
        // (no harmonic shaping without a pitch lag: the taps are not staged then; the analysis hands over zero prediction taps for
        // an unvoiced frame, which is not relied on)
        SX_PAR(j, SX_LPC) w->coef[RW_CA + j] = sx_pre16(A_Q12[j]);
        SX_PAR(j, SX_SHAPE_ORDER) w->coef[RW_CAR + j] = sx_pre16(AR_shp_Q13[j]);
        SX_PAR(j, SX_LTP_ORDER) w->coef[RW_CB + j] = voiced ? sx_pre16(B_Q14[j]) : 0;
        RW_FORK(l) {
            if (l == 0) {
                w->coef[RW_CM + 0] = sx_pre16(SX_WARPING_Q16);
                w->coef[RW_CM + 1] = sx_pre16(Tilt_Q14);
                w->coef[RW_CM + 2] = sx_pre16(LF_shp_Q14);
                w->coef[RW_CM + 3] = (i32)((u32)LF_shp_Q14 & 0xFFFF0000u);
                w->coef[RW_CM + 4] = lagC > 0 ? sx_pre16(HarmShapeFIRPacked_Q14) : 0;
                w->coef[RW_CM + 5] = lagC > 0 ? (i32)((u32)HarmShapeFIRPacked_Q14 & 0xFFFF0000u) : 0;
                w->coef[RW_CM + 8] = Gain_Q16;
            }
        }
        wv_sync();                                    // factors visible; the emission stores of the previous subframe have landed

        // ---- the per-sample trellis (SKP_Silk_md_noise_shape_quantizer_del_dec, NSQ_del_dec.c:1341) ----
        const int odd = subfr & 1;
        const int shp_base = sLTP_shp_buf_idx, pred_base = sLTP_buf_idx;
        // The long-term prediction / harmonic-shaping histories live in HBM, but the sample loop never reads HBM for them: the entries
        // this subframe's taps can reach are staged in LDS now (the emission stores of the previous subframes have landed: wv_sync
        // above).  Entries that are only emitted during this very subframe are not valid yet: the emission writes them into the
        // window as well (iteration ip emits window entry ip + D + 5 of the prediction history, ip + D + 4 of the shaping history,
        // D = lag - decisionDelay - 3; the first tap to read it belongs to iteration ip + 1 + D or later).
        {
            constexpr int NL = (SX_TAPL_N + SX_NLANES - 1) / SX_NLANES, NS = (SX_TAPS_N + SX_NLANES - 1) / SX_NLANES;
            // A history entry is stored once, unscaled.  What the reference's rescaling passes would have done to it by now is applied
            // here, in the reference's order (every smulww rounds): the factor of subframe start j for every j between the entry's own
            // subframe and this one -- shaping history: the newest SX_FRAME entries are rescaled at every start, so all of them
            // (an entry that a tap can reach is at most four subframes old); prediction history: only the newest lag_j + 2 entries
            // are, and only at starts that did not re-whiten it.
            // (unvoiced frame: the side tracks keep THEIR previous lag -- it differs from the centre's only in the first frames
            // after a reset, where the reference's set-up gives the centre 100 and the sides 0)
            int lagU[SX_N_TRACKS];
#pragma unroll
            for (int t = 0; t < SX_N_TRACKS; t++) lagU[t] = voiced ? lagC : P->trk[t].s.lagPrev;
#pragma unroll
            for (int t = 0; t < SX_N_TRACKS; t++) {               // per track: all loads of the lane first, then the LDS writes
                i32 vl[NL], vs[NS];
                const int iL0 = pred_base - lagU[t] - SX_LTP_ORDER / 2, iS0 = shp_base - lagU[t] - 1;
                const i32* srcL = &P->trk[t].sLTP_Q16[iL0];                                    // tap j of iteration i sits at srcL[i - j + 4]
                const i32* shpT = P->trk[t].shp;                                               // tap j of iteration i: logical entry iS0 + i - j + 2
                const bool stageL = voiced && !rewhite;          // (a re-whitening start has just written its window itself)
#pragma unroll
                for (int u = 0; u < NL; u++) { const int n = SX_LANE + u * SX_NLANES; vl[u] = (stageL && n < SX_TAPL_N) ? srcL[n] : 0; }
#pragma unroll
                for (int u = 0; u < NS; u++) { const int n = SX_LANE + u * SX_NLANES; vs[u] = (lagC > 0 && n < SX_TAPS_N) ? shpT[rw_phys(iS0 + n, hbase)] : 0; }
                if (stageL) {
#pragma unroll
                    for (int u = 0; u < NL; u++) {
                        const int n = SX_LANE + u * SX_NLANES, idx = iL0 + n;
                        const int ep = idx < SX_FRAME ? rewhite_k : rewhite_k + (idx - SX_FRAME) / SX_SUBFR;      // the entry's own subframe
#pragma unroll
                        for (int j = 1; j < SX_NB_SUBFR; j++) {
                            const bool ap = (j > rewhite_k) & (j <= k) & (j > ep) & (idx >= SX_FRAME + SX_SUBFR * (j - rewhite_k) - (w->lagk[j] + SX_LTP_ORDER / 2));
                            const i32 sc = sx_smulww(w->gfac[t][SX_NB_SUBFR + j], vl[u]);
                            vl[u] = ap ? sc : vl[u];
                        }
                    }
                }
                if (lagC > 0) {
#pragma unroll
                    for (int u = 0; u < NS; u++) {
                        const int n = SX_LANE + u * SX_NLANES, idx = iS0 + n;
                        const int r = idx / SX_SUBFR - SX_NB_SUBFR;                    // the entry's subframe relative to this frame: -4 .. 3
#pragma unroll
                        for (int d = SX_NB_SUBFR - 1; d >= 0; d--) {                   // subframe starts k - 3 .. k, in time order
                            const int j = k - d;
                            const i32 sc = sx_smulww(w->gfac[t][SX_NB_SUBFR + j], vs[u]);
                            vs[u] = (j > r) ? sc : vs[u];
                        }
                    }
                }
                if (stageL) {
#pragma unroll
                    for (int u = 0; u < NL; u++) { const int n = SX_LANE + u * SX_NLANES; if (n < SX_TAPL_N) w->win[t].tapL[n] = vl[u]; }
                } else if (!voiced) {
#pragma unroll
                    for (int u = 0; u < NL; u++) { const int n = SX_LANE + u * SX_NLANES; if (n < SX_TAPL_N) w->win[t].tapL[n] = 0; }
                }
#pragma unroll
                for (int u = 0; u < NS; u++) { const int n = SX_LANE + u * SX_NLANES; if (n < SX_TAPS_N) w->win[t].tapS[n] = vs[u]; }
            }
        }
        wv_sync();
        // per-lane constants of the subframe:
        //   sd      the lane's side of the MD gain split (MD1 takes the p1 share on even subframes, MD2 on odd ones; the centre's lanes
        //           compute side candidates nobody reads), as an offset into w->mdc
        //   dL4     4 x (entry of the prediction history - entry of the circular histories) of an emitted sample
        //   wb      window entry of the prediction history that the sample emitted at iteration 0 goes to
        //   lamT    the track's weight in the joint cost beyond 1 (centre: 0, sides: INTERNAL_JOINT_LAMBDA - 1), pre-shifted
        int sd[RW_NL], wb[RW_NL], lamT[RW_NL];
        const int dL4 = 4 * (pred_base - k * SX_SUBFR - cur0);
        RW_FORK(l) {
            const int li = RW_LI(l);
            lamT[li] = RW_T(l) == 0 ? 0 : (i32)((u32)(SX_JOINT_LAMBDA - 65536) << 16);
            sd[li] = ((RW_T(l) == 1) != (odd != 0)) ? 0 : 4;
            wb[li] = lagT[li] - decisionDelay + (SX_LTP_ORDER - SX_LTP_ORDER / 2 - 1);
        }
        // one sample of the trellis; qc: the register set that holds the ring cell this sample consumes, qn: the set that takes the next sample's
        auto sample_step = [&](const int i, SxRowCell (&qc)[RW_NL], SxRowCell (&qn)[RW_NL]) SX_LAMBDA_INLINE {
            const bool emitted = subfr > 0 || i >= decisionDelay;
            const int smpl_new = (smpl_buf_idx - 1) & SX_DD_MASK;                  // ring position this sample writes
            const int last_smple_idx = (smpl_new + decisionDelay) & SX_DD_MASK;    // ring position this sample emits
            // The request for the cell the NEXT sample emits / tests: a whole sample's work lies between the request and its first
            // use, and the request stands in front of this sample's stores (vector memory operations complete in order).  The
            // requested cell was written decisionDelay - 1 >= 12 samples ago.
            RW_MARK("R")
            RW_FORK(l) { if (RW_LIVE(l)) RW_CELL_LD(qn[RW_LI(l)], (last_smple_idx - 1) & SX_DD_MASK, l) }       // (the spare quad: no ring traffic)
            RW_MARK("A")
            // phase A: predictions, shaping, residual, dither of the lane's track and state
            i32 my_inv_gain[RW_NL], my_inv_of_delta[RW_NL], my_offset[RW_NL], my_DG[RW_NL];
            // the subframe's coefficients and the stream's scalars: 16-byte reads through an offset the compiler cannot see
            // through (it would otherwise hoist the loads out of the sample loop and keep the values in registers)
            u32 co_ = 0;
            SX_OPAQUE(co_);
            const SxRowV4* cv = (const SxRowV4*)__builtin_assume_aligned((const char*)w->coef + co_, 16);
            i32 cf[RW_NCOEF];
#define RW_CF_LOAD(j0_, j1_) _Pragma("unroll") for (int j = (j0_); j < (j1_); j++) { const SxRowV4 v_ = cv[j]; cf[4 * j] = v_.x; cf[4 * j + 1] = v_.y; cf[4 * j + 2] = v_.z; cf[4 * j + 3] = v_.w; }
            // (a row per stream: in three groups, each requested one stage before its use: all at once they would be 40 live registers
            // at the point of the sample step where the filter states are live as well)
#ifdef RW_NO_CF_FENCE
#define RW_CF_STAGE(j0_, j1_) RW_CF_LOAD(j0_, j1_)
#else
#define RW_CF_STAGE(j0_, j1_) SX_SCHED_FENCE(); RW_CF_LOAD(j0_, j1_) SX_SCHED_FENCE();
#endif
            const i32 *Apre = cf + RW_CA, *ARpre = cf + RW_CAR, *Bpre = cf + RW_CB;
            constexpr int G1 = (RW_CAR + 3) / 4, G2 = (RW_CB + 3) / 4;      // 16-byte groups that hold A | the rest of AR | the rest
            // (written as a sequence of short steps over "the lanes of the stream": with three tracks per lane the steps of the three
            // tracks then alternate in program order, so that the tracks' dependent chains -- the warped all-pass sections above all --
            // fill each other's issue gaps: a dependent vector instruction issues every ~8 cycles, an independent one every ~5)
            i32 curL[RW_NL][SX_LTP_ORDER], curS[RW_NL][3], tmp1[RW_NL], tmp2[RW_NL], nAR_[RW_NL];
            const i32 x_sc_Q10 = w->xsc[i];
            RW_FORK(l) {
                const int li = RW_LI(l), si = RW_SI(l), tt = RW_TT(l);
                // The taps of this sample (long-term prediction: 5, harmonic shaping: 3): one LDS read each at a fixed place of the
                // staged windows.  (Unvoiced frame: the prediction coefficients are zero; no pitch lag: the shaping gains were
                // zeroed -- whatever the windows hold.)
#pragma unroll
                for (int j = 0; j < SX_LTP_ORDER; j++) curL[li][j] = w->win[tt].tapL[i + (SX_LTP_ORDER - 1) - j];
#pragma unroll
                for (int j = 0; j < 3; j++) curS[li][j] = w->win[tt].tapS[i + 2 - j];
                if (RW_ONCE(l)) {
                    Seed2[si] = sx_rand(Seed2[si]);                                            // Agora_Silk_Dither (NSQ_del_dec.c:520)
                    dith[si] = Seed2[si] >> 31;
                }
                const SxRowV4 v_ = *(const SxRowV4*)__builtin_assume_aligned((const char*)w->mdc + co_ + 4 * sd[li], 16);
                my_inv_gain[li] = v_.x; my_inv_of_delta[li] = v_.y; my_offset[li] = v_.z; my_DG[li] = v_.w;
                LPC_pred[li] = 0;
            }
            RW_CF_STAGE(0, G1)
            RW_CF_STAGE(G1, G2)
#pragma unroll
            for (int j = 0; j < SX_LPC; j++) {
                RW_FORK(l) { const int li = RW_LI(l); LPC_pred[li] = sx_smlaw_pre(LPC_pred[li], sLPC[li][j], Apre[j]); }
            }
            RW_CF_STAGE(G2, RW_NCOEF / 4)
            const i32 warp_pre = cf[RW_CM + 0], Tilt_pre = cf[RW_CM + 1], LFb_pre = cf[RW_CM + 2], LFt_pre = cf[RW_CM + 3], Hb_pre = cf[RW_CM + 4], Ht_pre = cf[RW_CM + 5];
            // Agora_Silk_STS (Agora_SILK_func.c:85): warped shaping filter, state updated in place.  One all-pass section is
            // difference -> multiply -> add, each waiting for the one before; RW_STEP issues one such operation for every track of
            // the lane before the next (three tracks per lane: the order is pinned, the tracks hide each other's result latency)
#define RW_STEP(body_) { RW_FORK(l) { const int li = RW_LI(l); body_ } }
            i32 dd_[RW_NL], mm_[RW_NL];
            RW_STEP(dd_[li] = sx_smulw_pre(sAR2[li][0], warp_pre);)
            RW_STEP(tmp2[li] = sx_add(sLPC[li][0], dd_[li]);)
            RW_STEP(dd_[li] = sx_sub(sAR2[li][1], tmp2[li]); mm_[li] = sx_smulw_pre(tmp2[li], ARpre[0]);)
            RW_STEP(dd_[li] = sx_smulw_pre(dd_[li], warp_pre); nAR_[li] = mm_[li];)
            RW_STEP(tmp1[li] = sx_add(sAR2[li][0], dd_[li]); sAR2[li][0] = tmp2[li];)
#pragma unroll
            for (int j = 2; j < SX_SHAPE_ORDER; j += 2) {
                RW_STEP(dd_[li] = sx_sub(sAR2[li][j], tmp1[li]); mm_[li] = sx_smulw_pre(tmp1[li], ARpre[j - 1]);)
                RW_STEP(dd_[li] = sx_smulw_pre(dd_[li], warp_pre); nAR_[li] = sx_add(nAR_[li], mm_[li]);)
                RW_STEP(tmp2[li] = sx_add(sAR2[li][j - 1], dd_[li]); sAR2[li][j - 1] = tmp1[li];)
                RW_STEP(dd_[li] = sx_sub(sAR2[li][j + 1], tmp2[li]); mm_[li] = sx_smulw_pre(tmp2[li], ARpre[j]);)
                RW_STEP(dd_[li] = sx_smulw_pre(dd_[li], warp_pre); nAR_[li] = sx_add(nAR_[li], mm_[li]);)
                RW_STEP(tmp1[li] = sx_add(sAR2[li][j], dd_[li]); sAR2[li][j] = tmp2[li];)
            }
            RW_FORK(l) {
                const int li = RW_LI(l), si = RW_SI(l);
                const i32 dither = dith[si];
                sAR2[li][SX_SHAPE_ORDER - 1] = tmp1[li];
                i32 n_AR_Q10 = sx_smlaw_pre(nAR_[li], tmp1[li], ARpre[SX_SHAPE_ORDER - 1]);
                n_AR_Q10 = n_AR_Q10 >> 1;
                n_AR_Q10 = sx_smlaw_pre(n_AR_Q10, LF_AR[li], Tilt_pre);
                // newest shaping sample of this state's lineage
                i32 n_LF_Q10 = sx_shl(sx_smulw_pre(lastShp[li], LFb_pre), 2);
                n_LF_Q10 = sx_smlaw_pre(n_LF_Q10, LF_AR[li], LFt_pre);
                n_AR[li] = n_AR_Q10;
                n_LF[li] = n_LF_Q10;
                // long-term prediction and harmonic shaping (taps are zero in an unvoiced frame / without a pitch lag: no branch;
                // the reference tests the CENTRE lag for every track, NSQ_del_dec.c:1436-1446)
                i32 LTP_pred_Q14 = 0;
#pragma unroll
                for (int j = 0; j < SX_LTP_ORDER; j++) LTP_pred_Q14 = sx_smlaw_pre(LTP_pred_Q14, curL[li][j], Bpre[j]);
                i32 n_LTP_Q14 = sx_smulw_pre(sx_add(curS[li][0], curS[li][2]), Hb_pre);
                n_LTP_Q14 = sx_smlaw_pre(n_LTP_Q14, curS[li][1], Ht_pre);
                n_LTP_Q14 = sx_shl(n_LTP_Q14, 6);
                // Agora_Silk_DoPred_And_Shap (Agora_SILK_func.c:143)
                i32 tmp = sx_sub(LTP_pred_Q14, n_LTP_Q14) >> 4;
                tmp = sx_add(tmp, LPC_pred[li]);
                tmp = sx_sub(tmp, n_AR_Q10);
                tmp = sx_sub(tmp, n_LF_Q10);
                i32 r_Q10 = sx_sub(x_sc_Q10, tmp);
                Seed[li] = sx_rand(Seed[li]);
                r_Q10 = (r_Q10 ^ dither) - dither;
                LTP_pred[li] = LTP_pred_Q14;
                rD[li] = r_Q10;
            }
            RW_MARK("B")
            // phase B: the two candidates of each side state (Agora_Silk_RDCx1) from the side's share of the CENTRE residual
            const i32 Lambda_Q10 = cf[RW_CM + 6], offsum = cf[RW_CM + 7], Gain_s = cf[RW_CM + 8];      // (the stream's scalars of the sample step)
            RWT_FROM(rC, rD, 0)
            RW_FORK(l) {
                const int li = RW_LI(l), si = RW_SI(l);
                if (RW_IS_S(l)) {
                    const i32 r_md_Q10 = sx_smulww(my_inv_gain[li], rC[si]);
                    sx_nsq_rdcx1(r_md_Q10, rD[li], my_inv_of_delta[li], Lambda_Q10, my_offset[li], cInc[li], cQ10[li]);
                }
            }
            RW_MARK("C")
            // phase C: Agora_Silk_CenterRD (NSQ_del_dec.c:1152) in the centre's lanes: the centre takes the best two of the four
            // combinations of side candidates; the side candidates are then re-ordered so that candidate j of every track belongs
            // to combination w_j
            i32 wpk[RW_NS], ccInc[RW_NS][2], ccQ10[RW_NS][2];
#if RW_ROW_C
            // the row's FOUR quads evaluate the four combinations of a state side by side (quad c: combination c; member of MD1
            // {0,1,0,1}, of MD2 {0,1,1,0}) instead of the centre's lanes evaluating all four: the members' values travel by
            // bank-masked row shifts (MD1 lives in quad 1, MD2 in quad 2), the four costs come back as keys (cost << 2 | c) by three row
            // rotations, and every lane of the row finds the two smallest on its own -- which also replaces the broadcast of the
            // winners.  The costs are sums of three terms that each went through >> 10: |cost| < 2^23, the key cannot overflow; equal
            // costs order by combination index, which is the reference's "first minimum".
            {
                const u32 t4_ = threadIdx.x & 12u;
                const i32 q0_ = cQ10[0][0], q1_ = cQ10[0][1], r0_ = cInc[0][0], r1_ = cInc[0][1];
                i32 x1q = q1_, x1r = r1_, x2q = q1_, x2r = r1_;       // (quad 1 / quad 2: the quad's own second candidate)
                x1q = __builtin_amdgcn_update_dpp(x1q, q0_, RW_SHL(4), 0xf, 0x1, false);  x1r = __builtin_amdgcn_update_dpp(x1r, r0_, RW_SHL(4), 0xf, 0x1, false);
                x2q = __builtin_amdgcn_update_dpp(x2q, q0_, RW_SHL(8), 0xf, 0x1, false);  x2r = __builtin_amdgcn_update_dpp(x2r, r0_, RW_SHL(8), 0xf, 0x1, false);
                x1q = __builtin_amdgcn_update_dpp(x1q, q0_, RW_SHR(4), 0xf, 0x4, false);  x1r = __builtin_amdgcn_update_dpp(x1r, r0_, RW_SHR(4), 0xf, 0x4, false);
                x2q = __builtin_amdgcn_update_dpp(x2q, q1_, RW_SHL(4), 0xf, 0x2, false);  x2r = __builtin_amdgcn_update_dpp(x2r, r1_, RW_SHL(4), 0xf, 0x2, false);
                x1q = __builtin_amdgcn_update_dpp(x1q, q1_, RW_SHR(8), 0xf, 0x8, false);  x1r = __builtin_amdgcn_update_dpp(x1r, r1_, RW_SHR(8), 0xf, 0x8, false);
                x2q = __builtin_amdgcn_update_dpp(x2q, q0_, RW_SHR(4), 0xf, 0x8, false);  x2r = __builtin_amdgcn_update_dpp(x2r, r0_, RW_SHR(4), 0xf, 0x8, false);
                const i32 rc3_ = RW_DPP(rD[0], RW_SHR(12));                              // the centre's residual, for the fourth quad as well
                const i32 r_temp = sx_sub(t4_ == 12u ? rc3_ : rC[0], offsum);
                const i32 qx = sx_add(x1q, x2q);
                const i32 rd_ = sx_add(sx_add(sx_nsq_center_rd1(qx, r_temp, offsum, Lambda_Q10), sx_mul_lambda(x1r)), sx_mul_lambda(x2r));
                const i32 k0_ = (i32)(((u32)rd_ << 2) | (t4_ >> 2));
                const i32 ka_ = RW_DPP(k0_, RW_ROR(4)), kb_ = RW_DPP(k0_, RW_ROR(8)), kc_ = RW_DPP(k0_, RW_ROR(12));
                const i32 lo1 = sx_min(k0_, ka_), hi1 = sx_max(k0_, ka_), lo2 = sx_min(kb_, kc_), hi2 = sx_max(kb_, kc_);
                const i32 m1 = sx_min(lo1, lo2), m2 = sx_min(sx_max(lo1, lo2), sx_min(hi1, hi2));
                const u32 w1 = (u32)m1 & 3u, w2 = (u32)m2 & 3u;
                tS[0] = (i32)(w1 | (w2 << 2));
                // the centre's lanes: the winners' quantised values from their members (x1q / x2q hold the first candidates there)
                ccInc[0][0] = m1 >> 2;
                ccInc[0][1] = m2 >> 2;
#ifdef RW_CQ_SELECT
                const i32 p1q1_ = RW_DPP(q1_, RW_SHL(4)), p2q1_ = RW_DPP(q1_, RW_SHL(8));
                ccQ10[0][0] = sx_add(((0xAu >> w1) & 1u) ? p1q1_ : x1q, ((0x6u >> w1) & 1u) ? p2q1_ : x2q);
                ccQ10[0][1] = sx_add(((0xAu >> w2) & 1u) ? p1q1_ : x1q, ((0x6u >> w2) & 1u) ? p2q1_ : x2q);
#else
                // (the winners' sums wait in the quads that evaluated them: one crossbar read each)
                const u32 lb_ = (threadIdx.x & 0x33u) << 2;
                ccQ10[0][0] = __builtin_amdgcn_ds_bpermute((int)((w1 << 4) | lb_), qx);
                ccQ10[0][1] = __builtin_amdgcn_ds_bpermute((int)((w2 << 4) | lb_), qx);
#endif
                (void)wpk;
            }
#else
            { i32 a_[RW_NL], b_[RW_NL], c_[RW_NL], d_[RW_NL];
#pragma unroll
              for (int q_ = 0; q_ < RW_NL; q_++) { a_[q_] = cQ10[q_][0]; b_[q_] = cQ10[q_][1]; c_[q_] = cInc[q_][0]; d_[q_] = cInc[q_][1]; }
              RWT_TO0(p1q0, a_, 1) RWT_TO0(p1q1, b_, 1) RWT_TO0(p1r0, c_, 1) RWT_TO0(p1r1, d_, 1)
              RWT_TO0(p2q0, a_, 2) RWT_TO0(p2q1, b_, 2) RWT_TO0(p2r0, c_, 2) RWT_TO0(p2r1, d_, 2) }
            RW_FORK(l) {
              if (RW_IS_C(l)) {
                const int li = RW_LI(l), si = RW_SI(l);
                const i32 off = offsum;
                const i32 qx0 = p1q0[si] + p2q0[si], qx1 = p1q1[si] + p2q1[si], qx2 = p1q0[si] + p2q1[si], qx3 = p1q1[si] + p2q0[si];
                const i32 r_temp = sx_sub(rD[li], off);
                const i32 l1r0 = sx_mul_lambda(p1r0[si]), l1r1 = sx_mul_lambda(p1r1[si]), l2r0 = sx_mul_lambda(p2r0[si]), l2r1 = sx_mul_lambda(p2r1[si]);
                i32 rdx0 = sx_nsq_center_rd1(qx0, r_temp, off, Lambda_Q10), rdx1 = sx_nsq_center_rd1(qx1, r_temp, off, Lambda_Q10);
                i32 rdx2 = sx_nsq_center_rd1(qx2, r_temp, off, Lambda_Q10), rdx3 = sx_nsq_center_rd1(qx3, r_temp, off, Lambda_Q10);
                rdx0 = sx_add(sx_add(rdx0, l1r0), l2r0);
                rdx1 = sx_add(sx_add(rdx1, l1r1), l2r1);
                rdx2 = sx_add(sx_add(rdx2, l1r0), l2r1);
                rdx3 = sx_add(sx_add(rdx3, l1r1), l2r0);
                // best combination (first minimum) and best of the remaining three (first minimum among them): selects, no branches.
                // (the costs are sums of three terms that each went through >> 10: far from INT32_MAX, which can stand for "taken")
                const i32 m1 = sx_min(sx_min(rdx0, rdx1), sx_min(rdx2, rdx3));
                const bool e0 = rdx0 == m1, e1 = rdx1 == m1, e2 = rdx2 == m1;
                const int w1 = e0 ? 0 : (e1 ? 1 : (e2 ? 2 : 3));
                const i32 q_w1 = e0 ? qx0 : (e1 ? qx1 : (e2 ? qx2 : qx3));
                const i32 x0 = w1 == 0 ? SX_I32_MAX : rdx0, x1 = w1 == 1 ? SX_I32_MAX : rdx1, x2 = w1 == 2 ? SX_I32_MAX : rdx2, x3 = w1 == 3 ? SX_I32_MAX : rdx3;
                const i32 m2 = sx_min(sx_min(x0, x1), sx_min(x2, x3));
                const bool f0 = x0 == m2, f1 = x1 == m2, f2 = x2 == m2;
                const int w2 = f0 ? 0 : (f1 ? 1 : (f2 ? 2 : 3));
                const i32 q_w2 = f0 ? qx0 : (f1 ? qx1 : (f2 ? qx2 : qx3));
                ccInc[si][0] = m1;
                ccInc[si][1] = m2;
                ccQ10[si][0] = q_w1;
                ccQ10[si][1] = q_w2;
                wpk[si] = w1 | (w2 << 2);
              }
            }
            RWT_FROM(tS, wpk, 0)
#endif
            RW_FORK(l) {
                const int li = RW_LI(l), si = RW_SI(l), t = RW_T(l);
                // the reference's 12-way memcpy case table (NSQ_del_dec.c:1266-1336) is this selection;
                // member of combination w: MD1 {0,1,0,1}, MD2 {0,1,1,0} -- bit w of a lane constant
                const u32 mtab = t == 1 ? 0xAu : 0x6u;
                const bool ca = ((mtab >> (tS[si] & 3)) & 1u) != 0, cb = ((mtab >> ((tS[si] >> 2) & 3)) & 1u) != 0;
                const i32 a0 = cInc[li][0], a1 = cInc[li][1], d0 = cQ10[li][0], d1 = cQ10[li][1];
                const bool ctr = t == 0;
                cInc[li][0] = ctr ? ccInc[si][0] : (ca ? a1 : a0);  cInc[li][1] = ctr ? ccInc[si][1] : (cb ? a1 : a0);
                cQ10[li][0] = ctr ? ccQ10[si][0] : (ca ? d1 : d0); cQ10[li][1] = ctr ? ccQ10[si][1] : (cb ? d1 : d0);
                // the track's share of the joint cost of candidate [0] (Agora_Silk_JudgeWinner, NSQ_del_dec.c:671)
                const i32 cand0 = sx_add(RD[li], cInc[li][0]);
                tT[li] = sx_add(cand0, sx_smulw_pre(cand0, lamT[li]));
            }
            RWT_SUM(jv, tT)
            smpl_buf_idx = smpl_new;
            RW_MARK("E")
            // phase E: Agora_Silk_JudgeWinner.  States whose decisionDelay-old ancestor differs from the joint winner's, in any
            // track, are expired (their centre costs are pushed up); then up to (number of expired states) rounds of "the best
            // second candidate replaces the worst first candidate" -- played on three index registers:
            //   par   whose filter state the lane continues from,  csrc / csel   whose candidate (and which one) it takes
            // (the centre's lanes play it; the plan then crosses to the side tracks' lanes as one packed word)
            i32 pen[RW_NS];
            {
                const i32 PEN = SX_I32_MAX >> 4;
                RWS_ARGMIN(jv, mv, mi)                                   // mi = the joint winner
                // the delayed random state of this state's lineage: held by the lane that owns the lineage's slot of that ring position;
                // before the frame has written that position (first decisionDelay samples) the reference reads its zero-initialised ring
                const bool written = k * SX_SUBFR + i >= decisionDelay;
                RW_FORK(l) {
                    if (RW_ONCE(l)) gq[RW_SI(l)] = RW_LIN_SLOT(linLo[RW_SI(l)], linHi[RW_SI(l)], last_smple_idx);
                    tT[RW_LI(l)] = qc[RW_LI(l)].w3;
                }
                RWK_GATHER(tT2, tT, gq)
                RW_FORK(l) { myRand[RW_LI(l)] = written ? tT2[RW_LI(l)] : 0; }
                RWK_PICK(tT2, myRand, mi)                                // the winner's
                RW_FORK(l) { tT[RW_LI(l)] = (myRand[RW_LI(l)] ^ tT2[RW_LI(l)]) != 0 ? 1 : 0; }
                RWT_OR(tS2, tT)                                          // expired: differs in any track
                RW_FORK(l) {
                    if (RW_IS_C(l)) {                                    // (only the centre's costs carry the penalty and take part in the rounds)
                        const int li = RW_LI(l), si = RW_SI(l);
                        pen[si] = tS2[si] ? PEN : 0;
                        par[si] = RW_K(l); csrc[si] = RW_K(l); csel[si] = 0;
                        const i32 rp_ = sx_add(RD[li], pen[si]);
                        c0[si] = sx_add(rp_, cInc[li][0]); c1[si] = sx_add(rp_, cInc[li][1]);
                    }
                }
                RWS_SUM(tS2, nrep)                                       // number of expired states
                RWS_ARGMIN(c1, mv2, mi2)                                 // best candidate [1] (first minimum): the [1] entries never change
#if SX_NLANES == 1
                int RandSyncCtl = RW_UNI(nrep);
                do {
                    RWS_ARGMAX(c0, mv, mi)                               // worst candidate [0] (first maximum)
                    RWS_PICK(gq, par, mi2)                               // the state lane mi2 holds NOW (it may itself have been replaced)
                    RW_FORK(l) {
                        const int si = RW_SI(l);
                        const bool rep_ = (mv2[si] < mv[si]) & (RW_K(l) == mi[si]);
                        par[si] = rep_ ? gq[si] : par[si]; csrc[si] = rep_ ? mi2[si] : csrc[si]; csel[si] = rep_ ? 1 : csel[si]; c0[si] = rep_ ? mv2[si] : c0[si];
                    }
                } while (--RandSyncCtl > 0);
#else
                i32 RandSyncCtl = nrep[0];
                do {
                    RWS_ARGMAX(c0, mv, mi)
                    RWS_PICK(gq, par, mi2)
                    const bool rep_ = (mv2[0] < mv[0]) & ((i32)(threadIdx.x & 3u) == mi[0]);
                    par[0] = rep_ ? gq[0] : par[0]; csrc[0] = rep_ ? mi2[0] : csrc[0]; csel[0] = rep_ ? 1 : csel[0]; c0[0] = rep_ ? mv2[0] : c0[0];
                } while (--RandSyncCtl > 0);
#endif
                // (a row per stream: the plan crosses from the centre's lanes to the side tracks' as one packed word; in the emulation
                // every virtual lane's copy of the centre's costs was garbage except the centre's)
                RW_FORK(l) { if (RW_ONCE(l)) tS[RW_SI(l)] = par[RW_SI(l)] | (csrc[RW_SI(l)] << 2) | (csel[RW_SI(l)] << 4); }
                RWT_FROM(tS2, tS, 0)
                RW_FORK(l) { if (RW_ONCE(l)) { const int si = RW_SI(l); par[si] = tS2[si] & 3; csrc[si] = (tS2[si] >> 2) & 3; csel[si] = (tS2[si] >> 4) & 1; } }
            }
            RW_MARK("M")
            // the survivors move: SKP_Silk_copy_del_dec_state (NSQ_del_dec.c:1668), every register once.
            // The last-sample memories shift by one on the way (sLPC[0] takes the new sample in phase G).
            i32 cand1RD[RW_NL], cand1Q10[RW_NL], penT[RW_NL];
            {
                // the chosen candidate [1] and the predictions it was built on come from the lane that produced it
                // (the long-term prediction is the same in the four states of a track)
                RW_FORK(l) {
                    const int li = RW_LI(l);
                    penT[li] = RW_T(l) == 0 ? pen[RW_SI(l)] : 0;
                    cand1RD[li] = sx_add(sx_add(RD[li], penT[li]), cInc[li][1]); cand1Q10[li] = cQ10[li][1];
                }
#define RW_LV_C1RD(q_) cand1RD[q_]
#define RW_LV_C1Q10(q_) cand1Q10[q_]
#define RW_LV_LPCP(q_) LPC_pred[q_]
#define RW_LV_NAR(q_) n_AR[q_]
#define RW_LV_NLF(q_) n_LF[q_]
#define RW_LV_DITH(q_) dith[q_]
                RWK_PERM(RW_LV_C1RD, csrc) RWK_PERM(RW_LV_C1Q10, csrc)
                RWK_PERM(RW_LV_LPCP, csrc) RWK_PERM(RW_LV_NAR, csrc) RWK_PERM(RW_LV_NLF, csrc) RWS_PERM(RW_LV_DITH, csrc)
#define RW_LV_SEED2(q_) Seed2[q_]
#define RW_LV_SEEDI(q_) SeedInit2[q_]
#define RW_LV_LINLO(q_) linLo[q_]
#define RW_LV_LINHI(q_) linHi[q_]
#define RW_LV_SEED(q_) Seed[q_]
                RWS_PERM(RW_LV_SEED2, par) RWS_PERM(RW_LV_SEEDI, par) RWS_PERM(RW_LV_LINLO, par) RWS_PERM(RW_LV_LINHI, par) RWK_PERM(RW_LV_SEED, par)
#pragma unroll
                for (int j = 0; j < SX_SHAPE_ORDER; j++) {
#define RW_LV_SAR2(q_) sAR2[q_][j]
                    RWK_PERM(RW_LV_SAR2, par)
                }
#pragma unroll
                for (int j = SX_LPC - 1; j > 0; j--) {
#if SX_NLANES == 1
                    { i32 o_[12]; for (int q_ = 0; q_ < 12; q_++) o_[q_] = sLPC[q_][j - 1]; for (int q_ = 0; q_ < 12; q_++) sLPC[q_][j] = o_[(q_ & ~3) | par[q_]]; }
#else
#pragma unroll
                    for (int q_ = 0; q_ < RW_NL; q_++) sLPC[q_][j] = rwk_from(sLPC[q_][j - 1], par[0]);
#endif
                }
            }
            RW_MARK("D")
            // phase D: undo dither, re-apply the side gains, simulate the decoder (Agora_Silk_UndoPred_And_Shap, NSQ_del_dec.c:482)
            // for the candidate the lane keeps; joint cost of the survivors
            i32 fRD[RW_NL], fQ0[RW_NL], cXq14[RW_NL], cShp[RW_NL], cExc10[RW_NL], cX[RW_NL];
            RW_FORK(l) {
                const int li = RW_LI(l), si = RW_SI(l), t = RW_T(l);
                const i32 dither = dith[si];
                const bool sel = csel[si] != 0;
                const i32 Q10 = sel ? cand1Q10[li] : cQ10[li][0];
                fRD[li] = sel ? cand1RD[li] : sx_add(sx_add(RD[li], penT[li]), cInc[li][0]);
                i32 Q = (Q10 ^ dither) - dither;
                const bool ctr = t == 0;
                // the pulse: the quantised value (a side's without its offset) >> 10; what the coder gets: the centre's excitation / the side's pulse
                fQ0[li] = ctr ? Q10 >> 10 : (i32)(i8)(sx_sub(Q10, my_offset[li]) >> 10);
                cX[li] = ctr ? Q : fQ0[li];
                Q = ctr ? Q : sx_smulww(my_DG[li], Q);
                const i32 LPC_exc_Q10 = Q + sx_rshift_round(LTP_pred[li], 4);
                const i32 xq_Q10 = sx_add(LPC_exc_Q10, LPC_pred[li]);
                const i32 sLF_AR_shp_Q10 = sx_sub(xq_Q10, n_AR[li]);
                cShp[li] = sx_sub(sLF_AR_shp_Q10, n_LF[li]);
                LF_AR[li] = sx_shl(sLF_AR_shp_Q10, 2);
                cXq14[li] = sx_shl(xq_Q10, 4);
                cExc10[li] = LPC_exc_Q10;
                tT[li] = sx_add(fRD[li], sx_smulw_pre(fRD[li], lamT[li]));
            }
            RWT_SUM(jv, tT)
            RW_MARK("F")
            // phase F: Agora_Silk_GetWinner{,_Side} (NSQ_del_dec.c:757, 820): emit the delayed sample of the joint winner.  The lane
            // that owns the winner's slot of the emitted ring position holds the cell in its prefetch registers.
            RWS_ARGMIN(jv, mv, mi)
            {
                RW_FORK(l) { if (RW_ONCE(l)) tS[RW_SI(l)] = RW_LIN_SLOT(linLo[RW_SI(l)], linHi[RW_SI(l)], last_smple_idx); }
                RWS_PICK(gq, tS, mi)
                const bool crossed = subfr > 0 && i < decisionDelay;      // the cell was written before this subframe's gain change
                RW_FORK(l) {
                    const int li = RW_LI(l);
                    if (emitted && RW_K(l) == gq[RW_SI(l)] && RW_LIVE(l)) {
                        const u32 e4 = (u32)(4 * (cur0 + k * SX_SUBFR + i - decisionDelay));
                        const i32 p16 = rw_cell_pred_Q16(qc[li]);
                        const i32 gx = crossed ? gadj[li] : 65536;               // (x 65536 >> 16: exact)
                        // (HBM gets the cell as it is -- it belongs to the previous subframe, this start's factor reaches it when
                        // it is staged --, this subframe's windows get it with the factor applied; an entry no tap of this
                        // subframe can reach goes to the row's dump word)
                        RW_EMIT_OUT(l, li, qc[li], e4, true)
                        SX_AT(i32, Pu + offsetof(SxNsqTrack, sLTP_Q16), pTrk[li] + e4 + (u32)dL4) = p16;
                        const u32 iL = (u32)(i + wb[li]), iS = iL - 1u;
                        w->win[RW_T(l)].tapL[iL < (u32)SX_TAPL_N ? iL : (u32)SX_TAPL_N] = sx_smulww(gx, p16);
                        w->win[RW_T(l)].tapS[iS < (u32)SX_TAPS_N ? iS : (u32)SX_TAPS_N] = sx_smulww(gx, qc[li].w2);
                    }
                }
            }
            RW_MARK("G")
            // phase G: Agora_Silk_Update_DelDecState (NSQ_del_dec.c:862): every state pushes its candidate into its own cell
            RW_FORK(l) {
                const int li = RW_LI(l), si = RW_SI(l), kk = RW_K(l);
                sLPC[li][0] = cXq14[li];
                lastShp[li] = cShp[li];
                Seed[li] = sx_add(Seed[li], fQ0[li]);
                RD[li] = fRD[li];
                SxRowCell cell;
                const i32 xq16 = sx_sat16(sx_rshift_round(sx_smulww(cXq14[li] >> 4, Gain_s), 10));
                cell.w0 = (i32)(((u32)xq16 & 0xFFFFu) | ((u32)cX[li] << 16));
                cell.w1 = (i32)(((u32)cExc10[li] & 0x03FFFFFFu) | (((u32)cX[li] << 10) & 0xFC000000u));
                cell.w2 = cShp[li];
                cell.w3 = Seed[li];
                if (RW_LIVE(l)) RW_CELL_ST(smpl_buf_idx, l, cell)
                // the state's own slot now holds its newest ring entry
                if (RW_ONCE(l)) {
                    const u32 m = 3u << (2 * (smpl_buf_idx & 15));
                    if (smpl_buf_idx < 16) linLo[si] = (i32)(((u32)linLo[si] & ~m) | (((u32)kk * 0x55555555u) & m));
                    else linHi[si] = (i32)(((u32)linHi[si] & ~m) | (((u32)kk * 0x55555555u) & m));
                }
            }
            RW_MARK("Z")
            wv_sync_lds();
        };
        static_assert(SX_SUBFR % 2 == 0, "two samples per iteration");
        for (int i = 0; i < SX_SUBFR; i += 2) {
            sample_step(i, qA, qB);
            sample_step(i + 1, qB, qA);
        }
        sLTP_shp_buf_idx += SX_SUBFR;
        sLTP_buf_idx += SX_SUBFR;
        subfr++;
    }

    // Agora_Silk_DelDec_UpdateState_And_Output{,_Side} (NSQ_del_dec.c:175, 245)
    RWT_FROM(jv, RD, 0)
    RWS_ARGMIN(jv, mv, mi)
    RWS_PICK(tS, SeedInit2, mi)
    RW_FORK(l) { if (l == 0) out->Seed = tS[RW_SI(l)]; }
    RWS_PICK(tS, linLo, mi)
    RWS_PICK(tS2, linHi, mi)
    wv_sync();                                  // the ring cells of the last samples must have landed
    RW_FLUSH(tS[RW_SI(l)], tS2[RW_SI(l)], SX_FRAME - decisionDelay)
    wv_sync();
    RW_FORK(l) {
        const int li = RW_LI(l);
        if (RW_K(l) == mi[RW_SI(l)] && RW_LIVE(l)) {
            SxNSQ* n = &SX_AT(SxNSQ, Pu, pTrk[li] + (u32)offsetof(SxNsqTrack, s));
#pragma unroll
            for (int i = 0; i < SX_MAX_LPC; i++) n->sLPC_Q14[i] = (SX_MAX_LPC - 1 - i) < SX_LPC ? sLPC[li][SX_MAX_LPC - 1 - i] : 0;
#pragma unroll
            for (int i = 0; i < SX_SHAPE_ORDER; i++) n->sAR2_Q14[i] = sAR2[li][i];
            n->sLF_AR_shp_Q12 = LF_AR[li];
            n->lagPrev = c->pitchL[SX_NB_SUBFR - 1];
            n->prev_inv_gain_Q16 = prevInv[li];
            n->histBase = cur0;                      // the frame just written becomes "the previous frame"
#pragma unroll
            for (int kq = 0; kq < SX_NB_SUBFR; kq++) n->gadjPrev[kq] = w->gfac[RW_T(l)][SX_NB_SUBFR + kq];
        }
    }
    wv_sync();
#undef RW_EMIT_OUT
#undef RW_FLUSH
#undef RW_CELL
#undef RW_CELL_LD
#undef RW_CELL_ST
#undef SX_AT
}
