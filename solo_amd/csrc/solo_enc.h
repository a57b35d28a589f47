// solo_enc.h -- parameter / pulse entropy coding, high-band (BWE) encoder and the packet-level encoder
// `sx_encode_packet` = AGR_Sate_Encoder_Encode for one 40 ms packet of one stream.  Rows E0-E9 of SURVEY.md 8(a).
// Reference (JC1_SDK_SRC_ARM/src/): libBWE/AGR_BWE_SDK_API.c:129 (AGR_Sate_Encoder_Encode),
//   libBWE/AGR_BWE_encode_frame_FIX.c:8-179, libBWE/AGR_BWE_find_HB_LPC_FIX.c:4, libBWE/AGR_BWE_quant_highband.c:24-147,
//   libBWE/AGR_BWE_bits.c:77, libSATECodec/SKP_Silk_enc_API.c:104-275, SKP_Silk_encode_frame_FIX.c:33,
//   SKP_Silk_encode_parameters.c:33, SKP_Silk_encode_pulses.c:55, SKP_Silk_shell_coder.c:84, SKP_Silk_code_signs.c:40
#pragma once
#include <stddef.h>
#include "solo_enc_analysis.h"
// the quantiser (solo_enc_nsq_row.h: one lane per (track, state), a 16-lane row per stream) has its own kernel file
// (solo_nsq_row.hip); only the host emulation (tests/emu) runs it from here, in the fused single-stream form at the end of this file
#if defined(SOLO_HOST_EMU)
#include "solo_enc_nsq_row.h"
#endif
#include "solo_cdf.h"

// what the parameter coder needs of one frame (kept from both frames until the packet is assembled)
struct SxFrameIdx {
    i32 sigtype, QuantOffsetType;
    i32 GainsIndices[SX_NB_SUBFR], DeltaGainsIndices;
    i32 NLSFIndices[SX_NLSF_STAGES], NLSFInterpCoef_Q2;
    i32 lagIndex, contourIndex, PERIndex, LTPIndex[SX_NB_SUBFR], LTP_scaleIndex;
    i32 Seed, vadFlag;
    i32 inDTX, pad_;                 // DTX state after this frame (SKP_Silk_encode_frame_FIX.c:155-171); the packet is dropped if set after frame 1
};

// Work area of ONE description's range coder (one lane of the entropy-coding kernel; host emulation: one after the other).
// The entropy coder is a serial chain per description, so the GPU runs it lane-per-description: 64 descriptions (32 streams)
// per wavefront, every lane with its own staged pulses and pulse work area in LDS.  The rows are an ODD number of dwords long
// so that the 64 lanes, which mostly sit at the same offset of their own row, hit different LDS banks.
// work row of one description's pulse coder: |pulse| of the frame, block sums, block shifts (padded to a multiple of 8), the sign mask (one bit per
// sample); an odd number of dwords per row (LDS banks of neighbouring lanes)
#define SX_RC_PW_SIGNS (SX_FRAME + 2 * (SX_FRAME / 16) + 8 - ((SX_FRAME + 2 * (SX_FRAME / 16)) & 7))
#define SX_RC_PW_ROW (SX_RC_PW_SIGNS + SX_FRAME / 8 + (((SX_RC_PW_SIGNS + SX_FRAME / 8) / 4) & 1 ? 0 : 4))
static_assert(((SX_RC_PW_ROW / 4) & 1) == 1 && SX_RC_PW_ROW % 4 == 0, "odd dword rows");
#define SX_RC_BUF_STRIDE (SX_MAX_ARITHM_BYTES + 16)             // HBM: byte buffer of one description
struct SxRcInfo { i32 nBytes, error; };                        // HBM: what the coder of one description reports

struct SxCodeWork {                  // host emulation of the coding kernels: both descriptions one after the other
#if SX_NLANES == 1
    u8 buf[2][SX_RC_BUF_STRIDE];
    alignas(4) u8 pulses[SX_RC_PW_ROW];
    SxCdf cdf;
#else
    i32 unused_;
#endif
};

struct SxFrontWork {                 // LDS scratch of the per-frame analysis chain
    i16 x_buf[SX_XBUF];              // staged analysis buffer: [0,200) history | [200,360) high-passed new frame
    i16 res_pitch[2 * SX_FRAME + SX_LA_PITCH];
    i16 Wsig[SX_PITCH_LPC_WIN];      // also: VAD band buffer (4 x 80) and shaping window (120)
    union {
        SxPitchWork pitch;
        SxShapeWork shape;
        SxPredWork pred;
        SxPrefWork pref;
    } u;
};

struct SxHbWork {                    // LDS scratch of the high-band encoder
    i16 x_hb_buf[SX_HB_XBUF];
    i16 lpc_in[4 * (10 * SX_FS_KHZ + SX_HB_LPC)];
    i16 exc[2 * SX_FRAME];
    i32 NLSF_Q15[SX_MAX_LPC];
    i32 weight[SX_MAX_LPC];
    i16 A_Q12[SX_MAX_LPC];
    i32 ws[SX_NLSF2A_WS];
    SxLpcWork lpc;
};

// high-band bytes of the packets whose descriptions are not range-coded yet: the persistent pipeline's front kernel (solo_enc_kernels.h) runs
// its coder once per SX_HB_Q packets, one lane per (packet, description); everywhere else slot 0 is the only one used.  (The slots fit
// into what the LDS allocation granularity of 512 bytes leaves behind the work area: 8 632 of 8 704, 12 200 of 12 288 bytes.)
#define SX_HB_Q (SX_FS_KHZ == 8 ? 10 : 8)
struct SxEncWork {
    // persistent over the launch / packet
    SxEncState st;                   // the stream's compact state (HBM record -> LDS at launch start, back at the end)
    SxEncCtrl ctrl;
    i16 xfw[SX_FRAME];
    u8 hb_bytes[8 * SX_HB_Q];
    // phase-local
    union {
        i16 qmf_tl[63 + SX_PACKET];
        SxFrontWork front;
#if defined(SOLO_HOST_EMU)
        SxRowWork nsq;               // host emulation runs the three stages back to back in one work area
#endif
        SxCodeWork code;
        SxHbWork hb;
    } u;
};

// ---------------------------------------------------------------------------------------------------
// entropy coding
// ---------------------------------------------------------------------------------------------------
SX_HD void sx_enc_split(SxRangeEnc* rc, int p_child1, int p, const u16* shell_table, const SxCdf* cdf) {
    if (p > 0) sx_rc_enc(rc, p_child1, &shell_table[cdf->shell_offsets[p]]);
}

// SKP_Silk_shell_encoder, SKP_Silk_shell_coder.c:84
SX_HD void sx_shell_encoder(SxRangeEnc* rc, const u8* pa, int sh, const SxCdf* cdf) {      // pa: the block's magnitudes, coded after >> sh
    i32 p0[16], p1[8], p2[4], p3[2], p4;
    for (int k = 0; k < 16; k++) p0[k] = (i32)pa[k] >> sh;
    for (int k = 0; k < 8; k++) p1[k] = p0[2 * k] + p0[2 * k + 1];
    for (int k = 0; k < 4; k++) p2[k] = p1[2 * k] + p1[2 * k + 1];
    for (int k = 0; k < 2; k++) p3[k] = p2[2 * k] + p2[2 * k + 1];
    p4 = p3[0] + p3[1];
    sx_enc_split(rc, p3[0], p4, cdf->cdf_shell3, cdf);
    for (int h = 0; h < 2; h++) {
        sx_enc_split(rc, p2[2 * h], p3[h], cdf->cdf_shell2, cdf);
        for (int g = 0; g < 2; g++) {
            const int m = 2 * h + g;
            sx_enc_split(rc, p1[2 * m], p2[m], cdf->cdf_shell1, cdf);
            sx_enc_split(rc, p0[4 * m], p1[2 * m], cdf->cdf_shell0, cdf);
            sx_enc_split(rc, p0[4 * m + 2], p1[2 * m + 1], cdf->cdf_shell0, cdf);
        }
    }
}

// SKP_Silk_encode_pulses + SKP_Silk_encode_signs, SKP_Silk_encode_pulses.c:55, SKP_Silk_code_signs.c:40
// q: the frame's pulses where the quantiser left them (4-byte aligned row; HBM on the GPU -- read once, four samples a load).
// pw: SX_RC_PW_ROW bytes of work space (LDS on the GPU; 4-byte aligned): the UNSHIFTED magnitudes (|pulse| <= 128), the final block
// sums / shifts (they fit a byte), the signs as a bit mask behind them.  A block's magnitudes are shifted where they are used (the
// reference shifts its copy in place and goes back to the pulses for the bits it dropped), so the pulses need no row of their own.
SX_HD void sx_encode_pulses(SxRangeEnc* rc, int sigtype, int QuantOffsetType, const i8* q, const SxCdf* cdf, u8* pw) {
    SX_IN_LDS(cdf); SX_IN_LDS(pw);
    const int iter = SX_FRAME / 16;
#if defined(__HIP_DEVICE_COMPILE__)
    pw = (u8*)__builtin_assume_aligned(pw, 4);                 // (rows of dwords: SxRcWork::pw, SxNsqOut::q)
    q = (const i8*)__builtin_assume_aligned(q, 4);
#endif
    u8 *abs_pulses = pw, *sum_pulses = pw + SX_FRAME, *nRshifts = pw + SX_FRAME + SX_FRAME / 16;
    const i32 maxp0 = T_max_pulses[0], maxp1 = T_max_pulses[1], maxp2 = T_max_pulses[2], maxp3 = T_max_pulses[3];
    static_assert(SX_FRAME % 32 == 0 && SX_RC_PW_SIGNS % 4 == 0, "sign mask words");
    u32* neg = (u32*)(void*)(pw + SX_RC_PW_SIGNS);
    for (int w_ = 0; w_ < SX_FRAME / 32; w_++) {
        u32 m = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u32 v;
            v = sx_pub_ld((const u32*)(const void*)(q + 32 * w_ + 4 * j));       // (written by the quantiser's kernel, possibly while this one runs: solo_wave.h)
            const u32 sb = (v >> 7) & 0x01010101u;                      // 1 in the bytes that are negative
            // per byte: negative ? -b : b = (b ^ 0xFF) + 1; a negative byte is not 0, so the + 1 never carries into its neighbour
            const u32 a = (v ^ (sb * 0xFFu)) + sb;
            __builtin_memcpy(abs_pulses + 32 * w_ + 4 * j, &a, 4);
            m |= ((sb & 1u) | ((sb >> 7) & 2u) | ((sb >> 14) & 4u) | ((sb >> 21) & 8u)) << (4 * j);
        }
        neg[w_] = m;
    }
    for (int i = 0; i < iter; i++) {
        const u8* ap = &abs_pulses[i * 16];
        int nrs = 0;
        for (;;) {
            // combine_and_check: 1+1 (max 3), 2+2 (max 6), 4+4 (max 8), 8+8 (max 12); the reference aborts each level at
            // the first overflow, which leaves its `pulses_comb` partly stale -- but any overflow forces another
            // pass, so only the overflow count of a clean pass is observable
            i32 c1[8], c2[4], c3[2];
            int scale_down = 0, bad;
            bad = 0;
            for (int k = 0; k < 8; k++) { c1[k] = ((i32)ap[2 * k] >> nrs) + ((i32)ap[2 * k + 1] >> nrs); bad |= c1[k] > maxp0; }
            scale_down += bad;
            bad = 0;
            for (int k = 0; k < 4; k++) { c2[k] = c1[2 * k] + c1[2 * k + 1]; bad |= c2[k] > maxp1; }
            scale_down += bad;
            bad = 0;
            for (int k = 0; k < 2; k++) { c3[k] = c2[2 * k] + c2[2 * k + 1]; bad |= c3[k] > maxp2; }
            scale_down += bad;
            const i32 sum = c3[0] + c3[1];
            if (sum > maxp3) scale_down++;
            if (!scale_down) { sum_pulses[i] = (u8)sum; break; }
            nrs++;
        }
        nRshifts[i] = (u8)nrs;
    }
    int RateLevelIndex = 0;
    i32 minSumBits_Q6 = SX_I32_MAX;
    for (int k = 0; k < 9; k++) {
        const i16* nBits = &cdf->bits_pulses_per_block_Q6[k * 20];
        i32 sumBits_Q6 = cdf->bits_rate_levels_Q6[sigtype * 9 + k];
        for (int i = 0; i < iter; i++) sumBits_Q6 += nRshifts[i] > 0 ? nBits[18 + 1] : nBits[sum_pulses[i]];
        if (sumBits_Q6 < minSumBits_Q6) { minSumBits_Q6 = sumBits_Q6; RateLevelIndex = k; }
    }
    sx_rc_enc(rc, RateLevelIndex, &cdf->cdf_rate_levels[sigtype * 10]);
    const u16* cdf_ptr = &cdf->cdf_pulses_per_block[RateLevelIndex * 21];
    for (int i = 0; i < iter; i++) {
        if (nRshifts[i] == 0) {
            sx_rc_enc(rc, sum_pulses[i], cdf_ptr);
        } else {
            sx_rc_enc(rc, 18 + 1, cdf_ptr);
            for (int k = 0; k < nRshifts[i] - 1; k++) sx_rc_enc(rc, 18 + 1, &cdf->cdf_pulses_per_block[9 * 21]);
            sx_rc_enc(rc, sum_pulses[i], &cdf->cdf_pulses_per_block[9 * 21]);
        }
    }
    for (int i = 0; i < iter; i++)
        if (sum_pulses[i] > 0) sx_shell_encoder(rc, &abs_pulses[i * 16], nRshifts[i], cdf);
    // least significant bits of the blocks that were scaled down.  Few blocks are (at these rates), but some lane of a wavefront
    // nearly always has one: every lane walks ITS OWN list of (block, sample, bit) symbols, so the wavefront's trip count is the
    // longest list of a lane and not the union of the blocks
    const u32 p_lsb = cdf->cdf_lsb[1];
    {
        int i = 0;
        while (i < iter && nRshifts[i] == 0) i++;
        int k = 0, j = i < iter ? (int)nRshifts[i] - 1 : 0;
        while (i < iter) {
            const i32 abs_q = abs_pulses[i * 16 + k];
            sx_rc_enc_bin(rc, (abs_q >> j) & 1, p_lsb);
            if (--j < 0) {
                if (++k == 16) {
                    k = 0;
                    do { i++; } while (i < iter && nRshifts[i] == 0);
                }
                j = i < iter ? (int)nRshifts[i] - 1 : 0;
            }
        }
    }
    const u32 p_sign = cdf->cdf_sign[sx_smulbb(10 - 1, (sigtype << 1) + QuantOffsetType) + RateLevelIndex];
    for (int w_ = 0; w_ < SX_FRAME / 32; w_++) {
        const u32 m = neg[w_];
        for (int b_ = 0; b_ < 32; b_++)
            if (abs_pulses[32 * w_ + b_] != 0) sx_rc_enc_bin(rc, (i32)(((m >> b_) & 1u) ^ 1u), p_sign);
    }
}


// SKP_Silk_encode_parameters, SKP_Silk_encode_parameters.c:33 (md_type = 1 description `md`).  x: the frame's indices (HBM record;
// read once into registers)
SX_HD void sx_encode_parameters(SxRangeEnc* rc, const SxFrameIdx* xp, int Seed, int frame, int md, int writeMDIndex, int typeOffsetPrev, const i8* q,
                                const SxCdf* cdf, u8* pw) {
    SX_IN_LDS(cdf);
    const SxFrameIdx xv = *xp;
    const SxFrameIdx* x = &xv;
    if (frame == 0) {
        if (writeMDIndex == 1) sx_rc_enc(rc, md, cdf->cdf_mdindex);
        sx_rc_enc(rc, SX_FS_KHZ == 8 ? 0 : 2, cdf->cdf_fs);      // index of fs_kHz in SamplingRates_table = {8, 12, 16, 24}
    }
    const int typeOffset = 2 * x->sigtype + x->QuantOffsetType;
    if (frame == 0) sx_rc_enc(rc, typeOffset, cdf->cdf_type_offset);
    else sx_rc_enc(rc, typeOffset, &cdf->cdf_type_offset_joint[typeOffsetPrev * 5]);
    if (frame == 0) sx_rc_enc(rc, x->GainsIndices[0], &cdf->cdf_gain[x->sigtype * 65]);
    else sx_rc_enc(rc, x->GainsIndices[0], cdf->cdf_delta_gain);
#pragma unroll
    for (int i = 1; i < SX_NB_SUBFR; i++) sx_rc_enc(rc, x->GainsIndices[i], cdf->cdf_delta_gain);
    if (frame == 0) sx_rc_enc(rc, x->DeltaGainsIndices, cdf->cdf_md_delta_gain);
    {
        const i32 nvec0[SX_NLSF_STAGES] = T_NLSF_CB0_NVEC, nvec1[SX_NLSF_STAGES] = T_NLSF_CB1_NVEC;
        const u16* ncdf = x->sigtype == 0 ? cdf->nlsf_cb0_cdf : cdf->nlsf_cb1_cdf;
        int off = 0;
#pragma unroll
        for (int s = 0; s < SX_NLSF_STAGES; s++) {
            sx_rc_enc(rc, x->NLSFIndices[s], ncdf + off);
            off += (x->sigtype == 0 ? nvec0[s] : nvec1[s]) + 1;
        }
    }
    sx_rc_enc(rc, x->NLSFInterpCoef_Q2, cdf->cdf_nlsf_interp);
    if (x->sigtype == 0) {
        sx_rc_enc(rc, x->lagIndex, cdf->cdf_pitch_lag);
        sx_rc_enc(rc, x->contourIndex, cdf->cdf_pitch_contour);
        sx_rc_enc(rc, x->PERIndex, cdf->cdf_ltp_per);
        const u16* gcdf = x->PERIndex == 0 ? cdf->cdf_ltp_gain0 : (x->PERIndex == 1 ? cdf->cdf_ltp_gain1 : cdf->cdf_ltp_gain2);
#pragma unroll
        for (int k = 0; k < SX_NB_SUBFR; k++) sx_rc_enc(rc, x->LTPIndex[k], gcdf);
        sx_rc_enc(rc, x->LTP_scaleIndex, cdf->cdf_ltpscale);
    }
    sx_rc_enc(rc, Seed, cdf->cdf_seed);
    sx_encode_pulses(rc, x->sigtype, x->QuantOffsetType, q, cdf, pw);
    sx_rc_enc(rc, x->vadFlag, cdf->cdf_vadflag);
}

// The range coder of ONE description of one packet: both frames' parameters and pulses -> `buf` (SX_RC_BUF_STRIDE bytes of HBM).
// idx2: the two frames' indices (hand-over record of the analysis), Seed0 / Seed1: the dither seeds the quantiser chose,
// q0 / q1: the description's pulses of frame 0 / frame 1 where the quantiser left them (SxNsqOut::q rows).  Reports the byte count and the
// coder's error flag.
SX_HD void sx_code_description(const SxFrameIdx* idx2, int Seed0, int Seed1, const i8* q0, const i8* q1, int md, int writeMDIndex, const SxCdf* cdf, u8* pw,
                               u8* buf, SxRcInfo* info, int fpp = 2) {
    SxRangeEnc rc;
    sx_rc_enc_init(&rc, buf);
    const int prev = 2 * idx2[0].sigtype + idx2[0].QuantOffsetType;
    sx_encode_parameters(&rc, &idx2[0], Seed0, 0, md, writeMDIndex, 0, q0, cdf, pw);
    if (fpp > 1) {
        sx_rc_enc(&rc, 1, cdf->cdf_frame_term);                  // SKP_SILK_MORE_FRAMES = 1
        sx_encode_parameters(&rc, &idx2[1], Seed1, 1, md, writeMDIndex, prev, q1, cdf, pw);
    }
    sx_rc_enc(&rc, 0, cdf->cdf_frame_term);                      // SKP_SILK_LAST_FRAME = 0
    i32 nb;
    sx_rc_length_bits(rc.bufferIx, rc.range_Q16, &nb);
    sx_rc_enc_wrap_up(&rc);
    info->nBytes = nb;
    info->error = rc.error;
}

// ---------------------------------------------------------------------------------------------------
// high band
// ---------------------------------------------------------------------------------------------------
// AGR_Bwe_encode_frame_FIX (AGR_BWE_encode_frame_FIX.c:8) for one high-band frame of N samples (BWE_FrameSize: 160 = 20 ms, or
// 320 = the 40 ms frame of joint_mode 1, AGR_BWE_SDK_API.c:64-67); writes its 4 payload bytes.
// `high`: N new high-band samples; residue0 / residue1: centre excitation Q10 of the SILK frame(s) under it (HBM; sample n of the
// high-band frame takes residue0[n] for n < 160, residue1[n - 160] after)
#define SX_HB_DELAY (5 * SX_FS_KHZ)                 // lb_Delay * hb_KHz: the high band is delayed 5 ms to stay in step with SILK
#define SX_HB_LPCBLK (10 * SX_FS_KHZ + SX_HB_LPC)   // BWE_LPCFrameSize + BWE_LPCOrder: one 10 ms analysis block with its history
// (N is a template parameter: one instance per frame length -- 160, or 320 with joint_mode 1 -- each with constant loop bounds and divisors)
template <int N>
SX_FN void sx_hb_encode_frame(SxEncHist* hist, const i16* high, const i32* residue0, const i32* residue1, SxHbWork* hw, u8* out4) {
    SX_IN_LDS(hw); SX_IN_LDS(out4);
    i16* xb = hw->x_hb_buf;
    i16* lpc_in = hw->lpc_in;
    i16* exc = hw->exc;
    const int sub_len = N >> 2;                  // BWE_SubFrameSize
    SX_PAR(i, N + SX_HB_DELAY) xb[i] = hist->x_hb_buf[i];
    SX_PAR(i, N) xb[N + SX_HB_DELAY + i] = high[i];
    wv_sync();
    // AGR_Sate_find_HB_LPC_FIX: four 10 ms blocks, each with 8 samples of history; the window runs past the
    // written part of the reference's buffer (zeros)
    SX_PAR(t, 4 * SX_HB_LPCBLK) {
        const int k = t / SX_HB_LPCBLK, j = t - k * SX_HB_LPCBLK;
        const int src = N - SX_HB_LPC + k * (SX_HB_LPCBLK - SX_HB_LPC) + j;
        lpc_in[t] = src < 2 * N + SX_HB_DELAY ? xb[src] : (i16)0;
    }
    wv_sync();
    i32 res_nrg, res_nrg_Q;
    i32* weight = hw->weight;
    i32* a_Q16 = hw->lpc.a_Q16;
    i32* NLSF_Q15 = hw->NLSF_Q15;
    SX_S(57)
    sx_burg_modified<2>(&res_nrg, &res_nrg_Q, a_Q16, lpc_in, SX_HB_LPCBLK, 4, K_FIND_LPC_COND_FAC_Q32, SX_HB_LPC, &hw->lpc.burg);
    sx_bwexpander_32(a_Q16, SX_HB_LPC, K_FIND_LPC_CHIRP_Q16);
    wv_sync();
    SX_S(58)
    sx_a2nlsf<2>(NLSF_Q15, a_Q16, SX_HB_LPC, hw->lpc.P, hw->lpc.Q, &hw->lpc.u.grid);
    wv_sync();
    SX_S(59)
    // AGR_Sate_lsp_quant_highband (AGR_BWE_quant_highband.c:91): 256-entry first stage, weighted 16-entry second stage
#ifdef SX_LANE_STREAM
    sx_row_nlsf_weights_laroia(weight, NLSF_Q15, SX_HB_LPC, SX_LANE < 16);                       // (one division per lane of row 0)
#else
    if (SX_LANE == 0) sx_nlsf_weights_laroia(SX_VPTR(weight), SX_VPTR(NLSF_Q15), SX_HB_LPC);
#endif
    wv_sync();
    int idx1 = 0;
    {
        i32 my_best = SX_I32_MAX;
        int my_idx = 0;
        SX_PAR(i, 256) {
            i32 dist = 0;
            for (int j = 0; j < SX_HB_LPC; j++) {
                i32 tmp = NLSF_Q15[j] - T_hb_lsp_cb1[i * SX_HB_LPC + j];
                dist = sx_smlabb(dist, tmp, tmp);
            }
            if (dist < my_best) { my_best = dist; my_idx = i; }
        }
        wv_argmin(&my_best, &my_idx);
        idx1 = my_idx;
    }
    wv_sync();
    for (int j = 0; j < SX_HB_LPC; j++) NLSF_Q15[j] -= T_hb_lsp_cb1[idx1 * SX_HB_LPC + j];
    wv_sync();
    int idx2 = 0;
    {
        i32 my_best = SX_I32_MAX;
        int my_idx = SX_I32_MAX;
        SX_PAR(i, 16) {
            i32 dist = 0;
            for (int j = 0; j < SX_HB_LPC; j++) {
                i32 tmp = sx_sub(NLSF_Q15[j], T_hb_lsp_cb2[i * SX_HB_LPC + j]);
                dist = sx_smlawb(dist, sx_smulbb(tmp, tmp), weight[j]);
            }
            if (dist < my_best) { my_best = dist; my_idx = i; }
        }
        wv_argmin(&my_best, &my_idx);
        idx2 = my_idx;
    }
    wv_sync();
    for (int j = 0; j < SX_HB_LPC; j++) NLSF_Q15[j] = (i32)T_hb_lsp_cb1[idx1 * SX_HB_LPC + j] + (i32)T_hb_lsp_cb2[idx2 * SX_HB_LPC + j];
    wv_sync();
    SX_S(60)
    const i32 hb_lsp_idx = (idx2 << 8) + idx1;
    i16* A_Q12 = hw->A_Q12;
#if defined(SX_LANE_STREAM)
    // (every row converts the same vector -- sx_row_nlsf2a_stable_n, solo_enc_analysis.h; the serial form only for the reference's corrections)
    if (__builtin_amdgcn_ballot_w64(!sx_row_nlsf2a_stable_n<SX_HB_LPC>(A_Q12, NLSF_Q15)) != 0)
#endif
    {
        wv_sync();
        if (SX_LANE == 0) sx_nlsf2a_stable_ws(SX_VPTR(A_Q12), SX_VPTR(NLSF_Q15), SX_HB_LPC, SX_VPTR(hw->ws));
    }
    wv_sync();
    SX_S(61)
    u32 word = (u32)hb_lsp_idx << 20;
    // the four blocks of N / 4 samples are filtered from zero state (the reference calls the filter once per block)
    SX_PAR(t, N) {
        const int sub = t / sub_len, k = t - sub * sub_len;
        const i16* in = xb + N + sub * sub_len;
        i32 acc = 0;
        for (int j = 0; j < SX_HB_LPC; j++) {
            int u = k - 1 - j;
            if (u >= 0) acc = sx_smlabb(acc, in[u], A_Q12[j]);
        }
        i32 o = sx_sub_sat32(sx_shl((i32)in[k], 12), acc);
        exc[t] = (i16)sx_sat16(sx_rshift_round(o, 12));
    }
    wv_sync();
    SX_S(62)
#ifdef SX_LANE_STREAM
    {   // the four sub-frames side by side, one per 16-lane row: energies by row sums (wrapping adds: the order is free), the 32-entry
        // gain codebook two entries per lane, the row's first minimum by four DPP steps
        const int sub = SX_LANE >> 4, jl = SX_LANE & 15;
        i32 res_nrg0 = 0, res_nrg1 = 0;
        for (int i = jl; i < sub_len; i += 16) {
            const int n = sub * sub_len + i;
            const i32 e = exc[n];
            res_nrg0 = sx_add(res_nrg0, sx_mul(e, e));
            i32 tmp = sx_pub_ld(n < SX_FRAME ? &residue0[n] : &residue1[n - SX_FRAME]) >> 10;
            res_nrg1 = sx_smlabb(res_nrg1, tmp, tmp);
        }
        res_nrg0 = sx_sqrt_approx(wv_row_sum(res_nrg0));
        res_nrg1 = sx_sqrt_approx(wv_row_sum(res_nrg1));
        const i16 gain = (i16)(sx_shl(res_nrg0 + 1, 4) / (res_nrg1 + 1));
        i32 bv = SX_I32_MAX, bi = SX_I32_MAX;
        for (int i = jl; i < 32; i += 16) {
            i16 tmp = (i16)(gain - T_hb_gain_cb[i]);
            const i32 dist = sx_smulbb(tmp, tmp);
            if (dist < bv) { bv = dist; bi = i; }
        }
        SX_ARG_STEP(0xB1, <) SX_ARG_STEP(0x4E, <) SX_ARG_STEP(0x141, <) SX_ARG_STEP(0x140, <)
#pragma unroll
        for (int r = 0; r < 4; r++) word |= (u32)__builtin_amdgcn_readlane(bi, 16 * r) << (15 - 5 * r);
    }
#else
    for (int sub = 0; sub < 4; sub++) {
        i32 res_nrg0 = 0, res_nrg1 = 0;
        SX_PAR(i, sub_len) {
            const int n = sub * sub_len + i;
            const i32 e = exc[n];
            res_nrg0 = sx_add(res_nrg0, sx_mul(e, e));
            i32 tmp = sx_pub_ld(n < SX_FRAME ? &residue0[n] : &residue1[n - SX_FRAME]) >> 10;      // (the quantiser's output: solo_wave.h "publish / consume")
            res_nrg1 = sx_smlabb(res_nrg1, tmp, tmp);
        }
        res_nrg0 = wv_sum(res_nrg0);
        res_nrg1 = wv_sum(res_nrg1);
        res_nrg0 = sx_sqrt_approx(res_nrg0);
        res_nrg1 = sx_sqrt_approx(res_nrg1);
        const i16 gain = (i16)(sx_shl(res_nrg0 + 1, 4) / (res_nrg1 + 1));
        i32 min_dist = SX_I32_MAX;
        int gidx = SX_I32_MAX;
        SX_PAR(i, 32) {
            i16 tmp = (i16)(gain - T_hb_gain_cb[i]);
            const i32 dist = sx_smulbb(tmp, tmp);
            if (dist < min_dist) { min_dist = dist; gidx = i; }
        }
        wv_argmin(&min_dist, &gidx);
        word |= (u32)gidx << (15 - 5 * sub);
    }
#endif
    SX_S(45)
    out4[0] = (u8)(word >> 24); out4[1] = (u8)(word >> 16); out4[2] = (u8)(word >> 8); out4[3] = (u8)word;
    // slide the buffer: the last 200 samples are the next frame's history
    SX_PAR(i, N + SX_HB_DELAY) hist->x_hb_buf[i] = xb[N + i];
    wv_sync();
}

// ---------------------------------------------------------------------------------------------------
// frame and packet level
// ---------------------------------------------------------------------------------------------------
#ifndef SX_ENC_TAP
#define SX_ENC_TAP(stage, st, w, sig)   // test hook (tests/emu): nothing in the product build
#endif

// SKP_Silk_encode_frame_FIX (SKP_Silk_encode_frame_FIX.c:33) up to and including the NSQ; the range coding of both
// frames is deferred to the end of the packet (nothing in the analysis depends on it: DISABLE_BUF_RD)
SX_FNW void sx_enc_analyse_frame(SxEncStream* rec, SxEncWork* w, const i16* pIn, int frame, SxNsqIn* in, SxFrameIdx* x) {
    SX_IN_LDS(w);
    SxEncHist* hist = &rec->hist;
    SxEncState* st = &w->st;
    SxEncCtrl* c = &w->ctrl;
    SxFrontWork* f = &w->u.front;
    c->Seed = st->frameCounter++ & 3;
    i32 SNR_dB_Q7;
    SX_T_BEGIN
    // stage the analysis history and the new low-band frame
    SX_PAR(i, SX_FRAME + SX_LA_SHAPE) f->x_buf[i] = hist->x_buf[i];
    SX_PAR(i, SX_FRAME) f->res_pitch[i] = pIn[i];          // res_pitch doubles as the staging buffer of the raw input
    wv_sync();
    SX_STRETCH_LATENCY();
    // (the four bands -- at X + 0, 80, 160, 240, the last one SX_FRAME / 2 samples long -- start in Wsig and run on into the work area
    // behind it; the filter banks' raw sums follow them)
    constexpr int vad_bands_ = 240 + SX_FRAME / 2;
    static_assert(offsetof(SxFrontWork, u) == offsetof(SxFrontWork, Wsig) + sizeof(f->Wsig) && (offsetof(SxFrontWork, Wsig) + vad_bands_ * sizeof(i16)) % 16 == 0 &&
                  sizeof(f->Wsig) + sizeof(f->u) >= vad_bands_ * sizeof(i16) + SX_FRAME * sizeof(i32), "VAD bands + raw band sums in Wsig and the work area");
    sx_vad(st, c, f->res_pitch, f->Wsig, &SNR_dB_Q7, (i32*)(void*)(f->Wsig + vad_bands_));
    wv_sync();
    static_assert(sizeof(f->u) >= SX_HP_SCRATCH_WORDS * sizeof(i32), "the pitch work area doubles as the high-pass filter's scratch");
    SX_S(32)
    sx_hp_variable_cutoff(st, c, f->x_buf + SX_FRAME + SX_LA_SHAPE, f->res_pitch, (i32*)&f->u);
    wv_sync();
    SX_T(1)
    sx_find_pitch_lags(st, c, f->x_buf, f->res_pitch, f->Wsig, &f->u.pitch);
    wv_sync();
    SX_ENC_TAP(1, st, w, f->x_buf + SX_FRAME + SX_LA_SHAPE);
    SX_T(2)
    sx_noise_shape_analysis(st, c, f->res_pitch + SX_FRAME, f->x_buf + SX_FRAME, &f->u.shape);
    wv_sync();
    SX_ENC_TAP(2, st, w, f->res_pitch + SX_FRAME);
    SX_T(3)
    SX_PAR(i, SX_LTP_BUF) f->u.pref.ring[i] = hist->pf_sLTP_shp[i];
    wv_sync();
    sx_prefilter(st, c, w->xfw, f->x_buf + SX_FRAME, &f->u.pref);
    wv_sync();
    SX_PAR(i, SX_LTP_BUF) hist->pf_sLTP_shp[i] = f->u.pref.ring[i];
    wv_sync();
    SX_ENC_TAP(3, st, w, w->xfw);
    SX_T(4)
    static_assert(offsetof(SxFrontWork, Wsig) == offsetof(SxFrontWork, res_pitch) + sizeof(f->res_pitch) && offsetof(SxFrontWork, res_pitch) % 4 == 0,
                  "res_pitch and Wsig are one scratch area for the NLSF quantiser");
    sx_find_pred_coefs(st, c, f->x_buf, f->res_pitch, &f->u.pred);
    wv_sync();
    SX_ENC_TAP(4, st, w, w->xfw);
    SX_T(5)
    sx_process_gains(st, c);
    wv_sync();
    SX_ENC_TAP(5, st, w, w->xfw);
    SX_T(6)
    // the history of the next frame leaves LDS before the quantiser takes over the union
    SX_PAR(i, SX_FRAME + SX_LA_SHAPE) hist->x_buf[i] = f->x_buf[SX_FRAME + i];
    wv_sync();
    // hand-over record for the quantiser.  In the persistent pipeline the quantiser's wavefront reads it while this kernel still runs: every
    // word of it is stored write-through (sx_pub_st, solo_wave.h); the caller drains the stores and raises the stream's flag
    {
        if (SX_LANE == 0) {
            sx_pub_st(&in->sigtype, c->sigtype); sx_pub_st(&in->QuantOffsetType, c->QuantOffsetType); sx_pub_st(&in->NLSFInterpCoef_Q2, c->NLSFInterpCoef_Q2);
            sx_pub_st(&in->Seed, c->Seed); sx_pub_st(&in->Lambda_Q10, c->Lambda_Q10); sx_pub_st(&in->LTP_scale_Q14, c->LTP_scale_Q14); sx_pub_st(&in->DeltaGains_Q16, c->DeltaGains_Q16);
        }
        SX_PAR(i, SX_NB_SUBFR) {
            sx_pub_st(&in->pitchL[i], c->pitchL[i]); sx_pub_st(&in->Gains_Q16[i], c->Gains_Q16[i]); sx_pub_st(&in->LF_shp_Q14[i], c->LF_shp_Q14[i]);
            sx_pub_st(&in->Tilt_Q14[i], c->Tilt_Q14[i]); sx_pub_st(&in->HarmShapeGain_Q14[i], c->HarmShapeGain_Q14[i]);
        }
        SX_PAR(i, 2 * SX_MAX_LPC) sx_pub_st(&(&in->PredCoef_Q12[0][0])[i], (&c->PredCoef_Q12[0][0])[i]);
        SX_PAR(i, SX_LTP_ORDER * SX_NB_SUBFR) sx_pub_st(&in->LTPCoef_Q14[i], c->LTPCoef_Q14[i]);
        SX_PAR(i, SX_NB_SUBFR * SX_SHAPE_ORDER) sx_pub_st(&in->AR2_Q13[i], c->AR2_Q13[i]);
        SX_PAR(i, SX_FRAME) sx_pub_st(&in->xfw[i], w->xfw[i]);
        wv_sync();
    }
    SX_T(7)
    // VAD / DTX flags (encode_frame_FIX.c:155-171)
    if (st->speech_activity_Q8 < K_SPEECH_ACTIVITY_DTX_THRES_Q8) {
        st->vadFlag = 0;
        st->noSpeechCounter++;
        if (st->noSpeechCounter > 5) st->inDTX = 1;
        if (st->noSpeechCounter > 20 + 5) { st->noSpeechCounter = 5; st->inDTX = 0; }
    } else {
        st->noSpeechCounter = 0;
        st->inDTX = 0;
        st->vadFlag = 1;
    }
    x->sigtype = c->sigtype; x->QuantOffsetType = c->QuantOffsetType;
    for (int i = 0; i < 4; i++) { x->GainsIndices[i] = c->GainsIndices[i]; x->LTPIndex[i] = c->LTPIndex[i]; }
    x->DeltaGainsIndices = c->DeltaGainsIndices;
    for (int i = 0; i < SX_NLSF_STAGES; i++) x->NLSFIndices[i] = c->NLSFIndices[i];
    x->NLSFInterpCoef_Q2 = c->NLSFInterpCoef_Q2;
    x->lagIndex = c->lagIndex; x->contourIndex = c->contourIndex; x->PERIndex = c->PERIndex;
    x->LTP_scaleIndex = c->LTP_scaleIndex;
    x->Seed = c->Seed;                 // (replaced by the quantiser's winning seed in the coding stage)
    x->vadFlag = st->vadFlag;
    x->inDTX = st->inDTX; x->pad_ = 0;
    // cross-frame parameters (encode_frame_FIX.c:203-212)
    st->prev_sigtype = c->sigtype;
    st->prevLag = c->pitchL[SX_NB_SUBFR - 1];
    st->first_frame_after_reset = 0;
    st->nFramesInPayloadBuf = frame + 1 < st->fpp ? frame + 1 : 0;      // (the payload goes out after fpp frames: PacketSize_ms / 20)
    wv_sync();
    SX_T(8)
}

// what the coding stage needs of one analysed packet besides the quantiser's output
struct SxCodeIn {
    SxFrameIdx idx[2];
    i16 hi[SX_BAND];                 // high band of the packet (QMF output) for the BWE encoder
};

// Stage A of AGR_Sate_Encoder_Encode (AGR_BWE_SDK_API.c:129): QMF split and the analysis chain of both 20 ms frames.
// Nothing here depends on the quantiser's output (DISABLE_BUF_RD, SKP_Silk_define.h:53), so a whole launch of packets
// can be analysed before any is quantised.
SX_FNW void sx_enc_stage_a(SxEncStream* rec, SxEncWork* w, const i16* pcm, SxNsqIn* in2, SxCodeIn* cin) {
    SX_IN_LDS(w);
    SxEncHist* hist = &rec->hist;
    SX_T_BEGIN
    SX_STRETCH_DENSE();
    const int fpp = SX_UNI(w->st.fpp);
    sx_qmf_decomp(hist, pcm, w->u.qmf_tl, hist->lo, cin->hi, fpp * 2 * SX_FRAME);
    wv_sync();
    SX_T(0)
    for (int frame = 0; frame < fpp; frame++) {
        sx_enc_analyse_frame(rec, w, hist->lo + frame * SX_FRAME, frame, &in2[frame], &cin->idx[frame]);
        wv_sync();
    }
}

// Stage C: high-band encoder (needs the centre excitation of the quantiser), range coding of the two descriptions,
// payload assembly.  640 samples @ 16 kHz -> MD1 || MD2 || HB(8).  nBytesOut[0] = total, nBytesOut[1] = len(MD2) + 8.
// On the GPU the stage is two kernels: the range coder is a serial chain per description and runs LANE-per-description
// (sx_code_description; 32 streams per wavefront, solo_enc_rc_kernel), the high-band encoder and the payload assembly run
// wavefront-per-stream (sx_enc_stage_c_hb, sx_enc_stage_c_out; solo_enc_coding_kernel).  The host emulation calls the three in a row.
SX_FNW void sx_enc_stage_c_hb(SxEncStream* rec, SxEncWork* w, const SxCodeIn* cin, const SxNsqOut* out2, int hb_slot = 0) {      // (hb_slot: see SX_HB_Q)
    SX_IN_LDS(w);
    SxEncHist* hist = &rec->hist;
    SxEncState* st = &w->st;
    SX_T_BEGIN
    if (st->hb_joint) {
        sx_hb_encode_frame<2 * SX_FRAME>(hist, cin->hi, out2[0].r, out2[1].r, &w->u.hb, &w->hb_bytes[8 * hb_slot]);
        wv_sync();
        SX_T(9)
    } else {
        const int fpp = SX_UNI(st->fpp);
        for (int frame = 0; frame < fpp; frame++) {
            sx_hb_encode_frame<SX_FRAME>(hist, cin->hi + frame * SX_FRAME, out2[frame].r, out2[frame].r, &w->u.hb, &w->hb_bytes[8 * hb_slot + 4 * frame]);
            wv_sync();
            SX_T(9)
        }
    }
}

// the packet is dropped ("DTX simulation", SKP_Silk_enc_API.c:260-265): it was analysed, quantised and its high band encoded (all
// states moved on), but nothing is sent
SX_HD bool sx_enc_packet_in_dtx(int useDTX, const SxCodeIn* cin, int fpp = 2) { return useDTX && SX_UNI(cin->idx[fpp - 1].inDTX); }      // (the packet's LAST frame decides)

// Returns the total byte count, or a negative status if the payload does not fit `buf_size`.  buf0 / buf1: the range coder's
// bytes of MD1 / MD2 (HBM), info2: their byte counts and error flags.
SX_FN i32 sx_enc_stage_c_out(SxEncWork* w, const SxCodeIn* cin, const u8* buf0, const u8* buf1, const SxRcInfo* info2, u8* bits, i32 buf_size,
                             i16* nBytesOut) {
    SX_IN_LDS(w);
    SxEncState* st = &w->st;
    SX_T_BEGIN
    const int fpp = SX_UNI(st->fpp);
    const int hb_bytes = st->hb_joint ? 4 : 4 * fpp;
    if (sx_enc_packet_in_dtx(st->useDTX, cin, fpp)) {
        nBytesOut[0] = 0;                        // (The reference's bit buffer then holds just the high-band bytes and its Encode
        nBytesOut[1] = 0;                        // returns their count; mirrored: they sit at the start of the slot.)
        SX_PAR(i, hb_bytes) bits[i] = w->hb_bytes[i];
        wv_sync();
        return hb_bytes;
    }
    const i32 nb0 = SX_UNI(info2[0].nBytes), nb1 = SX_UNI(info2[1].nBytes);
    const i32 err = SX_UNI(info2[0].error | info2[1].error);
    const i32 total = nb0 + nb1 + hb_bytes;
    if (err || total > buf_size || nb0 > SX_MAX_ARITHM_BYTES || nb1 > SX_MAX_ARITHM_BYTES) {
        nBytesOut[0] = 0;
        nBytesOut[1] = 0;
        return -1;
    }
    SX_PAR(i, total) {
        u8 b;
        if (i < nb0) b = buf0[i];
        else if (i < nb0 + nb1) b = buf1[i - nb0];
        else b = w->hb_bytes[i - nb0 - nb1];
        bits[i] = b;
    }
    nBytesOut[0] = (i16)total;
    nBytesOut[1] = (i16)(nb1 + hb_bytes);
    wv_sync();
    SX_T(11)
    return total;
}

#if SX_NLANES == 1
// host emulation: high band, the two descriptions one after the other, assembly
SX_FN i32 sx_enc_stage_c(SxEncStream* rec, SxEncWork* w, const SxCodeIn* cin, const SxNsqOut* out2, u8* bits, i32 buf_size, i16* nBytesOut) {
    sx_enc_stage_c_hb(rec, w, cin, out2);
    SxRcInfo info[2] = {{0, 0}, {0, 0}};
    const int fpp = w->st.fpp;
    if (!sx_enc_packet_in_dtx(w->st.useDTX, cin, fpp)) {
        sx_cdf_load(&w->u.code.cdf);
        for (int md = 0; md < 2; md++)
            sx_code_description(cin->idx, out2[0].Seed, out2[1].Seed, out2[0].q[md], out2[1].q[md], md, w->st.useMDIndex, &w->u.code.cdf, w->u.code.pulses,
                                w->u.code.buf[md], &info[md], fpp);
    }
    return sx_enc_stage_c_out(w, cin, w->u.code.buf[0], w->u.code.buf[1], info, bits, buf_size, nBytesOut);
}
#endif

#if defined(SOLO_HOST_EMU)
// Fused single-stream form (host emulation / debugging): A, quantiser for both frames, C
SX_FN i32 sx_encode_packet(SxEncStream* rec, SxEncWork* w, SxCodeIn* cin, const i16* pcm, u8* bits, i32 buf_size, i16* nBytesOut) {
    sx_enc_stage_a(rec, w, pcm, rec->nsq_in, cin);
    wv_sync();
    for (int frame = 0; frame < w->st.fpp; frame++) {
        sx_nsq_del_dec((char*)&rec->nsq, 0u, &rec->nsq_in[frame], (char*)&rec->nsq_out[frame], 0u, &w->u.nsq, w->u.nsq.ring_emu, 0u, 12);
        wv_sync();
    }
    return sx_enc_stage_c(rec, w, cin, rec->nsq_out, bits, buf_size, nBytesOut);
}
#endif
