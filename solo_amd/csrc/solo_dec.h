// solo_dec.h -- SOLO decoder hot path (AGR_Sate_Decoder_Decode), one wavefront per stream.
//
// Rows D0-D8 of SURVEY.md section 8(a): range decoding of one or two descriptions, inverse NSQ
// (single-description rescale / two-description merge), LTP + LPC synthesis, PLC, CNG, high-band
// resynthesis and 64-tap QMF synthesis.  Specialised to the live configuration (16 kHz API rate,
// SILK running NB at 8 kHz, LPC order 10, 2 descriptions, 2 x 20 ms frames per 40 ms packet).
//
// Reference files restated here (JC1_SDK_SRC_ARM/src/...):
//   libBWE/AGR_BWE_SDK_API.c:249-280, libBWE/AGR_BWE_decode_frame_FIX.c:40-197, libBWE/AGR_BWE_qmf.c:86-182,
//   libBWE/AGR_BWE_LPC_synthesizer.c:56-136, libBWE/AGR_BWE_quant_highband.c:106-121, libBWE/AGR_BWE_bits.c:135
//   libSATECodec/SKP_Silk_dec_API.c:76-188, SKP_Silk_decode_frame.c:33-395, SKP_Silk_decode_parameters.c:31-261,
//   SKP_Silk_decode_pulses.c:33, SKP_Silk_shell_coder.c:59-155, SKP_Silk_code_signs.c:64, SKP_Silk_gain_quant.c:110,
//   SKP_Silk_NLSF_MSVQ_decode.c:31, SKP_Silk_decode_pitch.c:34, SKP_Silk_decode_core.c:43-288, SKP_Silk_PLC.c:36-417,
//   SKP_Silk_CNG.c:31-149, SKP_Silk_MA.c:40, SKP_Silk_LPC_synthesis_filter.c:37, SKP_Silk_decoder_set_fs.c:31,
//   SKP_Silk_create_init_destroy.c:34
#pragma once
#include "solo_common.h"
#include "solo_rc.h"
#include "solo_cdf.h"

#define SX_PACKET (80 * SX_FS_KHZ)   // 40 ms at the API rate (16 kHz: 640, 32 kHz: 1280)
#define SX_BAND (40 * SX_FS_KHZ)     // samples per band per packet
#define SX_HB_BYTES 8            // 2 x HB_BYTE (libBWE/AGR_BWE_defines.h:39)
#define SX_QMF_HIST 32           // synthesis memory per band (M2)
// first-failure trace of the range decoder (debug aid, costs one compare per stage)
#ifndef SX_TRACE_ALWAYS
#define SX_TRACE_ALWAYS 0
#endif
#define SX_TRACE(stage)                                                                            \
    do {                                                                                           \
        if (dbg && ((rc->error && dbg[0] == 0) || SX_TRACE_ALWAYS)) {                                                   \
            dbg[0] = (stage); dbg[1] = rc->error; dbg[2] = rc->bufferIx; dbg[3] = rc->bufferLength; \
            dbg[4] = (i32)rc->range_Q16; dbg[5] = (i32)rc->base_Q32; dbg[6] = nFramesDecoded; dbg[7] = kDesp; \
        }                                                                                          \
    } while (0)

// ---- persistent per-stream decoder state (HBM) ---------------------------------------------------
struct SxDecDesc {               // SKP_Silk_md_decoder_state, SKP_Silk_structs.h:295 (live fields)
    i32 LastGainIndex;
    i32 prevNLSF_Q15[SX_LPC];
    i32 typeOffsetPrev;
    i32 prevDeltaGainIndex;
    // the reference's range decoder of this description lives in its state (sMD[k].sRC, SKP_Silk_structs.h:296) and survives the
    // packet: see sx_decode_packet
    i32 rc_bufferLength, rc_bufferIx, rc_error;
    u32 rc_base_Q32, rc_range_Q16, rc_tail;
    i32 rc_stale;                // the slot's last description went through the batch path's records: only rc_bufferLength is current,
                                 // the registers are rebuilt from the shadow buffer if they are ever needed (sx_decode_packet)
};
struct SxPLC {                   // SKP_Silk_PLC_struct, SKP_Silk_structs.h:268
    i32 pitchL_Q8;
    i16 LTPCoef_Q14[SX_LTP_ORDER];
    i16 prevLPC_Q12[SX_LPC];
    i32 last_frame_lost;
    i32 rand_seed;
    i16 randScale_Q14;
    i16 prevLTP_scale_Q14;
    i32 conc_energy;
    i32 conc_energy_shift;
    i32 prevGain_Q16[SX_NB_SUBFR];
    i32 fs_kHz;
};
struct SxCNG {                   // SKP_Silk_CNG_struct, SKP_Silk_structs.h:283
    i32 exc_buf_Q10[SX_FRAME];
    i32 smth_NLSF_Q15[SX_LPC];
    i32 synth_state[SX_LPC];
    i32 smth_Gain_Q16;
    i32 rand_seed;
    i32 fs_kHz;
};
struct SxDecState {
    SxDecDesc md[2];
    i32 prev_inv_gain_Q16;
    i32 sLTP_Q16[2 * SX_FRAME];
    i32 sLPC_Q14[SX_MAX_LPC];
    i32 exc_Q10[SX_FRAME];
    i16 outBuf[2 * SX_FRAME];
    i32 lagPrev;
    i32 first_frame_after_reset;
    i32 nFramesDecoded;
    i32 moreInternalDecoderFrames;
    i32 FrameTermination;
    i32 vadFlag;
    i32 lossCnt;
    i32 prev_sigtype;
    i32 nBytesLeft0;
    i32 started;                 // 0 until the first received packet switched the decoder to 8 kHz
    i32 HPState[2];              // output high-pass (runs from a packet's third frame on: only after a corrupted payload, sx_dec_hp_output)
    SxCNG cng;
    SxPLC plc;
    // high band + QMF (AGR_Sate_decoder_hb_state_FIX / AGR_Sate_HB_decoder_control_FIX)
    i32 hb_lossCnt;
    i32 hb_first;
    i32 hb_joint;                // joint_mode 1: ONE 40 ms high-band frame per packet (4 HB bytes instead of 8)
    i32 fpp;                     // 2, or 1 with framesize_ms = 20: 320-sample packets, ONE 20 ms high-band frame (4 bytes) -- and, like the reference
                                 // (HB2LB_NUM = 2 low-band decoder calls per packet whatever its size, AGR_BWE_decode_frame_FIX.c:173), the packet's one
                                 // SILK frame decoded TWICE, the second time as a new packet whose output is dropped
    i32 HB_prev_NLSFq[SX_HB_LPC];
    i32 HB_synth_state[SX_HB_LPC];
    i32 HB_prev_Gain;
    i16 qmf_lo_hist[SX_QMF_HIST];   // last 32 low-band samples, time order (g0_mem of the reference, re-laid-out)
    i16 qmf_hi_hist[SX_QMF_HIST];
    i32 last_error;
#ifdef SX_RC_LOG
    i32 rclog[512];
#endif
    i32 dbg[8];                  // first-failure trace: {stage, rc error, bufferIx, bufferLength, range, base, frame, desc}
};

// Shadow of the reference's two internal range-decoder buffers (SKP_Silk_range_coder_state.buffer of sMD[0] / sMD[1],
// SKP_Silk_structs.h:85-92; sMD[0] takes whichever description is decoded first, SKP_Silk_decode_frame.c:93-99): only the four
// bytes behind the current description are ever read back (solo_rc.h: sx_rc_byte).  Lives in HBM next to the state record and is
// never staged in LDS: per packet the received description bytes are written once and eight bytes are read.
struct SxDecShadow {
    u8 b[2][SX_MAX_ARITHM_BYTES + 8];
    SxDecDesc keep[2];           // where sx_decode_packet parks a slot's state while it rebuilds the slot's coder registers
};
struct SxDecStream {             // one record per stream in HBM
    SxDecState st;
    SxDecShadow sh;
};

// ---- per-packet working set (LDS) ----------------------------------------------------------------
struct SxDecCtrl {               // SKP_Silk_decoder_control, SKP_Silk_structs.h:362
    i32 pitchL[SX_NB_SUBFR];
    i32 Gains_Q16[SX_NB_SUBFR];
    i32 DeltaGains_Q16;
    i32 Seed;
    i16 LTPCoef_Q14[SX_LTP_ORDER * SX_NB_SUBFR];    // (the prediction coefficients are not side information: SxDecWork::PredCoef_Q12)
    i32 LTP_scale_Q14;
    i32 PERIndex, RateLevelIndex, QuantOffsetType, sigtype, MDIndex, NLSFInterpCoef_Q2;
};
// symbols of one 20 ms frame of one description, in coding order (sx_extract_parameters)
struct SxFrameSyms {
    i32 fs_bad;                      // the sampling-rate symbol named another internal rate: nothing after it was read
    i32 MDIndex, typeOffset;
    i32 GainsIndices[SX_NB_SUBFR], DeltaGainIndices;
    i32 NLSFIndices[SX_NLSF_STAGES], NLSFInterpCoef_Q2;
    i32 NLSF_Q15[SX_LPC];            // the frame's NLSF vector (codebook sum, stabilised): needs nothing but the indices
    i32 lagIx, conIx, PERIndex, LTPIx[SX_NB_SUBFR], LTPscaleIx;
    i32 Seed, RateLevelIndex;
    i32 vadFlag, FrameTermination, left, error, bufferLength;     // after the frame: flags, bytes left, coder error, description length
};
// what the batch path's extraction kernel leaves for ONE description slot of one packet (HBM): see sx_extract_desc
struct alignas(16) SxExtracted {
    i32 usable;                      // both frames were read without a coder error and the symbols do not depend on the stream's history
    i32 pad_[3];
    SxFrameSyms y[2];
    i8 pulses[2][SX_FRAME];
    // What needs nothing but this packet's bits, for the slot the decoder takes its coefficients from (the last description slot in
    // use; it also carries the high-band bytes): NLSF -> prediction coefficients (stabilised) of each frame's own vector and of
    // frame 1's interpolated one (frame 0's interpolates with the previous packet: left to the decoder), and the high band's side
    // information (LSP vectors, their prediction coefficients, sub-frame gains).
    i32 have_A, have_hb;
    i16 A_final[2][SX_MAX_LPC];
    i16 A_interp1[SX_MAX_LPC];
    i32 hb_lsp[2][SX_HB_LPC];
    i16 hb_lpc[2][SX_HB_LPC];
    i16 hb_gain[2][4];
    // The side information of each frame DE-QUANTISED (sx_dequant_parameters with a blank description state: everything in it but
    // frame 0's interpolated NLSF vector and the first-frame override of the interpolation factor depends on this packet's symbols
    // only -- frame 0 codes its gains unconditionally, frame 1 relative to frame 0), and the gain index the description's state is
    // left with after each frame.  The decoder copies the block instead of walking the pitch / LTP / gain tables on two lanes with a
    // memory round trip per look-up (a third of a lone decoder wave's time, profiles/r05_decoder_sections_cycles.txt).
    SxDecCtrl ctl[2];
    i32 lastGain[2];
};
#define SX_DEC_PAYLOAD_LDS 252      // packets up to this size are staged in LDS (13.6 kbps packets are ~80 B; larger ones are read from HBM)
// High band of a packet, decoded up front (side information) and synthesised next to the low band: see sx_hb_decode_side
struct SxHbParams {
    i32 lsp[2][SX_HB_LPC];          // quantised LSPs of the (up to) two high-band frames
    i16 lpc[2][SX_HB_LPC];          // ... as prediction coefficients
    i16 gain[2][4];                 // sub-frame gains (codebook values)
    i32 S[SX_HB_LPC];               // synthesis filter state while the packet is being decoded (committed when the packet is done)
    i32 piggy, frame, lost, pad_;   // piggy: the low-band synthesis of frame `frame` also runs the high-band filter (lane 1)
};
// Phases of a packet reuse the same LDS (the decoder's occupancy is LDS-bound): see the lifetimes in the comments
struct SxDecWork {
    SxDecState st;                  // the stream's state: HBM record -> LDS at launch start, back at the end
    SxDecShadow* shadow;            // the stream's range-decoder buffer shadow (HBM)
    SxCdfDec cdf;                   // entropy-coding tables (loaded once per launch)
    u8 payload[SX_DEC_PAYLOAD_LDS + 4];
    SxDecCtrl ctrl;
    i16 PredCoef_Q12[2][SX_MAX_LPC];    // prediction coefficients of the two frame halves (from the NLSF vectors of the description in use)
    // frame scratch, in turn: parse (NLSF vectors [0,40) + per-description pulse-decoder scratch [40,80)), NLSF->LPC
    // workspace [40,122), LPC residual of decode_core / concealment signal of the PLC, high-band side information + workspace
    i32 res_Q10[SX_FRAME];
    union {
        struct {                    // parse .. inverse NSQ
            i16 pulses[2][SX_FRAME];
            SxDecCtrl ctrl2[2];     // side information as decoded from description 0 / 1 (one description per lane)
            i32 lane_out[2][4];     // per description: vadFlag, FrameTermination, bytes left, range-coder error
            i32 lane_len[2];        // per description: range-coder buffer length
        } parse;
        i32 ws1[SX_NLSF2A_WS];      // second NLSF->LPC workspace (after the inverse NSQ / before the high-band synthesis)
        struct {                    // decode_core, PLC, CNG
            i32 sLPC_Q14[SX_MAX_LPC + SX_SUBFR];
            i16 tmp16[SX_FRAME];    // exc_buf (PLC) / CNG_sig
        } syn;
        i16 hi[SX_QMF_HIST + SX_BAND];  // [history | packet] high band (high-band synthesis .. QMF)
    } u;
    SxHbParams hbp;
    i16 hi_out[SX_BAND];            // high band of the packet as it is synthesised
    i16 lo[SX_QMF_HIST + SX_BAND];  // [history | packet] low band
};

// the frame's symbols as read off the range coder, per description slot: they live in the tail of the frame scratch until they are
// de-quantised (the head holds the NLSF vectors and the pulse decoder's scratch at that time)
#define SX_SYMS_AT (SX_FRAME - 2 * (int)(sizeof(SxFrameSyms) / 4))
static_assert(SX_SYMS_AT >= 4 * SX_LPC + 4 * (SX_FRAME / 16), "frame scratch too small for the symbol records");
SX_HD SxFrameSyms* sx_dec_syms(SxDecWork* w, int md) { return (SxFrameSyms*)&w->res_Q10[SX_SYMS_AT] + md; }

// SKP_Silk_init_decoder + first decoder_set_fs(8) folded together (create_init_destroy.c:34,
// decoder_set_fs.c:31).  The reference starts at 24 kHz and switches to 8 kHz when the first
// payload is parsed; every field touched by that switch has the same value here, so the state after
// the first received packet is identical.  Packets LOST before that run the reference's 24 kHz concealment +
// resampler: sx_silk_decode_frame models exactly what survives of it (first_frame_after_reset doubles as "still at 24 kHz").
// hb_mode: bit 0 = joint_mode 1, bit 1 = framesize_ms 20
SX_HD void sx_dec_state_init(SxDecState* st, int hb_mode = 0) {
    const int hb_joint = hb_mode & 1;
    u8* p = (u8*)st;
    SX_PAR(i, (int)sizeof(SxDecState)) p[i] = 0;
    wv_sync();
    st->lagPrev = 100;
    st->first_frame_after_reset = 1;
    st->prev_inv_gain_Q16 = 65536;
    st->md[0].LastGainIndex = 1;
    st->md[1].LastGainIndex = 1;
    st->hb_first = 1;
    st->hb_joint = hb_joint;
    st->fpp = (hb_mode & 2) ? 1 : 2;
    // CNG / PLC are (re)initialised on the first call because their fs_kHz field is 0
    wv_sync();
}

// Output high-pass of the low-band decoder: SKP_Silk_decode_frame.c:381-383 applies SKP_Silk_biquad (SKP_Silk_biquad.c:43-72: second-order
// section, state in Q13, the output rounded and then incremented by one) with the coefficients of the internal rate
// (SKP_Silk_tables_other.c:72-81) once nFramesDecoded > 2.  With two frames per packet that only happens after a corrupted payload
// whose termination symbol announced more frames than it carried: the following calls decode on in the old buffer as frames 3, 4, 5.
// The state is zero at creation and never reset.  Serial on one lane: the case is rare.
SX_HD void sx_dec_hp_output(i32* S, i16* x, int len) {
    const i32 B0 = 8000, B1 = -16000, B2 = 8000;                                                    // MA part, Q13
    const i32 A0_neg = SX_FS_KHZ == 8 ? 15885 : 16127, A1_neg = SX_FS_KHZ == 8 ? -7710 : -7940;    // negated AR part, Q13
    SX_PAR(v, 1) {
        i32 s0 = S[0], s1 = S[1];
        for (int k = 0; k < len; k++) {
            const i32 in = x[k];
            const i32 o = sx_smlabb(s0, in, B0);
            s0 = sx_add(sx_smlabb(s1, in, B1), sx_shl(sx_smulwb(o, A0_neg), 3));
            s1 = sx_smlabb(sx_shl(sx_smulwb(o, A1_neg), 3), in, B2);
            x[k] = (i16)sx_sat16(sx_rshift_round(o, 13) + 1);
        }
        S[0] = s0;
        S[1] = s1;
    }
    wv_sync();
}

// SKP_Silk_gains_dequant, SKP_Silk_gain_quant.c:110 (md_enable = 1)
SX_HD void sx_gains_dequant(i32* gain_Q16, const i32* ind, i32* prev_ind, int conditional, int ind2, i32* DeltaGains_Q16) {
    const i32 OFFSET = (6 * 128) / 6 + 16 * 128;                                  // gain_quant.c:30
    const i32 INV_SCALE_Q16 = (65536 * (((86 - 6) * 128) / 6)) / (64 - 1);        // gain_quant.c:32
    for (int k = 0; k < SX_NB_SUBFR; k++) {
        if (k == 0 && conditional == 0) *prev_ind = ind[k];
        else *prev_ind += ind[k] + (-4);
        gain_Q16[k] = sx_log2lin(sx_min(sx_smulwb(INV_SCALE_Q16, *prev_ind) + OFFSET, 3967));
    }
    i32 inv_gain_Q16 = (ind2 + 1) * (32768 / 8) + 32767;                           // gain_quant.c:138-139
    *DeltaGains_Q16 = sx_inverse32_varQ(sx_max(inv_gain_Q16, 1), 32);
}

// decode_split + SKP_Silk_shell_decoder, SKP_Silk_shell_coder.c:59-155 (scalars stay in registers)
template <typename RC>
SX_HD void sx_shell_split(i32* c1, i32* c2, RC* rc, i32 p, const u16* table, const SxCdf* cdf) {
    if (p > 0) {
        *c1 = sx_rc_dec(rc, &table[cdf->shell_offsets[p]], p >> 1);
        *c2 = p - *c1;
    } else {
        *c1 = 0;
        *c2 = 0;
    }
}
template <typename RC, typename QT>
SX_HD void sx_shell_decoder(QT* q, RC* rc, i32 pulses4, const SxCdf* cdf) {
    i32 p3[2], p2[4], p1[8], a, b;
    sx_shell_split(&p3[0], &p3[1], rc, pulses4, cdf->cdf_shell3, cdf);
    sx_shell_split(&p2[0], &p2[1], rc, p3[0], cdf->cdf_shell2, cdf);
    sx_shell_split(&p1[0], &p1[1], rc, p2[0], cdf->cdf_shell1, cdf);
    sx_shell_split(&a, &b, rc, p1[0], cdf->cdf_shell0, cdf); q[0] = (QT)a; q[1] = (QT)b;
    sx_shell_split(&a, &b, rc, p1[1], cdf->cdf_shell0, cdf); q[2] = (QT)a; q[3] = (QT)b;
    sx_shell_split(&p1[2], &p1[3], rc, p2[1], cdf->cdf_shell1, cdf);
    sx_shell_split(&a, &b, rc, p1[2], cdf->cdf_shell0, cdf); q[4] = (QT)a; q[5] = (QT)b;
    sx_shell_split(&a, &b, rc, p1[3], cdf->cdf_shell0, cdf); q[6] = (QT)a; q[7] = (QT)b;
    sx_shell_split(&p2[2], &p2[3], rc, p3[1], cdf->cdf_shell2, cdf);
    sx_shell_split(&p1[4], &p1[5], rc, p2[2], cdf->cdf_shell1, cdf);
    sx_shell_split(&a, &b, rc, p1[4], cdf->cdf_shell0, cdf); q[8] = (QT)a; q[9] = (QT)b;
    sx_shell_split(&a, &b, rc, p1[5], cdf->cdf_shell0, cdf); q[10] = (QT)a; q[11] = (QT)b;
    sx_shell_split(&p1[6], &p1[7], rc, p2[3], cdf->cdf_shell1, cdf);
    sx_shell_split(&a, &b, rc, p1[6], cdf->cdf_shell0, cdf); q[12] = (QT)a; q[13] = (QT)b;
    sx_shell_split(&a, &b, rc, p1[7], cdf->cdf_shell0, cdf); q[14] = (QT)a; q[15] = (QT)b;
}

// SKP_Silk_decode_pulses (SKP_Silk_decode_pulses.c:33) + SKP_Silk_decode_signs (code_signs.c:64).
// tmp: 2 * SX_FRAME/16 words of per-description scratch (LDS)
// returns the rate level index.  QT / TT: storage of the pulses / of the per-block scratch -- int16 / int32 in the decoder proper; the
// extraction kernel keeps them in bytes (a lane's LDS row: what a well-formed stream holds always fits) and reports in *narrow
// whether anything did not fit (the record is then not used)
template <typename RC, typename QT, typename TT>
SX_HD i32 sx_decode_pulses(RC* rc, int sigtype, int QuantOffsetType, QT* q, const SxCdf* cdf, TT* tmp, i32* narrow) {
    SX_IN_LDS(q); SX_IN_LDS(cdf); SX_IN_LDS(tmp);
    const int iter = SX_FRAME / 16;
    TT *sum_pulses = tmp, *nLshifts = tmp + SX_FRAME / 16;
    const i32 RateLevelIndex = sx_rc_dec(rc, &cdf->cdf_rate_levels[sigtype * 10], T_CDF_MID_RATE_LEVELS);
    const u16* cdf_ptr = &cdf->cdf_pulses_per_block[RateLevelIndex * 21];
    for (int i = 0; i < iter; i++) {
        i32 nl = 0;
        i32 sp = sx_rc_dec(rc, cdf_ptr, T_CDF_MID_PULSES_PER_BLOCK);
        while (sp == 18 + 1) {
            nl++;
            sp = sx_rc_dec(rc, &cdf->cdf_pulses_per_block[9 * 21], T_CDF_MID_PULSES_PER_BLOCK);
            if (rc->error) break;   // (reference would spin on the zero returned after an error only until != 19; 0 != 19)
        }
        if (sizeof(TT) == 1 && nl > 7) *narrow = 1;
        nLshifts[i] = (TT)nl;
        sum_pulses[i] = (TT)sp;
    }
    const u32 p_lsb = cdf->cdf_lsb[1];
    for (int i = 0; i < iter; i++) {
        if (sum_pulses[i] > 0) {
            sx_shell_decoder(&q[i * 16], rc, sum_pulses[i], cdf);
        } else {
            for (int k = 0; k < 16; k++) q[i * 16 + k] = (QT)0;
        }
    }
    for (int i = 0; i < iter; i++) {
        const int nLS = nLshifts[i];
        if (nLS > 0) {
            for (int k = 0; k < 16; k++) {
                i32 abs_q = q[i * 16 + k];
                for (int j = 0; j < nLS; j++) {
                    abs_q = sx_shl(abs_q, 1);
                    abs_q += sx_rc_dec_bin(rc, p_lsb);
                }
                if (sizeof(QT) == 1 && abs_q > 127) *narrow = 1;
                q[i * 16 + k] = (QT)abs_q;
            }
        }
    }
    // signs
    const u32 p_sign = cdf->cdf_sign[sx_smulbb(10 - 1, (sigtype << 1) + QuantOffsetType) + RateLevelIndex];
    for (int i = 0; i < SX_FRAME; i++) {
        const i32 v = q[i];
        if (v > 0) {
            i32 data = sx_rc_dec_bin(rc, p_sign);
            q[i] = (QT)(v * ((data << 1) - 1));
        }
    }
    return RateLevelIndex;
}

// SKP_Silk_NLSF_MSVQ_decode, SKP_Silk_NLSF_MSVQ_decode.c:31 (order 10, 6 stages); codebook / spacing table wherever the caller keeps them
template <typename IDX>
SX_HD void sx_nlsf_msvq_decode_cb(i32* pNLSF_Q15, const IDX* idx, const i16* cb, const i32* nvec, const i32* ndelta_min_Q15) {
    const i16* e = &cb[idx[0] * SX_LPC];
    for (int i = 0; i < SX_LPC; i++) pNLSF_Q15[i] = e[i];
    int base = nvec[0];
    for (int s = 1; s < SX_NLSF_STAGES; s++) {
        e = &cb[(base + idx[s]) * SX_LPC];
        for (int i = 0; i < SX_LPC; i++) pNLSF_Q15[i] += e[i];
        base += nvec[s];
    }
    sx_nlsf_stabilize(pNLSF_Q15, ndelta_min_Q15, SX_LPC);
}
SX_HD void sx_nlsf_msvq_decode(i32* pNLSF_Q15, int sigtype, const i32* idx) {
    const i32 nvec0[SX_NLSF_STAGES] = T_NLSF_CB0_NVEC, nvec1[SX_NLSF_STAGES] = T_NLSF_CB1_NVEC;
    sx_nlsf_msvq_decode_cb(pNLSF_Q15, idx, sigtype == 0 ? T_nlsf_cb0_Q15 : T_nlsf_cb1_Q15, sigtype == 0 ? nvec0 : nvec1,
                           sigtype == 0 ? T_nlsf_cb0_ndelta_min_Q15 : T_nlsf_cb1_ndelta_min_Q15);
}

// SKP_Silk_decode_parameters, SKP_Silk_decode_parameters.c:31 (fullDecoding = 1, fs pinned to 8 kHz), in two steps:
//   sx_extract_parameters   reads the frame's symbols off the range coder, in coding order, into SxFrameSyms (+ the pulses).  Which
//                           tables it uses depends only on symbols of the same packet (signal type, PER index, the previous frame's
//                           type of the SAME packet), never on the decoder's history: the packets of a stream can be extracted
//                           independently of each other (sx_extract_desc, the batch path's extraction kernel).
//   sx_dequant_parameters   turns the symbols into the control block: gains (running index), NLSF vectors (interpolated with the
//                           previous frame's), pitch lags, LTP taps -- the part that needs the description's history.
// The reference interleaves the two; no symbol read depends on the arithmetic in between, and after a range-coder error every
// later symbol reads as 0 either way, so the control block and the state come out the same.
// typeOffsetPrev: signal type / offset symbol of the packet's previous frame (read when nFramesDecoded != 0, always written);
// q / tmp: pulses and pulse-decoder scratch (LDS); dbg: first-failure trace record or NULL
template <typename RC, typename QT, typename TT>
SX_HD void sx_extract_parameters(int nFramesDecoded, i32* typeOffsetPrev, i32* dbg, RC* rc_io, QT* q, int kDesp, int useMDIndex, const SxCdf* cdf,
                                 SxFrameSyms* y, TT* tmp, i32* narrow) {
    SX_IN_LDS(q); SX_IN_LDS(cdf); SX_IN_LDS(tmp);
    RC rc_local = *rc_io;
    RC* rc = &rc_local;
    i32 Ix;
    y->fs_bad = 0;
    y->MDIndex = 0;
    if (nFramesDecoded == 0) {
        if (useMDIndex == 1) y->MDIndex = sx_rc_dec(rc, cdf->cdf_mdindex, T_CDF_MID_MDINDEX);
        Ix = sx_rc_dec(rc, cdf->cdf_fs, T_CDF_MID_FS);
        if (Ix != (SX_FS_KHZ == 8 ? 0 : 2)) {  // index into {8, 12, 16, 24} kHz: this build decodes ONE internal rate (reference: decoder_set_fs)
            if (!rc->error) rc->error = SX_RC_ILLEGAL_SAMPLING_RATE;
            y->fs_bad = 1;
            y->error = rc->error;
            y->bufferLength = rc->bufferLength;
            *rc_io = rc_local;
            return;
        }
        Ix = sx_rc_dec(rc, cdf->cdf_type_offset, T_CDF_MID_TYPE_OFFSET);
    } else {
        Ix = sx_rc_dec(rc, &cdf->cdf_type_offset_joint[*typeOffsetPrev * 5], T_CDF_MID_TYPE_OFFSET);
    }
    SX_TRACE(1);
    y->typeOffset = Ix;
    *typeOffsetPrev = Ix;
    const int sigtype = Ix >> 1, QuantOffsetType = Ix & 1;

    if (nFramesDecoded == 0) y->GainsIndices[0] = sx_rc_dec(rc, &cdf->cdf_gain[sigtype * 65], T_CDF_MID_GAIN);
    else y->GainsIndices[0] = sx_rc_dec(rc, cdf->cdf_delta_gain, T_CDF_MID_DELTA_GAIN);
    for (int i = 1; i < SX_NB_SUBFR; i++) y->GainsIndices[i] = sx_rc_dec(rc, cdf->cdf_delta_gain, T_CDF_MID_DELTA_GAIN);
    y->DeltaGainIndices = 0;
    if (nFramesDecoded == 0) y->DeltaGainIndices = sx_rc_dec(rc, cdf->cdf_md_delta_gain, T_CDF_MID_MD_DELTA_GAIN);
    SX_TRACE(2);
    // NLSF path: 6 stages, per-stage CDFs laid out back to back (nvec+1 entries each)
    {
        const i32 nvec0[SX_NLSF_STAGES] = T_NLSF_CB0_NVEC, nvec1[SX_NLSF_STAGES] = T_NLSF_CB1_NVEC;
        const i32* nvec = sigtype == 0 ? nvec0 : nvec1;
        const u16* ncdf = sigtype == 0 ? cdf->nlsf_cb0_cdf : cdf->nlsf_cb1_cdf;
        const i32* mid = sigtype == 0 ? T_nlsf_cb0_cdf_mid : T_nlsf_cb1_cdf_mid;
        int off = 0;
        for (int s = 0; s < SX_NLSF_STAGES; s++) {
            y->NLSFIndices[s] = sx_rc_dec(rc, ncdf + off, mid[s]);
            off += nvec[s] + 1;
        }
    }
    SX_TRACE(3);
    sx_nlsf_msvq_decode(y->NLSF_Q15, sigtype, y->NLSFIndices);
    y->NLSFInterpCoef_Q2 = sx_rc_dec(rc, cdf->cdf_nlsf_interp, T_CDF_MID_NLSF_INTERP);
    y->lagIx = y->conIx = y->PERIndex = y->LTPscaleIx = 0;
    for (int k = 0; k < SX_NB_SUBFR; k++) y->LTPIx[k] = 0;
    if (sigtype == 0) {
        y->lagIx = sx_rc_dec(rc, cdf->cdf_pitch_lag, T_CDF_MID_PITCH_LAG);
        y->conIx = sx_rc_dec(rc, cdf->cdf_pitch_contour, T_CDF_MID_PITCH_CONTOUR);
        y->PERIndex = sx_rc_dec(rc, cdf->cdf_ltp_per, T_CDF_MID_LTP_PER);
        const u16* gcdf = y->PERIndex == 0 ? cdf->cdf_ltp_gain0 : (y->PERIndex == 1 ? cdf->cdf_ltp_gain1 : cdf->cdf_ltp_gain2);
        for (int k = 0; k < SX_NB_SUBFR; k++) y->LTPIx[k] = sx_rc_dec(rc, gcdf, T_cdf_mid_ltp_gain[y->PERIndex]);
        y->LTPscaleIx = sx_rc_dec(rc, cdf->cdf_ltpscale, T_CDF_MID_LTPSCALE);
    }
    SX_TRACE(4);
    y->Seed = sx_rc_dec(rc, cdf->cdf_seed, T_CDF_MID_SEED);
    SX_TRACE(5);
    y->RateLevelIndex = sx_decode_pulses(rc, sigtype, QuantOffsetType, q, cdf, tmp, narrow);
    SX_TRACE(6);
    y->vadFlag = sx_rc_dec(rc, cdf->cdf_vadflag, T_CDF_MID_VADFLAG);
    y->FrameTermination = sx_rc_dec(rc, cdf->cdf_frame_term, T_CDF_MID_FRAME_TERM);

    i32 nBytesUsed;
    sx_rc_length_bits(rc->bufferIx, rc->range_Q16, &nBytesUsed);
    const i32 left = rc->bufferLength - nBytesUsed;
    y->left = left;
    if (left < 0) rc->error = SX_RC_READ_BEYOND_BUFFER;
    if (left == 0) sx_rc_check_after_decoding(rc);
    y->error = rc->error;
    y->bufferLength = rc->bufferLength;
    SX_TRACE(7);
    *rc_io = rc_local;
}

// md: the state of the description slot being decoded; c: its control block; lane_out: {vadFlag, FrameTermination, bytes left, coder
// error}; nlsf_out[2][SX_LPC]: interpolated / final NLSF vector -- their conversion to prediction coefficients only matters for the
// description that is used and is done by the caller (all LDS)
// (sx_dequant_parameters_any: the same for records anywhere -- the extraction kernel runs it on private / HBM data)
template <bool IN_LDS>
SX_HD void sx_dequant_parameters_t(const SxFrameSyms* y, int nFramesDecoded, int first_frame_after_reset, int useMDIndex, SxDecDesc* md, SxDecCtrl* c,
                                 i32* lane_out, i32* nlsf_out) {
    if (IN_LDS) { SX_IN_LDS(y); SX_IN_LDS(md); SX_IN_LDS(c); SX_IN_LDS(lane_out); SX_IN_LDS(nlsf_out); }
    if (y->fs_bad) { lane_out[3] = y->error; return; }
    i32 *pNLSF0_Q15 = nlsf_out, *pNLSF_Q15 = nlsf_out + SX_LPC;
    if (nFramesDecoded == 0 && useMDIndex == 1) c->MDIndex = y->MDIndex;
    c->sigtype = y->typeOffset >> 1;
    c->QuantOffsetType = y->typeOffset & 1;
    md->typeOffsetPrev = y->typeOffset;
    i32 DeltaGainIndices;
    if (nFramesDecoded == 0) {
        DeltaGainIndices = y->DeltaGainIndices;
        md->prevDeltaGainIndex = DeltaGainIndices;
    } else {
        DeltaGainIndices = md->prevDeltaGainIndex;
    }
    sx_gains_dequant(c->Gains_Q16, y->GainsIndices, &md->LastGainIndex, nFramesDecoded, DeltaGainIndices, &c->DeltaGains_Q16);
    for (int i = 0; i < SX_LPC; i++) pNLSF_Q15[i] = y->NLSF_Q15[i];
    c->NLSFInterpCoef_Q2 = y->NLSFInterpCoef_Q2;
    if (first_frame_after_reset == 1) c->NLSFInterpCoef_Q2 = 4;
    if (c->NLSFInterpCoef_Q2 < 4) {
        for (int i = 0; i < SX_LPC; i++)
            pNLSF0_Q15[i] = md->prevNLSF_Q15[i] + (sx_mul(c->NLSFInterpCoef_Q2, pNLSF_Q15[i] - md->prevNLSF_Q15[i]) >> 2);
    }
    for (int i = 0; i < SX_LPC; i++) md->prevNLSF_Q15[i] = pNLSF_Q15[i];
    if (c->sigtype == 0) {
        const i32 lag = 2 * SX_FS_KHZ + y->lagIx;   // SKP_Silk_decode_pitch.c:43-58 (8 kHz: stage-2 contours, above: stage-3 contours)
        for (int i = 0; i < SX_NB_SUBFR; i++) c->pitchL[i] = lag + T_pitch_cb_dec[i * SX_PITCH_CB_N + y->conIx];
        c->PERIndex = y->PERIndex;
        const i16* cbk = c->PERIndex == 0 ? T_ltp_vq0_Q14 : (c->PERIndex == 1 ? T_ltp_vq1_Q14 : T_ltp_vq2_Q14);
        for (int k = 0; k < SX_NB_SUBFR; k++) {
            const i32 Ix = y->LTPIx[k];
            for (int i = 0; i < SX_LTP_ORDER; i++) c->LTPCoef_Q14[k * SX_LTP_ORDER + i] = cbk[Ix * SX_LTP_ORDER + i];
        }
        c->LTP_scale_Q14 = T_ltp_scales_Q14[y->LTPscaleIx];
    } else {
        for (int i = 0; i < SX_NB_SUBFR; i++) c->pitchL[i] = 0;
        for (int i = 0; i < SX_LTP_ORDER * SX_NB_SUBFR; i++) c->LTPCoef_Q14[i] = 0;
        c->PERIndex = 0;
        c->LTP_scale_Q14 = 0;
    }
    c->Seed = y->Seed;
    c->RateLevelIndex = y->RateLevelIndex;
    lane_out[0] = y->vadFlag;
    lane_out[1] = y->FrameTermination;
    lane_out[2] = y->left;
    lane_out[3] = y->error;
}
SX_HD void sx_dequant_parameters(const SxFrameSyms* y, int nFramesDecoded, int first_frame_after_reset, int useMDIndex, SxDecDesc* md, SxDecCtrl* c,
                                 i32* lane_out, i32* nlsf_out) {
    sx_dequant_parameters_t<true>(y, nFramesDecoded, first_frame_after_reset, useMDIndex, md, c, lane_out, nlsf_out);
}

// (the de-quantisation as a real call of its own: inlined into sx_decode_parameters the pair needs 129 vector registers -- one more
// than the four-waves-per-SIMD budget of the kernels that call it, i.e. 136 allocated and three waves per SIMD: the decoder's sixteen
// workgroups per compute unit then take two rounds)
SX_FN void sx_dequant_parameters_call(const SxFrameSyms* y, int nFramesDecoded, int first_frame_after_reset, int useMDIndex, SxDecDesc* md, SxDecCtrl* c,
                                      i32* lane_out, i32* nlsf_out) {
    SX_IN_LDS(y); SX_IN_LDS(md); SX_IN_LDS(c); SX_IN_LDS(lane_out); SX_IN_LDS(nlsf_out);
    sx_dequant_parameters(y, nFramesDecoded, first_frame_after_reset, useMDIndex, md, c, lane_out, nlsf_out);
}
// the two steps in a row for one description slot of the wave-per-stream decoder (a real call: two call sites per kernel)
SX_FN void sx_decode_parameters(int nFramesDecoded, int first_frame_after_reset, SxDecDesc* md, i32* dbg, SxDecCtrl* c, SxRangeDec* rc_io,
                                i16* q, int kDesp, int useMDIndex, const SxCdf* cdf, i32* lane_out, i32* nlsf_out, i32* tmp, SxFrameSyms* y) {
    SX_IN_LDS(md); SX_IN_LDS(c); SX_IN_LDS(q); SX_IN_LDS(cdf); SX_IN_LDS(lane_out); SX_IN_LDS(nlsf_out); SX_IN_LDS(tmp); SX_IN_LDS(y);
    i32 top = md->typeOffsetPrev, narrow = 0;
    sx_extract_parameters(nFramesDecoded, &top, dbg, rc_io, q, kDesp, useMDIndex, cdf, y, tmp, &narrow);
    sx_dequant_parameters_call(y, nFramesDecoded, first_frame_after_reset, useMDIndex, md, c, lane_out, nlsf_out);
}

#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64
// All-pole recursion with ONE TAP PER LANE, in its transposed form (see sx_decode_core): lane j of every 16-lane row holds coefficient j,
// pre-shifted (aj; zero past the order), and the partial sum r_j = sum over m of a_(j+m) h[t - m], seeded from the history the caller
// hands over (hj = the output of time t0 - 1 - j).  step(i, p) turns the prediction p = r_0 of sample i into the filter value v (and
// stores whatever the caller wants of it); the new state sample is v << 4, saturating if SAT.  Returns the lane's final history (the
// output of time t_end - j): kept beside the sums by a one-lane shift of the row per sample.
#define SX_IIR_SEED_(m_) r = sx_add(r, sx_smulw_pre(__builtin_amdgcn_update_dpp(0, hj, 0x150 + (m_), 0xF, 0xF, false), \
                                                   (m_) == 0 ? aj : __builtin_amdgcn_update_dpp(0, aj, 0x100 + ((m_) == 0 ? 1 : (m_)), 0xF, 0xF, true)));
#define SX_IIR_SEED(r, hj, aj) { SX_IIR_SEED_(0) SX_IIR_SEED_(1) SX_IIR_SEED_(2) SX_IIR_SEED_(3) SX_IIR_SEED_(4) SX_IIR_SEED_(5) SX_IIR_SEED_(6) SX_IIR_SEED_(7) \
                                 SX_IIR_SEED_(8) SX_IIR_SEED_(9) SX_IIR_SEED_(10) SX_IIR_SEED_(11) SX_IIR_SEED_(12) SX_IIR_SEED_(13) SX_IIR_SEED_(14) SX_IIR_SEED_(15) }
template <bool SAT, typename F>
__device__ __forceinline__ i32 sx_iir_rows(i32 hj, const i32 aj, int n, F step) {
    const int j = SX_LANE & 15;
    i32 r = 0;
    SX_IIR_SEED(r, hj, aj)
    for (int i = 0; i < n; i++) {
        const i32 p = __builtin_amdgcn_update_dpp(0, r, 0x150, 0xF, 0xF, false);            // row_newbcast:0
        const i32 v = step(i, p);
        const i32 hn = SAT ? sx_lshift_sat32(v, 4) : sx_shl(v, 4);
        r = sx_add(sx_smulw_pre(hn, aj), __builtin_amdgcn_update_dpp(0, r, 0x101, 0xF, 0xF, true));   // row_shl:1, zero into the row's last lane
        const i32 sh = __builtin_amdgcn_update_dpp(0, hj, 0x111, 0xF, 0xF, true);          // row_shr:1
        hj = j == 0 ? hn : sh;
    }
    return hj;
}
#endif

// SKP_Silk_decode_core, SKP_Silk_decode_core.c:43.  exc_Q10 = st->exc_Q10; writes outBuf[160..320).
SX_FN void sx_decode_core(SxDecState* st, SxDecWork* w, i16* xq) {
    SX_IN_LDS(st); SX_IN_LDS(w); SX_IN_LDS(xq);
    SxDecCtrl* c = &w->ctrl;
    const int NLSF_interpolation_flag = c->NLSFInterpCoef_Q2 < 4 ? 1 : 0;
    i32* pexc_Q10 = st->exc_Q10;
    i32* pres_Q10 = w->res_Q10;
    i16* pxq = &st->outBuf[SX_FRAME];
    int sLTP_buf_idx = SX_FRAME;
    int lag = 0;
    SX_PAR(i, SX_MAX_LPC) w->u.syn.sLPC_Q14[i] = st->sLPC_Q14[i];
    wv_sync();
    for (int k = 0; k < SX_NB_SUBFR; k++) {
        const i16* A_Q12 = w->PredCoef_Q12[k >> 1];
        i16* B_Q14 = &c->LTPCoef_Q14[k * SX_LTP_ORDER];
        i32 Gain_Q16 = c->Gains_Q16[k];
        int sigtype = c->sigtype;
        i32 inv_gain_Q16 = sx_inverse32_varQ(sx_max(Gain_Q16, 1), 32);
        inv_gain_Q16 = sx_min(inv_gain_Q16, 32767);
        i32 gain_adj_Q16 = 1 << 16;
        if (inv_gain_Q16 != st->prev_inv_gain_Q16) gain_adj_Q16 = sx_div32_varQ(inv_gain_Q16, st->prev_inv_gain_Q16, 16);

        if (st->lossCnt && st->prev_sigtype == 0 && c->sigtype == 1 && k < 2) {
            for (int i = 0; i < SX_LTP_ORDER; i++) B_Q14[i] = 0;
            B_Q14[SX_LTP_ORDER / 2] = (i16)(1 << 12);
            sigtype = 0;
            c->pitchL[k] = st->lagPrev;
        }
        if (sigtype == 0) {
            lag = c->pitchL[k];
            if ((k & (3 - (NLSF_interpolation_flag << 1))) == 0) {
                // re-whitening: only the last lag+2 outputs of the reference's MA_Prediction run are
                // consumed, and every one of them has a complete 10-sample history, so the filter is
                // evaluated directly per output sample (wrapping sums: tap order irrelevant)
                const i16* in = &st->outBuf[k * SX_SUBFR];   // in[j] == outBuf[j + k*40], j < 160
                i32 inv_gain_Q32 = sx_shl(inv_gain_Q16, 16);
                if (k == 0) inv_gain_Q32 = sx_shl(sx_smulwb(inv_gain_Q32, c->LTP_scale_Q14), 2);
                const int n = lag + SX_LTP_ORDER / 2;
                SX_PAR(i, n) {
                    int j = SX_FRAME - 1 - i;
                    i32 acc = 0;
                    for (int d = 0; d < SX_LPC; d++) acc = sx_smlabb(acc, in[j - 1 - d], A_Q12[d]);
                    i32 o = sx_rshift_round(sx_sub(sx_shl((i32)in[j], 12), acc), 12);
                    st->sLTP_Q16[sLTP_buf_idx - i - 1] = sx_smulwb(inv_gain_Q32, sx_sat16(o));
                }
                wv_sync();
            } else if (gain_adj_Q16 != (1 << 16)) {
                const int n = lag + SX_LTP_ORDER / 2;
                SX_PAR(i, n) st->sLTP_Q16[sLTP_buf_idx - i - 1] = sx_smulww(gain_adj_Q16, st->sLTP_Q16[sLTP_buf_idx - i - 1]);
                wv_sync();
            }
        }
        SX_PAR(i, SX_MAX_LPC) w->u.syn.sLPC_Q14[i] = sx_smulww(gain_adj_Q16, w->u.syn.sLPC_Q14[i]);
        wv_sync();
        st->prev_inv_gain_Q16 = inv_gain_Q16;

        if (sigtype == 0) {
            // long-term prediction (decode_core.c:150-186): sample i reads the synthesis buffer at most up to i - lag + 2, so
            // runs of lag - 2 consecutive samples are independent and go to the lanes
            const i32 b0 = sx_pre16(B_Q14[0]), b1 = sx_pre16(B_Q14[1]), b2 = sx_pre16(B_Q14[2]), b3 = sx_pre16(B_Q14[3]),
                      b4 = sx_pre16(B_Q14[4]);
            const int run = sx_min(sx_max(lag - SX_LTP_ORDER / 2, 1), SX_SUBFR);
            for (int s0 = 0; s0 < SX_SUBFR; s0 += run) {
                const int n = sx_min(run, SX_SUBFR - s0);
                SX_PAR(ii, n) {
                    const int i = s0 + ii;
                    const i32* pl = &st->sLTP_Q16[sLTP_buf_idx + i - lag + SX_LTP_ORDER / 2];
                    i32 p = sx_smulw_pre(pl[0], b0);
                    p = sx_smlaw_pre(p, pl[-1], b1);
                    p = sx_smlaw_pre(p, pl[-2], b2);
                    p = sx_smlaw_pre(p, pl[-3], b3);
                    p = sx_smlaw_pre(p, pl[-4], b4);
                    i32 r = sx_add(pexc_Q10[i], sx_rshift_round(p, 4));
                    pres_Q10[i] = r;
                    st->sLTP_Q16[sLTP_buf_idx + i] = sx_shl(r, 6);
                }
                wv_sync();
            }
            sLTP_buf_idx += SX_SUBFR;
        } else {
            SX_PAR(i, SX_SUBFR) pres_Q10[i] = pexc_Q10[i];
            wv_sync();
        }
        // short-term prediction (decode_core.c:188-288), serial recursion: coefficients (pre-shifted for the high-word multiply)
        // and the last SX_LPC outputs live in registers; slot r of the ring holds the sample of time t = r (mod SX_LPC).
        // The high band's synthesis filter (AGR_BWE_LPC_synthesizer.c, sx_hb_lpc_synthesis) is the same kind of recursion over the
        // same SX_SUBFR samples, fed with the excitation of this very subframe, and independent of the low band's: it runs in the
        // SAME instruction stream on lane 1 (order SX_HB_LPC <= SX_LPC: its two / eight oldest ring slots meet zero coefficients),
        // with its own input scaling, saturation and output; every other lane computes the low band (wave-uniform, as before).
        {
            const SxHbParams* hp = &w->hbp;
            const int piggy = hp->piggy;
            const int hf = st->hb_joint ? 0 : hp->frame;                                   // high-band frame / its subframe under these samples
            const int hk = st->hb_joint ? (hp->frame * 2 + (k >> 1)) : k;
            const i32 hbGain_Q16 = sx_mul(-2867, (i32)hp->gain[hf][hk]);
            i16* hb_out = &w->hi_out[hp->frame * SX_FRAME + k * SX_SUBFR];
#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64
            // GPU form: the TRANSPOSED recursion, one tap per lane.  Lane (row, j) of the wavefront holds coefficient a_j of the row's filter
            // (row 1 = the high band's, every other row a copy of the low band's) and the partial sum
            //     r_j[t] = sum over m >= 0 of a_(j+m) h[t - m]            (h: the filter's outputs, scaled; wrapping 32-bit sums)
            // which obeys  r_j[t] = a_j h[t] + r_(j+1)[t - 1]  and whose lane 0 IS the prediction of the next sample, p[t + 1] = r_0[t].
            // A sample is then: lane 0's sum to the row (one DPP row broadcast), the role's output arithmetic, ONE high-word multiply
            // and ONE add that takes the neighbour's sum (DPP row_shl:1) -- ~11 instructions.  (Until round 6 lane j held the output of time
            // t - 1 - j and every sample summed sixteen products across the row: a multiply, four dependent DPP adds with their wait states,
            // a one-lane shift of the row: ~18 instructions.  The products are the same sixteen, each rounded on its own; only the order of
            // the wrapping additions differs.)
            {
                const int row = SX_LANE >> 4, j = SX_LANE & 15;
                const bool hbl = piggy && row == 1;
                const i32 al = j < SX_LPC ? sx_pre16(A_Q12[j < SX_LPC ? j : 0]) : 0;
                const i32 ah = j < SX_HB_LPC ? sx_pre16(hp->lpc[hf][j < SX_HB_LPC ? j : 0]) : 0;
                const i32 hl = w->u.syn.sLPC_Q14[SX_MAX_LPC - 1 - j];                    // (all 16 lanes carry true history; taps past the order meet zeros)
                const i32 hh = j < SX_HB_LPC ? hp->S[SX_HB_LPC - 1 - (j < SX_HB_LPC ? j : 0)] : 0;
                const i32 aj = hbl ? ah : al;
                const i32 hj = hbl ? hh : hl;                                            // h[t0 - 1 - j]
                // the sums the subframe starts from: r_j = sum over m of a_(j+m) h[t0 - 1 - m] -- lane m's history to the row, lane j + m's
                // coefficient to lane j (zero past the row's end)
                i32 r = 0;
                SX_IIR_SEED(r, hj, aj)
                // What a sample step does NOT need stays outside the recursion: the high band's excitation is scaled for the whole
                // subframe beforehand (xs: the tail of sLPC_Q14, which only the other builds use), the filter values v are left
                // where their inputs were (the low band's residual, xs), and gain / rounding / saturation of both bands' outputs
                // follow for all samples side by side.  The two roles then differ in a saturating vs a wrapping add and a clamp
                // before the state's shift: selects, no branch.
                i32* xs = &w->u.syn.sLPC_Q14[SX_MAX_LPC];
                const bool hb_zero = hp->lost != 0;
                if (piggy) { SX_PAR(i, SX_SUBFR) xs[i] = sx_smulww(hbGain_Q16, hb_zero ? 0 : pexc_Q10[i]); }
                wv_sync();
                i32* io = hbl ? xs : pres_Q10;
                i32 clo = hbl ? (SX_I32_MIN >> 4) : SX_I32_MIN, chi = hbl ? (SX_I32_MAX >> 4) : SX_I32_MAX;   // lshift_sat32(v, 4) = clamp, then shift
                SX_VEC(clo); SX_VEC(chi);
                for (int i = 0; i < SX_SUBFR; i++) {
                    const i32 p = __builtin_amdgcn_update_dpp(0, r, 0x150, 0xF, 0xF, false);       // row_newbcast:0: r_0 = the prediction of this sample
                    const i32 x = io[i];
                    const i32 v = hbl ? sx_add_sat32(p, x) : sx_add(x, p);
                    const i32 hn = sx_shl(sx_max(sx_min(v, chi), clo), 4);
                    r = sx_add(sx_smulw_pre(hn, aj), __builtin_amdgcn_update_dpp(0, r, 0x101, 0xF, 0xF, true));   // r_j = a_j h[t] + r_(j+1); the row's last lane takes 0
                    io[i] = v;
                }
                wv_sync();
                SX_PAR(t, (piggy ? 2 : 1) * SX_SUBFR) {
                    const bool hb = t >= SX_SUBFR;
                    const int i = hb ? t - SX_SUBFR : t;
                    const i32 o = hb ? xs[i] : sx_smulww(pres_Q10[i], Gain_Q16);
                    (hb ? hb_out : pxq)[i] = (i16)sx_sat16(sx_rshift_round(o, 10));
                }
                // the last SX_MAX_LPC outputs, as the state's history (the next subframe starts from them): from the values just left in io
                static_assert(SX_SUBFR >= SX_MAX_LPC, "a subframe holds the whole history");
                const i32 hlast = sx_shl(sx_max(sx_min(io[SX_SUBFR - 1 - j], chi), clo), 4);
                wv_sync();
                if (row == 0) w->u.syn.sLPC_Q14[SX_MAX_LPC - 1 - j] = hlast;
                if (hbl && j < SX_HB_LPC) w->hbp.S[SX_HB_LPC - 1 - j] = hlast;
            }
#else
#if SX_NLANES == 1
            for (int role = 0; role < (piggy ? 2 : 1); role++) {
                const bool hbl = role == 1;
#else
            {
                const bool hbl = piggy && SX_LANE == 1;
#endif
                i32 a[SX_LPC], h[SX_LPC];
#pragma unroll
                for (int j = 0; j < SX_LPC; j++) {
                    const int jh = j - (SX_LPC - SX_HB_LPC);                                // ring slot j = time -SX_LPC + j
                    const i32 al = sx_pre16(A_Q12[j]), hl = w->u.syn.sLPC_Q14[SX_MAX_LPC - SX_LPC + j];
                    const i32 ah = j < SX_HB_LPC ? sx_pre16(hp->lpc[hf][j < SX_HB_LPC ? j : 0]) : 0;
                    const i32 hh = jh >= 0 ? hp->S[jh >= 0 ? jh : 0] : 0;
                    a[j] = hbl ? ah : al;
                    h[j] = hbl ? hh : hl;
                }
                const i32* in = hbl ? pexc_Q10 : pres_Q10;
                i16* out = hbl ? hb_out : pxq;
                const bool hb_zero = hbl && hp->lost;                                      // high band lost: zero excitation (decode_frame_FIX.c:65)
                for (int i0 = 0; i0 < SX_SUBFR; i0 += SX_LPC) {
#pragma unroll
                    for (int u = 0; u < SX_LPC; u++) {
                        i32 p = 0;
#pragma unroll
                        for (int j = 0; j < SX_LPC; j++) p = sx_smlaw_pre(p, h[(u - 1 - j + 2 * SX_LPC) % SX_LPC], a[j]);
                        const i32 x = hb_zero ? 0 : in[i0 + u];
                        // low band: v = x + p, state v << 4, output sat16(round(v * gain >> 10));
                        // high band: v = sat32(p + x * gain), state sat32(v << 4), output sat16(round(v >> 10))
                        const i32 xs = hbl ? sx_smulww(hbGain_Q16, x) : x;
                        const i32 v = hbl ? sx_add_sat32(p, xs) : sx_add(xs, p);
                        const i32 hn = hbl ? sx_lshift_sat32(v, 4) : sx_shl(v, 4);
                        const i32 o = hbl ? v : sx_smulww(v, Gain_Q16);
                        h[u] = hn;
                        if (!hbl) w->u.syn.sLPC_Q14[SX_MAX_LPC + i0 + u] = hn;
                        out[i0 + u] = (i16)sx_sat16(sx_rshift_round(o, 10));
                    }
                }
                if (hbl) {
#pragma unroll
                    for (int j = 0; j < SX_HB_LPC; j++) w->hbp.S[j] = h[SX_LPC - SX_HB_LPC + j];
                }
            }
            wv_sync();
            SX_PAR(i, SX_MAX_LPC) {
                i32 t = w->u.syn.sLPC_Q14[SX_SUBFR + i];
                w->u.syn.sLPC_Q14[i] = t;
            }
#endif
        }
        wv_sync();
        pexc_Q10 += SX_SUBFR;
        pres_Q10 += SX_SUBFR;
        pxq += SX_SUBFR;
    }
    SX_PAR(i, SX_MAX_LPC) st->sLPC_Q14[i] = w->u.syn.sLPC_Q14[i];
    SX_PAR(i, SX_FRAME) xq[i] = st->outBuf[SX_FRAME + i];
    wv_sync();
}

// SKP_Silk_PLC_update, SKP_Silk_PLC.c:75
SX_HD void sx_plc_update(SxDecState* st, SxDecCtrl* c, const i16* PredCoef1_Q12) {
    SxPLC* p = &st->plc;
    st->prev_sigtype = c->sigtype;
    i32 LTP_Gain_Q14 = 0;
    if (c->sigtype == 0) {
        for (int j = 0; j * SX_SUBFR < c->pitchL[SX_NB_SUBFR - 1]; j++) {
            i32 t = 0;
            for (int i = 0; i < SX_LTP_ORDER; i++) t += c->LTPCoef_Q14[(SX_NB_SUBFR - 1 - j) * SX_LTP_ORDER + i];
            if (t > LTP_Gain_Q14) {
                LTP_Gain_Q14 = t;
                for (int i = 0; i < SX_LTP_ORDER; i++) p->LTPCoef_Q14[i] = c->LTPCoef_Q14[(SX_NB_SUBFR - 1 - j) * SX_LTP_ORDER + i];
                p->pitchL_Q8 = sx_shl(c->pitchL[SX_NB_SUBFR - 1 - j], 8);
            }
        }
        // USE_SINGLE_TAP (SKP_Silk_PLC.h:38)
        for (int i = 0; i < SX_LTP_ORDER; i++) p->LTPCoef_Q14[i] = 0;
        p->LTPCoef_Q14[SX_LTP_ORDER / 2] = (i16)LTP_Gain_Q14;
        if (LTP_Gain_Q14 < 11469) {
            i32 scale_Q10 = (11469 << 10) / sx_max(LTP_Gain_Q14, 1);
            for (int i = 0; i < SX_LTP_ORDER; i++) p->LTPCoef_Q14[i] = (i16)(sx_smulbb(p->LTPCoef_Q14[i], scale_Q10) >> 10);
        } else if (LTP_Gain_Q14 > 15565) {
            i32 scale_Q14 = (15565 << 14) / sx_max(LTP_Gain_Q14, 1);
            for (int i = 0; i < SX_LTP_ORDER; i++) p->LTPCoef_Q14[i] = (i16)(sx_smulbb(p->LTPCoef_Q14[i], scale_Q14) >> 14);
        }
    } else {
        p->pitchL_Q8 = sx_shl(sx_smulbb(SX_FS_KHZ, 18), 8);
        for (int i = 0; i < SX_LTP_ORDER; i++) p->LTPCoef_Q14[i] = 0;
    }
    for (int i = 0; i < SX_LPC; i++) p->prevLPC_Q12[i] = PredCoef1_Q12[i];
    p->prevLTP_scale_Q14 = (i16)c->LTP_scale_Q14;
    for (int i = 0; i < SX_NB_SUBFR; i++) p->prevGain_Q16[i] = c->Gains_Q16[i];
}

// SKP_Silk_PLC_conceal, SKP_Silk_PLC.c:146
SX_FN void sx_plc_conceal(SxDecState* st, SxDecWork* w, i16* signal) {
    SX_IN_LDS(st); SX_IN_LDS(w); SX_IN_LDS(signal);
    SxPLC* p = &st->plc;
    SxDecCtrl* c = &w->ctrl;
    i16* exc_buf = w->u.syn.tmp16;
    i32* sig_Q10 = w->res_Q10;
    // shift LTP buffer (source and destination halves do not overlap)
    SX_PAR(i, SX_FRAME) st->sLTP_Q16[i] = st->sLTP_Q16[SX_FRAME + i];
    wv_sync();
    sx_bwexpander(p->prevLPC_Q12, SX_LPC, 64880);
    SX_PAR(t, 2 * SX_SUBFR) {
        int k = 2 + t / SX_SUBFR, i = t % SX_SUBFR;
        exc_buf[t] = (i16)(sx_smulww(st->exc_Q10[i + k * SX_SUBFR], p->prevGain_Q16[k]) >> 10);
    }
    wv_sync();
    i32 energy1, energy2, shift1, shift2;
    sx_sum_sqr_shift(&energy1, &shift1, exc_buf, SX_SUBFR, 0);
    sx_sum_sqr_shift(&energy2, &shift2, &exc_buf[SX_SUBFR], SX_SUBFR, 0);
    const i32* rand_ptr;
    if ((energy1 >> shift2) < (energy2 >> shift1)) rand_ptr = &st->exc_Q10[sx_max(0, 3 * SX_SUBFR - 128)];
    else rand_ptr = &st->exc_Q10[sx_max(0, SX_FRAME - 128)];

    i16* B_Q14 = p->LTPCoef_Q14;
    i32 rand_scale_Q14 = p->randScale_Q14;
    const int att = sx_min(1, st->lossCnt);
    const i32 HARM_ATT_Q15[2] = {32440, 31130}, RAND_ATT_V_Q15[2] = {31130, 26214}, RAND_ATT_UV_Q15[2] = {32440, 29491};
    i32 harm_Gain_Q15 = HARM_ATT_Q15[att];
    i32 rand_Gain_Q15 = st->prev_sigtype == 0 ? RAND_ATT_V_Q15[att] : RAND_ATT_UV_Q15[att];
    if (st->lossCnt == 0) {
        rand_scale_Q14 = 1 << 14;
        if (st->prev_sigtype == 0) {
            for (int i = 0; i < SX_LTP_ORDER; i++) rand_scale_Q14 = (i16)(rand_scale_Q14 - B_Q14[i]);
            rand_scale_Q14 = (i16)sx_max(3277, (i16)rand_scale_Q14);
            rand_scale_Q14 = (i16)(sx_smulbb(rand_scale_Q14, p->prevLTP_scale_Q14) >> 14);
        }
        if (st->prev_sigtype == 1) {
            i32 invGain_Q30, down_scale_Q30;
            sx_lpc_inv_pred_gain(&invGain_Q30, p->prevLPC_Q12, SX_LPC);
            down_scale_Q30 = sx_min((1 << 30) >> 3, invGain_Q30);
            down_scale_Q30 = sx_max((1 << 30) >> 8, down_scale_Q30);
            down_scale_Q30 = sx_shl(down_scale_Q30, 3);
            rand_Gain_Q15 = sx_smulwb(down_scale_Q30, rand_Gain_Q15) >> 14;
        }
    }
    i32 rand_seed = p->rand_seed;
    int lag = sx_rshift_round(p->pitchL_Q8, 8);
    int sLTP_buf_idx = SX_FRAME;
    i32* sig_ptr = sig_Q10;
    for (int k = 0; k < SX_NB_SUBFR; k++) {
        i32* pred_lag_ptr = &st->sLTP_Q16[sLTP_buf_idx - lag + SX_LTP_ORDER / 2];
        for (int i = 0; i < SX_SUBFR; i++) {
            rand_seed = sx_rand(rand_seed);
            int idx = (rand_seed >> 25) & 127;
            i32 pr = sx_smulwb(pred_lag_ptr[0], B_Q14[0]);
            pr = sx_smlawb(pr, pred_lag_ptr[-1], B_Q14[1]);
            pr = sx_smlawb(pr, pred_lag_ptr[-2], B_Q14[2]);
            pr = sx_smlawb(pr, pred_lag_ptr[-3], B_Q14[3]);
            pr = sx_smlawb(pr, pred_lag_ptr[-4], B_Q14[4]);
            pred_lag_ptr++;
            i32 e = sx_shl(sx_smulwb(rand_ptr[idx], rand_scale_Q14), 2);
            e = sx_add(e, sx_rshift_round(pr, 4));
            st->sLTP_Q16[sLTP_buf_idx] = sx_shl(e, 6);
            sLTP_buf_idx++;
            sig_ptr[i] = e;
        }
        sig_ptr += SX_SUBFR;
        for (int j = 0; j < SX_LTP_ORDER; j++) B_Q14[j] = (i16)(sx_smulbb(harm_Gain_Q15, B_Q14[j]) >> 15);
        rand_scale_Q14 = (i16)(sx_smulbb(rand_scale_Q14, rand_Gain_Q15) >> 15);
        p->pitchL_Q8 += sx_smulwb(p->pitchL_Q8, 655);
        p->pitchL_Q8 = sx_min(p->pitchL_Q8, sx_shl(sx_smulbb(18, SX_FS_KHZ), 8));
        lag = sx_rshift_round(p->pitchL_Q8, 8);
    }
    // LPC synthesis
    SX_PAR(i, SX_MAX_LPC) w->u.syn.sLPC_Q14[i] = st->sLPC_Q14[i];
    wv_sync();
#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64
    {   // one tap per lane (sx_iir_rows): the whole frame in one run, the last SX_MAX_LPC outputs go back to the state
        const int j = SX_LANE & 15;
        const i32 aj = j < SX_LPC ? sx_pre16(p->prevLPC_Q12[j < SX_LPC ? j : 0]) : 0;
        const i32 hj = sx_iir_rows<false>(w->u.syn.sLPC_Q14[SX_MAX_LPC - 1 - j], aj, SX_FRAME, [&](int i, i32 pr) {
            const i32 v = sx_add(sig_Q10[i], pr);
            sig_Q10[i] = v;
            return v;
        });
        wv_sync();
        if (SX_LANE < 16) w->u.syn.sLPC_Q14[SX_MAX_LPC - 1 - j] = hj;
        wv_sync();
    }
    (void)sig_ptr;
#else
    sig_ptr = sig_Q10;
    for (int k = 0; k < SX_NB_SUBFR; k++) {
        for (int i = 0; i < SX_SUBFR; i++) {
            i32 pr = 0;
            for (int j = 0; j < SX_LPC; j++) pr = sx_smlawb(pr, w->u.syn.sLPC_Q14[SX_MAX_LPC + i - j - 1], p->prevLPC_Q12[j]);
            i32 v = sx_add(sig_ptr[i], pr);
            sig_ptr[i] = v;
            w->u.syn.sLPC_Q14[SX_MAX_LPC + i] = sx_shl(v, 4);
        }
        sig_ptr += SX_SUBFR;
        for (int i = 0; i < SX_MAX_LPC; i++) {
            i32 t = w->u.syn.sLPC_Q14[SX_SUBFR + i];
            w->u.syn.sLPC_Q14[i] = t;
        }
    }
#endif
    SX_PAR(i, SX_MAX_LPC) st->sLPC_Q14[i] = w->u.syn.sLPC_Q14[i];
    SX_PAR(i, SX_FRAME) signal[i] = (i16)sx_sat16(sx_rshift_round(sx_smulww(sig_Q10[i], p->prevGain_Q16[SX_NB_SUBFR - 1]), 10));
    wv_sync();
    p->rand_seed = rand_seed;
    p->randScale_Q14 = (i16)rand_scale_Q14;
    for (int i = 0; i < SX_NB_SUBFR; i++) c->pitchL[i] = lag;
}

// SKP_Silk_PLC, SKP_Silk_PLC.c:43
SX_HD void sx_plc(SxDecState* st, SxDecWork* w, i16* signal, int lost) {
    if (st->plc.fs_kHz != SX_FS_KHZ) {
        st->plc.pitchL_Q8 = SX_FRAME >> 1;   // SKP_Silk_PLC_Reset
        st->plc.fs_kHz = SX_FS_KHZ;
    }
    if (lost) {
        sx_plc_conceal(st, w, signal);
        st->lossCnt++;
    } else {
        sx_plc_update(st, &w->ctrl, w->PredCoef_Q12[1]);
    }
}

// SKP_Silk_PLC_glue_frames, SKP_Silk_PLC.c:363.  `odd` = int16 offset of `signal` from a 4-byte
// aligned base, modulo 2 (sum_sqr_shift's alignment branch); the low-band buffer of the reference is
// a stack array advanced by 160 samples per frame, i.e. always even.  `ramp_len` = the reference's `length` argument: 160,
// except for the first frame ever decoded, where decode_frame.c:303 latched L = 480 before the payload switched the decoder
// to 8 kHz -- after lost leading packets that frame is faded in with the 480-sample slope (the 320 samples past the frame
// only enter the energy, which the zero concealed energy makes irrelevant: any non-zero frame gets gain 0, slope 4096/480).
SX_FN void sx_plc_glue_frames(SxDecState* st, i16* signal, int length, int ramp_len) {
    SX_IN_LDS(st); SX_IN_LDS(signal);
    SxPLC* p = &st->plc;
    if (st->lossCnt) {
        sx_sum_sqr_shift(&p->conc_energy, &p->conc_energy_shift, signal, length, 0);
        p->last_frame_lost = 1;
    } else {
        if (p->last_frame_lost) {
            i32 energy, energy_shift;
            sx_sum_sqr_shift(&energy, &energy_shift, signal, length, 0);
            if (energy_shift > p->conc_energy_shift) p->conc_energy = p->conc_energy >> (energy_shift - p->conc_energy_shift);
            else if (energy_shift < p->conc_energy_shift) energy = energy >> (p->conc_energy_shift - energy_shift);
            if (energy > p->conc_energy) {
                i32 LZ = sx_clz32(p->conc_energy) - 1;
                p->conc_energy = sx_shl(p->conc_energy, LZ);
                energy = energy >> sx_max(24 - LZ, 0);
                i32 frac_Q24 = p->conc_energy / sx_max(energy, 1);
                i32 gain_Q12 = sx_sqrt_approx(frac_Q24);
                i32 slope_Q12 = ((1 << 12) - gain_Q12) / ramp_len;
                // serial ramp (rare path: first good frame after a loss); uniform code, all lanes identical
                for (int i = 0; i < length; i++) {
                    signal[i] = (i16)(sx_mul(gain_Q12, signal[i]) >> 12);
                    gain_Q12 += slope_Q12;
                    gain_Q12 = sx_min(gain_Q12, 1 << 12);
                }
                wv_sync();
            }
        }
        p->last_frame_lost = 0;
    }
}

// SKP_Silk_LPC_synthesis_filter, SKP_Silk_LPC_synthesis_filter.c:37 (order 10), in-place capable
SX_HD void sx_lpc_synthesis_filter(const i16* in, const i16* A_Q12, i32 Gain_Q26, i32* S, i16* out, int len, int Order) {
    // S[Order-1] is the newest state sample; the reference shifts the delay line by swapping pairs
    i32 hist[SX_MAX_LPC];
    for (int j = 0; j < Order; j++) hist[j] = S[Order - 1 - j];   // hist[0] newest
    for (int k = 0; k < len; k++) {
        i32 acc = 0;
        for (int j = 0; j < Order; j++) acc = sx_smlawb(acc, hist[j], A_Q12[j]);
        acc = sx_add_sat32(acc, sx_smulwb(Gain_Q26, in[k]));
        out[k] = (i16)sx_sat16(sx_rshift_round(acc, 10));
        for (int j = Order - 1; j > 0; j--) hist[j] = hist[j - 1];
        hist[0] = sx_lshift_sat32(acc, 4);
    }
    for (int j = 0; j < Order; j++) S[Order - 1 - j] = hist[j];
}

// SKP_Silk_CNG, SKP_Silk_CNG.c:75
SX_FN void sx_cng(SxDecState* st, SxDecWork* w, i16* signal, int length) {
    SX_IN_LDS(st); SX_IN_LDS(w); SX_IN_LDS(signal);
    SxCNG* g = &st->cng;
    SxDecCtrl* c = &w->ctrl;
    if (g->fs_kHz != SX_FS_KHZ) {
        i32 step = 32767 / (SX_LPC + 1), acc = 0;      // SKP_Silk_CNG_Reset, CNG.c:58
        for (int i = 0; i < SX_LPC; i++) { acc += step; g->smth_NLSF_Q15[i] = acc; }
        g->smth_Gain_Q16 = 0;
        g->rand_seed = 3176576;
        g->fs_kHz = SX_FS_KHZ;
    }
    if (st->lossCnt == 0 && st->vadFlag == 0) {
        for (int i = 0; i < SX_LPC; i++)
            g->smth_NLSF_Q15[i] += sx_smulwb(st->md[0].prevNLSF_Q15[i] - g->smth_NLSF_Q15[i], 16348);
        i32 max_Gain_Q16 = 0;
        int subfr = 0;
        for (int i = 0; i < SX_NB_SUBFR; i++) {
            if (c->Gains_Q16[i] > max_Gain_Q16) { max_Gain_Q16 = c->Gains_Q16[i]; subfr = i; }
        }
        // memmove( &buf[subfr_length], buf, 3 * subfr_length ) then memcpy( buf, &exc[subfr * subfr_length], subfr_length ):
        // read everything first (3 * SX_SUBFR <= 4 * 64)
        const int l = SX_LANE;
        if (SX_NLANES == 1) {
            for (int i = 3 * SX_SUBFR - 1; i >= 0; i--) g->exc_buf_Q10[SX_SUBFR + i] = g->exc_buf_Q10[i];
        } else {
            i32 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (l + 64 * j < 3 * SX_SUBFR) ? g->exc_buf_Q10[l + 64 * j] : 0;
            wv_sync();
#pragma unroll
            for (int j = 0; j < 4; j++) if (l + 64 * j < 3 * SX_SUBFR) g->exc_buf_Q10[SX_SUBFR + l + 64 * j] = v[j];
        }
        wv_sync();
        SX_PAR(i, SX_SUBFR) g->exc_buf_Q10[i] = st->exc_Q10[subfr * SX_SUBFR + i];
        wv_sync();
        for (int i = 0; i < SX_NB_SUBFR; i++)
            g->smth_Gain_Q16 += sx_smulwb(c->Gains_Q16[i] - g->smth_Gain_Q16, 4634);
    }
    if (st->lossCnt) {
        i16* CNG_sig = w->u.syn.tmp16;
        int exc_mask = 255;
        while (exc_mask > length) exc_mask >>= 1;
        // CNG_exc (CNG.c:31): the seed recurrence is a pure LCG, so every lane regenerates it serially
        i32 seed = g->rand_seed;
        for (int i = 0; i < length; i++) {
            seed = sx_rand(seed);
            int idx = (seed >> 24) & exc_mask;
            CNG_sig[i] = (i16)sx_sat16(sx_rshift_round(sx_smulww(g->exc_buf_Q10[idx], g->smth_Gain_Q16), 10));
        }
        g->rand_seed = seed;
#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64
        {   // one tap per lane (sx_iir_rows); the coefficients go through the prediction-coefficient slot of the frame (free by now)
            i16* LPC_buf = w->PredCoef_Q12[0];
            sx_nlsf2a_stable(LPC_buf, g->smth_NLSF_Q15, SX_LPC);
            wv_sync();
            const int j = SX_LANE & 15;
            const i32 aj = j < SX_LPC ? sx_pre16(LPC_buf[j < SX_LPC ? j : 0]) : 0;
            const i32 h0 = j < SX_LPC ? g->synth_state[SX_LPC - 1 - (j < SX_LPC ? j : 0)] : 0;
            const i32 hj = sx_iir_rows<true>(h0, aj, length, [&](int i, i32 pr) {
                const i32 acc = sx_add_sat32(pr, sx_smulwb(1 << 26, CNG_sig[i]));
                CNG_sig[i] = (i16)sx_sat16(sx_rshift_round(acc, 10));
                return acc;
            });
            wv_sync();
            if (SX_LANE < SX_LPC) g->synth_state[SX_LPC - 1 - j] = hj;
            wv_sync();
        }
#else
        i16 LPC_buf[SX_MAX_LPC];
        sx_nlsf2a_stable(LPC_buf, g->smth_NLSF_Q15, SX_LPC);
        sx_lpc_synthesis_filter(CNG_sig, LPC_buf, 1 << 26, g->synth_state, CNG_sig, length, SX_LPC);
#endif
        SX_PAR(i, length) signal[i] = (i16)sx_sat16((i32)signal[i] + (i32)CNG_sig[i]);
        wv_sync();
    } else {
        for (int i = 0; i < SX_LPC; i++) g->synth_state[i] = 0;
    }
}

// One 20 ms low-band frame: SKP_Silk_SDK_Decode + SKP_Silk_decode_frame + AgoraSateDecodeTwoDesps.
// rc[] persists across the two frames of a packet.  Returns 0, or a negative SILK error code
// (SKP_Silk_errors.h) on a corrupt payload.
// pre2: NULL = read the symbols off the range coder here (description md on lane md); else the records of the packet's description
// slots from the extraction kernel, frame f of which holds this frame's symbols (the caller has checked that they can be used)
// The batch path's records of one frame -> where the serial parse would have left them (a call of its own: inlined it takes the 32 kHz
// build's synthesis kernel from 250 to 262 registers, i.e. from two waves per SIMD to one).
SX_FN void sx_dec_stage_records(SxDecState* st, SxDecWork* w, const SxExtracted* pre2, int ndesc, int f) {
    SX_IN_LDS(st); SX_IN_LDS(w);
    ndesc = SX_UNI(ndesc); f = SX_UNI(f);
    // The symbols were read ahead and de-quantised by the extraction kernel (SxExtracted::ctl): everything the frame takes from
    // the records is requested here in ONE batch of loads -- the control block of the description in use, the pulses, and per
    // description slot the few values its state moves on with (sx_dequant_parameters: the type offset, the delta-gain and
    // last gain indices, the NLSF vector; frame 0 interpolates with the slot's previous vector, which only the decoder has).
    const int fa = st->first_frame_after_reset == 1;
    for (int d = 0; d < ndesc; d++) { SX_PAR(i, SX_FRAME) w->u.parse.pulses[d][i] = (i16)pre2[d].pulses[f][i]; }
    {
        const i32* sc = (const i32*)&pre2[ndesc - 1].ctl[f];
        i32* dc = (i32*)&w->u.parse.ctrl2[ndesc - 1];
        SX_PAR(i, (int)(sizeof(SxDecCtrl) / 4)) dc[i] = sc[i];
    }
    static_assert(SX_LPC + 2 <= 32, "a row of 32 per description slot");
    SX_PAR(t, ndesc * 32) {
        const int d = t >> 5, i = t & 31;
        const SxFrameSyms* y = &pre2[d].y[f];
        SxDecDesc* m = &st->md[d];
        i32* nl = &w->res_Q10[d * 2 * SX_LPC];
        if (i < SX_LPC) {
            const i32 coef = fa ? 4 : y->NLSFInterpCoef_Q2;
            const i32 nq = y->NLSF_Q15[i], prev = m->prevNLSF_Q15[i];
            nl[SX_LPC + i] = nq;
            if (coef < 4) nl[i] = prev + (sx_mul(coef, nq - prev) >> 2);
            m->prevNLSF_Q15[i] = nq;
        } else if (i == SX_LPC) {
            m->typeOffsetPrev = y->typeOffset;
            if (st->nFramesDecoded == 0) m->prevDeltaGainIndex = y->DeltaGainIndices;
            m->LastGainIndex = pre2[d].lastGain[f];
        } else if (i == SX_LPC + 1) {
            w->u.parse.lane_out[d][0] = y->vadFlag;
            w->u.parse.lane_out[d][1] = y->FrameTermination;
            w->u.parse.lane_out[d][2] = y->left;
            w->u.parse.lane_out[d][3] = y->error;
            w->u.parse.lane_len[d] = y->bufferLength;
        }
    }
    if (fa) { wv_sync(); w->u.parse.ctrl2[ndesc - 1].NLSFInterpCoef_Q2 = 4; }
}

SX_HD int sx_silk_decode_frame(SxDecState* st, SxDecWork* w, SxRangeDec* rc, int action, const u8* payload,
                               i32 nB0, i32 nB1, int useMDIndex, i16* pOut, const SxExtracted* pre2 = 0, int f = 0) {
    SX_IN_LDS(st); SX_IN_LDS(w); SX_IN_LDS(pOut);
    int ret = 0;
    SxDecCtrl* c = &w->ctrl;
    SX_T_BEGIN
    SX_S(14)
    if (st->moreInternalDecoderFrames == 0) st->nFramesDecoded = 0;
    c->LTP_scale_Q14 = 0;
    int used = 0;
    if (action == 1 && st->first_frame_after_reset) {
        // Lost frame before any frame has been decoded: the reference decoder is still at its initial 24 kHz
        // (create_init_destroy.c:41), so SKP_Silk_PLC_conceal / glue_frames / CNG run on 480-sample all-zero state and
        // the 24 -> 8 kHz resampler (dec_API.c:158-176) turns 480 zeros into 160 zeros.  What survives the switch to 8 kHz
        // at the first decoded frame (decoder_set_fs.c:36-64 resets the rest): the concealment LCG advanced once per 24 kHz
        // sample (PLC.c:243), the loss counter, the zero concealed energy that makes glue_frames fade the first good frame
        // in (PLC.c:375-413), and the 24 kHz tags that force PLC_Reset / CNG_Reset on the first decoded frame.
        {   // (wave-uniform: every lane stores the same values)
            st->plc.fs_kHz = 24;
            st->plc.rand_seed = sx_rand_skip(st->plc.rand_seed, 480);
            st->plc.randScale_Q14 = 0;           // (1 << 14) * prevLTP_scale_Q14 (= 0) >> 14, PLC.c:213
            st->plc.conc_energy = 0;
            st->plc.conc_energy_shift = 0;
            st->plc.last_frame_lost = 1;
            st->lossCnt++;
            st->cng.fs_kHz = 24;
        }
        SX_PAR(i, SX_FRAME) pOut[i] = 0;
        wv_sync();
        return 0;
    }
    if (action == 1) {
        sx_plc(st, w, pOut, 1);
    } else {
        const int desp_type = action - 2;
        const int ndesc = desp_type > 1 ? 2 : 1;
        u32 tails[2] = {0u, 0u};
        if (st->nFramesDecoded == 0) {
            // what the reference's range_dec_init leaves behind (see SxDecShadow): the four stale bytes behind each description
            // are fetched first, then the description bytes overwrite the head of "its" buffer
            SxDecShadow* sh = w->shadow;
            const i32 len[2] = {nB0, ndesc > 1 ? nB1 : -1};
            const i32 off[2] = {0, nB0};
            if (!pre2) {
                for (int d = 0; d < 2; d++) {
                    if (len[d] >= 0 && len[d] <= SX_MAX_ARITHM_BYTES) {
                        const u8* t = &sh->b[d][len[d]];
                        tails[d] = (u32)t[0] | ((u32)t[1] << 8) | ((u32)t[2] << 16) | ((u32)t[3] << 24);
                    }
                }
                wv_sync();
            }
            for (int d = 0; d < 2; d++) {
                if (len[d] >= 0 && len[d] <= SX_MAX_ARITHM_BYTES) { SX_PAR(i, len[d]) sh->b[d][i] = payload[off[d] + i]; }
            }
            if (!pre2) wv_sync();        // (read ahead: nothing of this frame reads the shadow; its copy travels with the record loads below)
        }
        SX_S(13)
        if (pre2) {
            sx_dec_stage_records(st, w, pre2, ndesc, f);
        } else {
        // the two descriptions are independent range-coded streams: description md is parsed by lane md
        SX_PAR(md, ndesc) {
            SxRangeDec* r = &rc[SX_NLANES == 1 ? md : 0];
            {
                if (st->nFramesDecoded == 0) {
                    r->tail = tails[md];
                    if (md == 0) sx_rc_dec_init(r, payload, nB0);
                    else sx_rc_dec_init(r, payload + nB0, nB1);
                    st->md[md].rc_stale = 0;
                }
                sx_decode_parameters(st->nFramesDecoded, st->first_frame_after_reset, &st->md[md], st->dbg, &w->u.parse.ctrl2[md], r, w->u.parse.pulses[md],
                                     md, useMDIndex, (const SxCdf*)&w->cdf, w->u.parse.lane_out[md], &w->res_Q10[md * 2 * SX_LPC],
                                     &w->res_Q10[4 * SX_LPC + md * 2 * (SX_FRAME / 16)], sx_dec_syms(w, md));
                w->u.parse.lane_len[md] = r->bufferLength;
            }
        }
        }
        wv_sync();
        SX_T(0)
        {   // the reference parses description 1 after description 0 into the same control block: the last one wins
            const i32* src = (const i32*)&w->u.parse.ctrl2[ndesc - 1];
            i32* dst = (i32*)c;
            SX_PAR(i, (int)(sizeof(SxDecCtrl) / 4)) dst[i] = src[i];
            st->vadFlag = w->u.parse.lane_out[ndesc - 1][0];
            st->FrameTermination = w->u.parse.lane_out[ndesc - 1][1];
            st->nBytesLeft0 = w->u.parse.lane_out[0][2];
            wv_sync();
        }
        const i32 err0 = w->u.parse.lane_out[0][3], err1 = ndesc > 1 ? w->u.parse.lane_out[1][3] : 0;
        const i32 len0 = w->u.parse.lane_len[0];

        i32 inv_gain_Q16 = sx_inverse32_varQ(sx_max(c->DeltaGains_Q16, 1), 32);
        i32 inv_gain_p1_Q16 = inv_gain_Q16;
        i32 inv_gain_p2_Q16 = 65536 - inv_gain_Q16;
        i32 DeltaGains_p1_Q16 = sx_inverse32_varQ(sx_max(inv_gain_p1_Q16, 1), 32);
        i32 DeltaGains_p2_Q16 = sx_inverse32_varQ(sx_max(inv_gain_p2_Q16, 1), 32);
        i32 offset_Q10 = T_quant_offsets_Q10[c->sigtype * 2 + c->QuantOffsetType];
        i32 offset_p1_Q10 = sx_smulww(inv_gain_p1_Q16, offset_Q10);
        i32 offset_p2_Q10 = sx_smulww(inv_gain_p2_Q16, offset_Q10);

        if (err0 || err1) {
            st->nBytesLeft0 = 0;
            used = len0;
            ret = err0 == SX_RC_DEC_PAYLOAD_TOO_LONG ? -11 : -12;   // SKP_SILK_DEC_PAYLOAD_TOO_LARGE / _ERROR
            st->moreInternalDecoderFrames = 0;
        } else {
            st->nFramesDecoded++;
            used = len0 - st->nBytesLeft0;
            // inverse NSQ (decode_frame.c:166-264); the dither LCG is serial, regenerate it per lane
            const i32 seed0 = c->Seed;
            i32 sd = sx_lcg_first(seed0);                 // iterate i + 1 of the dither LCG, for the lane's samples i
            if (desp_type == 2) {
                SX_PAR(i, SX_FRAME) {
                    const i32 dither = sd >> 31;
                    sd = sx_lcg_next(sd);
                    i32 q_Q10 = sx_add(sx_shl(w->u.parse.pulses[0][i], 10), sx_shl(w->u.parse.pulses[1][i], 10));
                    q_Q10 = sx_add(offset_p1_Q10 + offset_p2_Q10, q_Q10);
                    st->exc_Q10[i] = (q_Q10 ^ dither) - dither;
                }
            } else {
                SX_PAR(i, SX_FRAME) {
                    const i32 dither = sd >> 31;
                    sd = sx_lcg_next(sd);
                    int first_half = (i % (SX_SUBFR << 1)) < SX_SUBFR;
                    int use_p1 = desp_type == 0 ? first_half : !first_half;
                    i32 q_Q10 = sx_add(use_p1 ? offset_p1_Q10 : offset_p2_Q10, sx_shl(w->u.parse.pulses[0][i], 10));
                    i32 e = (q_Q10 ^ dither) - dither;
                    st->exc_Q10[i] = sx_smulww(use_p1 ? DeltaGains_p1_Q16 : DeltaGains_p2_Q16, e);
                }
            }
            wv_sync();
            {
                // NLSF -> prediction coefficients of the description in use: the two frame halves on two lanes
                // (decode_parameters.c:108-131); runs after the inverse NSQ because its second workspace reuses the pulses' LDS
                const i32* nl = &w->res_Q10[(ndesc - 1) * 2 * SX_LPC];
                const int interp = c->NLSFInterpCoef_Q2 < 4;
                const SxExtracted* pa = (pre2 && SX_UNI(pre2[ndesc - 1].have_A)) ? &pre2[ndesc - 1] : 0;
                if (pa) {
                    // the conversions that need nothing but this packet's bits were done by the extraction kernel; frame 0's interpolated
                    // vector (it interpolates with the previous packet's) is converted here, on one lane
                    SX_PAR(i, SX_LPC) w->PredCoef_Q12[1][i] = pa->A_final[f][i];
                    if (interp && f == 1) { SX_PAR(i, SX_LPC) w->PredCoef_Q12[0][i] = pa->A_interp1[i]; }
                    if (interp && f == 0) {
#if defined(SX_LANE_STREAM) && defined(SX_HAVE_ROW_NLSF2A) && SX_LPC <= 14
                        // (every row converts the vector: sx_row_nlsf2a_stable, solo_common.h; the serial form only for the reference's corrections.
                        // Not in the 32 kHz build: inlined there it takes the synthesis kernel past 256 registers = one wave per SIMD)
                        if (__builtin_amdgcn_ballot_w64(!sx_row_nlsf2a_stable(w->PredCoef_Q12[0], nl)) != 0)
#endif
                        { wv_sync(); SX_PAR(v, 1) sx_nlsf2a_stable_ws(w->PredCoef_Q12[0], nl, SX_LPC, w->u.ws1); }
                    }
                } else {
                SX_PAR(v, 2) {
                    if (v == 1) sx_nlsf2a_stable_ws(w->PredCoef_Q12[1], nl + SX_LPC, SX_LPC, &w->res_Q10[4 * SX_LPC]);
                    else if (interp) sx_nlsf2a_stable_ws(w->PredCoef_Q12[0], nl, SX_LPC, w->u.ws1);
                }
                }
                wv_sync();
                if (!interp) {
                    SX_PAR(i, SX_LPC) w->PredCoef_Q12[0][i] = w->PredCoef_Q12[1][i];
                    wv_sync();
                }
                if (st->lossCnt) {
                    SX_PAR(v, 2) sx_bwexpander(w->PredCoef_Q12[v], SX_LPC, 63570);
                    wv_sync();
                }
            }
            SX_T(1)
            sx_decode_core(st, w, pOut);
            SX_T(2)
            sx_plc(st, w, pOut, 0);
            SX_T(3)
            st->lossCnt = 0;
            st->prev_sigtype = c->sigtype;
        }
    }
    if (ret < 0) return ret;   // corrupt payload: the reference returns before producing output
    SX_T_RESET
    SX_PAR(i, SX_FRAME) st->outBuf[i] = pOut[i];
    wv_sync();
    sx_plc_glue_frames(st, pOut, SX_FRAME, st->first_frame_after_reset ? 480 : SX_FRAME);
    st->first_frame_after_reset = 0;     // (decode_frame.c:281; cleared here because the ramp length above still needs it)
    SX_T(4)
    sx_cng(st, w, pOut, SX_FRAME);
    SX_T(5)
    if (SX_UNI(st->nFramesDecoded) > 2) sx_dec_hp_output(st->HPState, pOut, SX_FRAME);      // (decode_frame.c:381: received or concealed frame alike)
    st->lagPrev = c->pitchL[SX_NB_SUBFR - 1];
    // SKP_Silk_SDK_Decode bookkeeping, dec_API.c:125-150
    if (used) {
        if (st->nBytesLeft0 > 0 && st->FrameTermination == 1 && st->nFramesDecoded < 5) st->moreInternalDecoderFrames = 1;
        else st->moreInternalDecoderFrames = 0;
    }
    return 0;
}

// MSB-first bit reader over the 8 high-band bytes (libBWE/AGR_BWE_bits.c:135)
SX_HD u32 sx_hb_unpack(const u8* hb, int* bitpos, int nbBits) {
    u32 d = 0;
    for (int i = 0; i < nbBits; i++) {
        int bp = *bitpos + i;
        d = (d << 1) | ((hb[bp >> 3] >> (7 - (bp & 7))) & 1);
    }
    *bitpos += nbBits;
    return d;
}

// AGR_Sate_LPC_synthesis_filter_fix, libBWE/AGR_BWE_LPC_synthesizer.c:56 (order 8, int32 Q10 input)
SX_HD void sx_hb_lpc_synthesis(const i32* in_Q10, const i16* A_Q12, i32 Gain_Q16, i32* S, i16* out, int len) {
    // coefficients (pre-shifted for the high-word multiply) and the last SX_HB_LPC outputs in registers; slot r of the ring
    // holds the output of time t = r (mod SX_HB_LPC); len is a multiple of SX_HB_LPC
    i32 a[SX_HB_LPC], h[SX_HB_LPC];
#pragma unroll
    for (int j = 0; j < SX_HB_LPC; j++) { a[j] = sx_pre16(A_Q12[j]); h[j] = S[j]; }
    for (int k0 = 0; k0 < len; k0 += SX_HB_LPC) {
#pragma unroll
        for (int u = 0; u < SX_HB_LPC; u++) {
            i32 acc = 0;
#pragma unroll
            for (int j = 0; j < SX_HB_LPC; j++) acc = sx_smlaw_pre(acc, h[(u - 1 - j + 2 * SX_HB_LPC) % SX_HB_LPC], a[j]);
            acc = sx_add_sat32(acc, sx_smulww(Gain_Q16, in_Q10[k0 + u]));
            out[k0 + u] = (i16)sx_sat16(sx_rshift_round(acc, 10));
            h[u] = sx_lshift_sat32(acc, 4);
        }
    }
#pragma unroll
    for (int j = 0; j < SX_HB_LPC; j++) S[j] = h[j];
}

// AGR_Bwe_decode_frame_FIX, libBWE/AGR_BWE_decode_frame_FIX.c:40, for both 20 ms frames of a packet, in three steps:
//   sx_hb_decode_side   before the low-band frames: the side information of the (up to) two high-band frames, LSP -> LPC conversion
//                       included, on two lanes; nothing of the stream state changes
//   (synthesis)         inside sx_decode_core, on lane 1 of the low band's own synthesis loop, subframe by subframe, from the
//                       excitation of that subframe (hbp.piggy); packets whose low band is concealed instead of decoded have no
//                       such loop and no excitation: sx_hb_finish runs the filter on zeros for them
//   sx_hb_finish        after the low-band frames (not reached when the packet is abandoned, like the reference's high-band call):
//                       commits the filter state, the loss / previous-frame bookkeeping in frame order, hands the band to the QMF
// `hb` = the 8 high-band bytes (ignored when lost).
// pre: NULL, or the extraction record of the slot that carries the high-band bytes (side information already converted there)
SX_FN void sx_hb_decode_side(SxDecState* st, SxDecWork* w, const u8* hb, int lostflag, const SxExtracted* pre) {
    SX_IN_LDS(st); SX_IN_LDS(w);
    SxHbParams* hp = &w->hbp;
    const int lost = (lostflag == 1 || lostflag == 2);
    const int nf = st->hb_joint ? 1 : st->fpp;    // high-band frames per packet
    if (pre && !lost && SX_UNI(pre->have_hb)) {
        SX_PAR(i, nf * SX_HB_LPC) { (&hp->lsp[0][0])[i] = (&pre->hb_lsp[0][0])[i]; (&hp->lpc[0][0])[i] = (&pre->hb_lpc[0][0])[i]; }
        SX_PAR(i, nf * 4) (&hp->gain[0][0])[i] = (&pre->hb_gain[0][0])[i];
    } else
    SX_PAR(f, nf) {
        i32* l = hp->lsp[f];
        if (lost) {
            for (int i = 0; i < SX_HB_LPC; i++) l[i] = st->HB_prev_NLSFq[i];
            for (int k = 0; k < 4; k++) hp->gain[f][k] = (i16)st->HB_prev_Gain;
        } else {
            int bitpos = f * 32;
            u32 idx = sx_hb_unpack(hb, &bitpos, 12);
            u32 idx1 = idx & 0xFF, idx2 = idx >> 8;
            for (int i = 0; i < SX_HB_LPC; i++) l[i] = T_hb_lsp_cb1[idx1 * SX_HB_LPC + i] + T_hb_lsp_cb2[idx2 * SX_HB_LPC + i];
            for (int k = 0; k < 4; k++) hp->gain[f][k] = (i16)T_hb_gain_cb[sx_hb_unpack(hb, &bitpos, 5)];
        }
        sx_nlsf2a_stable_ws(hp->lpc[f], l, SX_HB_LPC, f == 0 ? &w->res_Q10[4 * SX_LPC] : w->u.ws1);   // same for all 4 subframes
    }
    SX_PAR(i, SX_HB_LPC) hp->S[i] = st->HB_synth_state[i];
    if (SX_LANE == 0) { hp->lost = lost; hp->piggy = 0; hp->frame = 0; }
    wv_sync();
}

// piggy_done: both low-band frames were decoded (and the high band synthesised next to them)
SX_FN void sx_hb_finish(SxDecState* st, SxDecWork* w, int lostflag, int piggy_done) {
    SX_IN_LDS(st); SX_IN_LDS(w);
    SxHbParams* hp = &w->hbp;
    const int lost = (lostflag == 1 || lostflag == 2);
    const int nf = st->hb_joint ? 1 : st->fpp;
    const int sub_len = st->hb_joint ? 2 * SX_SUBFR : SX_SUBFR;   // BWE_SubFrameSize
    SX_T_BEGIN
    if (!piggy_done) {
        // the low band was concealed: the high band is lost with it, its filter runs on zero excitation (decode_frame_FIX.c:65)
        i32* zero = w->res_Q10;
        SX_PAR(i, SX_SUBFR) zero[i] = 0;
        wv_sync();
#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64
        (void)sub_len;
        for (int f = 0; f < nf; f++) {   // zero input: the gains do not matter; one tap per lane (sx_iir_rows), a whole high-band frame per run
            const int j = SX_LANE & 15;
            const i32 aj = j < SX_HB_LPC ? sx_pre16(hp->lpc[f][j < SX_HB_LPC ? j : 0]) : 0;
            const i32 h0 = j < SX_HB_LPC ? hp->S[SX_HB_LPC - 1 - (j < SX_HB_LPC ? j : 0)] : 0;
            i16* out = &w->hi_out[f * SX_FRAME];
            const i32 hj = sx_iir_rows<true>(h0, aj, st->hb_joint ? 2 * SX_FRAME : SX_FRAME, [&](int i, i32 pr) {
                out[i] = (i16)sx_sat16(sx_rshift_round(pr, 10));
                return pr;
            });
            wv_sync();
            if (SX_LANE < SX_HB_LPC) hp->S[SX_HB_LPC - 1 - j] = hj;
            wv_sync();
        }
#else
        for (int f = 0; f < nf; f++)
            for (int k = 0; k < 4; k++)
                for (int h = 0; h < sub_len; h += SX_SUBFR)
                    sx_hb_lpc_synthesis(zero, hp->lpc[f], sx_mul(-2867, (i32)hp->gain[f][k]), hp->S, &w->hi_out[f * SX_FRAME + k * sub_len + h], SX_SUBFR);
        wv_sync();
#endif
    }
    SX_T(8)
    for (int f = 0; f < nf; f++) {
        const i32* QHB_LSP = hp->lsp[f];
        if (lost) {
            st->hb_lossCnt++;
        } else {
            if (st->hb_first) {
                for (int i = 0; i < SX_HB_LPC; i++) st->HB_prev_NLSFq[i] = QHB_LSP[i];
                st->HB_prev_Gain = hp->gain[f][3];
            }
            st->hb_lossCnt = 0;
        }
        if (lostflag == 0 || lostflag == 4 || lostflag == 3) {
            st->HB_prev_Gain = hp->gain[f][3];
            for (int i = 0; i < SX_HB_LPC; i++) st->HB_prev_NLSFq[i] = QHB_LSP[i];
        }
        st->hb_first = 0;
    }
    wv_sync();
    SX_PAR(i, SX_HB_LPC) st->HB_synth_state[i] = hp->S[i];
    SX_PAR(i, SX_QMF_HIST) w->u.hi[i] = st->qmf_hi_hist[i];      // (the high-band buffer of the QMF shares its LDS with the frame scratch)
    SX_PAR(i, st->fpp * SX_FRAME) w->u.hi[SX_QMF_HIST + i] = w->hi_out[i];
    wv_sync();
}

// AGR_Sate_qmf_synth, libBWE/AGR_BWE_qmf.c:86, as a direct polyphase form (wave-parallel):
//   y[2k]   = sat( pshr15( sum_m a[2m]   * s1[k-m] + (-a[2m]) * s2[k-m] ) )
//   y[2k+1] = sat( pshr15( sum_m a[2m+1] * s1[k-m] +   a[2m+1] * s2[k-m] ) )      m = 0..31
// lo/hi hold [32 history | 320 new] samples.  32-bit accumulation wraps, so summation order is free.
SX_FN void sx_qmf_synth(const i16* lo, const i16* hi, i16* y, int n = SX_BAND) {
    SX_IN_LDS(lo); SX_IN_LDS(hi);
    n = SX_UNI(n);
    SX_PAR(k, n) {
        i32 y0 = 0, y1 = 0;
        const i16* s1 = lo + SX_QMF_HIST + k;
        const i16* s2 = hi + SX_QMF_HIST + k;
        for (int m = 0; m < 32; m++) {
            i32 a0 = T_qmf_taps[2 * m], a1 = T_qmf_taps[2 * m + 1];
            i32 x1 = s1[-m], x2 = s2[-m];
            y0 = sx_add(y0, sx_mul(a0, x1));
            y0 = sx_add(y0, sx_mul((i32)(i16)(-a0), x2));
            y1 = sx_add(y1, sx_mul(a1, x1));
            y1 = sx_add(y1, sx_mul(a1, x2));
        }
        y[2 * k] = (i16)sx_saturate(sx_pshr32(y0, 15), 32767);
        y[2 * k + 1] = (i16)sx_saturate(sx_pshr32(y1, 15), 32767);
    }
}

// AGR_Sate_Decoder_Decode + AGR_Sate_decode_process for one 40 ms packet (libBWE/AGR_BWE_SDK_API.c:249,
// libBWE/AGR_BWE_decode_frame_FIX.c:118).  `bits`/`nBytes*` follow the reference's calling convention:
//   lostflag 4: bits = MD1|MD2|HB, nBytes0 = total, nBytes1 = len(MD2)+8
//   lostflag 2: bits = MD1,        nBytes0 = len(MD1), nBytes1 = 0
//   lostflag 3: bits = MD2|HB,     nBytes0 = len(MD2)+8, nBytes1 = 0
//   lostflag 1: packet lost (bits ignored)
// Returns 0 / -1 / negative SILK code like the reference.
// ext2: NULL, or the records of the packet's two description slots from the extraction kernel (batch path): used when the packet
// is an ordinary one -- see sx_extracted_usable; otherwise the symbols are read here, serially, like without them.
SX_HD bool sx_extracted_usable(const SxDecState* st, const SxExtracted* ext2, int lostflag) {
    if (!ext2 || lostflag < 2 || SX_UNI(st->fpp) != 2) return false;
    // the packet starts a new range-coder buffer (no frames left over from a payload that announced more than it carried)
    if (SX_UNI(st->moreInternalDecoderFrames) != 0) return false;
    const int ndesc = lostflag == 4 ? 2 : 1;
    if (!SX_UNI(ext2[0].usable) || (ndesc > 1 && !SX_UNI(ext2[1].usable))) return false;
    // ... and has the ordinary structure: frame 0 announces one more frame and leaves bytes for it, frame 1 is the last one
    // (SKP_Silk_dec_API.c:125-150; nFramesDecoded is 1 / 2 at these points)
    const SxExtracted* last = &ext2[ndesc - 1];
    const bool more0 = SX_UNI(ext2[0].y[0].left) > 0 && SX_UNI(last->y[0].FrameTermination) == 1;
    const bool more1 = SX_UNI(ext2[0].y[1].left) > 0 && SX_UNI(last->y[1].FrameTermination) == 1;
    return more0 && !more1;
}
SX_HD int sx_decode_packet(SxDecWork* w, const u8* bits, i32 nBytes0, i32 nBytes1, int lostflag,
                           int useMDIndex, i16* pcm_out, const SxExtracted* ext2 = 0) {
    SxDecState* st = &w->st;
    if (nBytes0 <= 0) return -1;
    const int fpp = SX_UNI(st->fpp);
    const i32 hb_bytes = st->hb_joint ? SX_HB_BYTES / 2 : (SX_HB_BYTES / 2) * fpp;     // (QMF_HB_FrameSize / BWE_FrameSize) * HB_BYTE
    i32 nB0 = (lostflag == 2) ? nBytes0 : nBytes0 - hb_bytes;
    i32 nB1 = nBytes1 ? nBytes1 - hb_bytes : 0;
    const i32 hb_pos = nB0;
    nB0 -= nB1;
    // The reference's two range decoders are part of its state and are only re-initialised when a call starts a new packet
    // (nFramesDecoded == 0).  A payload whose frame-termination symbol announces more than two frames -- only a corrupted one does --
    // makes the NEXT call go on decoding the OLD buffer (SKP_Silk_dec_API.c:125-150, SKP_Silk_decode_frame.c:93-99): the coder
    // registers are therefore kept in the state record, and the old buffer is the shadow (SxDecShadow).
    SxRangeDec rc[2];
    // (an ordinary packet whose symbols were read ahead: the serial coder is not used, and since such a packet ends without frames
    // left over, the next packet re-initialises the coder's registers before it reads them)
    const SxExtracted* pre2 = sx_extracted_usable(st, ext2, lostflag) ? ext2 : 0;
    if (!pre2) {
#if SX_NLANES == 1
    for (int d = 0; d < 2; d++) {
        const SxDecDesc* m = &st->md[d];
#else
    {
        const int d = 0;
        const SxDecDesc* m = &st->md[SX_LANE == 1 ? 1 : 0];              // description md is parsed by lane md
#endif
        rc[d].bufferLength = m->rc_bufferLength; rc[d].bufferIx = m->rc_bufferIx; rc[d].error = m->rc_error;
        rc[d].base_Q32 = m->rc_base_Q32; rc[d].range_Q16 = m->rc_range_Q16; rc[d].tail = m->rc_tail;
    }
    }
    // A packet that went through the records left the coder registers of its slots behind (the next packet normally starts new
    // buffers).  This call goes on in the OLD buffers (a corrupted payload announced more frames than it carried): rebuild the
    // registers of such a slot the way the reference got them -- range_dec_init on the description (its bytes, and the four behind
    // them, are still in the shadow buffer) and the symbol walk of its two frames; the slot's other state is put back afterwards.
    if (!pre2 && SX_UNI(st->moreInternalDecoderFrames) != 0 && (SX_UNI(st->md[0].rc_stale) | SX_UNI(st->md[1].rc_stale)) != 0) {
        SX_PAR(md, 2) {
            SxDecDesc* m = &st->md[md];
            if (m->rc_stale) {
                SxDecDesc* keep = &w->shadow->keep[md];          // (HBM: a private copy would cost every lane of the kernel scratch memory)
                *keep = *m;
                SxRangeDec* r = &rc[SX_NLANES == 1 ? md : 0];
                const u8* b = w->shadow->b[md];
                const i32 len = m->rc_bufferLength;
                r->tail = (u32)b[len] | ((u32)b[len + 1] << 8) | ((u32)b[len + 2] << 16) | ((u32)b[len + 3] << 24);
                sx_rc_dec_init(r, b, len);
                for (int f = 0; f < 2; f++)
                    sx_decode_parameters(f, 0, m, (i32*)0, &w->u.parse.ctrl2[md], r, w->u.parse.pulses[md], md, useMDIndex, (const SxCdf*)&w->cdf,
                                         w->u.parse.lane_out[md], &w->res_Q10[md * 2 * SX_LPC], &w->res_Q10[4 * SX_LPC + md * 2 * (SX_FRAME / 16)],
                                         sx_dec_syms(w, md));
                *m = *keep;
                m->rc_stale = 0;          // (the registers themselves are stored from rc[] after every frame below)
            }
        }
        wv_sync();
    }
    // the payload is read byte by byte by a serial coder: stage it in LDS
    if (lostflag != 1 && nBytes0 <= SX_DEC_PAYLOAD_LDS && !pre2) {
        SX_PAR(i, nBytes0) w->payload[i] = bits[i];
        bits = w->payload;
    }
#if SX_NLANES == 1
    rc[0].buf = w->shadow->b[0]; rc[1].buf = w->shadow->b[1];               // (replaced by the payload when the coder is initialised)
#else
    rc[0].buf = w->shadow->b[SX_LANE == 1 ? 1 : 0];
#endif
#ifdef SX_RC_LOG
    rc[0].log = st->rclog; rc[0].nlog = 0; rc[1].log = 0; rc[1].nlog = 0;
#endif
    SX_PAR(i, SX_QMF_HIST) w->lo[i] = st->qmf_lo_hist[i];
    wv_sync();
    sx_hb_decode_side(st, w, bits + hb_pos, lostflag, pre2 ? &pre2[lostflag == 4 ? 1 : 0] : 0);
    int piggy_frames = 0;
    for (int f = 0; f < 2; f++) {
        // a frame that is decoded (not concealed) runs the high band's synthesis filter next to its own (sx_decode_core)
        w->hbp.frame = f;
        w->hbp.piggy = lostflag != 1 && f < fpp;       // (framesize 20: the second decoder call of the packet has no high-band frame under it)
        wv_sync();
        if (lostflag != 1 && f < fpp) piggy_frames++;
        int ret = sx_silk_decode_frame(st, w, rc, lostflag, bits, nB0, nB1, useMDIndex, &w->lo[SX_QMF_HIST + f * SX_FRAME], pre2, f);
        wv_sync();
        if (!pre2) {
#if SX_NLANES == 1
        for (int d = 0; d < 2; d++) {
            SxDecDesc* m = &st->md[d];
#else
        if (SX_LANE < 2) {
            const int d = 0;
            SxDecDesc* m = &st->md[SX_LANE];
#endif
            m->rc_bufferLength = rc[d].bufferLength; m->rc_bufferIx = rc[d].bufferIx; m->rc_error = rc[d].error;
            m->rc_base_Q32 = rc[d].base_Q32; m->rc_range_Q16 = rc[d].range_Q16; m->rc_tail = rc[d].tail;
        }
        wv_sync();
        }
        if (ret < 0) { st->last_error = ret; return ret; }
    }
    if (pre2) {
        SX_PAR(md, lostflag == 4 ? 2 : 1) { st->md[md].rc_stale = 1; st->md[md].rc_bufferLength = md == 0 ? nB0 : nB1; }
        wv_sync();
    }
    SX_T_BEGIN
    sx_hb_finish(st, w, lostflag, piggy_frames == fpp);
    SX_T(6)
    sx_qmf_synth(w->lo, w->u.hi, pcm_out, fpp * SX_FRAME);
    SX_T(7)
    SX_PAR(i, SX_QMF_HIST) { st->qmf_lo_hist[i] = w->lo[fpp * SX_FRAME + i]; st->qmf_hi_hist[i] = w->u.hi[fpp * SX_FRAME + i]; }
    wv_sync();
    return 0;
}


// ---------------------------------------------------------------------------------------------------
// Batch path, step 1: the symbols of ONE description slot of ONE packet, read ahead of the serial decoder by the extraction kernel
// (one lane each: every description of every packet of the call at once).  The lane has the description's bytes and nothing else
// -- not the stream's history -- so it decodes with the interval form of the range decoder (solo_rc.h, SxRangeDec2) and marks the
// record usable only if both frames came out without a coder error and without depending on what lies behind the description.
// ---------------------------------------------------------------------------------------------------
// LDS row of one extraction lane: the per-block scratch and the pulses of one frame, in bytes; an odd number of dwords (the lanes
// of a wavefront sit at the same offset of their own row most of the time)
#define SX_EXTRACT_TMP (2 * (SX_FRAME / 16))
#define SX_EXTRACT_ROW (4 * (((SX_EXTRACT_TMP + SX_FRAME + 3) / 4) | 1))
struct SxExtractLane { u8 b[SX_EXTRACT_ROW]; };
static_assert(SX_EXTRACT_TMP % 4 == 0 && (SX_EXTRACT_ROW / 4) % 2 == 1, "row layout");

// where description slot md of a packet handed over as (nBytes0, nBytes1, lostflag) lies (see sx_decode_packet); false: no such slot
// *sel: this is the slot the decoder takes its coefficients from; *hb_off: where the high-band bytes lie (-1: not in the packet)
SX_HD bool sx_desc_span(int lostflag, i32 nBytes0, i32 nBytes1, int hb_joint, int md, i32* off, i32* len, int* sel = 0, i32* hb_off = 0) {
    if (lostflag < 2 || nBytes0 <= 0) return false;
    const i32 hb_bytes = hb_joint ? SX_HB_BYTES / 2 : SX_HB_BYTES;
    i32 nB0 = (lostflag == 2) ? nBytes0 : nBytes0 - hb_bytes;
    const i32 nB1 = nBytes1 ? nBytes1 - hb_bytes : 0;
    const i32 hb_pos = nB0;
    nB0 -= nB1;
    const int ndesc = lostflag == 4 ? 2 : 1;
    if (md >= ndesc) return false;
    *off = md == 0 ? 0 : nB0;
    *len = md == 0 ? nB0 : nB1;
    if (sel) *sel = md == ndesc - 1;
    if (hb_off) *hb_off = lostflag == 2 ? -1 : hb_pos;
    return true;
}

// src: the description's bytes inside the packet (HBM), len of them (read in place: a serial coder touches every byte once, the
// look-ups of the tables are what it waits for)
// sel: also convert the NLSF vectors; hb: the packet's high-band bytes or NULL (only looked at when sel)
SX_HD void sx_extract_desc(const u8* src, i32 len, int useMDIndex, const SxCdf* cdf, SxExtractLane* L, SxExtracted* rec, int sel = 0,
                           const u8* hb = 0, int hb_joint = 0) {
    rec->usable = 0;
    rec->have_A = 0;
    rec->have_hb = 0;
    if (len <= 0 || len > SX_MAX_ARITHM_BYTES) return;           // (the serial decoder reports what is wrong with it)
    SxRangeDec2 r;
    sx_rc_dec_init(&r, src, len);
    i32 top = 0, narrow = 0;
    i32* dbg = 0;
    for (int f = 0; f < 2; f++) {
        SxFrameSyms y;
        sx_extract_parameters(f, &top, dbg, &r, (i8*)&L->b[SX_EXTRACT_TMP], 0, useMDIndex, cdf, &y, &L->b[0], &narrow);
        { const i32* sy = (const i32*)&y; i32* dy = (i32*)&rec->y[f]; for (int i = 0; i < (int)(sizeof(SxFrameSyms) / 4); i++) dy[i] = sy[i]; }
        if (y.fs_bad || y.error || narrow) return;
        { const i32* sq = (const i32*)&L->b[SX_EXTRACT_TMP]; i32* dq = (i32*)&rec->pulses[f][0]; for (int i = 0; i < SX_FRAME / 4; i++) dq[i] = sq[i]; }
    }
    if (r.ambiguous) return;
    {   // the frames' side information de-quantised (see SxExtracted::ctl): blank description state, no first-frame override
        SxDecDesc m;
        SxDecCtrl c;
        i32 lo_[4], nl_[2 * SX_LPC];
        { i32* z = (i32*)&m; for (int i = 0; i < (int)(sizeof(SxDecDesc) / 4); i++) z[i] = 0; }
        { i32* z = (i32*)&c; for (int i = 0; i < (int)(sizeof(SxDecCtrl) / 4); i++) z[i] = 0; }
        for (int f = 0; f < 2; f++) {
            sx_dequant_parameters_t<false>(&rec->y[f], f, 0, useMDIndex, &m, &c, lo_, nl_);
            { const i32* sc = (const i32*)&c; i32* dc = (i32*)&rec->ctl[f]; for (int i = 0; i < (int)(sizeof(SxDecCtrl) / 4); i++) dc[i] = sc[i]; }
            rec->lastGain[f] = m.LastGainIndex;
        }
    }
    if (sel) {
        // (private workspace: every lane of the wavefront is at the same place of its own copy, which is how scratch memory is laid out)
        i32 ws[SX_NLSF2A_WS], nl[SX_LPC];
        i16 a[SX_MAX_LPC];
        for (int f = 0; f < 2; f++) {
            for (int i = 0; i < SX_LPC; i++) nl[i] = rec->y[f].NLSF_Q15[i];
            sx_nlsf2a_stable_ws(a, nl, SX_LPC, ws);
            for (int i = 0; i < SX_LPC; i++) rec->A_final[f][i] = a[i];
        }
        if (rec->y[1].NLSFInterpCoef_Q2 < 4) {                   // frame 1 interpolates between the two vectors of this packet (sx_dequant_parameters)
            const i32 coef = rec->y[1].NLSFInterpCoef_Q2;
            for (int i = 0; i < SX_LPC; i++) {
                const i32 prev = rec->y[0].NLSF_Q15[i];
                nl[i] = prev + (sx_mul(coef, rec->y[1].NLSF_Q15[i] - prev) >> 2);
            }
            sx_nlsf2a_stable_ws(a, nl, SX_LPC, ws);
            for (int i = 0; i < SX_LPC; i++) rec->A_interp1[i] = a[i];
        }
        rec->have_A = 1;
        if (hb) {                                                // AGR_Bwe_decode_frame_FIX side information (sx_hb_decode_side)
            const int nf = hb_joint ? 1 : 2;
            for (int f = 0; f < nf; f++) {
                int bitpos = f * 32;
                const u32 idx = sx_hb_unpack(hb, &bitpos, 12);
                const u32 idx1 = idx & 0xFF, idx2 = idx >> 8;
                for (int i = 0; i < SX_HB_LPC; i++) { nl[i] = T_hb_lsp_cb1[idx1 * SX_HB_LPC + i] + T_hb_lsp_cb2[idx2 * SX_HB_LPC + i]; rec->hb_lsp[f][i] = nl[i]; }
                for (int k = 0; k < 4; k++) rec->hb_gain[f][k] = (i16)T_hb_gain_cb[sx_hb_unpack(hb, &bitpos, 5)];
                sx_nlsf2a_stable_ws(a, nl, SX_HB_LPC, ws);
                for (int i = 0; i < SX_HB_LPC; i++) rec->hb_lpc[f][i] = a[i];
            }
            rec->have_hb = 1;
        }
    }
    rec->usable = 1;
}
