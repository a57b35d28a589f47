// solo_enc_state.h -- persistent per-stream encoder state (HBM) and per-packet working set of the
// SOLO encoder kernels.  Live subset of the reference's SKP_Silk_encoder_state_FIX (SKP_Silk_structs_FIX.h:79,
// SKP_Silk_structs.h:136) + AGR_Sate_encoder_hb_state_FIX / AGR_Sate_HB_encoder_control_FIX
// (libBWE/AGR_BWE_structs.h:14-47) at the benchmark configuration: 8 kHz NB, complexity 2,
// 2 descriptions, 2 x 20 ms frames per packet, no LBRR / DTX / bandwidth switching.
#pragma once
#include "solo_common.h"
#include "solo_consts.inc"
#include "solo_dec.h"     // SX_PACKET / SX_BAND, sx_nlsf_msvq_decode (shared with the decoder)

#define SX_SHAPE_ORDER 16            // shapingLPCOrder (setup_complexity.h:76)
#define SX_LA_SHAPE (5 * SX_FS_KHZ)   // la_shape = 5 * fs_kHz                         (control_codec_FIX.c:285)
#define SX_LA_PITCH (2 * SX_FS_KHZ)   // la_pitch = 2 * fs_kHz
#define SX_SHAPE_WIN (15 * SX_FS_KHZ) // shapeWinLength = 5*fs_kHz + 2*la_shape          (setup_complexity.h:78)
#define SX_PITCH_LPC_WIN (24 * SX_FS_KHZ)   // pitch_LPC_win_length = (20 + 2*2) * fs_kHz (setup_complexity.h:84)
#define SX_PITCH_LPC_ORDER SX_LPC    // min(16, predictLPCOrder)                        (setup_complexity.h:86)
#define SX_XBUF (2 * SX_FRAME + SX_LA_SHAPE)   // 360
#define SX_LTP_BUF 512
#define SX_LTP_MASK (SX_LTP_BUF - 1)
#define SX_DD_STATES 4               // nStatesDelayedDecision
#define SX_DD_DELAY 32               // DECISION_DELAY
#define SX_N_TRACKS 3                // centre, MD1, MD2
#define SX_WARPING_Q16 (SX_FS_KHZ * K_WARPING_MULTIPLIER_Q16)    // setup_complexity.h:82
#define SX_MSVQ_SURVIVORS 16
#define SX_HB_XBUF (85 * SX_FS_KHZ)   // x_hb_buf_fix: BWE_FrameSize*2 + lb_Delay*hb_KHz (360 live words with 20 ms high-band frames, 680 with joint_mode 1)

struct SxVAD {                       // SKP_Silk_VAD_state, SKP_Silk_structs.h:69
    i32 AnaState[2], AnaState1[2], AnaState2[2];
    i32 XnrgSubfr[4];
    i32 NrgRatioSmth_Q8[4];
    i32 HPstate;
    i32 NL[4], inv_NL[4], NoiseLevelBias[4];
    i32 counter;
};

struct SxNSQ {                       // SKP_Silk_nsq_state, SKP_Silk_structs.h:44: small per-track part (histories: SxNsqTrack)
    i32 sLPC_Q14[SX_MAX_LPC];        // newest 16 of the reference's 32-entry tail (only the last 10 are ever read)
    i32 sAR2_Q14[SX_SHAPE_ORDER];
    i32 sLF_AR_shp_Q12;
    i32 lagPrev;
    i32 prev_inv_gain_Q16;
    i32 gadjPrev[SX_NB_SUBFR];       // gain-adjustment factors of the previous frame's four subframe starts (65536 = none): the history
                                     // arrays are stored unscaled, the factors are applied when history is staged (solo_enc_nsq_row.h)
    i32 histBase;                    // 0 / SX_FRAME: where logical entry 0 of the two circular histories sits (toggles every frame)
};

// Compact per-stream encoder state: loaded into LDS when a launch starts, written back when it ends.
struct SxEncState {
    // --- SILK common ---
    i32 frameCounter;
    i32 prev_sigtype, prevLag, first_frame_after_reset;
    i32 nFramesInPayloadBuf;
    i32 vadFlag, noSpeechCounter, inDTX;
    i32 speech_activity_Q8;
    i32 LTPCorr_Q15;
    i32 SNR_dB_Q7, SNRPerMD_dB_Q7;
    i32 avgGain_Q16;
    i32 prevLTPredCodGain_Q7, HPLTPredCodGain_Q7;
    i32 variable_HP_smth1_Q15, variable_HP_smth2_Q15;
    i32 In_HP_State[2];
    i32 useMDIndex;
    i32 useDTX;                      // dtx_enable: packets are not sent while inDTX
    i32 hb_joint;                    // joint_mode 1: ONE 40 ms high-band frame per packet (4 HB bytes instead of 8)
    i32 fpp;                         // SILK frames per packet: 2 (framesize_ms = 40) or 1 (framesize_ms = 20: one frame, one 4-byte high-band frame)
    SxVAD vad;
    // shape / prefilter / prediction states (SKP_Silk_structs_FIX.h:44-73)
    i32 LastGainIndex, HarmBoost_smth_Q16, HarmShapeGain_smth_Q16, Tilt_smth_Q16;
    i32 pf_sAR_shp[SX_SHAPE_ORDER + 1];
    i32 pf_sLTP_shp_buf_idx, pf_sLF_AR_shp_Q12, pf_sLF_MA_shp_Q12, pf_sHarmHP, pf_lagPrev;
    i32 prev_NLSFq_Q15[SX_LPC];
};

// Quantiser state + histories of ONE track (centre, MD1, MD2), contiguous: a lane of the quantiser kernel addresses everything of
// its track from one base.  The quantised signal and the shaping history are kept for two frames (the reference: previous frame |
// current frame, shifted down by memcpy when a frame ends); here they are CIRCULAR over 2 SX_FRAME entries, logical entry L (0 =
// first sample of the previous frame) at physical (L + histBase) mod 2 SX_FRAME, and a frame end only toggles histBase: every
// entry is written exactly once.  sLTP_Q16 (the scaled prediction history) is frame-local: re-whitening regenerates it.
struct alignas(16) SxNsqTrack {
    i32 sLTP_Q16[2 * SX_FRAME];
    i32 shp[2 * SX_FRAME];           // sLTP_shp_Q10, circular
    i16 xq[2 * SX_FRAME];            // quantised signal, circular
    SxNSQ s;
};

// Per-stream history arrays: stay in HBM, staged through LDS by the phase that uses them.
struct SxEncHist {
    i16 x_buf[SX_FRAME + SX_LA_SHAPE];           // samples [0, 200) of the analysis buffer; [200, 360) is new every frame
    i16 pf_sLTP_shp[SX_LTP_BUF];                 // prefilter's harmonic-shaping ring
    i16 qmf_hist[63 + 1];                        // last 63 input samples >> 1 (h0_mem of the reference, time order)
    i16 x_hb_buf[2 * SX_FRAME + SX_LA_SHAPE];    // high-band analysis history (BWE_FrameSize + lb_Delay*hb_KHz = 200 samples; 360 with joint_mode 1)
    // hand-over between the phases of one packet
    i16 lo[SX_BAND], hi[SX_BAND];
};

// Hand-over records between the three stages of the encoder (analysis -> quantiser -> coder); they live in HBM.
struct SxNsqIn {                     // what the quantiser needs of one analysed 20 ms frame
    i32 sigtype, QuantOffsetType, NLSFInterpCoef_Q2, Seed, Lambda_Q10, LTP_scale_Q14, DeltaGains_Q16;
    i32 pitchL[SX_NB_SUBFR], Gains_Q16[SX_NB_SUBFR], LF_shp_Q14[SX_NB_SUBFR], Tilt_Q14[SX_NB_SUBFR], HarmShapeGain_Q14[SX_NB_SUBFR];
    i16 PredCoef_Q12[2][SX_MAX_LPC];
    i16 LTPCoef_Q14[SX_LTP_ORDER * SX_NB_SUBFR];
    i16 AR2_Q13[SX_NB_SUBFR * SX_SHAPE_ORDER];
    i16 xfw[SX_FRAME];               // prefiltered input
};
struct SxNsqOut {                    // what the quantiser produces for one frame
    i32 Seed;                        // dither seed of the winning path (coded)
    i32 r[SX_FRAME];                 // centre excitation Q10 (high-band gain reference)
    // pulses of MD1 / MD2 (the centre stream is never coded).  The quantiser emits every sample of every track with ONE 4-byte store:
    // a side track's lands at the pulse's byte, its upper three bytes (sign bytes) fall on the next three pulses of the row, which
    // are emitted -- and so overwritten -- after it; the last three of a row fall into its four bytes of padding.
    i8 q[2][SX_FRAME + 4];
};
struct alignas(16) SxNsqPersist {    // quantiser state of one stream
    SxNsqTrack trk[SX_N_TRACKS];
};
static_assert(sizeof(SxNsqTrack) % 16 == 0 && sizeof(SxNSQ) % 16 == 0, "track records stay 16-byte aligned");

struct alignas(16) SxEncStream {     // one record per stream in HBM
    SxEncState core;
    SxEncHist hist;
    SxNsqPersist nsq;
    SxNsqIn nsq_in[2];               // fused single-stream path: hand-over records of the current packet
    SxNsqOut nsq_out[2];
};

// Output of the analysis chain for one 20 ms frame = input of the NSQ and of the parameter coder
// (SKP_Silk_encoder_control + _FIX, SKP_Silk_structs.h:241, SKP_Silk_structs_FIX.h:113)
struct SxEncCtrl {
    i32 lagIndex, contourIndex, PERIndex;
    i32 LTPIndex[SX_NB_SUBFR];
    i32 NLSFIndices[SX_NLSF_STAGES];
    i32 NLSFInterpCoef_Q2;
    i32 GainsIndices[SX_NB_SUBFR];
    i32 DeltaGainsIndices;
    i32 Seed;
    i32 LTP_scaleIndex, QuantOffsetType, sigtype;
    i32 pitchL[SX_NB_SUBFR];
    i32 Gains_Q16[SX_NB_SUBFR];
    i32 DeltaGains_Q16;
    i16 PredCoef_Q12[2][SX_MAX_LPC];
    i16 LTPCoef_Q14[SX_LTP_ORDER * SX_NB_SUBFR];
    i32 LTP_scale_Q14;
    i16 AR1_Q13[SX_NB_SUBFR * SX_SHAPE_ORDER];
    i16 AR2_Q13[SX_NB_SUBFR * SX_SHAPE_ORDER];
    i32 LF_shp_Q14[SX_NB_SUBFR];
    i32 GainsPre_Q14[SX_NB_SUBFR], HarmBoost_Q14[SX_NB_SUBFR], Tilt_Q14[SX_NB_SUBFR], HarmShapeGain_Q14[SX_NB_SUBFR];
    i32 Lambda_Q10;
    i32 input_quality_Q14, coding_quality_Q14, pitch_freq_low_Hz;
    i32 current_SNR_dB_Q7, current_SNRPerMD_dB_Q7;
    float md_delta_gain_par;
    i32 sparseness_Q8, predGain_Q16, LTPredCodGain_Q7;
    i32 input_quality_bands_Q15[4], input_tilt_Q15;
    i32 ResNrg[SX_NB_SUBFR], ResNrgQ[SX_NB_SUBFR];
    i32 vadFlag;                     // copy of state->vadFlag for this frame (coded per frame)
};

// SKP_Silk_init_encoder_FIX (SKP_Silk_init_encoder_FIX.c:33) + the first SKP_Silk_control_encoder_FIX
// pass (control_codec_FIX.c:56-130: setup_fs(8), setup_rate, ...) + AGR_Sate_Encoder_Init
// (libBWE/AGR_BWE_SDK_API.c:11-126).  `silk_rate_bps` = targetRate_bps - 1600.
SX_FN void sx_enc_state_init(SxEncStream* rec, i32 silk_rate_bps, i32 useMDIndex, i32 hb_joint = 0, i32 useDTX = 0, i32 fpp = 2) {
    u8* p = (u8*)rec;
    SX_PAR(i, (int)sizeof(SxEncStream)) p[i] = 0;
    wv_sync();
    SxEncState* st = &rec->core;
    st->variable_HP_smth1_Q15 = 200844;
    st->variable_HP_smth2_Q15 = 200844;
    st->first_frame_after_reset = 1;
    st->useMDIndex = useMDIndex;
    st->hb_joint = hb_joint;
    st->useDTX = useDTX;
    st->fpp = fpp;
    // SKP_Silk_VAD_Init, SKP_Silk_VAD.c:39
    for (int b = 0; b < 4; b++) {
        st->vad.NoiseLevelBias[b] = sx_max(50 / (b + 1), 1);
        st->vad.NL[b] = 100 * st->vad.NoiseLevelBias[b];
        st->vad.inv_NL[b] = SX_I32_MAX / st->vad.NL[b];
        st->vad.NrgRatioSmth_Q8[b] = 100 * 256;
    }
    st->vad.counter = 15;
    for (int t = 0; t < SX_N_TRACKS; t++) {
        rec->nsq.trk[t].s.prev_inv_gain_Q16 = 65536;
        for (int k = 0; k < SX_NB_SUBFR; k++) rec->nsq.trk[t].s.gadjPrev[k] = 65536;
    }
    // setup_fs_FIX (control_codec_FIX.c:232): only the CENTRE nsq state gets lagPrev = 100
    st->prevLag = 100;
    st->prev_sigtype = 1;
    st->pf_lagPrev = 100;
    st->LastGainIndex = 1;
    rec->nsq.trk[0].s.lagPrev = 100;
    // setup_rate_FIX (control_codec_FIX.c:319): bitrate -> SNR tables, per description and total
    silk_rate_bps = sx_limit(silk_rate_bps, 5000, 100000);           // enc_API.c:187
    i32 md_rate = silk_rate_bps / 2;
    for (int k = 1; k < 8; k++) {
        if (md_rate < T_target_rate[k]) {
            i32 frac_Q6 = sx_shl(md_rate - T_target_rate[k - 1], 6) / (T_target_rate[k] - T_target_rate[k - 1]);
            st->SNRPerMD_dB_Q7 = sx_shl(T_snr_table_Q1[k - 1], 6) + sx_mul(frac_Q6, T_snr_table_Q1[k] - T_snr_table_Q1[k - 1]);
            break;
        }
    }
    for (int k = 1; k < 8; k++) {
        if (silk_rate_bps <= T_target_rate[k]) {
            i32 frac_Q6 = sx_shl(silk_rate_bps - T_target_rate[k - 1], 6) / (T_target_rate[k] - T_target_rate[k - 1]);
            st->SNR_dB_Q7 = sx_shl(T_snr_table_Q1[k - 1], 6) + sx_mul(frac_Q6, T_snr_table_Q1[k] - T_snr_table_Q1[k - 1]);
            break;
        }
    }
    wv_sync();
}
