/* Test infrastructure only: stage taps for the compiled reference encoder.
 *
 * Linked into oracle/_ref/libsolo_ref_fix_taps.so together with the UNMODIFIED reference sources, with
 * -Wl,--wrap=<stage function> for each analysis stage of SKP_Silk_encode_frame_FIX (encode_frame_FIX.c:100-143).
 * Every wrapper calls the real function and then appends a canonical int32 record of the encoder control block
 * to a global tap buffer that the tests read through ctypes and compare with the same record produced by the
 * host emulation of the kernel source (tests/emu).  Nothing here is part of the product. */
#include <string.h>
#include "SKP_Silk_main_FIX.h"

#define TAP_REC 1024
#define TAP_MAX 64
int solo_tap_n = 0;
int solo_tap_buf[TAP_MAX][TAP_REC];

static int *put(int *p, const void *src, int n, int elem) {
    int i;
    for (i = 0; i < n; i++) {
        if (elem == 4) *p++ = ((const int *)src)[i];
        else if (elem == 2) *p++ = ((const short *)src)[i];
        else *p++ = ((const signed char *)src)[i];
    }
    return p;
}

static void tap(int stage, SKP_Silk_encoder_state_FIX *psEnc, SKP_Silk_encoder_control_FIX *c, const short *xfw,
                const signed char *q, signed char **q_md, const int *r) {
    int *p, *p0;
    if (solo_tap_n >= TAP_MAX) return;
    p0 = p = solo_tap_buf[solo_tap_n++];
    memset(p, 0, sizeof(int) * TAP_REC);
    *p++ = stage;
    *p++ = c->sCmn.sigtype; *p++ = c->sCmn.QuantOffsetType; *p++ = c->sCmn.lagIndex; *p++ = c->sCmn.contourIndex;
    *p++ = c->sCmn.PERIndex;
    p = put(p, c->sCmn.LTPIndex, 4, 4);
    p = put(p, c->sCmn.NLSFIndices, 6, 4);
    *p++ = c->sCmn.NLSFInterpCoef_Q2;
    p = put(p, c->sCmn.GainsIndices, 4, 4);
    *p++ = c->sCmn.DeltaGainsIndices; *p++ = c->sCmn.Seed; *p++ = c->sCmn.LTP_scaleIndex;
    p = put(p, c->sCmn.pitchL, 4, 4);
    p = put(p, c->Gains_Q16, 4, 4);
    *p++ = c->DeltaGains_Q16;
    p = put(p, c->PredCoef_Q12[0], 10, 2);
    p = put(p, c->PredCoef_Q12[1], 10, 2);
    p = put(p, c->LTPCoef_Q14, 20, 2);
    *p++ = c->LTP_scale_Q14;
    p = put(p, c->AR1_Q13, 64, 2);
    p = put(p, c->AR2_Q13, 64, 2);
    p = put(p, c->LF_shp_Q14, 4, 4);
    p = put(p, c->GainsPre_Q14, 4, 4);
    p = put(p, c->HarmBoost_Q14, 4, 4);
    p = put(p, c->Tilt_Q14, 4, 4);
    p = put(p, c->HarmShapeGain_Q14, 4, 4);
    *p++ = c->Lambda_Q10; *p++ = c->input_quality_Q14; *p++ = c->coding_quality_Q14; *p++ = c->current_SNR_dB_Q7;
    *p++ = c->sparseness_Q8; *p++ = c->predGain_Q16; *p++ = c->LTPredCodGain_Q7;
    p = put(p, c->input_quality_bands_Q15, 4, 4);
    *p++ = c->input_tilt_Q15;
    p = put(p, c->ResNrg, 4, 4);
    p = put(p, c->ResNrgQ, 4, 4);
    *p++ = psEnc->speech_activity_Q8; *p++ = psEnc->LTPCorr_Q15;
    /* p0 + 256: signals */
    p = p0 + 256;
    if (xfw) put(p, xfw, 160, 2);
    p += 160;
    if (q) put(p, q, 160, 1);
    p += 160;
    if (q_md) { put(p, q_md[0], 160, 1); put(p + 160, q_md[1], 160, 1); }
    p += 320;
    if (r) put(p, r, 160, 4);
}

static SKP_Silk_encoder_state_FIX *cur_enc;
static const short *cur_xfw;

void __real_SKP_Silk_find_pitch_lags_FIX(SKP_Silk_encoder_state_FIX *, SKP_Silk_encoder_control_FIX *, SKP_int16 *, const SKP_int16 *);
void __wrap_SKP_Silk_find_pitch_lags_FIX(SKP_Silk_encoder_state_FIX *psEnc, SKP_Silk_encoder_control_FIX *c, SKP_int16 *res, const SKP_int16 *x) {
    cur_enc = psEnc;
    __real_SKP_Silk_find_pitch_lags_FIX(psEnc, c, res, x);
    tap(1, psEnc, c, x + 40, 0, 0, 0);          /* signal slot: the high-passed input frame */
}
void __real_SKP_Silk_noise_shape_analysis_FIX(SKP_Silk_encoder_state_FIX *, SKP_Silk_encoder_control_FIX *, const SKP_int16 *, const SKP_int16 *);
void __wrap_SKP_Silk_noise_shape_analysis_FIX(SKP_Silk_encoder_state_FIX *psEnc, SKP_Silk_encoder_control_FIX *c, const SKP_int16 *pr, const SKP_int16 *x) {
    __real_SKP_Silk_noise_shape_analysis_FIX(psEnc, c, pr, x);
    tap(2, psEnc, c, pr, 0, 0, 0);              /* signal slot: pitch residual of the frame */
}
void __real_SKP_Silk_prefilter_FIX(SKP_Silk_encoder_state_FIX *, const SKP_Silk_encoder_control_FIX *, SKP_int16 *, const SKP_int16 *);
void __wrap_SKP_Silk_prefilter_FIX(SKP_Silk_encoder_state_FIX *psEnc, const SKP_Silk_encoder_control_FIX *c, SKP_int16 *xw, const SKP_int16 *x) {
    __real_SKP_Silk_prefilter_FIX(psEnc, c, xw, x);
    cur_xfw = xw;
    tap(3, psEnc, (SKP_Silk_encoder_control_FIX *)c, xw, 0, 0, 0);
}
void __real_SKP_Silk_find_pred_coefs_FIX(SKP_Silk_encoder_state_FIX *, SKP_Silk_encoder_control_FIX *, const SKP_int16 *);
void __wrap_SKP_Silk_find_pred_coefs_FIX(SKP_Silk_encoder_state_FIX *psEnc, SKP_Silk_encoder_control_FIX *c, const SKP_int16 *res) {
    __real_SKP_Silk_find_pred_coefs_FIX(psEnc, c, res);
    tap(4, psEnc, c, cur_xfw, 0, 0, 0);
}
void __real_SKP_Silk_process_gains_FIX(SKP_Silk_encoder_state_FIX *, SKP_Silk_encoder_control_FIX *);
void __wrap_SKP_Silk_process_gains_FIX(SKP_Silk_encoder_state_FIX *psEnc, SKP_Silk_encoder_control_FIX *c) {
    __real_SKP_Silk_process_gains_FIX(psEnc, c);
    tap(5, psEnc, c, cur_xfw, 0, 0, 0);
}
/* ---- isolated quantiser check (tests/test_gpu_nsq_taps.py): the ARGUMENTS of every SKP_Silk_NSQ_del_dec call (SKP_Silk_NSQ_del_dec.c:925) in the
 * layout of the build's hand-over record SxNsqIn (solo_amd/csrc/solo_enc_state.h: 660 bytes at the 8 kHz internal rate), and its outputs. */
#define NSQ_TAP_MAX 512
struct nsq_tap_in {
    int sigtype, QuantOffsetType, NLSFInterpCoef_Q2, Seed, Lambda_Q10, LTP_scale_Q14, DeltaGains_Q16;
    int pitchL[4], Gains_Q16[4], LF_shp_Q14[4], Tilt_Q14[4], HarmShapeGain_Q14[4];
    short PredCoef_Q12[2][16];
    short LTPCoef_Q14[20];
    short AR2_Q13[64];
    short xfw[160];
};
struct nsq_tap_out { int Seed; signed char q[2][160]; int r[160]; };
int solo_nsq_tap_n = 0;
struct nsq_tap_in solo_nsq_tap_in[NSQ_TAP_MAX];
struct nsq_tap_out solo_nsq_tap_out[NSQ_TAP_MAX];
int solo_nsq_tap_sizeof_in(void) { return (int)sizeof(struct nsq_tap_in); }
int solo_nsq_tap_sizeof_out(void) { return (int)sizeof(struct nsq_tap_out); }
static void nsq_tap_before(SKP_Silk_encoder_state *psEncC, SKP_Silk_encoder_control *c, const SKP_int16 *x, int LSFInterpFactor_Q2,
                           const SKP_int16 *PredCoef_Q12, const SKP_int16 *LTPCoef_Q14, const SKP_int16 *AR2_Q13, const SKP_int *HarmShapeGain_Q14,
                           const SKP_int *Tilt_Q14, const SKP_int32 *LF_shp_Q14, const SKP_int32 *Gains_Q16, SKP_int32 DeltaGains_Q16, int Lambda_Q10,
                           int LTP_scale_Q14) {
    struct nsq_tap_in *t;
    if (solo_nsq_tap_n >= NSQ_TAP_MAX || psEncC->frame_length != 160) return;
    t = &solo_nsq_tap_in[solo_nsq_tap_n];
    memset(t, 0, sizeof(*t));
    t->sigtype = c->sigtype; t->QuantOffsetType = c->QuantOffsetType; t->NLSFInterpCoef_Q2 = LSFInterpFactor_Q2; t->Seed = c->Seed;
    t->Lambda_Q10 = Lambda_Q10; t->LTP_scale_Q14 = LTP_scale_Q14; t->DeltaGains_Q16 = DeltaGains_Q16;
    memcpy(t->pitchL, c->pitchL, sizeof(t->pitchL));
    memcpy(t->Gains_Q16, Gains_Q16, sizeof(t->Gains_Q16));
    memcpy(t->LF_shp_Q14, LF_shp_Q14, sizeof(t->LF_shp_Q14));
    memcpy(t->Tilt_Q14, Tilt_Q14, sizeof(t->Tilt_Q14));
    memcpy(t->HarmShapeGain_Q14, HarmShapeGain_Q14, sizeof(t->HarmShapeGain_Q14));
    memcpy(t->PredCoef_Q12, PredCoef_Q12, sizeof(t->PredCoef_Q12));          /* SKP_int16 PredCoef_Q12[2][MAX_LPC_ORDER] */
    memcpy(t->LTPCoef_Q14, LTPCoef_Q14, sizeof(t->LTPCoef_Q14));
    memcpy(t->AR2_Q13, AR2_Q13, sizeof(t->AR2_Q13));
    memcpy(t->xfw, x, sizeof(t->xfw));
}
static void nsq_tap_after(SKP_Silk_encoder_control *c, SKP_int8 **q_md, const SKP_int32 *r) {
    struct nsq_tap_out *t;
    if (solo_nsq_tap_n >= NSQ_TAP_MAX) return;
    t = &solo_nsq_tap_out[solo_nsq_tap_n++];
    t->Seed = c->Seed;
    memcpy(t->q[0], q_md[0], 160); memcpy(t->q[1], q_md[1], 160);
    memcpy(t->r, r, sizeof(t->r));
}
void __real_SKP_Silk_NSQ_del_dec(SKP_Silk_encoder_state *, SKP_Silk_encoder_control *, SKP_Silk_nsq_state *, SKP_Silk_nsq_state *,
    const SKP_int16 *, SKP_int8 *, SKP_int8 **, SKP_int32 *, const SKP_int, const SKP_int16 *, const SKP_int16 *, const SKP_int16 *,
    const SKP_int *, const SKP_int *, const SKP_int32 *, const SKP_int32 *, const SKP_int32 *, const SKP_int32, const SKP_int, const SKP_int);
void __wrap_SKP_Silk_NSQ_del_dec(SKP_Silk_encoder_state *psEncC, SKP_Silk_encoder_control *psEncCtrlC, SKP_Silk_nsq_state *NSQ,
    SKP_Silk_nsq_state *NSQ_md, const SKP_int16 *x, SKP_int8 *q, SKP_int8 **q_md, SKP_int32 *r, const SKP_int LSFInterpFactor_Q2,
    const SKP_int16 *PredCoef_Q12, const SKP_int16 *LTPCoef_Q14, const SKP_int16 *AR2_Q13, const SKP_int *HarmShapeGain_Q14,
    const SKP_int *Tilt_Q14, const SKP_int32 *LF_shp_Q14, const SKP_int32 *Gains_Q16, const SKP_int32 *MDGains_Q16,
    const SKP_int32 DeltaGains_Q16, const SKP_int Lambda_Q10, const SKP_int LTP_scale_Q14) {
    nsq_tap_before(psEncC, psEncCtrlC, x, LSFInterpFactor_Q2, PredCoef_Q12, LTPCoef_Q14, AR2_Q13, HarmShapeGain_Q14, Tilt_Q14, LF_shp_Q14, Gains_Q16,
                   DeltaGains_Q16, Lambda_Q10, LTP_scale_Q14);
    __real_SKP_Silk_NSQ_del_dec(psEncC, psEncCtrlC, NSQ, NSQ_md, x, q, q_md, r, LSFInterpFactor_Q2, PredCoef_Q12, LTPCoef_Q14, AR2_Q13,
        HarmShapeGain_Q14, Tilt_Q14, LF_shp_Q14, Gains_Q16, MDGains_Q16, DeltaGains_Q16, Lambda_Q10, LTP_scale_Q14);
    nsq_tap_after(psEncCtrlC, q_md, r);
    /* sCmn is the first member of the FIX control / state structs */
    tap(6, cur_enc, (SKP_Silk_encoder_control_FIX *)psEncCtrlC, x, q, q_md, r);
}
