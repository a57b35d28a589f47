// Host emulation build of the KERNEL SOURCE (solo_amd/csrc/*.h) -- test infrastructure only.
//
// The codec kernels are written wave-uniform (solo_wave.h); compiled without hipcc the same source
// runs with SX_NLANES == 1, which lets the CPU-side tests (-m "not gpu") check the kernel source
// against the reference / golden vectors in a container that has no GPU.  Nothing in the product
// library (solo_amd/csrc/solo_api.hip) links or calls this file.
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
// -DSX_FS_KHZ=16 -DEMU_DEC_ONLY builds the 32 kHz-mode decoder (libsolo_emu_wb.so): same entry points, 1280-sample packets.
#ifndef EMU_DEC_ONLY
struct SxEncState; struct SxEncWork;
static void emu_tap(int stage, const SxEncState* st, const SxEncWork* w, const int16_t* sig);
#define SX_ENC_TAP(stage, st, w, sig) emu_tap(stage, st, w, sig)
#endif
#include "../../solo_amd/csrc/solo_dec.h"
#include "../../solo_amd/csrc/solo_l0_probe.h"
#include "../../solo_amd/csrc/solo_recv.h"
#ifndef EMU_DEC_ONLY
#include "../../solo_amd/csrc/solo_enc.h"
#endif

extern "C" {

struct EmuDec { SxDecState st; SxDecWork w; SxDecShadow sh; int useMDIndex; SxExtractLane L; int two_step; };

void* emu_dec_create(int useMDIndex) {                    // bit 1 of the argument: joint_mode 1 (40 ms high-band frame), bit 3: framesize_ms 20
    EmuDec* d = (EmuDec*)calloc(1, sizeof(EmuDec));
    sx_dec_state_init(&d->st, ((useMDIndex >> 1) & 1) | (((useMDIndex >> 3) & 1) << 1));
    d->useMDIndex = useMDIndex & 1;
    return d;
}
void emu_dec_destroy(void* h) { free(h); }
// on: decode like the batch path's two kernels -- the packet's descriptions go through the history-free extraction step
// (sx_extract_desc) first, the decoder proper then continues from its records where they are usable
void emu_dec_set_split(void* h, int on) { ((EmuDec*)h)->two_step = on; }
static int emu_usable_count = 0, emu_fallback_count = 0;
int emu_dec_two_step_stats(int which) { return which ? emu_fallback_count : emu_usable_count; }
int emu_dec_packet(void* h, const uint8_t* bits, int nBytes0, int nBytes1, int lostflag, int16_t* pcm) {
    EmuDec* d = (EmuDec*)h;
    sx_cdf_load_dec(&d->w.cdf);
    SxExtracted ext2[2];
    if (d->two_step) {
        memset(ext2, 0xA5, sizeof(ext2));                 // (whatever the decoder takes from a record must have been written by the extraction)
        for (int md = 0; md < 2; md++) {                  // solo_dec_extract_kernel, one lane per description slot
            ext2[md].usable = 0;
            i32 off = 0, len = 0, hb_off = -1;
            int sel = 0;
            if (sx_desc_span(lostflag, nBytes0, nBytes1, d->st.hb_joint, md, &off, &len, &sel, &hb_off))
                sx_extract_desc(bits + off, len, d->useMDIndex, (const SxCdf*)&d->w.cdf, &d->L, &ext2[md], sel, hb_off >= 0 ? bits + hb_off : 0, d->st.hb_joint);
        }
        if (lostflag >= 2) { if (sx_extracted_usable(&d->st, ext2, lostflag)) emu_usable_count++; else emu_fallback_count++; }
    }
    d->w.st = d->st;                                      // the kernel keeps state + tables in LDS for a launch
    d->w.shadow = &d->sh;
    int r = sx_decode_packet(&d->w, bits, nBytes0, nBytes1, lostflag, d->useMDIndex, pcm, d->two_step ? ext2 : 0);
    d->st = d->w.st;
    return r;
}
// L0 vocabulary, operand by operand (tests/test_l0_primitives.py)
void emu_l0(int op, int n, const int32_t* a, const int32_t* b, const int32_t* c, int32_t* out) {
    for (int i = 0; i < n; i++) out[i] = sx_l0_probe(op, a[i], b[i], c[i]);
}
void emu_sum_sqr_shift(const int16_t* x, int len, int odd_start, int32_t* energy, int32_t* shift) { sx_sum_sqr_shift(energy, shift, x, len, odd_start); }
int emu_sizeof_dec_state() { return (int)sizeof(SxDecState); }
int emu_sizeof_dec_work() { return (int)sizeof(SxDecWork); }
int emu_packet_samples() { return SX_PACKET; }

// receiver staging ring: the bookkeeping of solo_recv_insert_kernel (window check, slot claim), one arrival after the other
void emu_recv_file(const int32_t* arrivals, int n, int n_streams, int depth, int slot, const uint8_t* payload, long long payload_bytes, int useMDIndex,
                   const int32_t* play, uint32_t* lens, int32_t* verdict, int32_t* slot_out) {
    for (int a = 0; a < n; a++) {
        SxRecvArrival r = {arrivals[5 * a], arrivals[5 * a + 1], arrivals[5 * a + 2], arrivals[5 * a + 3], arrivals[5 * a + 4]};
        int sl = -1;
        int v = sx_recv_file(&r, payload, payload_bytes, n_streams, depth, slot, useMDIndex, play, lens, 1, &sl);
        if (v == SX_RECV_INSERTED && sl < 0) v = SX_RECV_DUP;
        verdict[a] = v; slot_out[a] = sl;
    }
}

}  // extern "C"

// debug aid: raw view of the decoder state (tests only)
extern "C" const void* emu_dec_state_ptr(void* h) { return &((EmuDec*)h)->st; }

#ifndef EMU_DEC_ONLY
// ---- encoder ----
extern "C" {
struct EmuEnc { SxEncStream rec; SxEncWork w; SxCodeIn cin; };
void* emu_enc_create(int rate_bps, int useMDIndex) {        // bit 1 of the second argument: joint_mode 1, bit 2: DTX, bit 3: framesize_ms 20
    EmuEnc* e = (EmuEnc*)calloc(1, sizeof(EmuEnc));
    const int joint = (useMDIndex >> 1) & 1;
    sx_enc_state_init(&e->rec, rate_bps - (joint ? 800 : 1600), useMDIndex & 1, joint, (useMDIndex >> 2) & 1, ((useMDIndex >> 3) & 1) ? 1 : 2);   // AGR_BWE_SDK_API.c:119
    return e;
}
void emu_enc_destroy(void* h) { free(h); }
int emu_enc_packet(void* h, const int16_t* pcm, uint8_t* bits, int buf_size, int16_t* nBytesOut) {
    EmuEnc* e = (EmuEnc*)h;
    e->w.st = e->rec.core;                                    // the kernel keeps the compact state in LDS for a launch
    int r = sx_encode_packet(&e->rec, &e->w, &e->cin, pcm, bits, buf_size, nBytesOut);
    e->rec.core = e->w.st;
    return r;
}
// the quantiser alone (tests/test_nsq_taps.py): n frames of ONE freshly initialised stream, in = SxNsqIn[n], out = SxNsqOut[n]
int emu_nsq_frames(const void* in, int n, void* out) {
    EmuEnc* e = (EmuEnc*)emu_enc_create(13600, 0);
    for (int f = 0; f < n; f++)
        sx_nsq_del_dec((char*)&e->rec.nsq, 0u, (const SxNsqIn*)in + f, (char*)((SxNsqOut*)out + f), 0u, &e->w.u.nsq, e->w.u.nsq.ring_emu, 0u, 12);
    emu_enc_destroy(e);
    return (int)sizeof(SxNsqOut);
}
// the pulse coder alone (tests/test_pulse_coder.py): one frame of pulses through a fresh range coder -> its bytes (SKP_Silk_encode_pulses.c:55)
int emu_encode_pulses(int sigtype, int QuantOffsetType, const int8_t* q, uint8_t* out, int* error) {
    static SxCdf cdf;
    sx_cdf_load(&cdf);
    alignas(4) u8 pw[SX_RC_PW_ROW];
    alignas(4) i8 qa[SX_FRAME + 4];
    memcpy(qa, q, SX_FRAME);
    static u8 buf[SX_RC_BUF_STRIDE];
    memset(buf, 0, sizeof(buf));
    SxRangeEnc rc;
    sx_rc_enc_init(&rc, buf);
    sx_encode_pulses(&rc, sigtype, QuantOffsetType, qa, &cdf, pw);
    i32 nb;
    sx_rc_length_bits(rc.bufferIx, rc.range_Q16, &nb);
    sx_rc_enc_wrap_up(&rc);
    if (nb > 0 && nb <= (i32)sizeof(buf)) memcpy(out, buf, (size_t)nb);
    *error = rc.error;
    return nb;
}
int emu_frame_samples() { return SX_FRAME; }
int emu_sizeof_nsq_in() { return (int)sizeof(SxNsqIn); }
int emu_sizeof_enc_state() { return (int)sizeof(SxEncStream); }
int emu_sizeof_enc_work() { return (int)sizeof(SxEncWork); }
// debug taps (tests only): last frame's control block, pulses and residual
const void* emu_enc_ctrl_ptr(void* h) { return &((EmuEnc*)h)->w.ctrl; }
int emu_sizeof_enc_ctrl() { return (int)sizeof(SxEncCtrl); }
const void* emu_enc_q_ptr(void* h) { return &((EmuEnc*)h)->rec.nsq_out[0].q[0][0]; }
const void* emu_enc_idx_ptr(void* h) { return &((EmuEnc*)h)->cin.idx[0]; }
const void* emu_enc_state_ptr(void* h) { return &((EmuEnc*)h)->rec; }
}

// ---- stage taps: same canonical record as oracle/ref_taps.c ----
#define TAP_REC 1024
#define TAP_MAX 64
extern "C" { int solo_tap_n = 0; int solo_tap_buf[TAP_MAX][TAP_REC]; }
template <typename T> static int* tput(int* p, const T* src, int n) { for (int i = 0; i < n; i++) *p++ = (int)src[i]; return p; }
static void emu_tap(int stage, const SxEncState* st, const SxEncWork* w, const int16_t* sig) {
    if (solo_tap_n >= TAP_MAX) return;
    const SxEncCtrl* c = &w->ctrl;
    const int frame = stage >> 4;
    stage &= 15;
    int* p0 = solo_tap_buf[solo_tap_n++];
    int* p = p0;
    memset(p, 0, sizeof(int) * TAP_REC);
    *p++ = stage;
    *p++ = c->sigtype; *p++ = c->QuantOffsetType; *p++ = c->lagIndex; *p++ = c->contourIndex; *p++ = c->PERIndex;
    p = tput(p, c->LTPIndex, 4); p = tput(p, c->NLSFIndices, 6); *p++ = c->NLSFInterpCoef_Q2;
    p = tput(p, c->GainsIndices, 4); *p++ = c->DeltaGainsIndices; *p++ = c->Seed; *p++ = c->LTP_scaleIndex;
    p = tput(p, c->pitchL, 4); p = tput(p, c->Gains_Q16, 4); *p++ = c->DeltaGains_Q16;
    p = tput(p, c->PredCoef_Q12[0], 10); p = tput(p, c->PredCoef_Q12[1], 10); p = tput(p, c->LTPCoef_Q14, 20);
    *p++ = c->LTP_scale_Q14;
    p = tput(p, c->AR1_Q13, 64); p = tput(p, c->AR2_Q13, 64); p = tput(p, c->LF_shp_Q14, 4); p = tput(p, c->GainsPre_Q14, 4);
    p = tput(p, c->HarmBoost_Q14, 4); p = tput(p, c->Tilt_Q14, 4); p = tput(p, c->HarmShapeGain_Q14, 4);
    *p++ = c->Lambda_Q10; *p++ = c->input_quality_Q14; *p++ = c->coding_quality_Q14; *p++ = c->current_SNR_dB_Q7;
    *p++ = c->sparseness_Q8; *p++ = c->predGain_Q16; *p++ = c->LTPredCodGain_Q7;
    p = tput(p, c->input_quality_bands_Q15, 4); *p++ = c->input_tilt_Q15;
    p = tput(p, c->ResNrg, 4); p = tput(p, c->ResNrgQ, 4);
    *p++ = st->speech_activity_Q8; *p++ = st->LTPCorr_Q15;
    p = p0 + 256;
    if (sig) tput(p, sig, 160);
    p += 160;
    if (stage == 6) {
        p += 160;                                   // centre pulses are not kept by the kernel source
        p += 320 + 160;                             // pulses / excitation now live in the stream's HBM record, see test hooks
    }
}

#endif  // EMU_DEC_ONLY
