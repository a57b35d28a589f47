// solo_enc_kernels.h -- the encoder's analysis / coding kernels and the launch table of one build (the quantiser kernel lives in
// solo_nsq_row.hip).  Compiled once per internal rate like solo_dec_kernels.h: solo_api.hip (SX_FS_KHZ = 8, 16 kHz API rate) and
// solo_api_wb.hip (SX_FS_KHZ = 16, 32 kHz API rate).
#pragma once
#include <hip/hip_runtime.h>
#include "solo_enc.h"
// streams per wavefront of the analysis / coding kernels (solo_enc_k.hip: SX_ENC_GROUP lanes per stream; a constant of both
// compilation passes -- SX_NLANES is 1 in the host pass)
#ifndef SX_ENC_GROUP
#define SX_ENC_GROUP 64
#endif
#define SX_ENC_PER_WAVE (64 / SX_ENC_GROUP)

__global__ void __launch_bounds__(64) SX_K(solo_enc_init_kernel)(SxEncStream* states, int n_streams, int silk_rate_bps, int useMDIndex, int hb_joint, int useDTX, int fpp) {
    const int s = blockIdx.x * SX_ENC_PER_WAVE + (int)(threadIdx.x / SX_ENC_GROUP);
    if (s >= n_streams) return;
    sx_enc_state_init(&states[s], silk_rate_bps, useMDIndex, hb_joint, useDTX, fpp);
}

// Encoder, rows E0-E9, as a three-stage pipeline over HBM hand-over records:
//   A  solo_enc_analysis_kernel  one wavefront per stream: QMF split + analysis chain of every frame of the launch
//   B  solo_nsq_kernel           four streams per wavefront (solo_nsq_row.hip): the delayed-decision quantiser
//   C  solo_enc_coding_kernel    one wavefront per stream: high-band encoder, range coding, payload assembly
#ifdef SX_OUTLINE_WRAPPERS
#define SX_ENTER_FN static __device__ __attribute__((noinline))
#else
#define SX_ENTER_FN __device__ __forceinline__
#endif
SX_ENTER_FN void SX_K(solo_enc_enter)(SxEncWork* w, const SxEncStream* rec) {
    const i32* src = (const i32*)&rec->core;
    i32* dst = (i32*)&w->st;
    SX_PAR(i, (int)(sizeof(SxEncState) / 4)) dst[i] = src[i];
    wv_sync();
}
SX_ENTER_FN void SX_K(solo_enc_leave)(SxEncWork* w, SxEncStream* rec) {
    wv_sync();
    const i32* src = (const i32*)&w->st;
    i32* dst = (i32*)&rec->core;
    SX_PAR(i, (int)(sizeof(SxEncState) / 4)) dst[i] = src[i];
}

// waves-per-SIMD target of the analysis kernel = its register budget (5: <= 96 VGPRs, 4: <= 128).  All 16 analysis workgroups
// of a compute unit (4096 streams / 256 CUs) must be resident BESIDE the quantiser's four (one wave per SIMD, 128 registers,
// solo_nsq_row.hip), or the last ones run as a second round that costs a whole single-wave latency:
// 4 x 96 + 128 registers = the 512 of a SIMD, and 16 x 8 704 B of LDS + the quantiser's 4 x 6 144 B = 160 KB.
#ifndef SX_ANALYSIS_WAVES
#if SX_ENC_GROUP == 64
#define SX_ANALYSIS_WAVES 5
#else
#define SX_ANALYSIS_WAVES 2          // two streams per wavefront: half as many waves per SIMD, twice the registers each
#endif
#endif
#ifndef SX_ANALYSIS_PRIO
#define SX_ANALYSIS_PRIO 3
#endif
__global__ void __launch_bounds__(64, SX_ANALYSIS_WAVES) SX_K(solo_enc_analysis_kernel)(SxEncStream* states, const i16* __restrict__ pcm, int n_streams,
                                                                  int n_packets, int p0, int pc, SxNsqIn* __restrict__ nsq_in,
                                                                  SxCodeIn* __restrict__ code_in) {
#if SX_ENC_GROUP == 64
    __shared__ SxEncWork w;
    const int s = blockIdx.x;
#else
    __shared__ SxEncWork wg_[SX_ENC_PER_WAVE];
    SxEncWork& w = wg_[threadIdx.x / SX_ENC_GROUP];
    const int s = blockIdx.x * SX_ENC_PER_WAVE + (int)(threadIdx.x / SX_ENC_GROUP);
#endif
    if (s >= n_streams) return;
    // the same issue priority as the quantiser's wave (solo_nsq_row.hip): with the quantiser above the analysis waves the encoder is
    // 0.5 % slower, below them 18 % (the quantiser starves); the range coder / coding kernels of older chunks stay at 0
    __builtin_amdgcn_s_setprio(SX_ANALYSIS_PRIO);
#ifdef SX_EXP_STAGGER     // timing experiment: the waves that share a SIMD start a fraction of a frame apart
    {
        const int q = (blockIdx.x >> SX_EXP_STAGGER_SHIFT) & 3;
        for (int i = 0; i < q * SX_EXP_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
    }
#endif
    SxEncStream* rec = &states[s];
#if defined(SX_PROF) && defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long hist_t0_ = wall_clock64();
#endif
#if defined(SX_STOPS) && defined(__HIP_DEVICE_COMPILE__)
    SX_STOPS_ENTER(0)
#endif
    SX_K(solo_enc_enter)(&w, rec);
    for (int p = p0; p < p0 + pc; p++) {          // packets [p0, p0 + pc) of a call of n_packets
        const size_t pk = (size_t)s * n_packets + p;
        sx_enc_stage_a(rec, &w, pcm + pk * (size_t)(SX_FRAME * 2 * SX_UNI(w.st.fpp)), nsq_in + pk * 2, code_in + pk);      // (a packet: fpp frames of 2 x SX_FRAME input samples)
        wv_sync();
    }
    SX_K(solo_enc_leave)(&w, rec);
#if defined(SX_PROF) && defined(__HIP_DEVICE_COMPILE__)
    if (threadIdx.x == 0) {     // histogram of the waves' lifetimes in 50 us bins (100 MHz clock), per SIMD of the CU: g_sx_hist[simd][bin]
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const unsigned long long dt = wall_clock64() - hist_t0_;
        const unsigned bin = (unsigned)(dt / 5000u);
        atomicAdd(&g_sx_hist[(hw >> 4) & 3][bin < 63 ? bin : 63], 1ull);
    }
#endif
}

// Entropy coding, LANE per description: lane l of workgroup g codes description (l & 1) of stream 32 g + (l >> 1).  The coder is a
// serial chain of table look-ups and byte writes; 64 of them advance together.  rcbuf / rcinfo: the launch's scratch, indexed
// [(stream * pc + (p - p0)) * 2 + md] -- consumed by the coding kernel that follows on the same HIP stream.
#ifndef SX_RC_LANES
#define SX_RC_LANES 64               // descriptions per workgroup (an experiment knob: the workgroup's LDS is 2.7 KB + 188 B per description)
#endif
struct SxRcWork {
    SxCdf cdf;
    u32 pw[SX_RC_LANES][SX_RC_PW_ROW / 4];
#ifdef SX_RC_PAD_LDS
    u32 pad[SX_RC_PAD_LDS / 4];
#endif
};
// SX_RC_VGPR_CAP = n: the coder's wave may use 2 n of the SIMD's 512 registers (as SX_NSQ_VGPR_CAP, solo_nsq_row.hip).  64 serial coders with a
// few dozen live values each: the allocator takes what the LDS-limited occupancy leaves (256) unless told otherwise, and every 96 registers
// it holds keep an analysis wave out of the SIMD while it runs
#ifndef SX_RC_VGPR_CAP
#define SX_RC_VGPR_CAP 0
#endif
#if SX_RC_VGPR_CAP > 0
#define SX_RC_CAP_ATTR __attribute__((amdgpu_num_vgpr(SX_RC_VGPR_CAP)))
#else
#define SX_RC_CAP_ATTR
#endif
__global__ void SX_RC_CAP_ATTR __launch_bounds__(64) SX_K(solo_enc_rc_kernel)(const SxEncStream* states, const SxCodeIn* __restrict__ code_in,
                                                               const SxNsqOut* __restrict__ nsq_out, int n_streams, int n_packets, int p0, int pc,
                                                               u8* __restrict__ rcbuf, SxRcInfo* __restrict__ rcinfo, const u8* __restrict__ hbout,
                                                               int slot_bytes, u8* __restrict__ bits, i16* __restrict__ nbytes, i32* status) {
    __shared__ SxRcWork w;
#if defined(__HIP_DEVICE_COMPILE__) && defined(SX_RC_PRIO)
    // 128 wavefronts of 64 serial coders each, holding 35 KB of LDS and 176 registers while they run: the sooner they are gone the sooner
    // the analysis workgroups they keep out are back
    __builtin_amdgcn_s_setprio(SX_RC_PRIO);
#endif
    for (int i = threadIdx.x; i < (int)(sizeof(SxCdf) / 4); i += 64) ((u32*)&w.cdf)[i] = 0;      // (padding entries)
    __syncthreads();
    {
        SxCdf* c = &w.cdf;
#define X(type, name, n) for (int i = threadIdx.x; i < (n); i += 64) c->name[i] = T_##name[i];
        SX_CDF_LIST(X) SX_CDF_LIST_ENC(X)
#undef X
    }
    __syncthreads();
    const int lane = threadIdx.x, md = lane & 1;
    const int s = blockIdx.x * (SX_RC_LANES / 2) + (lane >> 1);
#ifdef SX_RC_PAD_LDS
    if (n_streams < 0) w.pad[lane] = 1;
#endif
    if (lane >= SX_RC_LANES || s >= n_streams) return;
    const SxEncState* st = &states[s].core;
    const int useDTX = st->useDTX, useMDIndex = st->useMDIndex, fpp = st->fpp;
    i32 first_err = 0;
    for (int p = p0; p < p0 + pc; p++) {
        const size_t pk = (size_t)s * n_packets + p;
        const size_t slot = ((size_t)s * pc + (size_t)(p - p0)) * 2 + (size_t)md;
        const SxCodeIn* cin = code_in + pk;
        const SxNsqOut* out2 = nsq_out + pk * 2;
        SxRcInfo info = {0, 0};
        if (!(useDTX && cin->idx[fpp - 1].inDTX))
            sx_code_description(cin->idx, out2[0].Seed, out2[1].Seed, out2[0].q[md], out2[1].q[md], md, useMDIndex, &w.cdf, (u8*)&w.pw[lane][0],
                                rcbuf + slot * SX_RC_BUF_STRIDE, &info, fpp);
        rcinfo[slot] = info;
        if (hbout) {
            // Payload assembly (sx_enc_stage_c_out) by the two lanes that coded the packet's descriptions: each copies the bytes it wrote itself
            // (a lane reads its own stores back), the second one appends the high band's (written by the launch before this one); the byte
            // counts cross between the neighbouring lanes.  A packet in DTX carries the high-band bytes only.
            const int hb_bytes = st->hb_joint ? 4 : 4 * fpp;
            const i32 nbo = __shfl_xor(info.nBytes, 1), ero = __shfl_xor(info.error, 1);
            const i32 nb0 = md ? nbo : info.nBytes, nb1 = md ? info.nBytes : nbo;
            const i32 total = nb0 + nb1 + hb_bytes;
            u8* out = bits + pk * (size_t)slot_bytes;
            const u8* hb = hbout + (slot & ~(size_t)1) * 4;
            i32 ret;
            if (useDTX && cin->idx[fpp - 1].inDTX) {
                if (md) for (int i = 0; i < hb_bytes; i++) out[i] = hb[i];
                else { nbytes[pk * 2] = 0; nbytes[pk * 2 + 1] = 0; }
                ret = hb_bytes;
            } else if ((info.error | ero) || total > slot_bytes || nb0 > SX_MAX_ARITHM_BYTES || nb1 > SX_MAX_ARITHM_BYTES) {
                if (!md) { nbytes[pk * 2] = 0; nbytes[pk * 2 + 1] = 0; }
                ret = -1;
            } else {
                const u8* mine = rcbuf + slot * SX_RC_BUF_STRIDE;
                u8* dst = out + (md ? nb0 : 0);
                for (int i = 0; i < info.nBytes; i++) dst[i] = mine[i];
                if (md) for (int i = 0; i < hb_bytes; i++) out[nb0 + nb1 + i] = hb[i];
                else { nbytes[pk * 2] = (i16)total; nbytes[pk * 2 + 1] = (i16)(nb1 + hb_bytes); }
                ret = total;
            }
            if (ret < 0 && first_err == 0) first_err = ret;
        }
    }
    if (hbout && status && md == 0) {               // first error of the call (the chunks of a call run in order)
        if (p0 == 0) status[s] = first_err;
        else if (first_err != 0 && status[s] == 0) status[s] = first_err;
    }
}

// High-band encoder and payload assembly, one wavefront per stream; the descriptions' bytes come from solo_enc_rc_kernel.
// hbout != NULL: the high band ONLY -- its bytes (8 per packet) go to hbout and solo_enc_out_kernel assembles the payloads after the
// range coder: the launch order of the third stage is then high band, range coder, assembly (the high band needs the quantiser's
// excitation, not the coder's bytes, and its 4096 workgroups are the ones that have to find room between the analysis kernel's; the
// coder's 0.35 ms of serial latency in front of them made them miss the turn of the analysis launches)
__global__ void __launch_bounds__(64, SX_ANALYSIS_WAVES) SX_K(solo_enc_coding_kernel)(SxEncStream* states, const SxCodeIn* __restrict__ code_in,
                                                                const SxNsqOut* __restrict__ nsq_out, int n_streams, int n_packets, int p0,
                                                                int pc, int slot, u8* __restrict__ bits, i16* __restrict__ nbytes, i32* status,
                                                                const u8* __restrict__ rcbuf, const SxRcInfo* __restrict__ rcinfo, u8* __restrict__ hbout) {
#if SX_ENC_GROUP == 64
    __shared__ SxEncWork w;
    const int s = blockIdx.x;
#else
    __shared__ SxEncWork wg_[SX_ENC_PER_WAVE];
    SxEncWork& w = wg_[threadIdx.x / SX_ENC_GROUP];
    const int s = blockIdx.x * SX_ENC_PER_WAVE + (int)(threadIdx.x / SX_ENC_GROUP);
#endif
    if (s >= n_streams) return;
    SxEncStream* rec = &states[s];
#if defined(SX_STOPS) && defined(__HIP_DEVICE_COMPILE__)
    SX_STOPS_ENTER(1)
#endif
    SX_K(solo_enc_enter)(&w, rec);
    i32 first_err = 0;
    for (int p = p0; p < p0 + pc; p++) {
        const size_t pk = (size_t)s * n_packets + p;
        const size_t rs = ((size_t)s * pc + (size_t)(p - p0)) * 2;
        sx_enc_stage_c_hb(rec, &w, code_in + pk, nsq_out + pk * 2);
        wv_sync();
        if (hbout) {
            if (SX_LANE < 8) hbout[rs * 4 + SX_LANE] = w.hb_bytes[SX_LANE];
            wv_sync();
            continue;
        }
        i32 ret = sx_enc_stage_c_out(&w, code_in + pk, rcbuf + rs * SX_RC_BUF_STRIDE, rcbuf + (rs + 1) * SX_RC_BUF_STRIDE, rcinfo + rs,
                                     bits + pk * (size_t)slot, slot, nbytes + pk * 2);
        if (ret < 0 && first_err == 0) first_err = ret;
        wv_sync();
    }
    if (!hbout && status && SX_LANE == 0) {        // first error of the call (the chunks of a call run in order)
        if (p0 == 0) status[s] = first_err;
        else if (first_err != 0 && status[s] == 0) status[s] = first_err;
    }
}

// Payload assembly of the reordered third stage (sx_enc_stage_c_out with everything in global memory): one 16-lane row per stream
__global__ void __launch_bounds__(64) SX_K(solo_enc_out_kernel)(const SxEncStream* __restrict__ states, const SxCodeIn* __restrict__ code_in, int n_streams,
                                                                int n_packets, int p0, int pc, int slot, u8* __restrict__ bits, i16* __restrict__ nbytes,
                                                                i32* status, const u8* __restrict__ rcbuf, const SxRcInfo* __restrict__ rcinfo,
                                                                const u8* __restrict__ hbout) {
    const int s = blockIdx.x * 4 + (int)(threadIdx.x >> 4), l = (int)(threadIdx.x & 15);
    if (s >= n_streams) return;
    const SxEncState* st = &states[s].core;
    const int fpp = st->fpp, useDTX = st->useDTX;
    const int hb_bytes = st->hb_joint ? 4 : 4 * fpp;
    i32 first_err = 0;
    for (int p = p0; p < p0 + pc; p++) {
        const size_t pk = (size_t)s * n_packets + p;
        const size_t rs = ((size_t)s * pc + (size_t)(p - p0)) * 2;
        const SxCodeIn* cin = code_in + pk;
        const u8* hb = hbout + rs * 4;
        u8* out = bits + pk * (size_t)slot;
        i16* nBytesOut = nbytes + pk * 2;
        if (useDTX && cin->idx[fpp - 1].inDTX) {
            if (l == 0) { nBytesOut[0] = 0; nBytesOut[1] = 0; }
            if (l < hb_bytes) out[l] = hb[l];
            continue;
        }
        const i32 nb0 = rcinfo[rs].nBytes, nb1 = rcinfo[rs + 1].nBytes;
        const i32 err = rcinfo[rs].error | rcinfo[rs + 1].error;
        const i32 total = nb0 + nb1 + hb_bytes;
        if (err || total > slot || nb0 > SX_MAX_ARITHM_BYTES || nb1 > SX_MAX_ARITHM_BYTES) {
            if (l == 0) { nBytesOut[0] = 0; nBytesOut[1] = 0; }
            if (first_err == 0) first_err = -1;
            continue;
        }
        const u8* buf0 = rcbuf + rs * SX_RC_BUF_STRIDE;
        const u8* buf1 = rcbuf + (rs + 1) * SX_RC_BUF_STRIDE;
        for (int i = l; i < total; i += 16) out[i] = i < nb0 ? buf0[i] : (i < nb0 + nb1 ? buf1[i - nb0] : hb[i - nb0 - nb1]);
        if (l == 0) { nBytesOut[0] = (i16)total; nBytesOut[1] = (i16)(nb1 + hb_bytes); }
    }
    if (status && l == 0) {
        if (p0 == 0) status[s] = first_err;
        else if (first_err != 0 && status[s] == 0) status[s] = first_err;
    }
}


extern "C" int SX_K(solo_launch_nsq)(void* states, const void* in, void* out, int n_streams, int n_packets, int p0, int pc, unsigned int* started,
                                     void* ring, void* hip_stream);   // solo_nsq_row.hip / solo_nsq_row_wb.hip
extern "C" int SX_K(solo_nsq_workgroups)(int n_streams);
extern "C" size_t SX_K(solo_nsq_ring_bytes)(int n_streams);

#include "solo_enc_ops.h"
static hipError_t SX_K(solo_enc_launch_init)(void* states, int n_streams, int silk_rate_bps, int useMDIndex, int hb_joint, int useDTX, int fpp, hipStream_t s) {
    hipLaunchKernelGGL(SX_K(solo_enc_init_kernel), dim3((n_streams + SX_ENC_PER_WAVE - 1) / SX_ENC_PER_WAVE), dim3(64), 0, s, (SxEncStream*)states, n_streams, silk_rate_bps, useMDIndex, hb_joint, useDTX, fpp);
    return hipGetLastError();
}
static hipError_t SX_K(solo_enc_launch_analysis)(void* states, const int16_t* pcm, int n_streams, int n_packets, int p0, int pc, void* nsq_in,
                                                 void* code_in, hipStream_t s) {
    hipLaunchKernelGGL(SX_K(solo_enc_analysis_kernel), dim3((n_streams + SX_ENC_PER_WAVE - 1) / SX_ENC_PER_WAVE), dim3(64), 0, s, (SxEncStream*)states, pcm, n_streams, n_packets, p0, pc,
                       (SxNsqIn*)nsq_in, (SxCodeIn*)code_in);
    return hipGetLastError();
}
// rc_scratch: [n_streams * pc * 2] byte buffers of SX_RC_BUF_STRIDE, then as many SxRcInfo, then 8 high-band bytes per packet
static size_t SX_K(solo_enc_rc_scratch_bytes)(int n_streams, int pc) {
    return (size_t)n_streams * (size_t)pc * 2 * (SX_RC_BUF_STRIDE + sizeof(SxRcInfo) + 4) + 128;
}
static SxRcInfo* SX_K(solo_enc_rcinfo_of)(u8* rcbuf, int n_streams, int pc) {
    return (SxRcInfo*)(rcbuf + (((size_t)n_streams * (size_t)pc * 2 * SX_RC_BUF_STRIDE + 63) & ~(size_t)63));
}
static u8* SX_K(solo_enc_hbout_of)(u8* rcbuf, int n_streams, int pc) {
    return (u8*)(SX_K(solo_enc_rcinfo_of)(rcbuf, n_streams, pc) + (size_t)n_streams * (size_t)pc * 2);
}
static hipError_t SX_K(solo_enc_launch_rc_)(const void* states, const void* code_in, const void* nsq_out, int n_streams, int n_packets, int p0, int pc,
                                            void* rc_scratch, int assemble, int slot, uint8_t* bits, int16_t* nbytes, int32_t* status, hipStream_t s) {
    u8* rcbuf = (u8*)rc_scratch;
    hipLaunchKernelGGL(SX_K(solo_enc_rc_kernel), dim3((n_streams + SX_RC_LANES / 2 - 1) / (SX_RC_LANES / 2)), dim3(64), 0, s, (const SxEncStream*)states, (const SxCodeIn*)code_in,
                       (const SxNsqOut*)nsq_out, n_streams, n_packets, p0, pc, rcbuf, SX_K(solo_enc_rcinfo_of)(rcbuf, n_streams, pc),
                       assemble ? (const u8*)SX_K(solo_enc_hbout_of)(rcbuf, n_streams, pc) : (const u8*)NULL, slot, bits, nbytes, status);
    return hipGetLastError();
}
static hipError_t SX_K(solo_enc_launch_rc)(const void* states, const void* code_in, const void* nsq_out, int n_streams, int n_packets, int p0, int pc,
                                           void* rc_scratch, hipStream_t s) {
    return SX_K(solo_enc_launch_rc_)(states, code_in, nsq_out, n_streams, n_packets, p0, pc, rc_scratch, 0, 0, NULL, NULL, NULL, s);
}
static hipError_t SX_K(solo_enc_launch_hb_out)(void* states, const void* code_in, const void* nsq_out, int n_streams, int n_packets, int p0, int pc,
                                               int slot, uint8_t* bits, int16_t* nbytes, int32_t* status, const void* rc_scratch, hipStream_t s) {
    u8* rcbuf = (u8*)rc_scratch;
    hipLaunchKernelGGL(SX_K(solo_enc_coding_kernel), dim3((n_streams + SX_ENC_PER_WAVE - 1) / SX_ENC_PER_WAVE), dim3(64), 0, s, (SxEncStream*)states, (const SxCodeIn*)code_in,
                       (const SxNsqOut*)nsq_out, n_streams, n_packets, p0, pc, slot, bits, nbytes, status, (const u8*)rcbuf,
                       (const SxRcInfo*)SX_K(solo_enc_rcinfo_of)(rcbuf, n_streams, pc), (u8*)NULL);
    return hipGetLastError();
}
static hipError_t SX_K(solo_enc_launch_hb)(void* states, const void* code_in, const void* nsq_out, int n_streams, int n_packets, int p0, int pc, void* rc_scratch,
                                           hipStream_t s) {
    u8* rcbuf = (u8*)rc_scratch;
    hipLaunchKernelGGL(SX_K(solo_enc_coding_kernel), dim3((n_streams + SX_ENC_PER_WAVE - 1) / SX_ENC_PER_WAVE), dim3(64), 0, s, (SxEncStream*)states, (const SxCodeIn*)code_in,
                       (const SxNsqOut*)nsq_out, n_streams, n_packets, p0, pc, 0, (u8*)NULL, (i16*)NULL, (i32*)NULL, (const u8*)rcbuf,
                       (const SxRcInfo*)SX_K(solo_enc_rcinfo_of)(rcbuf, n_streams, pc), SX_K(solo_enc_hbout_of)(rcbuf, n_streams, pc));
    return hipGetLastError();
}
static hipError_t SX_K(solo_enc_launch_out)(const void* states, const void* code_in, int n_streams, int n_packets, int p0, int pc, int slot, uint8_t* bits,
                                            int16_t* nbytes, int32_t* status, const void* rc_scratch, hipStream_t s) {
    u8* rcbuf = (u8*)rc_scratch;
    hipLaunchKernelGGL(SX_K(solo_enc_out_kernel), dim3((n_streams + 3) / 4), dim3(64), 0, s, (const SxEncStream*)states, (const SxCodeIn*)code_in, n_streams, n_packets,
                       p0, pc, slot, bits, nbytes, status, (const u8*)rcbuf, (const SxRcInfo*)SX_K(solo_enc_rcinfo_of)(rcbuf, n_streams, pc),
                       (const u8*)SX_K(solo_enc_hbout_of)(rcbuf, n_streams, pc));
    return hipGetLastError();
}
// order 0: range coder, then high band + assembly (one kernel); order 1: high band, then range coder + assembly (one kernel); order 2: high
// band, range coder, assembly kernel
static hipError_t SX_K(solo_enc_launch_coding)(void* states, const void* code_in, const void* nsq_out, int n_streams, int n_packets, int p0, int pc,
                                               int slot, uint8_t* bits, int16_t* nbytes, int32_t* status, void* rc_scratch, int order, hipStream_t s) {
    if (!order) {
        const hipError_t e = SX_K(solo_enc_launch_rc)(states, code_in, nsq_out, n_streams, n_packets, p0, pc, rc_scratch, s);
        if (e != hipSuccess) return e;
        return SX_K(solo_enc_launch_hb_out)(states, code_in, nsq_out, n_streams, n_packets, p0, pc, slot, bits, nbytes, status, rc_scratch, s);
    }
    hipError_t e = SX_K(solo_enc_launch_hb)(states, code_in, nsq_out, n_streams, n_packets, p0, pc, rc_scratch, s);
    if (e != hipSuccess) return e;
    if (order == 1) return SX_K(solo_enc_launch_rc_)(states, code_in, nsq_out, n_streams, n_packets, p0, pc, rc_scratch, 1, slot, bits, nbytes, status, s);
    e = SX_K(solo_enc_launch_rc)(states, code_in, nsq_out, n_streams, n_packets, p0, pc, rc_scratch, s);
    if (e != hipSuccess) return e;
    return SX_K(solo_enc_launch_out)(states, code_in, n_streams, n_packets, p0, pc, slot, bits, nbytes, status, rc_scratch, s);
}
static const solo_enc_ops SX_K(solo_enc_ops_table) = {
    sizeof(SxEncStream), sizeof(SxNsqIn), sizeof(SxNsqOut), sizeof(SxCodeIn), SX_PACKET,
    SX_K(solo_enc_launch_init), SX_K(solo_enc_launch_analysis), SX_K(solo_launch_nsq), SX_K(solo_enc_launch_coding), SX_K(solo_enc_launch_rc), SX_K(solo_enc_launch_hb_out), SX_K(solo_enc_launch_hb), SX_K(solo_enc_launch_out), SX_K(solo_enc_rc_scratch_bytes), SX_K(solo_nsq_workgroups), SX_K(solo_nsq_ring_bytes)};
