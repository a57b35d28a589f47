// solo_enc_kernels.h -- the encoder's analysis / coding kernels and the launch table of one build (the quantiser kernel lives in
// solo_nsq_row.hip).  Compiled once per internal rate like solo_dec_kernels.h: solo_api.hip (SX_FS_KHZ = 8, 16 kHz API rate) and
// solo_api_wb.hip (SX_FS_KHZ = 16, 32 kHz API rate).
#pragma once
#include <hip/hip_runtime.h>
#include "solo_enc.h"

#ifndef SX_TU_FRONT       // (the front kernel lives in a translation unit of its own, solo_enc_front_k.hip: see there)
__global__ void __launch_bounds__(64) SX_K(solo_enc_init_kernel)(SxEncStream* states, int n_streams, int silk_rate_bps, int useMDIndex, int hb_joint, int useDTX, int fpp) {
    const int s = blockIdx.x;
    if (s >= n_streams) return;
    sx_enc_state_init(&states[s], silk_rate_bps, useMDIndex, hb_joint, useDTX, fpp);
}

// Encoder, rows E0-E9, over HBM hand-over records.  Two schedules of the same stage functions (solo_api.hip picks one per call):
//  * launch per chunk (the default):
//      A  solo_enc_analysis_kernel  one wavefront per stream: QMF split + analysis chain of the chunk's packets
//      B  solo_nsq_kernel           four streams per wavefront (solo_nsq_row.hip): the delayed-decision quantiser
//      C  solo_enc_coding_kernel    one wavefront per stream: high-band encoder (8 bytes per packet into the launch's scratch)
//      D  solo_enc_rc_kernel        one LANE per description: range coding, then the payload assembly by the two lanes of a packet
//  * persistent (SOLO_ENC_PERSIST=1, calls of two or more packets): ONE launch of solo_enc_front_kernel (one wavefront per stream:
//    analysis of every packet, and -- as soon as the quantiser has published a packet -- its high band, range coding and payload) beside
//    ONE launch of solo_nsq_persist_kernel (solo_nsq_row.hip); the two hand packets to each other through per-stream flags in HBM while
//    they run (solo_wave.h: "publish / consume"), so no launch ever waits for the stragglers of another.  Measured slower than the
//    launch-per-chunk schedule, kept as the measured answer to "remove the launch tails".
#endif
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
#define SX_ANALYSIS_WAVES 5
#endif
#ifndef SX_ANALYSIS_PRIO
#define SX_ANALYSIS_PRIO 3
#endif
#ifndef SX_TU_FRONT
__global__ void __launch_bounds__(64, SX_ANALYSIS_WAVES) SX_K(solo_enc_analysis_kernel)(SxEncStream* states, const i16* __restrict__ pcm, int n_streams,
                                                                  int n_packets, int p0, int pc, SxNsqIn* __restrict__ nsq_in,
                                                                  SxCodeIn* __restrict__ code_in) {
    __shared__ SxEncWork w;
    const int s = blockIdx.x;
    if (s >= n_streams) return;
    // the same issue priority as the quantiser's wave (solo_nsq_row.hip): with the quantiser above the analysis waves the encoder is
    // 0.5 % slower, below them 18 % (the quantiser starves); the range coder / coding kernels of older chunks stay at 0
    __builtin_amdgcn_s_setprio(SX_ANALYSIS_PRIO);
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

#endif
// Entropy coding, LANE per description: lane l of workgroup g codes description (l & 1) of stream 32 g + (l >> 1).  The coder is a
// serial chain of table look-ups and byte writes; 64 of them advance together.  rcbuf / rcinfo: the launch's scratch, indexed
// [(stream * pc + (p - p0)) * 2 + md]; hbout: the high band's bytes of the launch before this one on the same HIP stream.
#ifndef SX_RC_LANES
#define SX_RC_LANES 64               // descriptions per workgroup (the workgroup's LDS is 3.1 KB of tables + 204 B per description)
#endif
struct SxRcWork {
    SxCdf cdf;
    u32 pw[SX_RC_LANES][SX_RC_PW_ROW / 4];
};
// the coder's tables into LDS, by the 64 lanes of ONE wavefront (the front kernel's workgroups hold several, each with its own copy)
__device__ __forceinline__ void SX_K(sx_cdf_stage)(SxCdf* c) {
    const int lane = (int)(threadIdx.x & 63u);
    for (int i = lane; i < (int)(sizeof(SxCdf) / 4); i += 64) ((u32*)c)[i] = 0;      // (padding entries)
    wv_sync();
#define X(type, name, n) for (int i = lane; i < (n); i += 64) c->name[i] = T_##name[i];
    SX_CDF_LIST(X) SX_CDF_LIST_ENC(X)
#undef X
    wv_sync();
}

// One lane: codes description md of a packet (unless the packet is in DTX), then -- with the neighbouring lane, which coded the packet's
// other description -- assembles the payload MD1 || MD2 || HB (sx_enc_stage_c_out): each lane copies the bytes it wrote itself (it reads
// its own stores back), the second one appends the high band's; the byte counts cross between the two lanes.  A packet in DTX carries the
// high-band bytes only.  Returns what AGR_Sate_Encoder_Encode returns for the packet (a negative value: it does not fit / coder error).
__device__ __forceinline__ i32 SX_K(sx_rc_code_and_assemble)(const SxFrameIdx* idx2, const SxNsqOut* out2, int md, int useDTX, int useMDIndex, int fpp, int hb_bytes,
                                                             const SxCdf* cdf, u8* pw, u8* mine, const u8* hb, int slot_bytes, u8* out, i16* nb_out, SxRcInfo* info_out) {
    SxRcInfo info = {0, 0};
    const bool dtx = useDTX && idx2[fpp - 1].inDTX;
    if (!dtx) sx_code_description(idx2, sx_pub_ld(&out2[0].Seed), sx_pub_ld(&out2[1].Seed), out2[0].q[md], out2[1].q[md], md, useMDIndex, cdf, pw, mine, &info, fpp);
    if (info_out) *info_out = info;
    const i32 nbo = __shfl_xor(info.nBytes, 1), ero = __shfl_xor(info.error, 1);
    const i32 nb0 = md ? nbo : info.nBytes, nb1 = md ? info.nBytes : nbo;
    const i32 total = nb0 + nb1 + hb_bytes;
    if (dtx) {
        if (md) for (int i = 0; i < hb_bytes; i++) out[i] = hb[i];
        else { nb_out[0] = 0; nb_out[1] = 0; }
        return hb_bytes;
    }
    if ((info.error | ero) || total > slot_bytes || nb0 > SX_MAX_ARITHM_BYTES || nb1 > SX_MAX_ARITHM_BYTES) {
        if (!md) { nb_out[0] = 0; nb_out[1] = 0; }
        return -1;
    }
    u8* dst = out + (md ? nb0 : 0);
    for (int i = 0; i < info.nBytes; i++) dst[i] = mine[i];
    if (md) for (int i = 0; i < hb_bytes; i++) out[nb0 + nb1 + i] = hb[i];
    else { nb_out[0] = (i16)total; nb_out[1] = (i16)(nb1 + hb_bytes); }
    return total;
}

#ifndef SX_TU_FRONT
__global__ void __launch_bounds__(64) SX_K(solo_enc_rc_kernel)(const SxEncStream* states, const SxCodeIn* __restrict__ code_in,
                                                               const SxNsqOut* __restrict__ nsq_out, int n_streams, int n_packets, int p0, int pc,
                                                               u8* __restrict__ rcbuf, SxRcInfo* __restrict__ rcinfo, const u8* __restrict__ hbout,
                                                               int slot_bytes, u8* __restrict__ bits, i16* __restrict__ nbytes, i32* status) {
    __shared__ SxRcWork w;
    SX_K(sx_cdf_stage)(&w.cdf);
    const int lane = threadIdx.x, md = lane & 1;
    const int s = blockIdx.x * (SX_RC_LANES / 2) + (lane >> 1);
    if (lane >= SX_RC_LANES || s >= n_streams) return;
    const SxEncState* st = &states[s].core;
    const int useDTX = st->useDTX, useMDIndex = st->useMDIndex, fpp = st->fpp;
    const int hb_bytes = st->hb_joint ? 4 : 4 * fpp;
    i32 first_err = 0;
    for (int p = p0; p < p0 + pc; p++) {
        const size_t pk = (size_t)s * n_packets + p;
        const size_t slot = ((size_t)s * pc + (size_t)(p - p0)) * 2 + (size_t)md;
        const i32 ret = SX_K(sx_rc_code_and_assemble)(code_in[pk].idx, nsq_out + pk * 2, md, useDTX, useMDIndex, fpp, hb_bytes, &w.cdf, (u8*)&w.pw[lane][0],
                                                      rcbuf + slot * SX_RC_BUF_STRIDE, hbout + (slot & ~(size_t)1) * 4, slot_bytes, bits + pk * (size_t)slot_bytes,
                                                      nbytes + pk * 2, &rcinfo[slot]);
        if (ret < 0 && first_err == 0) first_err = ret;
    }
    if (status && md == 0) {               // first error of the call (the chunks of a call run in order)
        if (p0 == 0) status[s] = first_err;
        else if (first_err != 0 && status[s] == 0) status[s] = first_err;
    }
}

// High-band encoder, one wavefront per stream: its bytes (8 per packet) go to hbout; the launch of solo_enc_rc_kernel behind it assembles
// the payloads.  (The high band needs the quantiser's excitation, not the coder's bytes, and its 4096 workgroups are the ones that have
// to find room between the analysis kernel's: they are in line as soon as the quantiser is through, DESIGN_NOTES.md section 10.)
__global__ void __launch_bounds__(64, SX_ANALYSIS_WAVES) SX_K(solo_enc_coding_kernel)(SxEncStream* states, const SxCodeIn* __restrict__ code_in,
                                                                const SxNsqOut* __restrict__ nsq_out, int n_streams, int n_packets, int p0,
                                                                int pc, u8* __restrict__ hbout) {
    __shared__ SxEncWork w;
    const int s = blockIdx.x;
    if (s >= n_streams) return;
    SxEncStream* rec = &states[s];
#if defined(SX_STOPS) && defined(__HIP_DEVICE_COMPILE__)
    SX_STOPS_ENTER(1)
#endif
    SX_K(solo_enc_enter)(&w, rec);
    for (int p = p0; p < p0 + pc; p++) {
        const size_t pk = (size_t)s * n_packets + p;
        const size_t rs = ((size_t)s * pc + (size_t)(p - p0)) * 2;
        sx_enc_stage_c_hb(rec, &w, code_in + pk, nsq_out + pk * 2);
        wv_sync();
        if (SX_LANE < 8) hbout[rs * 4 + SX_LANE] = w.hb_bytes[SX_LANE];
        wv_sync();
    }
}

#endif
// ---------------------------------------------------------------------------------------------------------------------------------------
// The PERSISTENT pipeline's front kernel: one wavefront per stream, ONE launch per call and launch group (mode 0), beside one launch of
// solo_nsq_persist_kernel (solo_nsq_row.hip).  The wavefront analyses its stream's packets one after the other; after each it publishes
// the hand-over records (stored write-through by sx_enc_analyse_frame; drained here; ana_flag[stream] = ticket0 + packets analysed).  The
// quantiser's wavefront of the stream's group of four raises nsq_flag[group] to ticket0 + packets quantised.  Whenever that flag says the
// next uncoded packet is through the quantiser, the wavefront runs that packet's high-band encoder (its 8 bytes wait in LDS), and once
// SX_FRONT_RB packets wait, the range coder: one LANE per (packet, description), tables + work rows in the analysis work area, byte
// buffers in HBM scratch, then the payload assembly by the same lanes.  The coder is a serial chain of ~42 k instructions whatever the
// number of its lanes: batching it over SX_FRONT_RB packets of the SAME stream (instead of 32 streams of one packet: solo_enc_rc_kernel)
// keeps it inside the wavefront that owns the packets, so no third kernel has to find registers and LDS beside 16 front + 4 quantiser
// workgroups per compute unit (there are none left: DESIGN.md section 2).
//
// Progress without assumptions about dispatch order or residency: the check of the quantiser's flag between two packets NEVER waits (a
// packet that is not through yet is coded later), so the analysis of every stream completes whatever else is or is not resident, which
// is all the quantiser's wavefronts wait for.  Only after its last packet does the wavefront wait for the quantiser -- bounded by
// final_wait_ticks of the 100 MHz clock; what is left then (prog[stream] = packets coded) is done by a second launch of this kernel
// (mode 1: no analysis, the flags are final because solo_api.hip issues it behind the quantiser's launch) -- a launch that finds nothing
// to do in the normal case.  (mode 2, tests: the first launch codes nothing at all, the second everything.)
#define SX_FRONT_RB SX_HB_Q
#ifndef SX_FRONT_PRIO
#define SX_FRONT_PRIO SX_ANALYSIS_PRIO
#endif
#ifndef SX_FRONT_POLL_SLEEPS
#define SX_FRONT_POLL_SLEEPS 4       // x 127 x 64 clocks: ~15 us between two looks at the quantiser's flag
#endif
struct SxFrontRcWork {
    SxCdf cdf;
    u32 pw[2 * SX_FRONT_RB][SX_RC_PW_ROW / 4];
};
#if defined(__HIP_DEVICE_COMPILE__)
static_assert(sizeof(SxFrontRcWork) <= sizeof(((SxEncWork*)0)->u), "the in-wave coder's tables + work rows take the place of the analysis work area");
#endif
#if defined(SX_PIPE_TRACE) && defined(__HIPCC__)       // debug builds (tools/debug/pipe_trace.py): per front wavefront {start, analysis done, exit, packets coded before the last analysis ended}
static __device__ unsigned long long g_sx_front_trace[8192][6];
#define SX_FRONT_TRACE(s_, k_, v_) { if ((threadIdx.x & 63u) == 0 && (s_) < 8192) g_sx_front_trace[s_][k_] = (v_); }
#else
#define SX_FRONT_TRACE(s_, k_, v_) {}
#endif
#define SX_FRONT_ERR_HANDOVER (-64)  // status of a stream whose packets never came back from the quantiser (a bounded wait expired)

// A workgroup = ALL the front wavefronts of one compute unit (SX_FRONT_WAVES: sixteen streams at the 16 kHz rate, nine at 32 kHz), each
// with its own work area: 136 KB of the unit's 160 KB of LDS, so that exactly ONE such workgroup fits a unit -- and beside it exactly one
// of the quantiser's (four wavefronts, 24 KB; solo_nsq_row.hip).  The hardware deals a workgroup's wavefronts out over the four SIMDs in
// turn: four front wavefronts of 96 registers per SIMD, which leaves the 128 registers of one quantiser wavefront on each.  With
// single-wavefront workgroups the same plan has no slack at all (16 x 8.5 + 4 x 6 KB = 160 KB; 4 x 96 + 128 = 512 registers) and the
// dispatcher does not balance: some units took 18 front workgroups, their quantiser wavefronts found no room and only started when
// front wavefronts had finished ALL their packets (tools/debug/pipe_trace.py).  The wavefronts of a workgroup never meet at a barrier
// (wv_sync() of this translation unit is wave-local: solo_wave.h).
#ifndef SX_FRONT_WAVES
#define SX_FRONT_WAVES (SX_FS_KHZ == 8 ? 16 : 9)
#endif
// bytes of one wavefront's work area AS THE DEVICE PASS LAYS IT OUT (the host pass of this header sees the emulation's larger unions:
// its sizeof must not size the launch's dynamic LDS)
#define SX_FRONT_WORK_BYTES (SX_FS_KHZ == 8 ? 8640 : 12192)
#if defined(__HIP_DEVICE_COMPILE__)
static_assert(sizeof(SxEncWork) == SX_FRONT_WORK_BYTES, "SX_FRONT_WORK_BYTES = sizeof(SxEncWork) of the device pass");
#endif
#define SX_FRONT_PER_CU (SX_FS_KHZ == 8 ? 16 : 9)       // front wavefronts a compute unit holds beside the quantiser's four (LDS-bound)
#ifdef SX_TU_FRONT
// Registers: 96 of a SIMD's 512, like the analysis kernel (the waves-per-SIMD hint of __launch_bounds__).  The work areas are DYNAMIC
// shared memory on purpose: with 136 KB of static LDS the backend concludes that no more than four wavefronts of this kernel ever share a
// SIMD and pads the kernel's register allocation up to the largest size that still allows four (97 -> 104 registers in the kernel
// descriptor, while the metadata notes and the register allocator both say 96) -- and 4 x 104 leave 96, not 128, for the OTHER kernel's
// wavefront on that SIMD: not one quantiser wavefront was resident beside the front wavefronts until those began to exit
// (tools/debug/pipe_trace.py; tools/kernel_resources.py prints what the descriptor allocates).
__global__ void __launch_bounds__(64 * SX_FRONT_WAVES, SX_ANALYSIS_WAVES) SX_K(solo_enc_front_kernel)(SxEncStream* states, const i16* __restrict__ pcm, int n_streams, int n_packets,
                                                               SxNsqIn* nsq_in, SxCodeIn* code_in, const SxNsqOut* nsq_out, unsigned int* ana_flag,
                                                               const unsigned int* nsq_flag, unsigned int* prog, unsigned int ticket0, int mode,
                                                               unsigned int final_wait_ticks, int slot_bytes, u8* __restrict__ bits,
                                                               i16* __restrict__ nbytes, i32* status, u8* __restrict__ rcbuf, unsigned int* started) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sx_front_lds_[];
    SxEncWork& w = ((SxEncWork*)(void*)sx_front_lds_)[threadIdx.x >> 6];
    SX_IN_LDS(&w);
    const int lane = (int)(threadIdx.x & 63u);
    if (started && threadIdx.x == 0) atomicAdd(started, 1u);     // (the quantiser's launch waits for this count: solo_nsq_row.hip, "Residency")
    if ((int)(blockIdx.x * SX_FRONT_WAVES + (threadIdx.x >> 6)) >= n_streams) return;
    const int s = (int)(blockIdx.x * SX_FRONT_WAVES + (threadIdx.x >> 6));
    __builtin_amdgcn_s_setprio(SX_FRONT_PRIO);
    if (!(mode & 1)) SX_FRONT_TRACE(s, 0, wall_clock64())
#if defined(SX_PIPE_TRACE)
    if (!(mode & 1)) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); SX_FRONT_TRACE(s, 4, (unsigned long long)hw) SX_FRONT_TRACE(s, 5, (unsigned long long)xcc) }
#endif
    SxEncStream* rec = &states[s];
    SX_K(solo_enc_enter)(&w, rec);
    const int useDTX = SX_UNI(w.st.useDTX), useMDIndex = SX_UNI(w.st.useMDIndex), fpp = SX_UNI(w.st.fpp);
    const int hb_bytes = SX_UNI(w.st.hb_joint) ? 4 : 4 * fpp;
    const size_t pk0 = (size_t)s * n_packets;
    const unsigned int* my_nsq_flag = &nsq_flag[s >> 2];
    int hb_done = (mode & 1) ? (int)SX_UNI(prog[s]) : 0, rc_done = hb_done;
    i32 first_err = 0;
    // the range coder over packets [rc_done, hb_done): lane = (packet, description)
    auto rc_batch = [&]() __attribute__((always_inline)) {
        const int cnt = hb_done - rc_done;
        if (cnt <= 0) return;
#ifdef SX_FRONT_NO_RC            // (resource experiments)
        rc_done = hb_done; return;
#endif
        SxFrontRcWork* rw = (SxFrontRcWork*)(void*)&w.u;
        wv_sync();
        SX_K(sx_cdf_stage)(&rw->cdf);
        const int md = lane & 1, j = lane >> 1;
        i32 ret = 0;
        if (j < cnt) {
            const size_t pk = pk0 + (size_t)(rc_done + j);
            ret = SX_K(sx_rc_code_and_assemble)(code_in[pk].idx, nsq_out + pk * 2, md, useDTX, useMDIndex, fpp, hb_bytes, &rw->cdf, (u8*)&rw->pw[lane][0],
                                                rcbuf + ((size_t)s * (2 * SX_FRONT_RB) + (size_t)lane) * SX_RC_BUF_STRIDE, &w.hb_bytes[8 * j], slot_bytes,
                                                bits + pk * (size_t)slot_bytes, nbytes + pk * 2, (SxRcInfo*)nullptr);
        }
        // the call's status: the first packet that failed (all failures report -1)
        if (first_err == 0 && __builtin_amdgcn_ballot_w64(ret < 0) != 0ull) first_err = -1;
        rc_done = hb_done;
        wv_sync();
    };
    auto code_one = [&]() __attribute__((always_inline)) {       // high band of packet hb_done (the quantiser has published it: its outputs are read with sx_pub_ld)
        const size_t pk = pk0 + (size_t)hb_done;
#ifndef SX_FRONT_NO_HB
        sx_enc_stage_c_hb(rec, &w, code_in + pk, nsq_out + pk * 2, hb_done - rc_done);
#endif
        wv_sync();
        hb_done++;
        if (hb_done - rc_done >= SX_FRONT_RB) rc_batch();
    };
    if (!(mode & 1)) {
        for (int p = 0; p < n_packets; p++) {
            const size_t pk = pk0 + (size_t)p;
            sx_enc_stage_a(rec, &w, pcm + pk * (size_t)(SX_FRAME * 2 * fpp), nsq_in + pk * 2, code_in + pk);
            wv_sync();
            sx_pub_drain();
            if (lane == 0) sx_flag_st(&ana_flag[s], ticket0 + (unsigned int)p + 1u);
            // whatever the quantiser has finished meanwhile (at most two packets a turn, so that the analysis never falls far behind)
            for (int it = 0; it < 2 && hb_done <= p && !(mode & 2); it++) {
                const unsigned int f = (unsigned int)SX_UNI(sx_flag_ld(my_nsq_flag));
                if ((int)(f - (ticket0 + (unsigned int)hb_done + 1u)) < 0) break;
                code_one();
            }
        }
        SX_K(solo_enc_leave)(&w, rec);
        wv_sync();
        SX_FRONT_TRACE(s, 1, wall_clock64())
        SX_FRONT_TRACE(s, 3, (unsigned long long)hb_done)
    }
    // the rest: wait for the quantiser, packet by packet (mode 0: bounded; mode 1: the flags are final)
    {
        const unsigned long long t0 = wall_clock64();
        while (hb_done < n_packets && !(mode & 2)) {
            const unsigned int f = (unsigned int)SX_UNI(sx_flag_ld(my_nsq_flag));
            if ((int)(f - (ticket0 + (unsigned int)hb_done + 1u)) >= 0) { code_one(); continue; }
            if (mode & 1) { first_err = first_err ? first_err : SX_FRONT_ERR_HANDOVER; break; }
            if (wall_clock64() - t0 >= (unsigned long long)final_wait_ticks) break;
            // (a slow poll: thousands of wavefronts wait here at the end of a call, all reading the same few cache lines of flags through
            // the fabric -- polled every microsecond they slowed the quantiser they were waiting for by 20 %)
#pragma unroll
            for (int z = 0; z < SX_FRONT_POLL_SLEEPS; z++) __builtin_amdgcn_s_sleep(127);
        }
    }
    rc_batch();
    if (!(mode & 1)) SX_FRONT_TRACE(s, 2, wall_clock64())
    if (lane == 0) {
        prog[s] = (unsigned int)hb_done;
        if (status) {               // first error of the call
            if (!(mode & 1)) status[s] = first_err;
            else if (first_err != 0 && status[s] == 0) status[s] = first_err;
        }
    }
}

#endif
extern "C" int SX_K(solo_launch_nsq)(void* states, const void* in, void* out, int n_streams, int n_packets, int p0, int pc, unsigned int* started,
                                     void* ring, void* hip_stream);   // solo_nsq_row.hip / solo_nsq_row_wb.hip
extern "C" int SX_K(solo_launch_nsq_persist)(void* states, const void* in, void* out, int n_streams, int n_packets, unsigned int* started, void* ring,
                                             const unsigned int* ana_flag, unsigned int* nsq_flag, unsigned int ticket0, unsigned int* err, void* stage,
                                             void* hip_stream);
extern "C" size_t SX_K(solo_nsq_stage_bytes)(int n_streams);
extern "C" int SX_K(solo_nsq_persist_workgroups)(int n_streams);
extern "C" int SX_K(solo_nsq_workgroups)(int n_streams);
extern "C" size_t SX_K(solo_nsq_ring_bytes)(int n_streams);

#include "solo_enc_ops.h"
#ifndef SX_TU_FRONT
static hipError_t SX_K(solo_enc_launch_init)(void* states, int n_streams, int silk_rate_bps, int useMDIndex, int hb_joint, int useDTX, int fpp, hipStream_t s) {
    hipLaunchKernelGGL(SX_K(solo_enc_init_kernel), dim3(n_streams), dim3(64), 0, s, (SxEncStream*)states, n_streams, silk_rate_bps, useMDIndex, hb_joint, useDTX, fpp);
    return hipGetLastError();
}
static hipError_t SX_K(solo_enc_launch_analysis)(void* states, const int16_t* pcm, int n_streams, int n_packets, int p0, int pc, void* nsq_in,
                                                 void* code_in, hipStream_t s) {
    hipLaunchKernelGGL(SX_K(solo_enc_analysis_kernel), dim3(n_streams), dim3(64), 0, s, (SxEncStream*)states, pcm, n_streams, n_packets, p0, pc,
                       (SxNsqIn*)nsq_in, (SxCodeIn*)code_in);
    return hipGetLastError();
}
// rc_scratch of a launch-per-chunk coding stage: [n_streams * pc * 2] byte buffers of SX_RC_BUF_STRIDE, then as many SxRcInfo, then 8
// high-band bytes per packet
static size_t SX_K(solo_enc_rc_scratch_bytes)(int n_streams, int pc) {
    return (size_t)n_streams * (size_t)pc * 2 * (SX_RC_BUF_STRIDE + sizeof(SxRcInfo) + 4) + 128;
}
static SxRcInfo* SX_K(solo_enc_rcinfo_of)(u8* rcbuf, int n_streams, int pc) {
    return (SxRcInfo*)(rcbuf + (((size_t)n_streams * (size_t)pc * 2 * SX_RC_BUF_STRIDE + 63) & ~(size_t)63));
}
static u8* SX_K(solo_enc_hbout_of)(u8* rcbuf, int n_streams, int pc) {
    return (u8*)(SX_K(solo_enc_rcinfo_of)(rcbuf, n_streams, pc) + (size_t)n_streams * (size_t)pc * 2);
}
// third stage of a chunk: high band, then range coder + payload assembly
static hipError_t SX_K(solo_enc_launch_coding)(void* states, const void* code_in, const void* nsq_out, int n_streams, int n_packets, int p0, int pc,
                                               int slot, uint8_t* bits, int16_t* nbytes, int32_t* status, void* rc_scratch, hipStream_t s) {
    u8* rcbuf = (u8*)rc_scratch;
    hipLaunchKernelGGL(SX_K(solo_enc_coding_kernel), dim3(n_streams), dim3(64), 0, s, (SxEncStream*)states, (const SxCodeIn*)code_in,
                       (const SxNsqOut*)nsq_out, n_streams, n_packets, p0, pc, SX_K(solo_enc_hbout_of)(rcbuf, n_streams, pc));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(SX_K(solo_enc_rc_kernel), dim3((n_streams + SX_RC_LANES / 2 - 1) / (SX_RC_LANES / 2)), dim3(64), 0, s, (const SxEncStream*)states, (const SxCodeIn*)code_in,
                       (const SxNsqOut*)nsq_out, n_streams, n_packets, p0, pc, rcbuf, SX_K(solo_enc_rcinfo_of)(rcbuf, n_streams, pc),
                       (const u8*)SX_K(solo_enc_hbout_of)(rcbuf, n_streams, pc), slot, bits, nbytes, status);
    return hipGetLastError();
}
#endif
#ifdef SX_TU_FRONT
// persistent pipeline: scratch of the in-wave coder (2 SX_FRONT_RB byte buffers per stream) and the front kernel's launch
extern "C" size_t SX_K(solo_enc_front_scratch_bytes)(int n_streams) { return (size_t)n_streams * (2 * SX_FRONT_RB) * SX_RC_BUF_STRIDE + 128; }
extern "C" hipError_t SX_K(solo_enc_launch_front)(void* states, const int16_t* pcm, int n_streams, int n_packets, void* nsq_in, void* code_in, const void* nsq_out,
                                              unsigned int* ana_flag, const unsigned int* nsq_flag, unsigned int* prog, unsigned int ticket0, int mode,
                                              unsigned int final_wait_ticks, int slot, uint8_t* bits, int16_t* nbytes, int32_t* status, void* scratch,
                                              unsigned int* started, hipStream_t s) {
    static bool lds_set = false;           // (more than 64 KB of dynamic LDS has to be asked for once)
    if (!lds_set) {
        const hipError_t e = hipFuncSetAttribute((const void*)SX_K(solo_enc_front_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SX_FRONT_WAVES * SX_FRONT_WORK_BYTES));
        if (e != hipSuccess) return e;
        lds_set = true;
    }
    hipLaunchKernelGGL(SX_K(solo_enc_front_kernel), dim3((n_streams + SX_FRONT_WAVES - 1) / SX_FRONT_WAVES), dim3(64 * SX_FRONT_WAVES), SX_FRONT_WAVES * SX_FRONT_WORK_BYTES, s, (SxEncStream*)states, pcm, n_streams, n_packets, (SxNsqIn*)nsq_in,
                       (SxCodeIn*)code_in, (const SxNsqOut*)nsq_out, ana_flag, nsq_flag, prog, ticket0, mode, final_wait_ticks, slot, bits, nbytes, status,
                       (u8*)scratch, started);
    return hipGetLastError();
}
#else
extern "C" size_t SX_K(solo_enc_front_scratch_bytes)(int n_streams);                                            // solo_enc_front_k.hip
extern "C" hipError_t SX_K(solo_enc_launch_front)(void* states, const int16_t* pcm, int n_streams, int n_packets, void* nsq_in, void* code_in, const void* nsq_out,
                                                  unsigned int* ana_flag, const unsigned int* nsq_flag, unsigned int* prog, unsigned int ticket0, int mode,
                                                  unsigned int final_wait_ticks, int slot, uint8_t* bits, int16_t* nbytes, int32_t* status, void* scratch,
                                                  unsigned int* started, hipStream_t s);
static const solo_enc_ops SX_K(solo_enc_ops_table) = {
    sizeof(SxEncStream), sizeof(SxNsqIn), sizeof(SxNsqOut), sizeof(SxCodeIn), SX_PACKET,
    SX_K(solo_enc_launch_init), SX_K(solo_enc_launch_analysis), SX_K(solo_launch_nsq), SX_K(solo_enc_launch_coding), SX_K(solo_enc_rc_scratch_bytes),
    SX_K(solo_nsq_workgroups), SX_K(solo_nsq_ring_bytes), SX_K(solo_enc_launch_front), SX_K(solo_launch_nsq_persist), SX_K(solo_enc_front_scratch_bytes), SX_K(solo_nsq_persist_workgroups), SX_K(solo_nsq_stage_bytes),
    SX_FRONT_WAVES, SX_FRONT_PER_CU};

#endif