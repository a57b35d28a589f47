// solo_nsq_row.hip -- stage B of the encoder: the multiple-description delayed-decision quantiser (solo_enc_nsq_row.h), compiled
// with FOUR streams per wavefront: one lane = one delayed-decision state of one track, a stream = one 16-lane DPP row (twelve live
// lanes: 3 tracks x 4 states).  Everything that is "wave-uniform" in the one-stream-per-wave model is uniform within a row here, and
// the cross-lane exchanges stay inside the row.  The recursion is serial in time, so one row works through its stream's frames in
// order: packet by packet, two frames each.  4096 streams = 1024 wavefronts = one per SIMD: the quantiser's wave is a dependent
// chain that issues every ~5 cycles; the analysis / coding kernels of the neighbouring chunks of the pipeline fill the rest of
// every SIMD's issue slots.
#define SX_GROUP 16
#define SX_PER_WAVE (64 / SX_GROUP)
// wavefronts per workgroup of the persistent kernel (4: one workgroup per compute unit, one wavefront per SIMD; > 1: wv_sync() of this translation unit must not contain a workgroup
// barrier -- the wavefronts' control flow differs, each follows its own streams -- see solo_wave.h)
#ifndef SX_NSQ_PERSIST_WAVES
#define SX_NSQ_PERSIST_WAVES 4
#endif
#if SX_NSQ_PERSIST_WAVES > 1
#define SX_SYNC_WAVE_ONLY 1
#endif
#include <hip/hip_runtime.h>
#include "solo_enc_nsq_row.h"

#ifndef SX_NSQ_PRIO
#define SX_NSQ_PRIO 3
#endif
// SX_NSQ_VGPR_CAP = n: the kernel may use 2 n of the SIMD's 512 registers.  64 -> 128 registers: a SIMD that holds a quantiser
// wave still takes four analysis waves of 96 (128 + 4 x 96 = 512), so all sixteen analysis workgroups of a compute unit stay
// resident beside its four quantiser waves.
#ifndef SX_NSQ_VGPR_CAP
#if SX_FS_KHZ == 16
#define SX_NSQ_VGPR_CAP 80           // (32 kHz build: order-16 prediction spills inside the sample loop at 128 registers -- 160: encode 72.0 -> 67.0 ms
                                     // per 4096 x 25 packets; its analysis workgroups are LDS-bound to nine per compute unit, the registers are free)
#else
#define SX_NSQ_VGPR_CAP 64
#endif
#endif
#if SX_NSQ_VGPR_CAP > 0
#define SX_NSQ_CAP_ATTR __attribute__((amdgpu_num_vgpr(SX_NSQ_VGPR_CAP)))
#else
#define SX_NSQ_CAP_ATTR
#endif
// ring: SX_DD_DELAY rows of 64 cells per workgroup and track-per-lane (the emission ring of its streams, rows of 64 lanes = 1 KB)
#define SX_NSQ_RING_CELLS (SX_DD_DELAY * 64)
extern "C" __global__ void SX_NSQ_CAP_ATTR __launch_bounds__(64) SX_K(solo_nsq_kernel)(SxEncStream* states, const SxNsqIn* __restrict__ in,
                                                                 SxNsqOut* __restrict__ out, int n_streams, int n_packets, int p0, int pc,
                                                                 unsigned int* started, SxRowCell* __restrict__ ring) {
    __shared__ SxRowWork w[SX_PER_WAVE];
    const int g = threadIdx.x / SX_GROUP;
    const int s = blockIdx.x * SX_PER_WAVE + g;
    if (started && threadIdx.x == 0) atomicAdd(started, 1u);     // lets the host-side pipeline start the next analysis chunk once this kernel is resident
    if (s >= n_streams) return;
    // a latency-bound wave that shares its SIMD with the analysis / coding kernels of neighbouring chunks: issue first
    __builtin_amdgcn_s_setprio(SX_NSQ_PRIO);
    // wave-uniform bases + 32-bit lane offsets (solo_enc_nsq_row.h): the states / records of the wavefront's four streams
    char* Pu = (char*)&states[(size_t)blockIdx.x * SX_PER_WAVE];
    const u32 pOff = (u32)g * (u32)sizeof(SxEncStream) + (u32)offsetof(SxEncStream, nsq);
    const u32 rec_stride = (u32)n_packets * 2u;                     // hand-over records between consecutive streams
    SxRowCell* rgu = ring + (size_t)blockIdx.x * SX_NSQ_RING_CELLS;
    const int fpp = __builtin_amdgcn_readfirstlane(states[(size_t)blockIdx.x * SX_PER_WAVE].core.fpp);      // frames per packet: the same for every stream of a handle
    for (int p = p0; p < p0 + pc; p++) {          // packets [p0, p0 + pc) of a launch of n_packets (row stride of the records)
        for (int f = 0; f < fpp; f++) {
            const size_t r0 = ((size_t)blockIdx.x * SX_PER_WAVE * n_packets + p) * 2 + f;       // record of the wavefront's first stream
            sx_nsq_del_dec(Pu, pOff, &in[r0 + (size_t)g * rec_stride], (char*)&out[r0], (u32)g * rec_stride * (u32)sizeof(SxNsqOut), &w[g], rgu,
                           (u32)(g * SX_GROUP), 64);
            wv_sync();
        }
    }
}

// The quantiser of the PERSISTENT pipeline (solo_api.hip, solo_enc_kernels.h: solo_enc_front_kernel): one launch per call and launch group.
// A wavefront works through ALL packets of its four streams; before packet p it waits until the analysis of packet p of each of its
// streams is published (ana_flag[stream] has reached ticket0 + p + 1: tickets count packets over the handle's lifetime, so the flag
// words never need a reset between calls), after it -- outputs stored write-through, stores drained -- it raises its own flag to the
// same ticket, which the front kernel's waves of those streams poll before they code the packet.  Waits are bounded (~2 s of the
// 100 MHz clock): a wavefront that gives up sets err[0], raises nothing more and leaves; the front kernel then reports the streams as
// failed instead of hanging the device.
#if defined(SX_PIPE_TRACE)       // debug builds (tools/debug/pipe_trace.py): per quantiser wavefront {start, first packet done, exit, ticks spent waiting for flags}
static __device__ unsigned long long g_sx_nsq_trace[2048][6];
#define SX_NSQ_TRACE(k_, v_) { if (lane == 0 && wave < 2048) g_sx_nsq_trace[wave][k_] = (v_); }
#else
#define SX_NSQ_TRACE(k_, v_) {}
#endif
#ifndef SX_NSQ_POLL_SLEEP
#define SX_NSQ_POLL_SLEEP 100        // x 64 clocks: ~3 us between two looks at the streams' flags
#endif
#ifndef SX_NSQ_WAIT_TICKS
#define SX_NSQ_WAIT_TICKS 200000000ull
#endif
// Residency: the front kernel is launched FIRST -- one workgroup per compute unit, 136 of its 160 KB of LDS, four wavefronts of 96 registers
// on every SIMD (solo_enc_kernels.h) -- and this kernel once those workgroups have started (solo_api.hip gates the launch on their
// count).  What a unit has left then is 24 KB of LDS and 128 registers per SIMD: ONE workgroup of this kernel, one wavefront per SIMD.
extern "C" __global__ void SX_NSQ_CAP_ATTR __launch_bounds__(64 * SX_NSQ_PERSIST_WAVES) SX_K(solo_nsq_persist_kernel)(SxEncStream* states, const SxNsqIn* in, SxNsqOut* out,
                                                                         int n_streams, int n_packets, unsigned int* started, SxRowCell* __restrict__ ring,
                                                                         const unsigned int* ana_flag, unsigned int* nsq_flag, unsigned int ticket0,
                                                                         unsigned int* err, SxNsqIn* stage) {
    __shared__ SxRowWork w[SX_PER_WAVE * SX_NSQ_PERSIST_WAVES];
    const int wave = (int)(blockIdx.x * SX_NSQ_PERSIST_WAVES + threadIdx.x / 64);     // the wavefront's number in the launch = its flag word
    const int lane = (int)(threadIdx.x & 63u);
    const int g = lane / SX_GROUP;
    const int s = wave * SX_PER_WAVE + g;
    __builtin_amdgcn_s_setprio(SX_NSQ_PRIO);
    SX_NSQ_TRACE(0, wall_clock64())
    if (started && threadIdx.x == 0) atomicAdd(started, 1u);
    if (s >= n_streams) return;
    char* Pu = (char*)&states[(size_t)wave * SX_PER_WAVE];
    const u32 pOff = (u32)g * (u32)sizeof(SxEncStream) + (u32)offsetof(SxEncStream, nsq);
    const u32 rec_stride = (u32)n_packets * 2u;
    SxRowCell* rgu = ring + (size_t)wave * SX_NSQ_RING_CELLS;
    SxRowWork* wg = &w[threadIdx.x / SX_GROUP];
    const int fpp = __builtin_amdgcn_readfirstlane(states[(size_t)wave * SX_PER_WAVE].core.fpp);
#if defined(SX_PIPE_TRACE)
    { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); SX_NSQ_TRACE(4, (unsigned long long)hw) SX_NSQ_TRACE(5, (unsigned long long)xcc) }
#endif
    unsigned long long waited_ = 0;
    for (int p = 0; p < n_packets; p++) {
        const unsigned int target = ticket0 + (unsigned int)p + 1u;
        {   // every lane of a row polls its stream's flag (one word per row: the loads coalesce)
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                const bool ok = (int)(sx_flag_ld(&ana_flag[s]) - target) >= 0;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                if (wall_clock64() - t0 > SX_NSQ_WAIT_TICKS) { if (lane == 0) atomicOr(err, 1u); return; }
                __builtin_amdgcn_s_sleep(SX_NSQ_POLL_SLEEP);
            }
            waited_ += wall_clock64() - t0;
        }
        // the packet's hand-over records into the wavefront's own staging records, read with sc1 loads (solo_wave.h: the unit's L1 may hold
        // stale lines of them and is NOT invalidated); the quantiser proper then reads the copies with plain loads
        static_assert(sizeof(SxNsqIn) % 4 == 0, "dword copy of the hand-over records");
        SxNsqIn* mine = stage + ((size_t)wave * SX_PER_WAVE + g) * 2;
        for (int f = 0; f < fpp; f++) {
            const u32* src = (const u32*)&in[((size_t)wave * SX_PER_WAVE * n_packets + p) * 2 + f + (size_t)g * rec_stride];
            u32* dst = (u32*)&mine[f];
            constexpr int NW = (int)(sizeof(SxNsqIn) / 4), NU = (NW + SX_GROUP - 1) / SX_GROUP;
            u32 v[NU];
#pragma unroll
            for (int u = 0; u < NU; u++) { const int i = u * SX_GROUP + SX_LANE; v[u] = i < NW ? sx_pub_ld(&src[i]) : 0u; }
#pragma unroll
            for (int u = 0; u < NU; u++) { const int i = u * SX_GROUP + SX_LANE; if (i < NW) dst[i] = v[u]; }
        }
        wv_sync();
        for (int f = 0; f < fpp; f++) {
            const size_t r0 = ((size_t)wave * SX_PER_WAVE * n_packets + p) * 2 + f;
            sx_nsq_del_dec(Pu, pOff, &mine[f], (char*)&out[r0], (u32)g * rec_stride * (u32)sizeof(SxNsqOut), wg, rgu, (u32)(g * SX_GROUP), 64);
            wv_sync();
        }
        // Publish the packet's output records.  The sample loop stored them with plain stores: a write-through store there would sit in
        // front of the ring-cell loads of the following samples (vector memory operations complete in order), and every sample would wait
        // for a trip to memory.  So each row copies its stream's records once more, write-through, in one burst (the wavefront reads its own
        // stores back: wv_sync above has drained them), drains, and only then raises the flag.
        static_assert(sizeof(SxNsqOut) % 4 == 0, "dword copy of the output records");
        for (int f = 0; f < fpp; f++) {
            u32* rec = (u32*)&out[((size_t)wave * SX_PER_WAVE * n_packets + p) * 2 + f + (size_t)g * rec_stride];
            u32 v[(sizeof(SxNsqOut) / 4 + SX_GROUP - 1) / SX_GROUP];
#pragma unroll
            for (int u = 0; u < (int)(sizeof(v) / 4); u++) { const int i = u * SX_GROUP + SX_LANE; v[u] = i < (int)(sizeof(SxNsqOut) / 4) ? rec[i] : 0u; }
#pragma unroll
            for (int u = 0; u < (int)(sizeof(v) / 4); u++) { const int i = u * SX_GROUP + SX_LANE; if (i < (int)(sizeof(SxNsqOut) / 4)) sx_pub_st(&rec[i], v[u]); }
        }
        sx_pub_drain();
        if (lane == 0) sx_flag_st(&nsq_flag[wave], target);
        if (p == 0) SX_NSQ_TRACE(1, wall_clock64())
    }
    SX_NSQ_TRACE(2, wall_clock64())
    SX_NSQ_TRACE(3, waited_)
}

#if SX_FS_KHZ == 8
// gate: holds a stream until `*flag` has reached `target` (modulo 2^32), i.e. until all workgroups of the quantiser launch that
// counts into it are resident; gives up after ~20 ms so that a runtime that serialises the streams cannot hang
extern "C" __global__ void __launch_bounds__(64) solo_gate_kernel(const unsigned int* flag, unsigned int target) {
    if (threadIdx.x == 0) {
        for (int it = 0; it < 20000; it++) {
            if (__atomic_load_n(flag, __ATOMIC_RELAXED) - target < 0x80000000u) break;
            __builtin_amdgcn_s_sleep(32);
        }
    }
}
extern "C" int solo_launch_gate(const unsigned int* flag, unsigned int target, void* hip_stream) {
    hipLaunchKernelGGL(solo_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, flag, target);
    return (int)hipGetLastError();
}

// Unit check of the row exchanges (tests/test_gpu_nsq_row.py): out[0..7][lane] = the primitives applied to in[lane]
extern "C" __global__ void __launch_bounds__(64) solo_debug_rowops_kernel(const i32* in, const i32* idx, i32* out) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = threadIdx.x;
    i32 v[1] = {in[lane]}, ix[1] = {idx[lane] & 3}, d[1] = {-1}, mv[1], mi[1];
    RWT_FROM(d, v, 0) out[0 * 64 + lane] = d[0];
    RWT_FROM(d, v, 1) out[1 * 64 + lane] = d[0];
    RWT_FROM(d, v, 2) out[2 * 64 + lane] = d[0];
    d[0] = -1; RWT_TO0(d, v, 1) out[3 * 64 + lane] = d[0];
    d[0] = -1; RWT_TO0(d, v, 2) out[4 * 64 + lane] = d[0];
    RWT_SUM(d, v) out[5 * 64 + lane] = d[0];
    RWT_OR(d, v) out[6 * 64 + lane] = d[0];
    RWK_GATHER(d, v, ix) out[7 * 64 + lane] = d[0];
    RWK_ARGMIN(v, mv, mi) out[8 * 64 + lane] = mv[0]; out[9 * 64 + lane] = mi[0];
    RWK_ARGMAX(v, mv, mi) out[10 * 64 + lane] = mv[0]; out[11 * 64 + lane] = mi[0];
    RWS_SUM(v, d) out[12 * 64 + lane] = d[0];
    out[13 * 64 + lane] = rwk_from(v[0], ix[0]);
#endif
}
// Effective shader clock under vector load: every SIMD of the chip gets two waves that spin ~`iters` x 64 dependent vector
// instructions; wave 0 reports how far the shader-clock counter (s_memtime) and the constant 100 MHz counter advanced meanwhile.
extern "C" __global__ void __launch_bounds__(64) solo_debug_clock_kernel(unsigned long long* out2, int iters, int* sink) {
    int a = threadIdx.x, b = blockIdx.x | 1;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int r = 0; r < iters; r++) {
#pragma unroll
        for (int u = 0; u < 64; u++) a = a * b + u;
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (a == 0x7F123457) *sink = a;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out2[0] = c1 - c0; out2[1] = w1 - w0; }
}
extern "C" int32_t solo_debug_clock(double* mhz_out) {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    unsigned long long* d = NULL;
    if (hipMalloc((void**)&d, 32) != hipSuccess) return -1;
    unsigned long long h[2] = {0, 0};
    hipLaunchKernelGGL(solo_debug_clock_kernel, dim3(ncu * 8), dim3(64), 0, 0, d, 4000, (int*)(d + 2));
    const hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess || h[1] == 0) return -1;
    *mhz_out = 100.0 * (double)h[0] / (double)h[1];
    return 0;
}
extern "C" int solo_debug_rowops(const int32_t* d_in, const int32_t* d_idx, int32_t* d_out, void* hip_stream) {
    hipLaunchKernelGGL(solo_debug_rowops_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, d_in, d_idx, d_out);
    return (int)hipGetLastError();
}
#endif
extern "C" int SX_K(solo_launch_nsq_persist)(void* states, const void* in, void* out, int n_streams, int n_packets, unsigned int* started, void* ring,
                                              const unsigned int* ana_flag, unsigned int* nsq_flag, unsigned int ticket0, unsigned int* err, void* stage,
                                              void* hip_stream) {
    const int per_wg = SX_PER_WAVE * SX_NSQ_PERSIST_WAVES;
    hipLaunchKernelGGL(SX_K(solo_nsq_persist_kernel), dim3((n_streams + per_wg - 1) / per_wg), dim3(64 * SX_NSQ_PERSIST_WAVES), 0, (hipStream_t)hip_stream, (SxEncStream*)states,
                       (const SxNsqIn*)in, (SxNsqOut*)out, n_streams, n_packets, started, (SxRowCell*)ring, ana_flag, nsq_flag, ticket0, err, (SxNsqIn*)stage);
    return (int)hipGetLastError();
}
// the persistent kernel's staging records: two SxNsqIn per stream
extern "C" int SX_K(solo_nsq_persist_workgroups)(int n_streams) { return (n_streams + SX_PER_WAVE * SX_NSQ_PERSIST_WAVES - 1) / (SX_PER_WAVE * SX_NSQ_PERSIST_WAVES); }
extern "C" size_t SX_K(solo_nsq_stage_bytes)(int n_streams) { return (size_t)((n_streams + SX_PER_WAVE - 1) / SX_PER_WAVE) * SX_PER_WAVE * 2 * sizeof(SxNsqIn); }
#if defined(SX_PIPE_TRACE) && SX_FS_KHZ == 8
extern "C" int32_t solo_debug_nsq_trace(unsigned long long* out, int32_t n_waves) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sx_nsq_trace), (size_t)n_waves * 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int SX_K(solo_nsq_workgroups)(int n_streams) { return (n_streams + SX_PER_WAVE - 1) / SX_PER_WAVE; }
// host-side launcher (called from solo_api.hip); ring: SX_K(solo_nsq_ring_bytes)(n_streams) bytes of device memory (scratch of a launch)
extern "C" size_t SX_K(solo_nsq_ring_bytes)(int n_streams) {
    return (size_t)((n_streams + SX_PER_WAVE - 1) / SX_PER_WAVE) * SX_NSQ_RING_CELLS * sizeof(SxRowCell);
}
extern "C" int SX_K(solo_launch_nsq)(void* states, const void* in, void* out, int n_streams, int n_packets, int p0, int pc, unsigned int* started,
                               void* ring, void* hip_stream) {
    hipLaunchKernelGGL(SX_K(solo_nsq_kernel), dim3((n_streams + SX_PER_WAVE - 1) / SX_PER_WAVE), dim3(64), 0, (hipStream_t)hip_stream, (SxEncStream*)states,
                       (const SxNsqIn*)in, (SxNsqOut*)out, n_streams, n_packets, p0, pc, started, (SxRowCell*)ring);
    return (int)hipGetLastError();
}
