// solo_nsq16.hip -- stage B of the encoder: the multiple-description delayed-decision quantiser (solo_enc_nsq.h), compiled
// with SIXTEEN streams per wavefront: one lane = one delayed-decision state of one stream (all three tracks in its registers),
// a stream = one DPP quad.  Everything that is "wave-uniform" in the one-stream-per-wave model is uniform within a quad here, and
// the cross-lane exchanges stay inside the quad.  The recursion is serial in time, so one quad works through its stream's frames in
// order: packet by packet, two frames each.  4096 streams = 256 wavefronts = one per CU: the quantiser leaves three SIMDs of every
// CU (and most issue slots of the fourth) to the analysis / coding kernels of the neighbouring chunks of the pipeline.
#ifndef SX_NSQ_GROUP
#define SX_NSQ_GROUP 4        // lanes per stream: 4 -> sixteen streams per wavefront
#endif
#define SX_GROUP SX_NSQ_GROUP
#define SX_PER_WAVE (64 / SX_GROUP)
#include <hip/hip_runtime.h>
#include "solo_enc_nsq.h"

#ifndef SX_EXP_NSQ_FRAMES
#define SX_EXP_NSQ_FRAMES 2          // (timing experiments: 1 = only the first frame of a packet is quantised -- wrong output)
#endif
#ifndef SX_NSQ_PRIO
#define SX_NSQ_PRIO 3
#endif
#ifndef SX_NSQ_WAVES
#define SX_NSQ_WAVES 1
#endif
// ring: SX_NSQ_RING_CELLS(64) cells per workgroup (the emission ring of its sixteen streams, rows of 64 lanes = 1 KB)
// SX_NSQ_VGPR_CAP = n: the kernel may use 2 n of the SIMD's 512 registers (n accumulation registers as spill space on top of the
// 256 architectural ones).  176 -> 352 registers: the sample loop still has no scratch access (it has at 160), and 160 registers
// of the quantiser's SIMD are left for one wave of the analysis kernel (96), see solo_enc_kernels.h.
#ifndef SX_NSQ_VGPR_CAP
#define SX_NSQ_VGPR_CAP 176
#endif
#ifdef SX_NSQ_VGPR_CAP
#define SX_NSQ_CAP_ATTR __attribute__((amdgpu_num_vgpr(SX_NSQ_VGPR_CAP)))
#else
#define SX_NSQ_CAP_ATTR
#endif
extern "C" __global__ void SX_NSQ_CAP_ATTR __launch_bounds__(64, SX_NSQ_WAVES) SX_K(solo_nsq_kernel)(SxEncStream* states, const SxNsqIn* __restrict__ in,
                                                                 SxNsqOut* __restrict__ out, int n_streams, int n_packets, int p0, int pc,
                                                                 unsigned int* started, SxNsqCell* __restrict__ ring) {
    __shared__ SxNsqWork w[SX_PER_WAVE];
    const int g = threadIdx.x / SX_GROUP;
    const int s = blockIdx.x * SX_PER_WAVE + g;
    if (started && threadIdx.x == 0) atomicAdd(started, 1u);     // lets the host-side pipeline start the next analysis chunk once this kernel is resident
    if (s >= n_streams) return;
    // a latency-bound wave that shares its SIMD with the analysis / coding kernels of neighbouring chunks: issue first
    __builtin_amdgcn_s_setprio(SX_NSQ_PRIO);
    // wave-uniform bases + 32-bit lane offsets (solo_enc_nsq.h): the states / records of the wavefront's sixteen streams
    char* Pu = (char*)&states[(size_t)blockIdx.x * SX_PER_WAVE];
    const u32 pOff = (u32)g * (u32)sizeof(SxEncStream) + (u32)offsetof(SxEncStream, nsq);
    const u32 rec_stride = (u32)n_packets * 2u;                     // hand-over records between consecutive streams
    SxNsqCell* rgu = ring + (size_t)blockIdx.x * SX_NSQ_RING_CELLS(64);
    for (int p = p0; p < p0 + pc; p++) {          // packets [p0, p0 + pc) of a launch of n_packets (row stride of the records)
        for (int f = 0; f < SX_EXP_NSQ_FRAMES; f++) {
            const size_t r0 = ((size_t)blockIdx.x * SX_PER_WAVE * n_packets + p) * 2 + f;       // record of the wavefront's first stream
            sx_nsq_del_dec(Pu, pOff, &in[r0 + (size_t)g * rec_stride], (char*)&out[r0], (u32)g * rec_stride * (u32)sizeof(SxNsqOut), &w[g], rgu,
                           (u32)(g * SX_GROUP), 64);
            wv_sync();
        }
    }
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

#endif
extern "C" int SX_K(solo_nsq_workgroups)(int n_streams) { return (n_streams + SX_PER_WAVE - 1) / SX_PER_WAVE; }
// host-side launcher (called from solo_api.hip); ring: SX_K(solo_nsq_ring_bytes)(n_streams) bytes of device memory (scratch of a launch)
extern "C" size_t SX_K(solo_nsq_ring_bytes)(int n_streams) {
    return (size_t)((n_streams + SX_PER_WAVE - 1) / SX_PER_WAVE) * SX_NSQ_RING_CELLS(64) * sizeof(SxNsqCell);
}
extern "C" int SX_K(solo_launch_nsq)(void* states, const void* in, void* out, int n_streams, int n_packets, int p0, int pc, unsigned int* started,
                               void* ring, void* hip_stream) {
    hipLaunchKernelGGL(SX_K(solo_nsq_kernel), dim3((n_streams + SX_PER_WAVE - 1) / SX_PER_WAVE), dim3(64), 0, (hipStream_t)hip_stream, (SxEncStream*)states,
                       (const SxNsqIn*)in, (SxNsqOut*)out, n_streams, n_packets, p0, pc, started, (SxNsqCell*)ring);
    return (int)hipGetLastError();
}

#if defined(SX_PROF) && SX_FS_KHZ == 8
extern "C" int32_t solo_debug_prof_nsq(unsigned long long* out32, int32_t reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_sx_prof), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sx_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
