// solo_nsq16.hip -- stage B of the encoder: the multiple-description delayed-decision quantiser (solo_enc_nsq.h), compiled
// with FOUR streams per wavefront.  The quantiser's per-sample work occupies 12 lanes (3 tracks x 4 survivor states), so a
// 64-lane wavefront carries four independent streams in its four 16-lane rows; everything that is "wave-uniform" in the
// one-stream-per-wave model is uniform within a row here, and shuffles stay inside a row.  The recursion is serial in time,
// so one row works through its stream's frames in order: packet by packet, two frames each.
#ifndef SX_NSQ_GROUP
#define SX_NSQ_GROUP 16       // lanes per stream: 16 -> four streams per wavefront
#endif
#define SX_GROUP SX_NSQ_GROUP
#define SX_PER_WAVE (64 / SX_GROUP)
#include <hip/hip_runtime.h>
#include "solo_enc_nsq.h"

#ifndef SX_NSQ_WAVES
#define SX_NSQ_WAVES 1
#endif
extern "C" __global__ void __launch_bounds__(64, SX_NSQ_WAVES) solo_nsq_kernel(SxEncStream* states, const SxNsqIn* __restrict__ in,
                                                                 SxNsqOut* __restrict__ out, int n_streams, int n_packets) {
    __shared__ SxNsqWork w[SX_PER_WAVE];
    const int g = threadIdx.x / SX_GROUP;
    const int s = blockIdx.x * SX_PER_WAVE + g;
    if (s >= n_streams) return;
    SxNsqPersist* P = &states[s].nsq;
    for (int p = 0; p < n_packets; p++) {
        for (int f = 0; f < 2; f++) {
            const size_t r = ((size_t)s * n_packets + p) * 2 + f;
            sx_nsq_del_dec(P, &in[r], &out[r], &w[g]);
            wv_sync();
        }
    }
}

// host-side launcher (called from solo_api.hip)
extern "C" int solo_launch_nsq(void* states, const void* in, void* out, int n_streams, int n_packets, void* hip_stream) {
    hipLaunchKernelGGL(solo_nsq_kernel, dim3((n_streams + SX_PER_WAVE - 1) / SX_PER_WAVE), dim3(64), 0, (hipStream_t)hip_stream, (SxEncStream*)states,
                       (const SxNsqIn*)in, (SxNsqOut*)out, n_streams, n_packets);
    return (int)hipGetLastError();
}

#if defined(SX_PROF)
extern "C" int32_t solo_debug_prof_nsq(unsigned long long* out32, int32_t reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_sx_prof), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sx_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
