// solo_nsq16.hip -- stage B of the encoder: the multiple-description delayed-decision quantiser (solo_enc_nsq.h), compiled
// with FOUR streams per wavefront.  The quantiser's per-sample work occupies 12 lanes (3 tracks x 4 survivor states), so a
// 64-lane wavefront carries four independent streams in its four 16-lane rows; everything that is "wave-uniform" in the
// one-stream-per-wave model is uniform within a row here, and shuffles stay inside a row.  The recursion is serial in time,
// so one row works through its stream's frames in order: packet by packet, two frames each.
#define SX_GROUP16 1
#include <hip/hip_runtime.h>
#include "solo_enc_nsq.h"

extern "C" __global__ void __launch_bounds__(64) solo_nsq_kernel(SxEncStream* states, const SxNsqIn* __restrict__ in,
                                                                 SxNsqOut* __restrict__ out, int n_streams, int n_packets) {
    __shared__ SxNsqWork w[4];
    const int g = threadIdx.x >> 4;
    const int s = blockIdx.x * 4 + g;
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
    hipLaunchKernelGGL(solo_nsq_kernel, dim3((n_streams + 3) / 4), dim3(64), 0, (hipStream_t)hip_stream, (SxEncStream*)states,
                       (const SxNsqIn*)in, (SxNsqOut*)out, n_streams, n_packets);
    return (int)hipGetLastError();
}
