// micro-benchmark: do the latency-bound phases of one wave hide the VALU-dense phases of the other waves of its SIMD?
// Every wave alternates  D: 256 dependent v_mul_hi / v_add (a chain that keeps the vector unit ~70 % busy by itself)  and
// L: 16 dependent LDS reads (each waits for the previous one: ~2000 cycles with the vector unit idle).
//   hipcc --offload-arch=gfx950 -O2 -o build/mb_overlap tools/debug/mb_overlap.hip ; gpurun -- ./build/mb_overlap
// Perfect overlap: time(W waves per SIMD) = max(time(1), W x vector-unit time).  No overlap of a wave's waits with the others' work:
// time(1) + (W - 1) x vector-unit time -- the law the analysis kernel follows.
// Variant "skewed": wave w starts with (w mod 8) / 8 of a period of s_sleep, so that the waves of a SIMD are out of phase.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
template <int MODE, int PRIO = 0> __global__ void __launch_bounds__(64) kern(int* out, int iters, int skew_sleeps) {
    __shared__ int lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 37 + 11) & 1023;
    int v0 = threadIdx.x * 2654435 + 12345, v1 = v0 * 3 + 7, idx = threadIdx.x;
    if (skew_sleeps) for (int i = 0; i < (int)(blockIdx.x >> 8 & 7) * skew_sleeps; i++) __builtin_amdgcn_s_sleep(127);
    __syncthreads();
    for (int it = 0; it < iters; it++) {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);          // dense phase: lowest priority
        if (PRIO == 2) __builtin_amdgcn_s_setprio(3);          // (control: the other way round)
        if (MODE != 2) asm volatile(R64("v_mul_hi_i32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_mul_hi_i32 %0, %0, %1\n v_add_u32 %0, 0x12345, %0\n") : "+v"(v0) : "v"(v1));
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);          // latency phase: its few instructions go first
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        if (MODE != 1) {
#pragma unroll
            for (int j = 0; j < 16; j++) { idx = lds[idx & 1023]; asm volatile("" : "+v"(idx)); }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = v0 + idx;
}
template <int MODE, int PRIO = 0> float run(int* d_out, int nwg, int iters, int skew) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kern<MODE, PRIO>), dim3(nwg), dim3(64), 0, 0, d_out, 2, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((kern<MODE, PRIO>), dim3(nwg), dim3(64), 0, 0, d_out, iters, skew);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3f * 2.4e9f / iters;          // cycles per iteration
}
int main() {
    int* d_out; CK(hipMalloc(&d_out, 256 * 64 * 64 * 4));
    const int iters = 4000;
    printf("cycles per iteration (2.4 GHz) of one wave; D = 256 dependent VALU, L = 16 dependent LDS reads\n");
    printf("waves/SIMD    D only    L only    D + L   D + L skewed   prio L>D   prio D>L   (D+L expected: perfect overlap max(T1, W x busy) / none T1 + (W-1) x busy)\n");
    float d1 = 0, dl1 = 0, busy = 0;
    for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
        const int nwg = 256 * 4 * wps;             // 256 CUs x 4 SIMDs x waves per SIMD
        const float d = run<1>(d_out, nwg, iters, 0), l = run<2>(d_out, nwg, iters, 0), dl = run<0>(d_out, nwg, iters, 0), dls = run<0>(d_out, nwg, iters, 24),
                    dlp = run<0, 1>(d_out, nwg, iters, 0), dlq = run<0, 2>(d_out, nwg, iters, 0);
        if (wps == 1) { d1 = d; dl1 = dl; }
        if (wps == 8) busy = d / 8;                // vector-unit time of one wave's D phase = saturated D-only time / waves
        printf("%6d      %8.0f  %8.0f  %8.0f  %8.0f  %8.0f  %8.0f\n", wps, d, l, dl, dls, dlp, dlq);
    }
    printf("vector-unit time of one D phase (from 8 waves/SIMD): %.0f cycles; single-wave D %.0f, D + L %.0f\n", busy, d1, dl1);
    for (int wps : {2, 3, 4, 5, 6, 8}) printf("  W=%d: perfect %.0f, none %.0f\n", wps, dl1 > wps * busy ? dl1 : wps * busy, dl1 + (wps - 1) * busy);
    return 0;
}
