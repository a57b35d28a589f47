// Do a workgroup of kernel A (16 wavefronts, a given amount of LDS, optionally a large private segment) and one of kernel B (4 wavefronts,
// 24 320 B of LDS) share a compute unit?  A spins for 30 ms; B, launched on another stream once A is resident, records when it starts.
//   hipcc --offload-arch=gfx950 -O2 tools/debug/mb_coresident.hip -o /tmp/mb_coresident && /tmp/mb_coresident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <unistd.h>
template <int SCR>
__global__ void __launch_bounds__(1024) kA(unsigned long long* t, int* sink, int spin_ticks) {
    extern __shared__ int lds[];
    asm volatile("v_mov_b32 v95, 0" ::: "v95");      // 96 vector registers, like the front kernel
    int priv[SCR > 0 ? SCR : 1];
    lds[threadIdx.x] = threadIdx.x;
    if (SCR > 0) for (int i = 0; i < SCR; i++) priv[i] = i * threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) t[blockIdx.x] = t0;
    int acc = 0;
    if (spin_ticks < 0) {        // busy: dense dependent vector arithmetic + LDS traffic at issue priority 3, like an analysis wavefront
        __builtin_amdgcn_s_setprio(3);
        int a = threadIdx.x, b = blockIdx.x | 1;
        while (wall_clock64() - t0 < (unsigned long long)(-spin_ticks)) {
#pragma unroll
            for (int u = 0; u < 64; u++) { a = a * b + u; if ((u & 15) == 0) a += lds[(a & 1023)]; }
        }
        acc = a;
    } else
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) { __builtin_amdgcn_s_sleep(64); if (SCR > 0) acc += priv[(acc + threadIdx.x) % SCR]; }
    if (acc == 0x7fffffff) *sink = acc + lds[(threadIdx.x + 1) & 1023];
}
template <int SCR>
__global__ void __launch_bounds__(256) kB(unsigned long long* t, int* sink) {
    extern __shared__ int lds[];
    int priv[SCR > 0 ? SCR : 1];
    if (SCR > 0) { for (int i = 0; i < SCR; i++) priv[i] = i * threadIdx.x; if (priv[(threadIdx.x * 7) % SCR] == -5) *sink = 2; }
    asm volatile("v_mov_b32 v127, 0" ::: "v127");     // 128 vector registers, like the quantiser's kernel
    lds[threadIdx.x] = threadIdx.x;
    if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64();
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 255] == -1) *sink = 1;
}
int main() {
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long *ta, *tb; int* sink;
    hipMalloc(&ta, ncu * 8); hipMalloc(&tb, ncu * 8); hipMalloc(&sink, 4);
    hipStream_t s1, s2; int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, lo); hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi);
    hipFuncSetAttribute((const void*)kA<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)kA<400>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int busy = 0; busy < 2; busy++)
    for (int scr = 0; scr < 2; scr++)
        for (int ldsA : {138240}) {
            for (int ldsB : {24320, 20480}) {
                hipMemset(ta, 0, ncu * 8); hipMemset(tb, 0, ncu * 8);
                if (scr) hipLaunchKernelGGL(kA<400>, dim3(ncu), dim3(1024), ldsA, s1, ta, sink, busy ? -3000000 : 3000000);
                else hipLaunchKernelGGL(kA<0>, dim3(ncu), dim3(1024), ldsA, s1, ta, sink, busy ? -3000000 : 3000000);
                                // (give A half a millisecond to become resident)
                std::vector<unsigned long long> ha(ncu), hb(ncu);
                usleep(2000);
                if (scr) hipLaunchKernelGGL(kB<140>, dim3(ncu), dim3(256), ldsB, s2, tb, sink); else hipLaunchKernelGGL(kB<0>, dim3(ncu), dim3(256), ldsB, s2, tb, sink);
                hipDeviceSynchronize();
                hipMemcpy(ha.data(), ta, ncu * 8, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), tb, ncu * 8, hipMemcpyDeviceToHost);
                unsigned long long a0 = ~0ull, amax = 0; for (auto v : ha) { if (v < a0) a0 = v; if (v > amax) amax = v; }
                int early = 0; for (auto v : hb) if (v - a0 < 2900000ull) early++;
                printf("A %s: LDS %6d B, private %4d B/lane | B: LDS %5d B -> %3d of %d B-workgroups started while A was still spinning (A starts spread over %.3f ms)\n",
                       busy ? "busy prio 3" : "sleeping   ", ldsA, scr ? 1600 : 0, ldsB, early, ncu, (amax - a0) / 1e5);
            }
        }
    return 0;
}
