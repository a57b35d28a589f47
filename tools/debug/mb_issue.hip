// micro-benchmark: instruction issue on gfx950 -- how fast ONE wave issues dependent / independent VALU and SALU instructions, what
// several waves on one SIMD get together, and whether one wave's SALU work overlaps another wave's VALU work.
//   hipcc --offload-arch=gfx950 -O2 -o build/mb_issue tools/debug/mb_issue.hip ; gpurun -- ./build/mb_issue
// Kernel kinds (per iteration 64 instructions of the kind, plus the loop's 2-3 scalar instructions):
//   0 VALU dependent chain (v_add_u32)          1 VALU 4 independent chains        2 VALU v_mul_hi_i32 dependent
//   3 SALU dependent chain (s_add_u32)          4 SALU 4 independent chains        5 SALU s_mul_hi_i32 dependent
//   6 alternating dependent v_add / s_add (scalar feeds nothing vector: independent streams inside one wave)
//   7 even waves run kind 0, odd waves kind 3 (do SALU waves and VALU waves share an issue port?)
//   8 v_readlane -> s_add -> v_add dependent round trip (the lane-register recursions of the analysis kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
template <int K> __global__ void __launch_bounds__(64) kern(int* out, int iters, int seed) {
    int v0 = threadIdx.x + seed, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7;
    int s0 = seed, s1 = seed * 3, s2 = seed * 5, s3 = seed * 7;
    const int kind = K == 7 ? (((blockIdx.x) & 1) ? 3 : 0) : K;
    for (int it = 0; it < iters; it++) {
        if (kind == 0) { asm volatile(R64("v_add_u32 %0, %0, %1\n") : "+v"(v0) : "v"(v1)); }
        else if (kind == 1) { asm volatile(R16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(it)); }
        else if (kind == 2) { asm volatile(R64("v_mul_hi_i32 %0, %0, %1\n") : "+v"(v0) : "v"(v1)); }
        else if (kind == 3) { asm volatile(R64("s_add_u32 %0, %0, %1\n") : "+s"(s0) : "s"(s1) : "scc"); }
        else if (kind == 4) { asm volatile(R16("s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(it) : "scc"); }
        else if (kind == 5) { asm volatile(R64("s_mul_hi_i32 %0, %0, %1\n") : "+s"(s0) : "s"(s1)); }
        else if (kind == 6) { asm volatile(R16("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, %3\n v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, %3\n") : "+v"(v0), "+s"(s0) : "v"(v1), "s"(s1) : "scc"); }
        else if (kind == 8) { asm volatile(R16("v_readlane_b32 %1, %0, 3\n s_add_u32 %1, %1, %2\n v_add_u32 %0, %1, %0\n s_nop 0\n") : "+v"(v0), "+s"(s0) : "s"(s1) : "scc"); }
    }
    out[blockIdx.x * 64 + threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3;
}
template <int K> float run(int* d_out, int nwg, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern<K>, dim3(nwg), dim3(64), 0, 0, d_out, 8, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern<K>, dim3(nwg), dim3(64), 0, 0, d_out, iters, 1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    int* d_out; CK(hipMalloc(&d_out, 256 * 64 * 64 * 4));
    const char* nm[9] = {"VALU add dependent", "VALU add 4 chains", "VALU mul_hi dependent", "SALU add dependent", "SALU add 4 chains", "SALU mul_hi dependent",
                         "v_add/s_add alternating", "even waves VALU, odd SALU", "readlane->s_add->v_add"};
    const int per_it[9] = {64, 64, 64, 64, 64, 64, 64, 64, 48};
    printf("cycles per instruction PER WAVE at 2.4 GHz (time x 2.4e9 / instructions issued by one wave); waves/CU = workgroups / 256\n");
    for (int wpc : {1, 4, 8, 16, 32}) {          // waves per CU (1: one wave on one SIMD of every CU; 4: one per SIMD; 16: four per SIMD)
        printf("waves/CU %2d:", wpc);
        float t[9];
        t[0] = run<0>(d_out, 256 * wpc, iters); t[1] = run<1>(d_out, 256 * wpc, iters); t[2] = run<2>(d_out, 256 * wpc, iters);
        t[3] = run<3>(d_out, 256 * wpc, iters); t[4] = run<4>(d_out, 256 * wpc, iters); t[5] = run<5>(d_out, 256 * wpc, iters);
        t[6] = run<6>(d_out, 256 * wpc, iters); t[7] = run<7>(d_out, 256 * wpc, iters); t[8] = run<8>(d_out, 256 * wpc, iters);
        for (int k = 0; k < 9; k++) printf("  [%d] %6.2f", k, t[k] * 1e-3 * 2.4e9 / ((double)iters * per_it[k]));
        printf("\n");
    }
    for (int k = 0; k < 9; k++) printf("[%d] %s\n", k, nm[k]);
    return 0;
}
