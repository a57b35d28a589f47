// micro-benchmark: does the size of a loop body matter to ONE wave per CU?  The same dependent v_mul_hi / v_add chain (7 bytes per
// instruction on average) in loop bodies of 0.5 ... 112 KB, one wave per CU (256 workgroups), and 4 / 16 waves per CU.
//   hipcc --offload-arch=gfx950 -O2 -o build/mb_loopsize tools/debug/mb_loopsize.hip ; gpurun -- ./build/mb_loopsize
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define R2(x) x x
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define UNIT "v_mul_hi_i32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_mul_hi_i32 %0, %0, %1\n v_add_u32 %0, 0x12345, %0\n"       /* 28 bytes */
#define B64 R16(UNIT)                  /* 64 instructions, 448 B */
template <int K> __global__ void __launch_bounds__(64) kern(int* out, int iters) {
    int v0 = threadIdx.x * 2654435 + 12345, v1 = v0 * 3 + 7;
    for (int it = 0; it < iters; it++) {
        if (K == 0) asm volatile(B64 : "+v"(v0) : "v"(v1));
        if (K == 1) asm volatile(R2(B64) : "+v"(v0) : "v"(v1));
        if (K == 2) asm volatile(R4(B64) : "+v"(v0) : "v"(v1));
        if (K == 3) asm volatile(R2(R4(B64)) : "+v"(v0) : "v"(v1));
        if (K == 4) asm volatile(R16(B64) : "+v"(v0) : "v"(v1));
        if (K == 5) asm volatile(R2(R16(B64)) : "+v"(v0) : "v"(v1));
        if (K == 6) asm volatile(R4(R16(B64)) : "+v"(v0) : "v"(v1));
        if (K == 7) asm volatile(R2(R4(R16(B64))) : "+v"(v0) : "v"(v1));
        if (K == 8) asm volatile(R16(R16(B64)) : "+v"(v0) : "v"(v1));
    }
    out[blockIdx.x * 64 + threadIdx.x] = v0;
}
template <int K> void run(int* d_out, int nwg, long instr_total) {
    const int per_iter = 64 << K;
    const int iters = (int)(instr_total / per_iter);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern<K>, dim3(nwg), dim3(64), 0, 0, d_out, 2);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern<K>, dim3(nwg), dim3(64), 0, 0, d_out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %6.1f KB: %6.2f", per_iter * 7.0 / 1024, ms * 1e-3 * 2.4e9 / ((double)iters * per_iter));
}
int main() {
    int* d_out; CK(hipMalloc(&d_out, 256 * 64 * 64 * 4));
    printf("cycles per instruction per wave at 2.4 GHz, by loop-body size\n");
    for (int wpc : {1, 2, 4, 16}) {
        printf("waves/CU %2d:", wpc);
        const long n = 1 << 21;
        run<0>(d_out, 256 * wpc, n); run<1>(d_out, 256 * wpc, n); run<2>(d_out, 256 * wpc, n); run<3>(d_out, 256 * wpc, n); run<4>(d_out, 256 * wpc, n);
        run<5>(d_out, 256 * wpc, n); run<6>(d_out, 256 * wpc, n); run<7>(d_out, 256 * wpc, n); run<8>(d_out, 256 * wpc, n);
        printf("\n");
    }
    return 0;
}
