// mb_mix.hip -- do instruction TYPES overlap on a gfx950 SIMD?  Every wave runs REPS x 64 "slots"; a slot is one instruction of type A,
// or one of type A followed by one of type B (different waves are at different places, so the SIMD sees both types all the time).
// If the types issue side by side, time(A+B) ~= max(time(A), time(B)); if the SIMD takes them one after the other, ~= the sum.
//   hipcc --offload-arch=gfx950 -O2 tools/debug/mb_mix.hip -o build/mb_mix && build/mb_mix
// Output: ns per SLOT per SIMD (kernel time by events / slots per SIMD) at 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define REPS 256
#define CH 8
// A / B: 0 none, 1 v_mul_hi (slow VALU), 2 v_add (fast VALU), 3 s_add/s_mul chain (SALU), 4 ds_read_b32, 5 v_cndmask e64, 6 s_nop 0,
// 7 v_add with SGPR operand, 8 ds_bpermute, 9 v_readlane, 10 v_lshlrev imm, 11 v_mov_dpp, 12 v_add3, 13 v_ashrrev imm, 14 v_mad_i64_i32
template <int T>
__device__ __forceinline__ void emit(int& a, int b, int& sa, int sb, unsigned long long m64, int ldsaddr, long long& w) {
    if (T == 1) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (T == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (T == 3) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(sa) : "s"(sb));
    if (T == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(a) : "v"(ldsaddr));
    if (T == 5) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(m64));
    if (T == 6) asm volatile("s_nop 0");
    if (T == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "s"(sb));
    if (T == 8) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a) : "v"(ldsaddr));
    if (T == 9) { int t_; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(t_) : "v"(a)); sa += t_; }
    if (T == 10) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a));
    if (T == 11) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));
    if (T == 12) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    if (T == 13) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a));
    if (T == 14) { unsigned long long dm_; asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(w), "=s"(dm_) : "v"(b), "v"(a)); }
    if (T == 15) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sa) : "s"(sb));
    if (T == 16) asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m64) : "v"(a), "v"(b));
    if (T == 17) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a) : "v"(b));
    if (T == 18) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a) : "v"(b));
}
template <int A, int B>
__global__ void __launch_bounds__(64) k(int* out, int seed) {
    __shared__ int lds[256];
    lds[threadIdx.x] = seed; lds[threadIdx.x + 64] = seed; lds[threadIdx.x + 128] = seed; lds[threadIdx.x + 192] = seed;
    __syncthreads();
    int a[CH], sa[CH];
    long long w[CH];
    int b = seed * 3 + threadIdx.x, sb = seed | 1;
    unsigned long long m64 = 0x5555AAAA3333CCCCull * (unsigned)seed;
    const int ldsaddr = (threadIdx.x * 4 + seed * 8) & 252;
#pragma unroll
    for (int j = 0; j < CH; j++) { a[j] = threadIdx.x * (j + 1) + seed; sa[j] = seed + j; w[j] = j; }
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int j = 0; j < CH; j++) {
                emit<A>(a[j], b, sa[j], sb, m64, ldsaddr, w[j]);
                emit<B>(a[(j + 4) & 7], b, sa[(j + 4) & 7], sb, m64, ldsaddr, w[(j + 4) & 7]);
            }
            if (A == 4 || A == 8 || B == 4 || B == 8) asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    int s = 0;
#pragma unroll
    for (int j = 0; j < CH; j++) s += a[j] + sa[j] + (int)w[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int A, int B> void run(const char* name, int* d_out, int ncu) {
    printf("%-44s", name);
    for (int wps = 4; wps <= 8; wps *= 2) {
        const int nb = ncu * 4 * wps;
        hipLaunchKernelGGL((k<A, B>), dim3(nb), dim3(64), 0, 0, d_out, 3);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int t = 0; t < 3; t++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<A, B>), dim3(nb), dim3(64), 0, 0, d_out, 5 + t);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("  %dw/simd: %6.3f ms = %5.2f ns/slot/SIMD |", wps, best, best * 1e6 / ((double)wps * REPS * 64.0));
    }
    printf("\n");
}
// ONE dependent chain per wave, W waves per SIMD: how well does the SIMD interleave the chains of its waves?
template <int T>
__global__ void __launch_bounds__(64) kdep(int* out, int seed) {
    int a = threadIdx.x + seed, b = seed * 3 + threadIdx.x;
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int u = 0; u < 64; u++) {
            if (T == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
            if (T == 1) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a) : "v"(b));
            if (T == 2) { asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
template <int T> void rundep(const char* name, int* d_out, int ncu) {
    printf("%-44s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int nb = ncu * 4 * wps;
        hipLaunchKernelGGL((kdep<T>), dim3(nb), dim3(64), 0, 0, d_out, 3);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int t = 0; t < 3; t++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((kdep<T>), dim3(nb), dim3(64), 0, 0, d_out, 5 + t);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double n = (T == 2 ? 2.0 : 1.0) * REPS * 64.0;
        printf(" %dw: %5.2f ns/instr/wave %5.2f /SIMD |", wps, best * 1e6 / n, best * 1e6 / n / wps);
    }
    printf("\n");
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    int* d_out;
    hipMalloc(&d_out, (size_t)ncu * 32 * 64 * sizeof(int));
    rundep<0>("dependent v_add chain", d_out, ncu);
    rundep<1>("dependent v_mul_hi chain", d_out, ncu);
    rundep<2>("dependent mul_hi+add chain", d_out, ncu);
    run<0, 0>("empty loop", d_out, ncu);
    run<1, 0>("v_mul_hi", d_out, ncu);
    run<2, 0>("v_add", d_out, ncu);
    run<17, 0>("v_sub", d_out, ncu);
    run<13, 0>("v_ashrrev imm", d_out, ncu);
    run<10, 0>("v_lshlrev imm", d_out, ncu);
    run<7, 0>("v_add sgpr", d_out, ncu);
    run<12, 0>("v_add3", d_out, ncu);
    run<18, 0>("v_min", d_out, ncu);
    run<5, 0>("v_cndmask e64", d_out, ncu);
    run<16, 0>("v_cmp e64", d_out, ncu);
    run<11, 0>("v_mov_dpp", d_out, ncu);
    run<14, 0>("v_mad_i64_i32", d_out, ncu);
    run<3, 0>("s_mul_i32", d_out, ncu);
    run<15, 0>("s_add_u32", d_out, ncu);
    run<6, 0>("s_nop 0", d_out, ncu);
    run<4, 0>("ds_read_b32", d_out, ncu);
    run<8, 0>("ds_bpermute", d_out, ncu);
    run<9, 0>("v_readlane", d_out, ncu);
    run<1, 1>("v_mul_hi + v_mul_hi", d_out, ncu);
    run<1, 2>("v_mul_hi + v_add", d_out, ncu);
    run<2, 2>("v_add + v_add", d_out, ncu);
    run<1, 3>("v_mul_hi + s_mul", d_out, ncu);
    run<2, 3>("v_add + s_mul", d_out, ncu);
    run<1, 15>("v_mul_hi + s_add", d_out, ncu);
    run<1, 6>("v_mul_hi + s_nop", d_out, ncu);
    run<1, 4>("v_mul_hi + ds_read", d_out, ncu);
    run<2, 4>("v_add + ds_read", d_out, ncu);
    run<1, 8>("v_mul_hi + ds_bpermute", d_out, ncu);
    run<3, 4>("s_mul + ds_read", d_out, ncu);
    run<1, 9>("v_mul_hi + v_readlane", d_out, ncu);
    return 0;
}
