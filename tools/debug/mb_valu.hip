// mb_valu.hip -- VALU pipe time per instruction type on gfx950, as a function of the waves per SIMD that issue it.
// Every wave runs REPS x 64 instructions of one type on 8 independent dependency chains and reports its own cycle count;
// `hipcc --offload-arch=gfx950 -O2 tools/debug/mb_valu.hip -o /tmp/mb_valu && /tmp/mb_valu`
// Output: cycles per instruction seen by one wave, and SIMD cycles per instruction (= wave cycles / waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define REPS 256
#define CHAINS 8
template <int OP>
__global__ void __launch_bounds__(64) k(int* out, unsigned long long* cyc, int seed, unsigned long long* wall = nullptr) {
    const unsigned long long w0 = wall_clock64();
    int a[CHAINS];
    int b = (OP == 39 || OP == 46) ? (int)((threadIdx.x * 4 + seed * 8) & 252) : seed * 3 + threadIdx.x, c = seed + 7;
    unsigned long long m64 = 0x5555AAAA3333CCCCull * (unsigned)seed, m2[2] = {0, 0}, w64[4] = {1, 2, 3, 4};
#pragma unroll
    for (int j = 0; j < CHAINS; j++) a[j] = threadIdx.x * (j + 1) + seed;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int j = 0; j < CHAINS; j++) {
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 1) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 3) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b));
                if (OP == 5) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(b));
                if (OP == 6) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                if (OP == 7) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[j]));
                if (OP == 8) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 9) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[j]) : "v"(b));
                if (OP == 10) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 11) asm volatile("s_nop 0");
                if (OP == 12) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                if (OP == 13) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 14) asm volatile("v_mad_i32_i16 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                if (OP == 15) asm volatile("v_bfe_i32 %0, %0, 0, 16" : "+v"(a[j]));
                if (OP == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "s"(m64));
                if (OP == 17) asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m2[j & 1]) : "v"(a[j]), "v"(b));
                if (OP == 18) { asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m2[j & 1]) : "v"(a[j]), "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c), "s"(m2[j & 1])); }
                if (OP == 19) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[j]) : "v"(c), "v"(b), "s"(m64));
                if (OP == 20) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 21) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 22) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 23) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[j]));
                if (OP == 24) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 25) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 26) asm volatile("v_mov_b32 %0, %1" : "=v"(a[j]) : "v"(b));
                if (OP == 27) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 28) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                if (OP == 29) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(w64[j & 3]), "=s"(m2[0]) : "v"(b), "v"(c));
                if (OP == 30) asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(w64[j & 3]) : "v"(b), "s"(m64));
                if (OP == 31) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(b));
                if (OP == 32) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == 33) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "s"(seed));
                if (OP == 34) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[j]));
                if (OP == 35) { asm volatile("v_cmp_lt_i32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[j]) : "v"(a[(j + 1) & 7]), "v"(b), "v"(c) : "vcc"); }
                if (OP == 36) { asm volatile("v_cmp_lt_i32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[j]) : "v"(a[(j + 1) & 7]), "v"(b), "v"(c) : "vcc"); }
                if (OP == 37) { asm volatile("v_cmp_lt_i32_e64 %4, %1, %2\n\tv_cndmask_b32_e64 %0, %0, %3, %4" : "+v"(a[j]) : "v"(a[(j + 1) & 7]), "v"(b), "v"(c), "s"(m2[0])); }
                if (OP == 38) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b) : );
                if (OP == 39) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[j]) : "v"(b));
                if (OP == 40) asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(w64[j & 3]), "=s"(m2[0]) : "v"(b), "v"(c));
                if (OP == 41) asm volatile("v_ashrrev_i64 %0, 18, %0" : "+v"(w64[j & 3]));
                if (OP == 42) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w64[j & 3]) : "v"(w64[(j + 1) & 3]));
                if (OP == 43) { int sl_; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sl_) : "v"(a[j])); m2[j & 1] += sl_; }
                if (OP == 44) asm volatile("s_add_u32 %0, %0, %1" : "+s"(seed) : "s"(c));
                if (OP == 45) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[j]), "+v"(a[(j + 1) & 7]));
                if (OP == 46) asm volatile("ds_read_b32 %0, %1" : "=v"(a[j]) : "v"(b));
                if (OP == 47) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a[j]) : "s"(seed));
            }
            if (OP == 39 || OP == 46) asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    int s = (int)(m2[0] + m2[1] + w64[0] + w64[1] + w64[2] + w64[3]);
#pragma unroll
    for (int j = 0; j < CHAINS; j++) s += a[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (threadIdx.x == 0 && wall) wall[blockIdx.x] = wall_clock64() - w0;
}
// a dependent chain: ONE accumulator (latency of the instruction as one wave sees it)
template <int OP>
__global__ void __launch_bounds__(64) kd(int* out, unsigned long long* cyc, int seed) {
    int a = threadIdx.x + seed, b = seed * 3 + threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int u = 0; u < 64; u++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
            if (OP == 1) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a) : "v"(b));
            if (OP == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
            if (OP == 3) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a) : "v"(b));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name, int* d_out, unsigned long long* d_cyc, int ncu) {
    printf("%-16s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int nb = ncu * 4 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(nb), dim3(64), 0, 0, d_out, d_cyc, 3);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k<OP>, dim3(nb), dim3(64), 0, 0, d_out, d_cyc, 5);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(nb);
        hipMemcpy(h.data(), d_cyc, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        const double per = s / nb / (REPS * 64.0);
        printf("  %dw/simd: %5.2f wave-cyc  %5.2f simd-cyc |", wps, per, per / wps);
    }
    printf("\n");
    {   // effective shader clock under this load: cycle counter ticks per 100 MHz wall-clock tick, and the kernel's wall time by events
        const int nb = ncu * 4 * 8;
        unsigned long long* d_wall;
        hipMalloc(&d_wall, nb * sizeof(unsigned long long));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(nb), dim3(64), 0, 0, d_out, d_cyc, 9, d_wall);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nb), hw(nb);
        hipMemcpy(h.data(), d_cyc, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        hipMemcpy(hw.data(), d_wall, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double s = 0, sw = 0;
        for (int i = 0; i < nb; i++) { s += (double)h[i]; sw += (double)hw[i]; }
        printf("      8w/simd: counter %.1f ticks per 100 MHz tick; kernel %.3f ms by events = %.2f ns per instruction per SIMD\n", s / sw, ms, ms * 1e6 / (8.0 * REPS * 64.0));
        hipFree(d_wall);
    }
}
template <int OP> void rund(const char* name, int* d_out, unsigned long long* d_cyc, int ncu) {
    const int nb = ncu * 4;
    hipLaunchKernelGGL(kd<OP>, dim3(nb), dim3(64), 0, 0, d_out, d_cyc, 3);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb);
    hipMemcpy(h.data(), d_cyc, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-16s dependent chain, one wave per SIMD: %5.2f cycles per instruction\n", name, s / nb / (REPS * 64.0));
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz (s_memtime / readcyclecounter ticks may differ from the shader clock)\n", p.name, ncu, p.clockRate);
    int* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_out, (size_t)ncu * 32 * 64 * sizeof(int));
    hipMalloc(&d_cyc, (size_t)ncu * 32 * sizeof(unsigned long long));
    run<0>("v_add_u32", d_out, d_cyc, ncu);
    run<1>("v_mul_hi_i32", d_out, d_cyc, ncu);
    run<2>("v_mul_lo_u32", d_out, d_cyc, ncu);
    run<3>("v_mad_i32_i24", d_out, d_cyc, ncu);
    run<8>("v_mul_i32_i24", d_out, d_cyc, ncu);
    run<12>("v_mad_u32_u24", d_out, d_cyc, ncu);
    run<14>("v_mad_i32_i16", d_out, d_cyc, ncu);
    run<13>("v_pk_mul_lo_u16", d_out, d_cyc, ncu);
    run<4>("v_cndmask_b32", d_out, d_cyc, ncu);
    run<5>("v_mov_b32_dpp", d_out, d_cyc, ncu);
    run<6>("v_add3_u32", d_out, d_cyc, ncu);
    run<7>("v_ashrrev_i32", d_out, d_cyc, ncu);
    run<9>("v_alignbit_b32", d_out, d_cyc, ncu);
    run<10>("v_min_i32", d_out, d_cyc, ncu);
    run<15>("v_bfe_i32", d_out, d_cyc, ncu);
    run<11>("s_nop 0", d_out, d_cyc, ncu);
    run<16>("cndmask_e64 sgpr", d_out, d_cyc, ncu);
    run<35>("cmp+cnd vcc /2", d_out, d_cyc, ncu);
    run<36>("cmp+nop+cnd vcc", d_out, d_cyc, ncu);
    run<37>("cmp+cnd e64 /2", d_out, d_cyc, ncu);
    run<19>("cndmask indep", d_out, d_cyc, ncu);
    run<17>("v_cmp_e64", d_out, d_cyc, ncu);
    run<18>("v_cmp+cndmask /2", d_out, d_cyc, ncu);
    run<20>("v_max_i32", d_out, d_cyc, ncu);
    run<21>("v_xor_b32", d_out, d_cyc, ncu);
    run<22>("v_sub_u32", d_out, d_cyc, ncu);
    run<23>("v_lshlrev_b32", d_out, d_cyc, ncu);
    run<24>("v_and_b32", d_out, d_cyc, ncu);
    run<25>("v_or_b32", d_out, d_cyc, ncu);
    run<26>("v_mov_b32", d_out, d_cyc, ncu);
    run<27>("v_lshl_add_u32", d_out, d_cyc, ncu);
    run<28>("v_med3_i32", d_out, d_cyc, ncu);
    run<29>("v_mad_u64_u32", d_out, d_cyc, ncu);
    run<30>("v_lshrrev_b64", d_out, d_cyc, ncu);
    run<31>("v_add_u32_dpp", d_out, d_cyc, ncu);
    run<32>("v_add_u32_e64", d_out, d_cyc, ncu);
    run<33>("v_add_u32 sgpr", d_out, d_cyc, ncu);
    run<34>("v_ffbl_b32", d_out, d_cyc, ncu);
    run<39>("ds_bpermute_b32", d_out, d_cyc, ncu);
    run<46>("ds_read_b32", d_out, d_cyc, ncu);
    run<40>("v_mad_i64_i32", d_out, d_cyc, ncu);
    run<41>("v_ashrrev_i64", d_out, d_cyc, ncu);
    run<42>("v_lshl_add_u64", d_out, d_cyc, ncu);
    run<43>("v_readlane_b32", d_out, d_cyc, ncu);
    run<44>("s_add_u32", d_out, d_cyc, ncu);
    run<45>("v_permlane32_swap", d_out, d_cyc, ncu);
    run<47>("v_mul_hi_i32 sgpr", d_out, d_cyc, ncu);
    rund<0>("v_add_u32", d_out, d_cyc, ncu);
    rund<1>("v_mul_hi_i32", d_out, d_cyc, ncu);
    rund<2>("v_mul_lo_u32", d_out, d_cyc, ncu);
    rund<3>("v_mad_i32_i24", d_out, d_cyc, ncu);
    return 0;
}
