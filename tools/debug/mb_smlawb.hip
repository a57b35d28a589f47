// micro-benchmark: cost of SMLAWB (acc + ((a32 * b16) >> 16)) forms on gfx950.   hipcc --offload-arch=gfx950 -O2 -o build/mb_smlawb tools/debug/mb_smlawb.hip; gpurun -- ./build/mb_smlawb 1024 20000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32; typedef unsigned u32; typedef short i16;
#define NCH 12
__device__ __forceinline__ i32 f_mulhi(i32 acc, i32 a, i32 bpre) { return acc + __mulhi(a, bpre); }
__device__ __forceinline__ i32 f_c(i32 acc, i32 a, i32 b16) { return acc + (a >> 16) * b16 + (((i32)(a & 0xffff) * b16) >> 16); }
__device__ __forceinline__ i32 f_asm(i32 acc, i32 a, i32 b16) {
    i32 m, r;
    asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(m) : "v"(a), "v"(b16));
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(a), "v"(b16), "v"(acc));
    asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(r), "v"(m));
    return r;
}
__device__ __forceinline__ i32 f_asm_s(i32 acc, i32 a, i32 b16) {      // coefficient in a scalar register
    i32 m, r;
    asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(m) : "v"(a), "s"(b16));
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(a), "s"(b16), "v"(acc));
    asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(r), "v"(m));
    return r;
}
__device__ __forceinline__ i32 f_ref(i32 acc, i32 a, i32 b16) { return acc + (i32)(((long long)a * (long long)b16) >> 16); }

template <int V> __global__ void __launch_bounds__(64) kern(i32* out, const i32* in, const i32* co, int iters) {
    i32 s[NCH], acc[NCH];
    for (int j = 0; j < NCH; j++) { s[j] = in[threadIdx.x + 64 * j]; acc[j] = 0; }
    i32 b[4];
    for (int j = 0; j < 4; j++) {
        b[j] = (i32)(i16)co[j];
        if (V == 3) b[j] = __builtin_amdgcn_readfirstlane(b[j]);
        if (V == 0) { b[j] = (i32)((u32)b[j] << 16); asm volatile("" : "+v"(b[j])); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            const i32 bb = b[j & 3];
            if (V == 0) acc[j] = f_mulhi(acc[j], s[j], bb);
            if (V == 1) acc[j] = f_c(acc[j], s[j], bb);
            if (V == 2) acc[j] = f_asm(acc[j], s[j], bb);
            if (V == 3) acc[j] = f_asm_s(acc[j], s[j], bb);
            if (V == 4) acc[j] = f_ref(acc[j], s[j], bb);
            s[j] = s[j] * 3 + acc[j];          // full-rate-ish mixing so that inputs change (v_mad_u32_u24? no: 32-bit mul -> use add/xor instead)
        }
    }
    i32 r = 0;
    for (int j = 0; j < NCH; j++) r ^= acc[j];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}
// same with cheap mixing (xor / add) so that the MAC dominates
template <int V> __global__ void __launch_bounds__(64) kern2(i32* out, const i32* in, const i32* co, int iters) {
    i32 s[NCH], acc[NCH];
    for (int j = 0; j < NCH; j++) { s[j] = in[threadIdx.x + 64 * j]; acc[j] = 0; }
    i32 b[4];
    for (int j = 0; j < 4; j++) {
        b[j] = (i32)(i16)co[j];
        if (V == 3) b[j] = __builtin_amdgcn_readfirstlane(b[j]);
        if (V == 0) { b[j] = (i32)((u32)b[j] << 16); asm volatile("" : "+v"(b[j])); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            const i32 bb = b[j & 3];
            if (V == 0) acc[j] = f_mulhi(acc[j], s[j], bb);
            if (V == 1) acc[j] = f_c(acc[j], s[j], bb);
            if (V == 2) acc[j] = f_asm(acc[j], s[j], bb);
            if (V == 3) acc[j] = f_asm_s(acc[j], s[j], bb);
            if (V == 4) acc[j] = f_ref(acc[j], s[j], bb);
            s[j] = (s[j] ^ acc[j]) + 0x9e3779b9;
        }
    }
    i32 r = 0;
    for (int j = 0; j < NCH; j++) r ^= acc[j];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int V> float run(i32* d_out, const i32* d_in, const i32* d_co, int iters, int nwg, std::vector<i32>& h) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern2<V>, dim3(nwg), dim3(64), 0, 0, d_out, d_in, d_co, 16);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern2<V>, dim3(nwg), dim3(64), 0, 0, d_out, d_in, d_co, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    h.resize(64); CK(hipMemcpy(h.data(), d_out, 64 * 4, hipMemcpyDeviceToHost));
    return ms;
}
int main(int argc, char** argv) {
    int nwg = argc > 1 ? atoi(argv[1]) : 1024, iters = argc > 2 ? atoi(argv[2]) : 20000;
    std::vector<i32> in(64 * NCH), co(4), h[5];
    srand(1);
    for (auto& v : in) v = (i32)(((u32)rand() << 16) ^ (u32)rand() ^ ((u32)rand() << 31));
    co[0] = -32768; co[1] = 32767; co[2] = -12345; co[3] = 7864;
    i32 *d_in, *d_co, *d_out;
    CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_co, 16)); CK(hipMalloc(&d_out, (size_t)nwg * 64 * 4));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_co, co.data(), 16, hipMemcpyHostToDevice));
    float t[5];
    t[4] = run<4>(d_out, d_in, d_co, iters, nwg, h[4]);
    t[0] = run<0>(d_out, d_in, d_co, iters, nwg, h[0]);
    t[1] = run<1>(d_out, d_in, d_co, iters, nwg, h[1]);
    t[2] = run<2>(d_out, d_in, d_co, iters, nwg, h[2]);
    t[3] = run<3>(d_out, d_in, d_co, iters, nwg, h[3]);
    const char* nm[5] = {"mul_hi+add", "C i24 split", "asm 3-instr (vgpr coef)", "asm 3-instr (sgpr coef)", "64-bit reference"};
    for (int v = 0; v < 5; v++) {
        int bad = 0;
        for (int i = 0; i < 64; i++) bad += h[v][i] != h[4][i];
        printf("%-26s %8.3f ms  %6.2f ns per MAC+mix per wave  mismatches %d\n", nm[v], t[v], t[v] * 1e6 / ((double)iters * NCH), bad);
    }
    return 0;
}
