// micro-benchmark: what slows a latency-bound wave (one per CU, like the quantiser's) when other waves fill the CU?
//   hipcc --offload-arch=gfx950 -O2 -o build/mb_corun tools/debug/mb_corun.hip ; gpurun -- ./build/mb_corun
// A PROBE kernel (256 workgroups of one wave, s_setprio 3) runs a fixed dependent chain and times itself with the shader clock (s_memtime)
// and the constant 100 MHz clock (s_memrealtime).  It runs alone and beside a LOAD kernel of W waves per CU on another HIP stream:
//   load 1 dependent v_mul_hi_i32 chains (quarter-rate VALU)      load 2 independent v_add chains (full-rate VALU issue)
//   load 3 LDS reads + writes                                      load 4 streaming global loads (L2 / HBM traffic)
//   load 5 a 256 KB straight-line VALU body (instruction cache)   load 6 SALU chains
//   load 7 every wave loops inside its OWN 8 KB region of a 512 KB body (waves out of step: the worst case for the instruction cache)
//   load 8 per-lane private arrays in global memory, dynamically indexed reads + writes (what scratch traffic looks like to the L1)
// Probe kinds: 0 v_mul_hi / v_add dependent chain   1 the same with a ds_bpermute every 8 instructions   2 dependent global loads (L2-resident ring)
//              3 probe 0 with a 16 KB loop body (instruction fetch matters)
// (the loads end by iteration caps sized for ~0.1 s; they do not poll a stop flag: polling host memory from every wave jams the L1s)
// Output: probe time alone and beside each load, in ns per step, and the shader clock seen by the probe (memtime ticks per realtime tick).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))

struct ProbeOut { unsigned long long dt_core, dt_real; unsigned hwid, pad; };

template <int K> __global__ void __launch_bounds__(64) probe(ProbeOut* out, int* sink, const int* ring, int iters) {
    __shared__ int lds[64];
    __builtin_amdgcn_s_setprio(3);
    int v0 = threadIdx.x * 2654435 + 12345, v1 = v0 * 3 + 7;
    int idx = threadIdx.x;
    lds[threadIdx.x] = v0;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if (K == 0) { asm volatile(R16("v_mul_hi_i32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_mul_hi_i32 %0, %0, %1\n v_add_u32 %0, 0x12345, %0\n") : "+v"(v0) : "v"(v1)); }
        else if (K == 1) {
            asm volatile(R4("v_mul_hi_i32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_mul_hi_i32 %0, %0, %1\n v_add_u32 %0, 0x12345, %0\n v_mul_hi_i32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_mul_hi_i32 %0, %0, %1\n"
                            "ds_bpermute_b32 %0, %2, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(v0) : "v"(v1), "v"((int)((threadIdx.x ^ 1) * 4)));
        } else if (K == 3) { asm volatile(R4(R4(R4(R16("v_mul_hi_i32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_mul_hi_i32 %0, %0, %1\n v_add_u32 %0, 0x12345, %0\n")))) : "+v"(v0) : "v"(v1)); }
        else {
            for (int j = 0; j < 8; j++) { idx = __builtin_nontemporal_load(ring + idx); asm volatile("" : "+v"(idx)); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x] = ProbeOut{t1 - t0, r1 - r0, hw, 0};
    }
    sink[blockIdx.x * 64 + threadIdx.x] = v0 + idx + lds[(threadIdx.x + 1) & 63];
}

template <int K> __global__ void __launch_bounds__(64) load(int* sink, const int* big, size_t big_words, int iters) {
    __shared__ int lds[2048];
    int v0 = threadIdx.x + 1, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, s0 = 3, s1 = 5;
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = i;
    size_t pos = ((size_t)blockIdx.x * 64 + threadIdx.x) * 4;
    for (int it = 0; it < iters; it++) {
        if (K == 1) { asm volatile(R16("v_mul_hi_i32 %0, %0, %1\n v_mul_hi_i32 %2, %2, %1\n v_mul_hi_i32 %0, %0, %3\n v_mul_hi_i32 %2, %2, %3\n") : "+v"(v0), "+v"(v1), "+v"(v2) : "v"(v3)); }
        else if (K == 2) { asm volatile(R16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(it)); }
        else if (K == 3) {
            for (int j = 0; j < 16; j++) { v0 += lds[(v0 + j * 64 + threadIdx.x) & 2047]; lds[(threadIdx.x + j * 64) & 2047] = v0; }
        } else if (K == 4) {
            for (int j = 0; j < 8; j++) { const int4 q = *(const int4*)(big + (pos & (big_words - 1))); v0 += q.x + q.y + q.z + q.w; pos += (size_t)gridDim.x * 64 * 4; }
        } else if (K == 5) { asm volatile(R4(R64(R64("v_add_u32 %0, %0, %1\n v_xor_b32 %0, 0x5a5a5a5a, %0\n"))) : "+v"(v0) : "v"(v1)); }
        else if (K == 6) { asm volatile(R16("s_mul_hi_i32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_mul_i32 %0, %0, %1\n s_add_u32 %0, %0, 77\n") : "+s"(s0) : "s"(s1) : "scc"); }
        else if (K == 7) {
            // 64 regions of 8 KB (1024 two-word instructions each); a wave stays in region (wave id & 63)
            const int region = (blockIdx.x * 7 + 3) & 63;
#define REGION(n) if (region == (n)) { asm volatile(R4(R256("v_add_u32 %0, %0, %1\n")) : "+v"(v0) : "v"(v1)); asm volatile(".p2align 6\n"); } else
            REGION(0) REGION(1) REGION(2) REGION(3) REGION(4) REGION(5) REGION(6) REGION(7) REGION(8) REGION(9) REGION(10) REGION(11) REGION(12) REGION(13) REGION(14) REGION(15)
            REGION(16) REGION(17) REGION(18) REGION(19) REGION(20) REGION(21) REGION(22) REGION(23) REGION(24) REGION(25) REGION(26) REGION(27) REGION(28) REGION(29) REGION(30) REGION(31)
            REGION(32) REGION(33) REGION(34) REGION(35) REGION(36) REGION(37) REGION(38) REGION(39) REGION(40) REGION(41) REGION(42) REGION(43) REGION(44) REGION(45) REGION(46) REGION(47)
            REGION(48) REGION(49) REGION(50) REGION(51) REGION(52) REGION(53) REGION(54) REGION(55) REGION(56) REGION(57) REGION(58) REGION(59) REGION(60) REGION(61) REGION(62) REGION(63)
            { }
        } else if (K == 8) {
            int* mine = (int*)big + ((size_t)blockIdx.x * 64 + threadIdx.x) * 64;           // 256 B per lane, like a scratch frame
            for (int j = 0; j < 8; j++) { const int a = (v0 + j * 5) & 63; v0 += mine[a]; mine[(a + 17) & 63] = v0; }
        }
    }
    sink[blockIdx.x * 64 + threadIdx.x] = v0 + v1 + v2 + v3 + s0;
}

static int* d_sink; static int* d_big; static size_t big_words = (size_t)1 << 28; static int* d_ring; static ProbeOut* d_out;
static hipStream_t sP, sL;

template <int PK> void run_probe(int iters, const char* tag, double steps_per_iter) {
    hipLaunchKernelGGL(probe<PK>, dim3(256), dim3(64), 0, sP, d_out, d_sink, d_ring, iters);
    CK(hipStreamSynchronize(sP));
    std::vector<ProbeOut> h(256);
    CK(hipMemcpy(h.data(), d_out, sizeof(ProbeOut) * 256, hipMemcpyDeviceToHost));
    std::vector<double> ns, mhz; std::vector<unsigned> cu;
    for (auto& o : h) { ns.push_back(o.dt_real * 10.0 / (iters * steps_per_iter)); mhz.push_back(100.0 * o.dt_core / (double)o.dt_real); cu.push_back(o.hwid & 0xffffff00u); }
    std::sort(ns.begin(), ns.end()); std::sort(mhz.begin(), mhz.end()); std::sort(cu.begin(), cu.end());
    const int distinct = (int)(std::unique(cu.begin(), cu.end()) - cu.begin());
    printf("   %-34s ns/instr median %7.3f  p90 %7.3f  max %7.3f | memtime/realtime x100 median %7.1f | distinct hw ids %d\n", tag, ns[128], ns[230], ns[255], mhz[128], distinct);
}

template <int PK, int LK> void corun(int wpc, int probe_iters, double spi, const char* tag) {
    // iteration caps: every load ends by itself after 0.05 - 0.3 s even if the stop flag were never seen
    const int cap[9] = {0, 150000, 600000, 150000, 20000, 600, 400000, 100000, 60000};
    const int iters_l = (int)(cap[LK] * (wpc <= 4 ? 2.5 : 1.0));
    hipLaunchKernelGGL(load<LK>, dim3(256 * wpc), dim3(64), 0, sL, d_sink + 65536, d_big, big_words, iters_l);
    // let the load kernel spread over the CUs before the probe arrives
    for (volatile int spin = 0; spin < 20000000; spin++) { }
    char name[96]; snprintf(name, sizeof name, "%s, %d waves/CU", tag, wpc);
    run_probe<PK>(probe_iters, name, spi);
    hipEvent_t e0; CK(hipEventCreate(&e0)); CK(hipEventRecord(e0, sL));
    const bool still_running = hipEventQuery(e0) == hipErrorNotReady;          // the load outlived the probe: the whole probe ran beside it
    CK(hipStreamSynchronize(sL)); CK(hipEventDestroy(e0));
    if (!still_running) printf("      (the load ended before the probe: ignore the line above)\n");
}

template <int PK> void suite(int iters, double spi, const char* pname) {
    printf("probe %d: %s\n", PK, pname);
    run_probe<PK>(iters, "alone", spi);
    run_probe<PK>(iters, "alone (again)", spi);
    for (int w : {4, 15}) {
        corun<PK, 1>(w, iters, spi, "beside v_mul_hi chains");
        corun<PK, 2>(w, iters, spi, "beside v_add x4 chains");
        corun<PK, 3>(w, iters, spi, "beside LDS read+write");
        corun<PK, 4>(w, iters, spi, "beside streaming global loads");
        corun<PK, 5>(w, iters, spi, "beside 256 KB code body");
        corun<PK, 6>(w, iters, spi, "beside SALU chains");
        corun<PK, 7>(w, iters, spi, "beside 64 x 8 KB code regions");
        corun<PK, 8>(w, iters, spi, "beside private-array traffic");
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    CK(hipMalloc(&d_sink, 4 * (65536 + 256 * 64 * 64)));
    CK(hipMalloc(&d_big, big_words * 4)); CK(hipMemset(d_big, 1, big_words * 4));
    CK(hipMalloc(&d_out, sizeof(ProbeOut) * 256));
    std::vector<int> ring(1 << 16);                         // 256 KB pointer ring: L2-resident
    for (int i = 0; i < (1 << 16); i++) ring[i] = (int)(((unsigned)i * 40503u + 64u * 977u) & 0xffffu);
    CK(hipMalloc(&d_ring, 4 << 16)); CK(hipMemcpy(d_ring, ring.data(), 4 << 16, hipMemcpyHostToDevice));
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&sP, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&sL, hipStreamNonBlocking, lo));
    suite<0>(iters, 64, "dependent v_mul_hi / v_add chain (64 instructions per step counted)");
    suite<1>(iters, 32, "the same with a ds_bpermute every 8 instructions (32 per step)");
    suite<2>(iters / 8, 8, "dependent non-temporal global loads from a 256 KB ring (8 per step)");
    suite<3>(iters / 32, 2048, "probe 0 with a 16 KB loop body");
    return 0;
}
