// rc_loop_repro.hip -- stand-alone check of the range decoder's CDF search in its two forms on gfx950:
//   form R: the reference's early-exit search with ++/-- overshoot (SKP_Silk_range_coder.c:136-170), transcribed 1:1
//   form S: the restructured search that solo_rc.h ships
// over random (base, range, start index) triples on a set of monotone CDFs, as a scalar per-lane loop (the way the decoder's
// parse lanes run it).  DESIGN.md section 2 records the outcome.   hipcc --offload-arch=gfx950 -O2 rc_loop_repro.hip -o rc_loop_repro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>

__device__ int search_R(const uint16_t* prob, int probIx, uint32_t range_Q16, uint32_t base_Q32, uint32_t* lo, uint32_t* hi) {
    uint32_t low_Q16, high_Q16, base_tmp;
    high_Q16 = prob[probIx];
    base_tmp = range_Q16 * high_Q16;
    if (base_tmp > base_Q32) {
        while (1) {
            low_Q16 = prob[--probIx];
            base_tmp = range_Q16 * low_Q16;
            if (base_tmp <= base_Q32) break;
            high_Q16 = low_Q16;
            if (high_Q16 == 0) return -1;
        }
    } else {
        while (1) {
            low_Q16 = high_Q16;
            high_Q16 = prob[++probIx];
            base_tmp = range_Q16 * high_Q16;
            if (base_tmp > base_Q32) { probIx--; break; }
            if (high_Q16 == 0xFFFF) return -1;
        }
    }
    *lo = low_Q16; *hi = high_Q16;
    return probIx;
}
__device__ int search_S(const uint16_t* prob, int probIx, uint32_t range_Q16, uint32_t base_Q32, uint32_t* lo, uint32_t* hi) {
    uint32_t low_Q16 = 0, high_Q16;
    high_Q16 = prob[probIx];
    if (range_Q16 * high_Q16 > base_Q32) {
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (;;) {
            probIx--;
            low_Q16 = prob[probIx];
            if (range_Q16 * low_Q16 <= base_Q32) break;
            high_Q16 = low_Q16;
            if (high_Q16 == 0) return -1;
        }
    } else {
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (;;) {
            low_Q16 = high_Q16;
            high_Q16 = prob[probIx + 1];
            if (range_Q16 * high_Q16 > base_Q32) break;
            probIx++;
            if (high_Q16 == 0xFFFF) return -1;
        }
    }
    *lo = low_Q16; *hi = high_Q16;
    return probIx;
}

// several symbols in a row per lane with DIVERGENT trip counts (that is how the parse lanes run); tables in LDS like the decoder's
template <int FORM>
__global__ void __launch_bounds__(64) k(const uint16_t* cdfs, int ncdf, int len, const uint32_t* base, const uint32_t* range, const int* start, int n, int per_lane, int* out) {
    __shared__ uint16_t tab[4096];
    for (int i = threadIdx.x; i < ncdf * len && i < 4096; i += 64) tab[i] = cdfs[i];
    __syncthreads();
    const int lane = blockIdx.x * 64 + threadIdx.x;
    for (int j = 0; j < per_lane; j++) {
        const int i = lane * per_lane + j;
        if (i >= n) break;
        uint32_t lo = 0, hi = 0;
        const uint16_t* prob = &tab[(i % ncdf) * len];
        const int s = FORM == 0 ? search_R(prob, start[i], range[i], base[i], &lo, &hi) : search_S(prob, start[i], range[i], base[i], &lo, &hi);
        out[3 * i] = s; out[3 * i + 1] = (int)lo; out[3 * i + 2] = (int)hi;
    }
}

int main() {
    const int ncdf = 24, len = 130, n = 1 << 20, per_lane = 16;
    std::vector<uint16_t> cdfs(ncdf * len);
    srand(12345);
    for (int c = 0; c < ncdf; c++) {                         // monotone CDFs 0 .. 65535 with very uneven steps (long scans both ways)
        std::vector<uint32_t> w(len - 1);
        uint64_t tot = 0;
        for (int i = 0; i < len - 1; i++) { w[i] = 1 + (rand() % ((c % 3 == 0) ? 4 : ((i % 17 == 0) ? 5000 : 40))); tot += w[i]; }
        uint64_t acc = 0;
        cdfs[c * len] = 0;
        for (int i = 1; i < len; i++) { acc += w[i - 1]; uint32_t v = (uint32_t)(acc * 65535 / tot); if (v <= cdfs[c * len + i - 1]) v = cdfs[c * len + i - 1] + 1; cdfs[c * len + i] = (uint16_t)(v > 65535 ? 65535 : v); }
        cdfs[c * len + len - 1] = 65535;
    }
    std::vector<uint32_t> base(n), range(n); std::vector<int> start(n);
    for (int i = 0; i < n; i++) {
        range[i] = 0x100 + (uint32_t)(rand() % 0xFF00);
        const uint32_t r = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
        base[i] = (uint32_t)(((uint64_t)r * (uint64_t)(range[i] * 65535u)) >> 32);      // < range * 65535: a decodable value
        start[i] = 1 + rand() % (len - 2);
    }
    uint16_t* d_c; uint32_t *d_b, *d_r; int *d_s, *d_o0, *d_o1;
    hipMalloc(&d_c, cdfs.size() * 2); hipMalloc(&d_b, n * 4); hipMalloc(&d_r, n * 4); hipMalloc(&d_s, n * 4); hipMalloc(&d_o0, n * 12); hipMalloc(&d_o1, n * 12);
    hipMemcpy(d_c, cdfs.data(), cdfs.size() * 2, hipMemcpyHostToDevice); hipMemcpy(d_b, base.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_r, range.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(d_s, start.data(), n * 4, hipMemcpyHostToDevice);
    const int lanes = (n + per_lane - 1) / per_lane;
    hipLaunchKernelGGL(k<0>, dim3((lanes + 63) / 64), dim3(64), 0, 0, d_c, ncdf, len, d_b, d_r, d_s, n, per_lane, d_o0);
    hipLaunchKernelGGL(k<1>, dim3((lanes + 63) / 64), dim3(64), 0, 0, d_c, ncdf, len, d_b, d_r, d_s, n, per_lane, d_o1);
    std::vector<int> o0(3 * n), o1(3 * n);
    hipMemcpy(o0.data(), d_o0, n * 12, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d_o1, n * 12, hipMemcpyDeviceToHost);
    // host truth: the symbol is the unique index with range*cdf[ix] <= base < range*cdf[ix+1]
    long badR = 0, badS = 0;
    for (int i = 0; i < n; i++) {
        const uint16_t* p = &cdfs[(i % ncdf) * len];
        int ix = 0;
        while (ix + 1 < len - 1 && (uint64_t)range[i] * p[ix + 1] <= base[i]) ix++;
        const int lo = p[ix], hi = p[ix + 1];
        if (o0[3 * i] != ix || o0[3 * i + 1] != lo || o0[3 * i + 2] != hi) { if (badR < 5) printf("form R wrong at %d: got %d (%d,%d) want %d (%d,%d) start %d\n", i, o0[3 * i], o0[3 * i + 1], o0[3 * i + 2], ix, lo, hi, start[i]); badR++; }
        if (o1[3 * i] != ix || o1[3 * i + 1] != lo || o1[3 * i + 2] != hi) { if (badS < 5) printf("form S wrong at %d: got %d (%d,%d) want %d (%d,%d) start %d\n", i, o1[3 * i], o1[3 * i + 1], o1[3 * i + 2], ix, lo, hi, start[i]); badS++; }
    }
    printf("rc_loop_repro: %d searches; reference-form mismatches %ld, shipped-form mismatches %ld\n", n, badR, badS);
    return 0;
}
