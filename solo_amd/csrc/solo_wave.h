// solo_wave.h -- execution model of the SOLO kernels: ONE 64-lane wavefront owns ONE stream.
//
// All codec code is written "wave-uniform": control flow and scalars are identical in every lane
// (the compiler keeps provably uniform values in SGPRs), arrays live in LDS / HBM, and the
// data-parallel loops are strided over the lanes:
//
//     SX_PAR(i, n) { out[i] = f(in[i]); }      // lane l handles i = l, l+64, ...
//     wv_sync();                               // make the stores visible to the whole wave
//     s = wv_sum(partial);                     // wrapping int32 add => bit-exact tree reduction
//
// The workgroup is exactly one wavefront (blockDim.x == 64), so wv_sync() is a wave-level
// barrier + LDS/global fence (hipcc drops the s_barrier for single-wave groups).
//
// Host build (SOLO_HOST_EMU): SX_NLANES == 1, the same source runs serially -- used only by the
// CPU-side tests to check the kernel source against the reference without a GPU.
#pragma once
#include "solo_fix.h"

#if defined(__HIP_DEVICE_COMPILE__) && defined(SX_GROUP)
// quantiser kernel: SEVERAL streams per wavefront, SX_GROUP (16 or 32) lanes each ("wave-uniform" then means uniform within
// the lane group; the hardware's exec masking serialises groups that take different branches)
#define SX_NLANES SX_GROUP
#define SX_LANE ((int)(threadIdx.x & (SX_GROUP - 1)))
#define SX_XOR_REDUCE(v, OP)                                              \
    _Pragma("unroll") for (int o_ = SX_GROUP / 2; o_ > 0; o_ >>= 1) { auto t_ = __shfl_xor(v, o_, SX_GROUP); v = OP; }
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(SX_FORCE_SERIAL)
#define SX_NLANES 64
#define SX_LANE ((int)(threadIdx.x & 63))
#define SX_XOR_REDUCE(v, OP)                                              \
    _Pragma("unroll") for (int o_ = 32; o_ > 0; o_ >>= 1) { auto t_ = __shfl_xor(v, o_, 64); v = OP; }
#else
#define SX_NLANES 1
#define SX_LANE 0
#define SX_XOR_REDUCE(v, OP)
#endif

// (under hipcc these are __host__ __device__ so that the host pass of a .hip file still parses
// kernels that call them; the host bodies are the 1-lane identities used by the emulation build)
SX_HD void wv_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#endif
}
// same, but only LDS traffic is waited for: for phases that talk through LDS / shuffles while global loads and stores stay in flight
SX_HD void wv_sync_lds() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
SX_HD i32 wv_sum(i32 v) { SX_XOR_REDUCE(v, sx_add(v, t_)) return v; }   // sum over the lanes, result in every lane
SX_HD i64 wv_sum64(i64 v) { SX_XOR_REDUCE(v, v + t_) return v; }
SX_HD i32 wv_max(i32 v) { SX_XOR_REDUCE(v, (t_ > v ? t_ : v)) return v; }
SX_HD i32 wv_min(i32 v) { SX_XOR_REDUCE(v, (t_ < v ? t_ : v)) return v; }
SX_HD i32 wv_bcast(i32 v, int src) {   // broadcast lane `src`'s value
#if defined(__HIP_DEVICE_COMPILE__)
    return __shfl(v, src, SX_NLANES);
#else
    (void)src;
    return v;
#endif
}
// (value, index) arg-min with "first index wins on ties" (matches a serial `<` scan)
SX_HD void wv_argmin(i32* v, i32* idx) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int o = SX_NLANES / 2; o > 0; o >>= 1) {
        i32 tv = __shfl_xor(*v, o, SX_NLANES), ti = __shfl_xor(*idx, o, SX_NLANES);
        if (tv < *v || (tv == *v && ti < *idx)) { *v = tv; *idx = ti; }
    }
#else
    (void)v; (void)idx;
#endif
}
// (value, index) arg-max with "first index wins on ties" (matches a serial `>` scan)
SX_HD void wv_argmax(i32* v, i32* idx) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int o = SX_NLANES / 2; o > 0; o >>= 1) {
        i32 tv = __shfl_xor(*v, o, SX_NLANES), ti = __shfl_xor(*idx, o, SX_NLANES);
        if (tv > *v || (tv == *v && ti < *idx)) { *v = tv; *idx = ti; }
    }
#else
    (void)v; (void)idx;
#endif
}

// lane-strided parallel loop
#define SX_PAR(i, n) for (int i = SX_LANE; i < (int)(n); i += SX_NLANES)

// wave-cooperative memcpy / memset / memmove helpers (element-wise, any POD type)
template <typename T>
SX_HD void wv_copy(T* dst, const T* src, int n) { SX_PAR(i, n) dst[i] = src[i]; }
template <typename T>
SX_HD void wv_fill(T* dst, T v, int n) { SX_PAR(i, n) dst[i] = v; }
// overlapping move towards LOWER addresses (dst < src): chunked so every lane reads before any lane
// of a later chunk writes.  Reads of chunk c happen-before writes of chunk c within the wave.
template <typename T>
SX_HD void wv_move_down(T* dst, const T* src, int n) {
    for (int base = 0; base < n; base += SX_NLANES) {
        int i = base + SX_LANE;
        T v = T();
        if (i < n) v = src[i];
        wv_sync();
        if (i < n) dst[i] = v;
        wv_sync();
    }
}
