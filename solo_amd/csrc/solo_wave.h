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
// quantiser kernel: SEVERAL streams per wavefront, SX_GROUP (16) lanes each ("wave-uniform" then means uniform within
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
#if defined(__HIP_DEVICE_COMPILE__) && defined(SX_SYNC_WAVE_ONLY)
    // workgroups of several wavefronts that each own their streams (solo_nsq_row.hip): what a single-wave __syncthreads() leaves once
    // the compiler has dropped its barrier -- all memory operations of the wavefront done, nothing moved across
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#elif defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#endif
}
// same, but only LDS traffic is waited for: for phases that talk through LDS / shuffles while global loads and stores stay in flight
SX_HD void wv_sync_lds() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
#if defined(__HIP_DEVICE_COMPILE__) && (SX_NLANES == 64 || SX_NLANES == 16)
// Reductions without the LDS crossbar: four DPP steps (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror) leave the
// result of each 16-lane row in all of its lanes; the four rows are then combined on the scalar unit (v_readlane), which also
// makes the result provably wave-uniform.  Wrapping adds / min / max are order-independent, so this equals the serial scan.
#define SX_DPP_(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
#define SX_ROW_REDUCE(v, OP)                                                                     \
    { i32 t_ = SX_DPP_(v, 0xB1); v = OP; } { i32 t_ = SX_DPP_(v, 0x4E); v = OP; }                \
    { i32 t_ = SX_DPP_(v, 0x141); v = OP; } { i32 t_ = SX_DPP_(v, 0x140); v = OP; }
#if SX_NLANES == 64
#define SX_ROWS_COMBINE(v, OP)                                                                   \
    { i32 a_ = __builtin_amdgcn_readlane(v, 0), b_ = __builtin_amdgcn_readlane(v, 16),           \
          c_ = __builtin_amdgcn_readlane(v, 32), d_ = __builtin_amdgcn_readlane(v, 48);          \
      { i32 t_ = b_; v = a_; a_ = OP; } { i32 t_ = d_; v = c_; c_ = OP; } { i32 t_ = c_; v = a_; v = OP; } }
#else
#define SX_ROWS_COMBINE(v, OP)
#endif
SX_HD i32 wv_row_sum(i32 v) { SX_ROW_REDUCE(v, sx_add(v, t_)) return v; }    // sum over each 16-lane row, in all of its lanes
#if SX_NLANES == 64
// sum over the four rows, column by column (lane j of every row gets v[j] + v[16 + j] + v[32 + j] + v[48 + j]): two gfx950 lane-swap
// instructions (v_permlane32_swap: upper half of one operand <-> lower half of the other; v_permlane16_swap: odd rows <-> even rows)
SX_HD i32 wv_col_sum(i32 v) {
    auto a = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    const i32 s = sx_add((i32)a[0], (i32)a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(s, s, false, false);
    return sx_add((i32)b[0], (i32)b[1]);
}
#endif
#if SX_NLANES == 64
// the four row results folded with the two row-broadcast DPP steps of gfx9 (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and
// 3: lane 63 then holds the wave's result) and ONE v_readlane -- 3 instructions less than four v_readlane + scalar / vector folding.
// (Written as assembly: the compiler does not fuse the masked-row form into the DPP operand of v_min / v_max / v_add.  The s_nop are
// the two wait states a DPP read needs after the VALU write of its source.)
#define SX_BCAST_FOLD(v, INSN)                                                                                           \
    asm volatile("s_nop 1\n\t" INSN " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"               \
                 INSN " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0" : "+v"(v));                       \
    v = __builtin_amdgcn_readlane(v, 63);
SX_HD i32 wv_sum(i32 v) { SX_ROW_REDUCE(v, sx_add(v, t_)) SX_BCAST_FOLD(v, "v_add_u32_dpp") return v; }
SX_HD i32 wv_max(i32 v) { SX_ROW_REDUCE(v, (t_ > v ? t_ : v)) SX_BCAST_FOLD(v, "v_max_i32_dpp") return v; }
SX_HD i32 wv_min(i32 v) { SX_ROW_REDUCE(v, (t_ < v ? t_ : v)) SX_BCAST_FOLD(v, "v_min_i32_dpp") return v; }
#else
SX_HD i32 wv_sum(i32 v) { SX_ROW_REDUCE(v, sx_add(v, t_)) SX_ROWS_COMBINE(v, sx_add(v, t_)) return v; }
SX_HD i32 wv_max(i32 v) { SX_ROW_REDUCE(v, (t_ > v ? t_ : v)) SX_ROWS_COMBINE(v, (t_ > v ? t_ : v)) return v; }
SX_HD i32 wv_min(i32 v) { SX_ROW_REDUCE(v, (t_ < v ? t_ : v)) SX_ROWS_COMBINE(v, (t_ < v ? t_ : v)) return v; }
#endif
SX_HD i64 wv_sum64(i64 v) {
    u32 lo = (u32)v, hi = (u32)((u64)v >> 32);
#pragma unroll
    for (int s_ = 0; s_ < 4; s_++) {
        u32 tl, th;
        switch (s_) {           // the DPP control is an immediate
            case 0: tl = SX_DPP_(lo, 0xB1); th = SX_DPP_(hi, 0xB1); break;
            case 1: tl = SX_DPP_(lo, 0x4E); th = SX_DPP_(hi, 0x4E); break;
            case 2: tl = SX_DPP_(lo, 0x141); th = SX_DPP_(hi, 0x141); break;
            default: tl = SX_DPP_(lo, 0x140); th = SX_DPP_(hi, 0x140); break;
        }
        u64 r = (((u64)hi << 32) | lo) + (((u64)th << 32) | tl);
        lo = (u32)r; hi = (u32)(r >> 32);
    }
    u64 r = ((u64)hi << 32) | lo;
#if SX_NLANES == 64
    u64 acc = 0;
#pragma unroll
    for (int row = 0; row < 4; row++)
        acc += ((u64)(u32)__builtin_amdgcn_readlane((i32)hi, row * 16) << 32) | (u32)__builtin_amdgcn_readlane((i32)lo, row * 16);
    r = acc;
#endif
    return (i64)r;
}
#else
SX_HD i32 wv_sum(i32 v) { SX_XOR_REDUCE(v, sx_add(v, t_)) return v; }   // sum over the lanes, result in every lane
SX_HD i64 wv_sum64(i64 v) { SX_XOR_REDUCE(v, v + t_) return v; }
SX_HD i32 wv_max(i32 v) { SX_XOR_REDUCE(v, (t_ > v ? t_ : v)) return v; }
SX_HD i32 wv_min(i32 v) { SX_XOR_REDUCE(v, (t_ < v ? t_ : v)) return v; }
#endif
SX_HD i32 wv_bcast(i32 v, int src) {   // broadcast lane `src`'s value
#if defined(__HIP_DEVICE_COMPILE__)
    return __shfl(v, src, SX_NLANES);
#else
    (void)src;
    return v;
#endif
}
// (value, index) arg-min with "first index wins on ties" (matches a serial `<` scan)
#if defined(__HIP_DEVICE_COMPILE__) && (SX_NLANES == 64 || SX_NLANES == 16)
#define SX_ARG_STEP(CTRL, CMP)                                                                   \
    { i32 tv = SX_DPP_(bv, CTRL), ti = SX_DPP_(bi, CTRL);                                        \
      const bool take_ = (tv CMP bv) | ((tv == bv) & (ti < bi));                                 \
      bv = take_ ? tv : bv; bi = take_ ? ti : bi; }
#if SX_NLANES == 64
#define SX_ARG_ROWS(CMP)                                                                         \
    { i32 rv = __builtin_amdgcn_readlane(bv, 0), ri = __builtin_amdgcn_readlane(bi, 0);          \
      _Pragma("unroll") for (int row = 1; row < 4; row++) {                                      \
          i32 tv = __builtin_amdgcn_readlane(bv, row * 16), ti = __builtin_amdgcn_readlane(bi, row * 16); \
          const bool take_ = (tv CMP rv) | ((tv == rv) & (ti < ri));                             \
          rv = take_ ? tv : rv; ri = take_ ? ti : ri; }                                          \
      bv = rv; bi = ri; }
#else
#define SX_ARG_ROWS(CMP)
#endif
SX_HD void wv_argmin(i32* v, i32* idx) {
    i32 bv = *v, bi = *idx;
    SX_ARG_STEP(0xB1, <) SX_ARG_STEP(0x4E, <) SX_ARG_STEP(0x141, <) SX_ARG_STEP(0x140, <) SX_ARG_ROWS(<)
    *v = bv; *idx = bi;
}
SX_HD void wv_argmax(i32* v, i32* idx) {
    i32 bv = *v, bi = *idx;
    SX_ARG_STEP(0xB1, >) SX_ARG_STEP(0x4E, >) SX_ARG_STEP(0x141, >) SX_ARG_STEP(0x140, >) SX_ARG_ROWS(>)
    *v = bv; *idx = bi;
}
#else
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

#endif

// SX_UNI(x): x is known to be identical in all lanes of a fully active wave (wave-uniform code outside SX_PAR bodies); moving
// it to an SGPR lets the serial scalar recursions that consume it run on the scalar unit instead of 64 redundant VALU lanes
#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64
#define SX_UNI(x) ((i32)__builtin_amdgcn_readfirstlane((i32)(x)))
#else
#define SX_UNI(x) ((i32)(x))
#endif

// a value as a type: argument of a generic lambda whose body is compiled once per value (a loop specialised on a wave-uniform regime)
template <typename T, T V>
struct SxConst { static constexpr T value = V; };

// SX_VPTR(p): the same pointer, but opaque to the compiler's uniformity analysis (an offset of zero that lives in a vector
// register).  Wave-uniform straight-line arithmetic on values loaded through it is emitted for the VECTOR unit instead of the
// scalar unit: one wave pays one issue slot per instruction either way, but the scalar unit takes one instruction per four cycles
// per SIMD for all of its waves together, the vector unit two (tools/debug/mb_issue.hip).
#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64
SX_HD int sx_vzero() { int z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); return z; }
#define SX_VPTR(p) ((p) + sx_vzero())
// pins a value to a vector register and hides its uniformity from the compiler (see solo_enc_front.h: recursions on the vector unit)
#define SX_VEC(x) asm volatile("" : "+v"(x))
#else
#define SX_VPTR(p) (p)
#define SX_VEC(x)
#endif

// Serial recursions over a block of samples (IIR sections that cannot be re-cut over the lanes) read and write their samples
// through lane registers instead of LDS: sample i lives in lane (i & 63) of register (i >> 6); the scalar loop fetches it with
// v_readlane and deposits results with a lane-select, so no LDS round trip sits on the recursion's critical path.
#if defined(__HIP_DEVICE_COMPILE__) && SX_NLANES == 64
#define SX_LANE_STREAM 1
#define SX_RDLANE(v, i) ((i32)__builtin_amdgcn_readlane((i32)(v), (i)))
#define SX_WRLANE(v, i, val) (v) = (SX_LANE == (i)) ? (i32)(val) : (v)
#endif

// Hand-over INSIDE a launch (the persistent encoder pipeline, solo_enc_kernels.h / solo_nsq_row.hip): a record one workgroup writes and
// another reads while both kernels run.  gfx950 has eight XCDs with private L2s and a vector L1 per compute unit that other units'
// stores never refresh, so (guide: "publish / consume"): the PAYLOAD is stored write-through (sc1: sx_pub_st, <= 8 bytes a store), every
// storing wave drains its stores (sx_pub_drain), ONE lane stores the flag (sx_flag_st); the consumer polls the flag relaxed
// (sx_flag_ld), then reads the payload with sc1 loads (sx_pub_ld: served by the L2 / memory, never by the unit's L1, which may hold a
// stale line -- neighbouring records share cache lines).  No release fence (a buffer_wbl2 per hand-over would sweep the XCD's L2
// thousands of times a millisecond) and NO acquire fence: buffer_inv sc1 drops the whole L1 of the compute unit, and with twenty
// wavefronts per unit doing that once per packet the analysis chains' table reads all went to the L2 (measured: + 0.24 ms per packet,
// 65 instead of 53 ms per 4096 x 50).  The host emulation: plain stores and loads.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) unsigned int sx_gu32;
template <typename T, typename V>
__device__ __forceinline__ void sx_pub_st(T* p, V v) { __hip_atomic_store(p, (T)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sx_pub_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void sx_flag_st(unsigned int* p, unsigned int v) { __hip_atomic_store((sx_gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned int sx_flag_ld(const unsigned int* p) { return __hip_atomic_load((sx_gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T sx_pub_ld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
template <typename T, typename V>
SX_HD void sx_pub_st(T* p, V v) { *p = (T)v; }
SX_HD void sx_pub_drain() {}
SX_HD void sx_flag_st(unsigned int* p, unsigned int v) { *p = v; }
SX_HD unsigned int sx_flag_ld(const unsigned int* p) { return *p; }
template <typename T>
SX_HD T sx_pub_ld(const T* p) { return *p; }
#endif

// lane-strided parallel loop
#define SX_PAR(i, n) for (int i = SX_LANE; i < (int)(n); i += SX_NLANES)
// ... over a serial LCG sequence: sx_lcg_first(seed) = iterate SX_LANE + 1 of sx_rand, sx_lcg_next(v) = SX_NLANES iterates further
SX_HD i32 sx_lcg_first(i32 seed) { return sx_rand_skip_bits<7>(seed, (u32)SX_LANE + 1u); }
SX_HD i32 sx_lcg_next(i32 v) { constexpr SxLcgMap m = sx_lcg_map(SX_NLANES); return (i32)(m.A * (u32)v + m.C); }

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

// Issue priority of the wave inside its SIMD.  The arbiter serves the waves of a SIMD in a fixed order: a wave that runs a dense chain
// of dependent vector instructions has one ready almost every time the vector unit frees up and keeps the unit; the occasional vector
// instruction of a wave in a latency-bound stretch (address arithmetic between LDS / memory round trips, lane-serial recursions) then
// waits for the dense stretch to END, so that one wave's waiting time does not hide the others' arithmetic (tools/debug/mb_overlap.hip:
// two waves alternating 1200 cycles of dependent v_mul_hi / v_add with 1200 cycles of dependent LDS reads take 3350 cycles per round,
// 2460 when the LDS stretch raises its priority).  The stages of the analysis chain therefore say which kind they are.
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef SX_PRIO_LAT
#define SX_PRIO_LAT 3
#endif
#ifndef SX_PRIO_DENSE
#define SX_PRIO_DENSE 3
#endif
#define SX_STRETCH_LATENCY() __builtin_amdgcn_s_setprio(SX_PRIO_LAT)
#define SX_STRETCH_DENSE() __builtin_amdgcn_s_setprio(SX_PRIO_DENSE)
#else
#define SX_STRETCH_LATENCY()
#define SX_STRETCH_DENSE()
#endif

// optional section timer (debug builds with -DSX_PROF: cycles per section accumulated into a device array), or section STOPS (debug
// builds with -DSX_STOPS, tools/debug/analysis_sections.py: a wave ends -- s_endpgm, nothing saved -- when it reaches site `id` for
// the hit-th time of its launch, g_sx_stop = id << 8 | hit; the SQ instruction counters of launches stopped at successive sites,
// subtracted, are the wave-instructions of each section.  g_sx_stop = 0: the launch runs through and counts the sites it passes)
#if defined(SX_PROF) && defined(__HIPCC__)
static __device__ unsigned long long g_sx_prof[64];
static __device__ unsigned long long g_sx_hist[4][64];
#endif
#if defined(SX_STOPS) && defined(__HIPCC__)
static __device__ int g_sx_stop;                          // kernel tag << 16 | site << 8 | hit (0: run through and count)
static __device__ unsigned long long g_sx_site_hits[2][64];
#endif
#if defined(SX_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define SX_T_BEGIN unsigned long long sx_t_last_ = __builtin_readcyclecounter();
#define SX_T_RESET sx_t_last_ = __builtin_readcyclecounter();
#define SX_T(id) { unsigned long long t_ = __builtin_readcyclecounter(); if (SX_LANE == 0) atomicAdd(&g_sx_prof[id], t_ - sx_t_last_); sx_t_last_ = __builtin_readcyclecounter(); }
#elif defined(SX_STOPS) && defined(__HIP_DEVICE_COMPILE__)
static __shared__ unsigned char sx_site_hits_[64];       // (one stream per workgroup; cleared by the kernel wrapper; [63] = the kernel's tag:
#define SX_T_BEGIN                                       //  stages shared by two kernels -- Burg, A2NLSF -- stop in the tagged one only)
#define SX_T_RESET
#define SX_STOPS_ENTER(tag_) { sx_site_hits_[threadIdx.x & 63] = (threadIdx.x & 63) == 63 ? (unsigned char)(tag_) : (unsigned char)0; }
#define SX_T(id) { const int h_ = sx_site_hits_[id] + 1; const int stop_ = g_sx_stop; const int tag_ = sx_site_hits_[63]; sx_site_hits_[id] = (unsigned char)h_; \
        if (stop_ == 0) { if (SX_LANE == 0) atomicAdd(&g_sx_site_hits[tag_ & 1][id], 1ull); }                                                 \
        else if (stop_ == ((tag_ << 16) | ((id) << 8) | h_)) __builtin_amdgcn_endpgm(); }
#else
#define SX_T_BEGIN
#define SX_T_RESET
#define SX_T(id)
#endif
// (sites that only the stop builds know: finer than the timer's sections)
#if defined(SX_STOPS) && defined(__HIP_DEVICE_COMPILE__)
#define SX_S(id) SX_T(id)
#else
#define SX_S(id)
#endif
