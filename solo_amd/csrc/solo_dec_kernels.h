// solo_dec_kernels.h -- the decoder's kernels and their launchers, compiled once per internal rate: solo_api.hip includes it
// with SX_FS_KHZ = 8 (16 kHz API rate), solo_api_wb.hip with SX_FS_KHZ = 16 (32 kHz API rate); SX_K() keeps the symbols apart.
#pragma once
#include <hip/hip_runtime.h>
#include "solo_dec.h"
#include "solo_recv.h"

__global__ void __launch_bounds__(64) SX_K(solo_dec_init_kernel)(SxDecStream* states, int n_streams, int hb_mode) {
    const int s = blockIdx.x;
    if (s >= n_streams) return;
    sx_dec_state_init(&states[s].st, hb_mode);
    u32* sh = (u32*)&states[s].sh;
    SX_PAR(i, (int)(sizeof(SxDecShadow) / 4)) sh[i] = 0;
}

// Decoder: rows D0-D8.  blockIdx.x = stream.
// state record HBM <-> LDS (whole launch) and the entropy tables
// ONE work area for all decoder kernels of the build (a file-scope LDS variable): the stage functions are real calls, and a function whose
// callers all pass the address of the same variable is compiled with that address as a constant -- LDS offsets become immediates.  With a
// __shared__ variable per kernel the five kernels passed five different variables and the functions they share got a run-time base.
static __shared__ SxDecWork SX_K(g_sx_dec_work);
__device__ __forceinline__ void SX_K(solo_dec_enter)(SxDecWork* w, SxDecStream* rec) {
    const i32* src = (const i32*)&rec->st;
    i32* dst = (i32*)&w->st;
    SX_PAR(i, (int)(sizeof(SxDecState) / 4)) dst[i] = src[i];
    w->shadow = &rec->sh;
    sx_cdf_load_dec(&w->cdf);
    wv_sync();
}
__device__ __forceinline__ void SX_K(solo_dec_leave)(SxDecWork* w, SxDecStream* rec) {
    wv_sync();
    const i32* src = (const i32*)&w->st;
    i32* dst = (i32*)&rec->st;
    SX_PAR(i, (int)(sizeof(SxDecState) / 4)) dst[i] = src[i];
}

// One (bits, nbytes, recv) record of the batched API -> the reference's (pointer, nBytes[2], lostflag) calling convention.
struct SxDecArgs { int lostflag, bad; i32 a0, a1, ptr_off; };
SX_HD SxDecArgs sx_dec_map_record(i32 n0, i32 n1, int slot, int recv_mask, int hb_joint) {
    // The lengths come from the network: a record that does not describe two descriptions inside its own slot
    // (0 <= len(MD2)+HB <= total <= slot, MD2 carrying at least its high-band bytes) is never dereferenced; the packet is
    // concealed as lost and the stream's status reports SKP_SILK_DEC_PAYLOAD_TOO_LARGE (-11) / _PAYLOAD_ERROR (-12).
    const int hbb = hb_joint ? SX_HB_BYTES / 2 : SX_HB_BYTES;
    SxDecArgs r;
    r.bad = 0;
    if (n0 > slot) r.bad = -11;
    else if (n0 > 0 && (n1 < 0 || n1 > n0 || (n1 > 0 && n1 < hbb))) r.bad = -12;
    // a record that is shorter than the high-band bytes every packet ends with cannot hold a first description + high band (recv
    // bit 0) -- the reference would compute a negative low-band length from it (AGR_BWE_decode_frame_FIX.c:150-169)
    else if (n0 > 0 && n0 < hbb && (recv_mask & 1)) r.bad = -12;
    if (r.bad) { n0 = 0; n1 = 0; }
    // DELIBERATE CONVENTION of the batched API (not reference behaviour): an EMPTY record (n0 <= 0, e.g. a DTX packet that
    // was never sent) is concealed like a lost packet, i.e. decoded with lostflag = 1.  The reference library itself returns
    // -1 for nBytes[0] <= 0 without touching its state (AGR_BWE_SDK_API.c:266) and its CLI then writes the previous output
    // buffer again (test/dec_main.c:365-381); the legacy AGR_Sate_Decoder_Decode symbol of this library keeps that behaviour.
    const int m = n0 <= 0 ? 0 : (recv_mask & 3);
    // receiver-side mapping of the reference harness (test/dec_main.c:255-378)
    r.ptr_off = 0;
    if (m == 3) { r.lostflag = 4; r.a0 = n0; r.a1 = n1; }
    else if (m == 1) { r.lostflag = 2; r.a0 = n0 - n1; r.a1 = 0; }
    else if (m == 2) { r.lostflag = 3; r.ptr_off = n0 - n1; r.a0 = n1; r.a1 = 0; }
    else { r.lostflag = 1; r.a0 = n0 > 0 ? n0 : 16; r.a1 = n0 > 0 ? n1 : 0; }
    if (r.lostflag == 2 && r.a0 <= 0) { r.lostflag = 1; r.a0 = 16; r.a1 = 0; r.ptr_off = 0; }          // nothing but the second description in the record
    if (r.lostflag == 3 && r.a0 <= hbb) { r.lostflag = 1; r.a0 = 16; r.a1 = 0; r.ptr_off = 0; }
    return r;
}

__global__ void __launch_bounds__(64, 4) SX_K(solo_decode_kernel)(SxDecStream* states, const u8* __restrict__ bits,
                                                         const i16* __restrict__ nbytes, const u8* __restrict__ recv,
                                                         int n_streams, int n_packets, int slot, int useMDIndex,
                                                         i16* __restrict__ pcm, i32* status) {
    SxDecWork& w = SX_K(g_sx_dec_work);
    const int s = blockIdx.x;
    if (s >= n_streams) return;
    SX_K(solo_dec_enter)(&w, &states[s]);
    i32 first_err = 0;
    for (int p = 0; p < n_packets; p++) {
        const size_t pk = (size_t)s * n_packets + p;
        const SxDecArgs a = sx_dec_map_record(nbytes[pk * 2 + 0], nbytes[pk * 2 + 1], slot, recv ? (int)recv[pk] : 3, w.st.hb_joint | (w.st.fpp == 1));
        const u8* ptr = bits + pk * (size_t)slot + a.ptr_off;
        const int lostflag = a.lostflag, bad = a.bad;
        const i32 a0 = a.a0, a1 = a.a1;
        i16* out = pcm + pk * (size_t)(SX_FRAME * 2 * SX_UNI(w.st.fpp));
        int ret = sx_decode_packet(&w, ptr, a0, a1, lostflag, useMDIndex, out);
        if (ret == 0 && bad) ret = bad;
        if (ret < 0 && first_err == 0) first_err = ret;
        wv_sync();
    }
    SX_K(solo_dec_leave)(&w, &states[s]);
    if (status && SX_LANE == 0) status[s] = first_err;
}

// ---- batch path in two kernels -----------------------------------------------------------------------------------------------
// Reading the symbols off the range coder is a serial chain per description: in the single kernel above it keeps 2 of 64 lanes
// busy for ~45 % of the decoder's time.  But which symbol comes next depends only on symbols of the same packet, never on the
// decoder's history, so the batch path reads ALL descriptions of ALL packets of a chunk at once, one LANE each
// (solo_dec_extract_kernel, sx_extract_desc), and the decoder proper (one wavefront per stream, packets in order) starts from the
// records: de-quantisation, synthesis, concealment, high band, QMF.  A record is only used for an ordinary packet
// (sx_extracted_usable); anything else -- coder errors, a packet that announces more frames than it carries, symbols that depend
// on the bytes behind the description -- is decoded serially as in the single kernel, with the exact history.
#define SX_EXTRACT_LANES (SX_FS_KHZ == 8 ? 64 : 32)        // lanes per workgroup (LDS: ~0.5 KB per lane at 8 kHz, ~0.9 KB at 16 kHz)
#ifndef SX_EXTRACT_WAVES
#define SX_EXTRACT_WAVES 4                                 // waves per SIMD the extraction kernel is compiled for (its register budget)
#endif
struct SxExtractWork {
    SxCdfDec cdf;
    SxExtractLane lane[SX_EXTRACT_LANES];
};
// recs: [(stream * pc + (p - p0)) * 2 + slot]; one lane per record.
// Which records there is something to read for is decided first (solo_dec_list_kernel): the description slots that carry bytes are
// numbered densely -- a wavefront's count by ballot, its place in the list by one atomic add --, the others get their `usable = 0` there.
// The extraction kernel's lane i then takes list entry i: with descriptions lost on the way (BASELINE configs[3]: 30 % of them) the
// wavefronts of 64 lock-stepped coders are all full and fewer, instead of each running with the lost slots' lanes switched off.
// (The order of the list depends on the order the wavefronts' atomic adds arrive in; every entry is extracted into its own record
// whatever lane takes it.)
static __device__ __forceinline__ bool SX_K(sx_extract_slot)(const SxDecStream* states, const i16* __restrict__ nbytes, const u8* __restrict__ recv, size_t idx, int n_packets,
                                                            int p0, int pc, int slot, size_t* pk_out, int* hb_joint_out, SxDecArgs* a_out, i32* off, i32* len, int* sel,
                                                            i32* hb_off) {
    const int md = (int)(idx & 1);
    const size_t sp = idx >> 1;
    const int s = (int)(sp / (size_t)pc), p = p0 + (int)(sp % (size_t)pc);
    const size_t pk = (size_t)s * n_packets + p;
    const int hb_joint = states[s].st.hb_joint | (states[s].st.fpp == 1);      // (what matters here: four high-band bytes instead of eight)
    const SxDecArgs a = sx_dec_map_record(nbytes[pk * 2 + 0], nbytes[pk * 2 + 1], slot, recv ? (int)recv[pk] : 3, hb_joint);
    *pk_out = pk; *hb_joint_out = hb_joint; *a_out = a;
    *off = 0; *len = 0; *hb_off = -1; *sel = 0;
    return sx_desc_span(a.lostflag, a.a0, a.a1, hb_joint, md, off, len, sel, hb_off);
}
// list: n_streams * pc * 2 entries, count: one word behind them, zeroed before the launch
__global__ void __launch_bounds__(64) SX_K(solo_dec_list_kernel)(const SxDecStream* states, const i16* __restrict__ nbytes, const u8* __restrict__ recv, int n_streams,
                                                                int n_packets, int p0, int pc, int slot, SxExtracted* __restrict__ recs, u32* __restrict__ list,
                                                                u32* __restrict__ count) {
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    bool present = false;
    if (idx < (size_t)n_streams * (size_t)pc * 2) {
        size_t pk; int hb_joint, sel; SxDecArgs a; i32 off, len, hb_off;
        present = SX_K(sx_extract_slot)(states, nbytes, recv, idx, n_packets, p0, pc, slot, &pk, &hb_joint, &a, &off, &len, &sel, &hb_off);
        if (!present) recs[idx].usable = 0;
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(present);
    if (m == 0) return;
    u32 base = 0;
    if (threadIdx.x == 0) base = atomicAdd(count, (u32)__builtin_popcountll(m));
    base = (u32)__builtin_amdgcn_readfirstlane((int)base);
    if (present) list[base + (u32)__builtin_popcountll(m & ((1ull << threadIdx.x) - 1ull))] = (u32)idx;
}
__global__ void __launch_bounds__(SX_EXTRACT_LANES, SX_EXTRACT_WAVES) SX_K(solo_dec_extract_kernel)(const SxDecStream* states, const u8* __restrict__ bits,
                                                                                 const i16* __restrict__ nbytes, const u8* __restrict__ recv,
                                                                                 int n_streams, int n_packets, int p0, int pc, int slot,
                                                                                 int useMDIndex, SxExtracted* __restrict__ recs, const u32* __restrict__ list,
                                                                                 const u32* __restrict__ count) {
    __shared__ SxExtractWork w;
    // (without a list -- the caller passed no reception flags, so nearly every slot carries bytes -- lane i takes slot i)
    const size_t n_listed = list ? (size_t)*count : (size_t)n_streams * (size_t)pc * 2;
    if ((size_t)blockIdx.x * SX_EXTRACT_LANES >= n_listed) return;                  // (the grid is sized for "every slot carries bytes")
    {
        SxCdfDec* c = &w.cdf;
#define X(type, name, n) for (int i = threadIdx.x; i < (n); i += SX_EXTRACT_LANES) c->name[i] = T_##name[i];
        SX_CDF_LIST(X)
#undef X
    }
    __syncthreads();
    const size_t li = (size_t)blockIdx.x * SX_EXTRACT_LANES + threadIdx.x;
    if (li >= n_listed) return;
    const size_t idx = list ? (size_t)list[li] : li;
    size_t pk; int hb_joint, sel; SxDecArgs a; i32 off, len, hb_off;
    SxExtracted* rec = &recs[idx];
    if (!SX_K(sx_extract_slot)(states, nbytes, recv, idx, n_packets, p0, pc, slot, &pk, &hb_joint, &a, &off, &len, &sel, &hb_off)) { rec->usable = 0; return; }
    const u8* pkt = bits + pk * (size_t)slot + a.ptr_off;
    sx_extract_desc(pkt + off, len, useMDIndex, (const SxCdf*)&w.cdf, &w.lane[threadIdx.x], rec, sel, hb_off >= 0 ? pkt + hb_off : 0, hb_joint);
}

#if !defined(SX_DEC_NO_PREFETCH) && defined(__HIP_DEVICE_COMPILE__)
// The records of a packet (2 x sizeof(SxExtracted), written by the extraction kernel: 0.45 GB per 4096 streams x 50 packets, so they
// come from HBM) are read in several dependent batches while the packet is decoded.  The wavefront therefore touches every 128-byte
// line of the NEXT packet's records when it starts a packet: one dword per line through the LDS-DMA path (global_load_lds), whose
// destination is a strip of LDS nobody reads -- no register waits for the data, the loads a packet later find their lines in L2
// (5.43 -> 5.30 ms per 4096 x 50 packets; profiles/r06_decoder_prefetch.txt).
#define SX_PF_LINES ((int)((2 * sizeof(SxExtracted) + 127 + 127) / 128))
static_assert(SX_PF_LINES <= 64, "one line per lane");
static __shared__ u32 SX_K(g_sx_dec_pf)[SX_PF_LINES];
static __device__ __forceinline__ void sx_prefetch_records(const SxExtracted* next2, u32* sink) {
    const unsigned long long a = (unsigned long long)next2, mine = (a & ~127ull) + (unsigned long long)SX_LANE * 128ull;
    if (mine < a + 2 * sizeof(SxExtracted))
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)mine, (__attribute__((address_space(3))) void*)sink, 4, 0, 0);
}
#endif

__global__ void __launch_bounds__(64, 4) SX_K(solo_dec_synth_kernel)(SxDecStream* states, const u8* __restrict__ bits,
                                                            const i16* __restrict__ nbytes, const u8* __restrict__ recv,
                                                            int n_streams, int n_packets, int p0, int pc, int slot, int useMDIndex,
                                                            const SxExtracted* __restrict__ recs, i16* __restrict__ pcm, i32* status) {
    SxDecWork& w = SX_K(g_sx_dec_work);
    const int s = blockIdx.x;
    if (s >= n_streams) return;
#if defined(SX_STOPS) && defined(__HIP_DEVICE_COMPILE__)
    SX_STOPS_ENTER(0)
#endif
    SX_K(solo_dec_enter)(&w, &states[s]);
    i32 first_err = 0;
    for (int p = p0; p < p0 + pc; p++) {
        const size_t pk = (size_t)s * n_packets + p;
#if !defined(SX_DEC_NO_PREFETCH) && defined(__HIP_DEVICE_COMPILE__)
        if (p + 1 < p0 + pc) sx_prefetch_records(recs + ((size_t)s * pc + (size_t)(p + 1 - p0)) * 2, SX_K(g_sx_dec_pf));
#endif
        const SxDecArgs a = sx_dec_map_record(nbytes[pk * 2 + 0], nbytes[pk * 2 + 1], slot, recv ? (int)recv[pk] : 3, w.st.hb_joint | (w.st.fpp == 1));
        int ret = sx_decode_packet(&w, bits + pk * (size_t)slot + a.ptr_off, a.a0, a.a1, a.lostflag, useMDIndex, pcm + pk * (size_t)(SX_FRAME * 2 * SX_UNI(w.st.fpp)),
                                   recs + ((size_t)s * pc + (size_t)(p - p0)) * 2);
        if (ret == 0 && a.bad) ret = a.bad;
        if (ret < 0 && first_err == 0) first_err = ret;
        wv_sync();
    }
    SX_K(solo_dec_leave)(&w, &states[s]);
    if (status && SX_LANE == 0) {
        if (p0 == 0) status[s] = first_err;
        else if (first_err != 0 && status[s] == 0) status[s] = first_err;
    }
}

// Receiver front end: the two descriptions of a 40 ms packet arrive as separate network packets (MD1, and
// MD2 || HB), possibly only one of them, possibly swapped.  descA / descB are the two arrival slots of every (stream, packet):
// uint8 [N][P][slot], lenA / lenB int16 [N][P] (0 = nothing arrived).  With useMDIndex = 1 every description carries its index
// as its first range-coded symbol (SKP_Silk_decode_parameters.c:55-57): the kernel reads it and sorts the arrivals itself
// (two copies of the same description count once); with useMDIndex = 0 slot A is MD1 and slot B is MD2 || HB.  The kernel then
// builds the (ptr, nBytes, lostflag) triple of test/dec_main.c:255-378 and decodes.  Packets above the LDS staging size
// (252 B; 13.6 kbps packets are ~80 B) are rejected with SKP_SILK_DEC_PAYLOAD_TOO_LARGE (-11).
// one packet of the receiver front end: what lies in the two arrival slots -> (ptr, nBytes, lostflag) -> the decoder
__device__ __forceinline__ int SX_K(sx_decode_split_packet)(SxDecWork& w, const u8* pa, i32 la, const u8* pb, i32 lb, int slot, int useMDIndex, i16* out) {
    if (la < 0 || la > slot) la = 0;
    if (lb < 0 || lb > slot) lb = 0;
    const u8 *p1 = pa, *p2 = pb;
    i32 l1 = la, l2 = lb;
    if (useMDIndex == 1) {
        int ia = -1, ib = -1;
        if (la > 0) { SxRangeDec r; r.error = 0; r.tail = 0; sx_rc_dec_init(&r, pa, sx_min(la, SX_MAX_ARITHM_BYTES)); ia = sx_rc_dec(&r, w.cdf.cdf_mdindex, T_CDF_MID_MDINDEX); if (r.error) ia = -1; }
        if (lb > 0) { SxRangeDec r; r.error = 0; r.tail = 0; sx_rc_dec_init(&r, pb, sx_min(lb, SX_MAX_ARITHM_BYTES)); ib = sx_rc_dec(&r, w.cdf.cdf_mdindex, T_CDF_MID_MDINDEX); if (r.error) ib = -1; }
        p1 = pa; l1 = 0; p2 = pb; l2 = 0;
        if (ia == 0) { p1 = pa; l1 = la; } else if (ib == 0) { p1 = pb; l1 = lb; }
        if (ia == 1) { p2 = pa; l2 = la; } else if (ib == 1) { p2 = pb; l2 = lb; }
    }
    const int hbb = (w.st.hb_joint | (w.st.fpp == 1)) ? SX_HB_BYTES / 2 : SX_HB_BYTES;
    if (l2 > 0 && l2 <= hbb) l2 = 0;                    // a second description always carries the high-band bytes
    if (l1 + l2 > SX_DEC_PAYLOAD_LDS) return -11;
    wv_sync();
    SX_PAR(i, l1) w.payload[i] = p1[i];
    SX_PAR(i, l2) w.payload[l1 + i] = p2[i];
    wv_sync();
    int lostflag;
    i32 a0, a1;
    if (l1 > 0 && l2 > 0) { lostflag = 4; a0 = l1 + l2; a1 = l2; }
    else if (l1 > 0) { lostflag = 2; a0 = l1; a1 = 0; }
    else if (l2 > 0) { lostflag = 3; a0 = l2; a1 = 0; }
    else { lostflag = 1; a0 = hbb + 1; a1 = 0; }
    return sx_decode_packet(&w, w.payload, a0, a1, lostflag, useMDIndex, out);
}

__global__ void __launch_bounds__(64, 4) SX_K(solo_decode_split_kernel)(SxDecStream* states, const u8* __restrict__ descA, const i16* __restrict__ lenA,
                                                               const u8* __restrict__ descB, const i16* __restrict__ lenB, int n_streams,
                                                               int n_packets, int slot, int useMDIndex, i16* __restrict__ pcm, i32* status) {
    SxDecWork& w = SX_K(g_sx_dec_work);
    const int s = blockIdx.x;
    if (s >= n_streams) return;
    SX_K(solo_dec_enter)(&w, &states[s]);
    i32 first_err = 0;
    for (int p = 0; p < n_packets; p++) {
        const size_t pk = (size_t)s * n_packets + p;
        const int ret = SX_K(sx_decode_split_packet)(w, descA + pk * (size_t)slot, lenA[pk], descB + pk * (size_t)slot, lenB[pk], slot, useMDIndex,
                                                     pcm + pk * (size_t)(SX_FRAME * 2 * SX_UNI(w.st.fpp)));
        if (ret < 0 && first_err == 0) first_err = ret;
        wv_sync();
    }
    SX_K(solo_dec_leave)(&w, &states[s]);
    if (status && SX_LANE == 0) status[s] = first_err;
}

// ---- receiver staging ring (solo_recv.h): arrivals filed by sequence number, decoded when their turn comes -----------------------
// one wavefront files one arrival
__global__ void __launch_bounds__(64) SX_K(solo_recv_insert_kernel)(const SxRecvArrival* __restrict__ arr, int n_arr, const u8* __restrict__ payload,
                                                               long long payload_bytes, int n_streams, int depth, int slot, int useMDIndex, u8* ring,
                                                               u32* lens, const i32* __restrict__ play, u32* stats) {
    const int a = blockIdx.x;
    if (a >= n_arr) return;
    const SxRecvArrival r = arr[a];
    int sl = -1;
    int verdict = sx_recv_file(&r, payload, payload_bytes, n_streams, depth, slot, useMDIndex, play, lens, SX_LANE == 0, &sl);
    if (verdict == SX_RECV_INSERTED) {
        sl = wv_bcast(sl, 0);
        if (sl < 0) verdict = SX_RECV_DUP;
    }
    const size_t e = sl >= 0 ? sx_recv_entry(r.stream, r.seq, depth) : 0;
    if (SX_LANE == 0) atomicAdd(&stats[verdict], 1u);
    if (sl >= 0) {
        u8* dst = ring + (e * 2 + (size_t)sl) * (size_t)slot;
        const u8* src = payload + r.offset;
        SX_PAR(i, r.len) dst[i] = src[i];
    }
}
__global__ void __launch_bounds__(64) SX_K(solo_recv_reset_kernel)(u32* lens, i32* play, u32* stats, int n_streams, int depth, i32 first_seq) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i < (size_t)n_streams * (size_t)depth) lens[i] = 0;
    if (i < (size_t)n_streams) play[i] = first_seq;
    if (i < SX_RECV_NSTATS) stats[i] = 0;
}
// the next n_packets sequence numbers of every stream: merge what has arrived (as the split kernel does), decode, free the entries
__global__ void __launch_bounds__(64, 4) SX_K(solo_decode_ring_kernel)(SxDecStream* states, const u8* ring, u32* lens, i32* play, int n_streams, int n_packets,
                                                              int depth, int slot, int useMDIndex, i16* __restrict__ pcm, i32* status) {
    SxDecWork& w = SX_K(g_sx_dec_work);
    const int s = blockIdx.x;
    if (s >= n_streams) return;
    SX_K(solo_dec_enter)(&w, &states[s]);
    i32 first_err = 0;
    const i32 play0 = play[s];
    for (int p = 0; p < n_packets; p++) {
        const size_t e = sx_recv_entry(s, play0 + p, depth);
        const u32 lw = lens[e];
        const u8* pa = ring + e * 2 * (size_t)slot;
        const int ret = SX_K(sx_decode_split_packet)(w, pa, (i32)(lw & 0xFFFFu), pa + slot, (i32)(lw >> 16), slot, useMDIndex,
                                                     pcm + ((size_t)s * n_packets + p) * (size_t)(SX_FRAME * 2 * SX_UNI(w.st.fpp)));
        if (ret < 0 && first_err == 0) first_err = ret;
        wv_sync();
        if (SX_LANE == 0) lens[e] = 0;
    }
    SX_K(solo_dec_leave)(&w, &states[s]);
    if (SX_LANE == 0) {
        play[s] = play0 + n_packets;
        if (status) status[s] = first_err;
    }
}

// single-packet decode with the reference's raw (ptr, nBytes, lostflag) convention
__global__ void __launch_bounds__(64, 4) SX_K(solo_decode_raw_kernel)(SxDecStream* st, const u8* bits, int n0, int n1, int lostflag,
                                                             int useMDIndex, i16* pcm, i32* status) {
    SxDecWork& w = SX_K(g_sx_dec_work);
    SX_K(solo_dec_enter)(&w, st);
    int ret = sx_decode_packet(&w, bits, n0, n1, lostflag, useMDIndex, pcm);
    SX_K(solo_dec_leave)(&w, st);
    if (SX_LANE == 0) *status = ret;
}

// launchers (host): same signature for both rates
static inline hipError_t SX_K(solo_dec_launch_init)(void* states, int n_streams, int hb_joint, hipStream_t s) {
    hipLaunchKernelGGL(SX_K(solo_dec_init_kernel), dim3(n_streams), dim3(64), 0, s, (SxDecStream*)states, n_streams, hb_joint);
    return hipGetLastError();
}
static inline hipError_t SX_K(solo_dec_launch)(void* states, const uint8_t* bits, const int16_t* nbytes, const uint8_t* recv, int n_streams,
                                               int n_packets, int slot, int useMDIndex, int16_t* pcm, int32_t* status, hipStream_t s) {
    hipLaunchKernelGGL(SX_K(solo_decode_kernel), dim3(n_streams), dim3(64), 0, s, (SxDecStream*)states, bits, nbytes, recv, n_streams,
                       n_packets, slot, useMDIndex, pcm, status);
    return hipGetLastError();
}
// recs: solo_dec_extracted_bytes() x n_streams x pc bytes + 256: the records, behind them the list of the slots that carry bytes and its count
static inline hipError_t SX_K(solo_dec_launch_extract)(const void* states, const uint8_t* bits, const int16_t* nbytes, const uint8_t* recv, int n_streams,
                                                       int n_packets, int p0, int pc, int slot, int useMDIndex, void* recs, hipStream_t s) {
    const size_t lanes = (size_t)n_streams * (size_t)pc * 2;
    u32* list = recv ? (u32*)((SxExtracted*)recs + lanes) : (u32*)0;     // (reception flags given: descriptions may be missing)
    u32* count = list ? list + lanes : (u32*)0;
    if (list) {
        hipError_t e = hipMemsetAsync(count, 0, sizeof(u32), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(SX_K(solo_dec_list_kernel), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s, (const SxDecStream*)states, nbytes, recv, n_streams,
                           n_packets, p0, pc, slot, (SxExtracted*)recs, list, count);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(SX_K(solo_dec_extract_kernel), dim3((unsigned)((lanes + SX_EXTRACT_LANES - 1) / SX_EXTRACT_LANES)), dim3(SX_EXTRACT_LANES), 0, s,
                       (const SxDecStream*)states, bits, nbytes, recv, n_streams, n_packets, p0, pc, slot, useMDIndex, (SxExtracted*)recs, list, count);
    return hipGetLastError();
}
static inline hipError_t SX_K(solo_dec_launch_synth)(void* states, const uint8_t* bits, const int16_t* nbytes, const uint8_t* recv, int n_streams,
                                                     int n_packets, int p0, int pc, int slot, int useMDIndex, const void* recs,
                                                     int16_t* pcm, int32_t* status, hipStream_t s) {
    hipLaunchKernelGGL(SX_K(solo_dec_synth_kernel), dim3(n_streams), dim3(64), 0, s, (SxDecStream*)states, bits, nbytes, recv, n_streams,
                       n_packets, p0, pc, slot, useMDIndex, (const SxExtracted*)recs, pcm, status);
    return hipGetLastError();
}
static inline size_t SX_K(solo_dec_extracted_bytes)() { return 2 * sizeof(SxExtracted) + 2 * sizeof(u32); }      // per packet: two records, two list entries
static inline hipError_t SX_K(solo_dec_launch_split)(void* states, const uint8_t* descA, const int16_t* lenA, const uint8_t* descB,
                                                     const int16_t* lenB, int n_streams, int n_packets, int slot, int useMDIndex,
                                                     int16_t* pcm, int32_t* status, hipStream_t s) {
    hipLaunchKernelGGL(SX_K(solo_decode_split_kernel), dim3(n_streams), dim3(64), 0, s, (SxDecStream*)states, descA, lenA, descB, lenB,
                       n_streams, n_packets, slot, useMDIndex, pcm, status);
    return hipGetLastError();
}
static inline hipError_t SX_K(solo_recv_launch_reset)(uint32_t* lens, int32_t* play, uint32_t* stats, int n_streams, int depth, int32_t first_seq, hipStream_t s) {
    size_t n = (size_t)n_streams * (size_t)depth;
    if (n < SX_RECV_NSTATS) n = SX_RECV_NSTATS;
    hipLaunchKernelGGL(SX_K(solo_recv_reset_kernel), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, lens, play, stats, n_streams, depth, first_seq);
    return hipGetLastError();
}
static inline hipError_t SX_K(solo_recv_launch_insert)(const void* arrivals, int n_arr, const uint8_t* payload, long long payload_bytes, int n_streams, int depth,
                                                       int slot, int useMDIndex, uint8_t* ring, uint32_t* lens, const int32_t* play, uint32_t* stats,
                                                       hipStream_t s) {
    hipLaunchKernelGGL(SX_K(solo_recv_insert_kernel), dim3(n_arr), dim3(64), 0, s, (const SxRecvArrival*)arrivals, n_arr, payload, payload_bytes, n_streams,
                       depth, slot, useMDIndex, ring, lens, play, stats);
    return hipGetLastError();
}
static inline hipError_t SX_K(solo_dec_launch_ring)(void* states, const uint8_t* ring, uint32_t* lens, int32_t* play, int n_streams, int n_packets, int depth,
                                                    int slot, int useMDIndex, int16_t* pcm, int32_t* status, hipStream_t s) {
    hipLaunchKernelGGL(SX_K(solo_decode_ring_kernel), dim3(n_streams), dim3(64), 0, s, (SxDecStream*)states, ring, lens, play, n_streams, n_packets, depth,
                       slot, useMDIndex, pcm, status);
    return hipGetLastError();
}
static inline hipError_t SX_K(solo_dec_launch_raw)(void* state, const uint8_t* bits, int n0, int n1, int lostflag, int useMDIndex, int16_t* pcm,
                                                   int32_t* status, hipStream_t s) {
    hipLaunchKernelGGL(SX_K(solo_decode_raw_kernel), dim3(1), dim3(64), 0, s, (SxDecStream*)state, bits, n0, n1, lostflag, useMDIndex, pcm, status);
    return hipGetLastError();
}
static inline size_t SX_K(solo_dec_state_bytes)() { return sizeof(SxDecStream); }
