// solo_recv.h -- the receiver's "cache queue" (README.md:52-58, imag/solo_neteq.png) as a device-resident staging ring.
//
// The reference leaves the time axis of the receiver to the engine that embeds it: "every time a packet needs to be inserted into
// the buffer, it needs to be determined whether the corresponding complementary packet already exists and which complementary
// bitstream the current packet contains", a lone description is inserted as it is, a description whose partner is already queued
// is merged with it, and at play-out time test/dec_main.c:255-378 turns what is there into (ptr, nBytes, lostflag).  Here that
// queue lives in HBM for all streams of a handle:
//
//     ring [N][D][2][slot]   payload of the two arrival slots of the D sequence numbers play .. play + D - 1 of every stream
//     lens [N][D]            one 32-bit word per entry: low half = bytes in slot A, high half = bytes in slot B (0 = empty)
//     play [N]               sequence number the stream decodes next
//
// Arrivals (stream, seq, description, payload) come in any order and any number per call; one wavefront files one arrival: it
// finds out which description the payload is when the transport does not say (desc = -1: "the engine can determine which
// complementary bitstream the current bitstream is based on the bitstream flag bit" -- the index a description carries as its first
// range-coded symbol with useMDIndex = 1, SKP_Silk_decode_parameters.c:55-57), checks the window (older than `play`: late,
// dropped; `play + D` or newer: ahead of the queue, dropped), claims the description's slot with a compare-and-swap on the entry's
// length word (a second copy of a description loses and is dropped) and copies the payload.  Decoding n packets walks entries play .. play + n - 1 of every stream through the same merge
// as solo_batch_decode_split (what has arrived BY THEN is decoded, the rest is concealed), clears them and advances `play`.
// A description that arrives after its partner but before its packet's turn is therefore merged on the GPU; one that arrives
// after the packet was played is counted as late.  This is the staging only -- no play-out adaptation, no time stretching.
#pragma once
#include "solo_rc.h"

struct SxRecvArrival { i32 stream, seq, desc, offset, len; };      // == solo_arrival_t (include/solo_mi355x.h)
static_assert(sizeof(SxRecvArrival) == 20, "solo_arrival_t layout");

// what became of an arrival (index into the handle's statistics)
#define SX_RECV_INSERTED 0
#define SX_RECV_LATE 1          // its packet has been played already
#define SX_RECV_AHEAD 2         // more than the queue's depth ahead of the play-out position
#define SX_RECV_DUP 3           // its slot already holds an arrival (a second copy of the description)
#define SX_RECV_BAD 4           // stream / description / length / payload range outside the handle's; desc = -1 without useMDIndex, or unreadable
#define SX_RECV_NSTATS 8

SX_HD size_t sx_recv_entry(int stream, i32 seq, int depth) { return (size_t)stream * (size_t)depth + (size_t)((u32)seq % (u32)depth); }

// window and argument check of one arrival against its stream's play-out position
SX_HD int sx_recv_check(const SxRecvArrival* r, int n_streams, int depth, int slot, long long payload_bytes, const i32* play) {
    if (r->stream < 0 || r->stream >= n_streams || r->desc < -1 || r->desc > 1 || r->len <= 0 || r->len > slot || r->len > 0x7FFF ||
        r->offset < 0 || (long long)r->offset + (long long)r->len > payload_bytes || r->seq < 0)
        return SX_RECV_BAD;
    const i32 p = play[r->stream];
    if (r->seq < p) return SX_RECV_LATE;
    if ((long long)r->seq >= (long long)p + (long long)depth) return SX_RECV_AHEAD;
    return SX_RECV_INSERTED;
}

// which description a payload is, read off its first range-coded symbol: 0 (MD1), 1 (MD2 || HB), or -1 (the coder rejects it)
SX_HD int sx_recv_md_index(const u8* p, int len) {
    SxRangeDec r;
    r.error = 0; r.tail = 0;
    sx_rc_dec_init(&r, p, sx_min(len, SX_MAX_ARITHM_BYTES));
    const int i = sx_rc_dec(&r, T_cdf_mdindex, T_CDF_MID_MDINDEX);
    return r.error ? -1 : i;
}

// Claim slot `desc` (0 = A: MD1, 1 = B: MD2 || HB) of an entry for `len` bytes; the first arrival wins.  Returns the slot or -1.
SX_HD int sx_recv_claim(u32* word, int desc, int len) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if ((old >> (16 * desc)) & 0xFFFFu) return -1;
        const u32 seen = atomicCAS(word, old, old | ((u32)len << (16 * desc)));
        if (seen == old) return desc;
        old = seen;
    }
#else
    const u32 old = *word;
    if ((old >> (16 * desc)) & 0xFFFFu) return -1;
    *word = old | ((u32)len << (16 * desc));
    return desc;
#endif
}

// the whole filing decision of one arrival (everything but the payload copy): verdict, and the slot when it is SX_RECV_INSERTED
SX_HD int sx_recv_file(const SxRecvArrival* r, const u8* payload, long long payload_bytes, int n_streams, int depth, int slot, int useMDIndex,
                       const i32* play, u32* lens, int leader, int* slot_out) {
    *slot_out = -1;
    int verdict = sx_recv_check(r, n_streams, depth, slot, payload_bytes, play);
    if (verdict != SX_RECV_INSERTED) return verdict;
    int desc = r->desc;
    if (desc < 0) {
        if (useMDIndex != 1) return SX_RECV_BAD;
        desc = sx_recv_md_index(payload + r->offset, r->len);
        if (desc < 0 || desc > 1) return SX_RECV_BAD;
    }
    int sl = -1;
    if (leader) sl = sx_recv_claim(&lens[sx_recv_entry(r->stream, r->seq, depth)], desc, r->len);
    *slot_out = sl;                 // (the other lanes of a wavefront take the leader's answer)
    return verdict;
}
