// solo_rc.h -- SILK range coder (encoder + decoder), wave-uniform scalar code.
//
// Restates  SKP_Silk_range_coder.c:31-372  (range_encoder, range_decoder, enc/dec_init, get_length,
// enc_wrap_up, check_after_decoding).  The coder state lives in registers (a small struct); the byte
// buffer is a plain pointer (LDS for the encoder, the packet itself in HBM for the decoder).
// Every lane executes the identical instruction stream on identical data (see solo_wave.h).
#pragma once
#include "solo_wave.h"

#define SX_MAX_ARITHM_BYTES 1024
// error codes: SKP_Silk_define.h:159-166
#define SX_RC_WRITE_BEYOND_BUFFER (-1)
#define SX_RC_CDF_OUT_OF_RANGE (-2)
#define SX_RC_NORMALIZATION_FAILED (-3)
#define SX_RC_ZERO_INTERVAL_WIDTH (-4)
#define SX_RC_DECODER_CHECK_FAILED (-5)
#define SX_RC_READ_BEYOND_BUFFER (-6)
#define SX_RC_ILLEGAL_SAMPLING_RATE (-7)
#define SX_RC_DEC_PAYLOAD_TOO_LONG (-8)

struct SxRangeDec {
    const u8* buf;      // payload bytes of this description (inside the packet, HBM)
    i32 bufferLength;   // bytes that belong to this description
    i32 bufferIx;       // read index, counted from byte 4 (the first 4 bytes pre-load base_Q32)
    u32 base_Q32;
    u32 range_Q16;
    i32 error;
    u32 tail;           // the four bytes that FOLLOW this description in the reference decoder's internal buffer (see sx_rc_byte)
#ifdef SX_RC_LOG
    i32* log; i32 nlog;
#endif
};

// The reference copies a description into its per-description buffer (range_dec_init: memcpy of bufferLength bytes) and its
// decoder then reads buffer[4 + ix] for ix < bufferLength (SKP_Silk_range_coder.c:129,205-216): up to FOUR bytes past the end
// of the description, which are whatever EARLIER packets left at those positions of that buffer.  A valid stream decodes the
// same for every continuation (the encoder's wrap-up guarantees it), a corrupted one need not -- so the decoder keeps a shadow
// of the reference's two buffers per stream (SxDecShadow, HBM) and hands the four bytes behind the current description over in
// `tail` (byte k = the byte at position bufferLength + k).  Probes that only read a first symbol leave tail = 0.
SX_HD u32 sx_rc_byte(const SxRangeDec* rc, i32 pos) {
    return pos < rc->bufferLength ? (u32)rc->buf[pos] : ((rc->tail >> (8 * ((pos - rc->bufferLength) & 3))) & 0xFFu);
}

// SKP_Silk_range_dec_init, SKP_Silk_range_coder.c:262
SX_HD void sx_rc_dec_init(SxRangeDec* rc, const u8* buf, i32 len) {
    rc->buf = buf;
    if (len > SX_MAX_ARITHM_BYTES || len < 0) { rc->error = SX_RC_DEC_PAYLOAD_TOO_LONG; rc->bufferLength = 0; return; }
    rc->bufferLength = len;
    rc->bufferIx = 0;
    rc->base_Q32 = (sx_rc_byte(rc, 0) << 24) | (sx_rc_byte(rc, 1) << 16) | (sx_rc_byte(rc, 2) << 8) | sx_rc_byte(rc, 3);
    rc->range_Q16 = 0x0000FFFF;
    rc->error = 0;
}

// SKP_Silk_range_decoder, SKP_Silk_range_coder.c:115.  Returns the decoded symbol (0 on error).
SX_HD i32 sx_rc_dec(SxRangeDec* rc, const u16* prob, i32 probIx) {
    u32 low_Q16 = 0, high_Q16, range_Q32;
    u32 base_Q32 = rc->base_Q32, range_Q16 = rc->range_Q16;
    i32 bufferIx = rc->bufferIx;
    if (rc->error) return 0;

    // CDF search from the start index.  probIx always names the entry holding low_Q16.  Written without the reference's ++/--
    // overshoot.  (Round 1 blamed hipcc for a symbol that "came out one too high" with the overshooting form; round 2 could not
    // reproduce that with either form -- stand-alone in tools/debug/rc_loop_repro.hip, or in this decoder built with
    // -DSX_RC_REFERENCE_LOOP, at -O2 / -O3, inlined or not: tests/test_alt_build.py keeps that build under test.  What was real:
    // the coder's registers were not kept across packets, so a corrupted payload announcing a third frame decoded from
    // uninitialised registers; see sx_decode_packet.)
    high_Q16 = prob[probIx];
#ifdef SX_RC_REFERENCE_LOOP
    if (range_Q16 * high_Q16 > base_Q32) {                    // the reference's form, SKP_Silk_range_coder.c:136-170
        while (1) {
            low_Q16 = prob[--probIx];
            if (range_Q16 * low_Q16 <= base_Q32) break;
            high_Q16 = low_Q16;
            if (high_Q16 == 0) { rc->error = SX_RC_CDF_OUT_OF_RANGE; return 0; }
        }
    } else {
        while (1) {
            low_Q16 = high_Q16;
            high_Q16 = prob[++probIx];
            if (range_Q16 * high_Q16 > base_Q32) { probIx--; break; }
            if (high_Q16 == 0xFFFF) { rc->error = SX_RC_CDF_OUT_OF_RANGE; return 0; }
        }
    }
#else
    if (range_Q16 * high_Q16 > base_Q32) {
        SX_PLAIN_LOOP
        for (;;) {
            probIx--;
            low_Q16 = prob[probIx];
            if (range_Q16 * low_Q16 <= base_Q32) break;
            high_Q16 = low_Q16;
            if (high_Q16 == 0) { rc->error = SX_RC_CDF_OUT_OF_RANGE; return 0; }
        }
    } else {
        SX_PLAIN_LOOP
        for (;;) {
            low_Q16 = high_Q16;
            high_Q16 = prob[probIx + 1];
            if (range_Q16 * high_Q16 > base_Q32) break;
            probIx++;
            if (high_Q16 == 0xFFFF) { rc->error = SX_RC_CDF_OUT_OF_RANGE; return 0; }
        }
    }
#endif
    base_Q32 -= range_Q16 * low_Q16;
    range_Q32 = range_Q16 * (high_Q16 - low_Q16);

    if (range_Q32 & 0xFF000000) {
        range_Q16 = range_Q32 >> 16;
    } else {
        if (range_Q32 & 0xFFFF0000) {
            range_Q16 = range_Q32 >> 8;
            if (base_Q32 >> 24) { rc->error = SX_RC_NORMALIZATION_FAILED; return 0; }
        } else {
            range_Q16 = range_Q32;
            // reference: SKP_RSHIFT( base_Q32, 16 ) on an unsigned value => logical shift
            if (base_Q32 >> 16) { rc->error = SX_RC_NORMALIZATION_FAILED; return 0; }
            base_Q32 <<= 8;
            if (bufferIx < rc->bufferLength) base_Q32 |= sx_rc_byte(rc, 4 + bufferIx++);
        }
        base_Q32 <<= 8;
        if (bufferIx < rc->bufferLength) base_Q32 |= sx_rc_byte(rc, 4 + bufferIx++);
    }
    if (range_Q16 == 0) { rc->error = SX_RC_ZERO_INTERVAL_WIDTH; return 0; }
    rc->base_Q32 = base_Q32;
    rc->range_Q16 = range_Q16;
    rc->bufferIx = bufferIx;
#ifdef SX_RC_LOG
    if (rc->log && rc->nlog < 500) { rc->log[rc->nlog++] = probIx; }
#endif
    return probIx;
}

// sx_rc_dec for a two-symbol model {0, p, 65535} started at index 1, the threshold kept in a register (the sign / LSB decoders):
// same arithmetic and the same error exits as the general search above
SX_HD i32 sx_rc_dec_bin(SxRangeDec* rc, u32 p) {
    u32 base_Q32 = rc->base_Q32, range_Q16 = rc->range_Q16, range_Q32, low_Q16, high_Q16;
    i32 bufferIx = rc->bufferIx, sym;
    if (rc->error) return 0;
    if (range_Q16 * p > base_Q32) {
        sym = 0; low_Q16 = 0; high_Q16 = p;        // (p == 0 cannot get here: 0 > base is false)
    } else {
        if (!(range_Q16 * 0xFFFFu > base_Q32)) { rc->error = SX_RC_CDF_OUT_OF_RANGE; return 0; }
        sym = 1; low_Q16 = p; high_Q16 = 0xFFFFu;
    }
    base_Q32 -= range_Q16 * low_Q16;
    range_Q32 = range_Q16 * (high_Q16 - low_Q16);
    if (range_Q32 & 0xFF000000) {
        range_Q16 = range_Q32 >> 16;
    } else {
        if (range_Q32 & 0xFFFF0000) {
            range_Q16 = range_Q32 >> 8;
            if (base_Q32 >> 24) { rc->error = SX_RC_NORMALIZATION_FAILED; return 0; }
        } else {
            range_Q16 = range_Q32;
            if (base_Q32 >> 16) { rc->error = SX_RC_NORMALIZATION_FAILED; return 0; }
            base_Q32 <<= 8;
            if (bufferIx < rc->bufferLength) base_Q32 |= sx_rc_byte(rc, 4 + bufferIx++);
        }
        base_Q32 <<= 8;
        if (bufferIx < rc->bufferLength) base_Q32 |= sx_rc_byte(rc, 4 + bufferIx++);
    }
    if (range_Q16 == 0) { rc->error = SX_RC_ZERO_INTERVAL_WIDTH; return 0; }
    rc->base_Q32 = base_Q32;
    rc->range_Q16 = range_Q16;
    rc->bufferIx = bufferIx;
#ifdef SX_RC_LOG
    if (rc->log && rc->nlog < 500) { rc->log[rc->nlog++] = sym; }
#endif
    return sym;
}

// SKP_Silk_range_coder_get_length, SKP_Silk_range_coder.c:288 (shared by both directions)
SX_HD i32 sx_rc_length_bits(i32 bufferIx, u32 range_Q16, i32* nBytes) {
    i32 nBits = (bufferIx << 3) + sx_clz32((i32)(range_Q16 - 1)) - 14;
    *nBytes = (nBits + 7) >> 3;
    return nBits;
}

// SKP_Silk_range_coder_check_after_decoding, SKP_Silk_range_coder.c:350.
// NOTE the reference indexes its internal buffer from byte 0 here (not from byte 4).
SX_HD void sx_rc_check_after_decoding(SxRangeDec* rc) {
    i32 nBytes;
    i32 bits = sx_rc_length_bits(rc->bufferIx, rc->range_Q16, &nBytes);
    if (nBytes - 1 >= rc->bufferLength) { rc->error = SX_RC_DECODER_CHECK_FAILED; return; }
    if (bits & 7) {
        i32 mask = 0xFF >> (bits & 7);
        if (((i32)sx_rc_byte(rc, nBytes - 1) & mask) != mask) { rc->error = SX_RC_DECODER_CHECK_FAILED; return; }
    }
}

// ---------------------------------------------------------------------------------------------------
// decoder, interval form: the same decoder run for EVERY possible value of the four bytes that follow the description at once.
// What those bytes are depends on the stream's history (see sx_rc_byte), which a decoder that works on the packets of a stream
// IN PARALLEL (solo_dec.h, sx_extract_desc) does not have.  The decoder's base register is an increasing function of the
// unknown bytes as long as the same symbols have been decoded, and every decision it takes is a threshold test on the base;
// so it carries the base for the all-zeros and for the all-ones continuation and checks at every decision that both ends of
// the interval fall on the same side.  If they always do, the decoded symbols are those of the reference decoder for ANY
// history (every well-formed description: its own bytes pin the symbols down); if they do not, `ambiguous` is set and the
// caller decodes that packet serially with the real history instead.
// ---------------------------------------------------------------------------------------------------
struct SxRangeDec2 {
    const u8* buf;
    i32 bufferLength, bufferIx;
    u32 base_Q32;       // continuation 00 00 00 00
    u32 base1_Q32;      // continuation FF FF FF FF
    u32 range_Q16;
    i32 error, ambiguous;
};
SX_HD void sx_rc_dec_init(SxRangeDec2* rc, const u8* buf, i32 len) {
    rc->buf = buf;
    rc->ambiguous = 0;
    if (len > SX_MAX_ARITHM_BYTES || len < 0) { rc->error = SX_RC_DEC_PAYLOAD_TOO_LONG; rc->bufferLength = 0; return; }
    rc->bufferLength = len;
    rc->bufferIx = 0;
    u32 b0 = 0, b1 = 0;
    for (int k = 0; k < 4; k++) {
        const bool in = k < len;
        const u32 v = in ? (u32)buf[k] : 0u;
        b0 = (b0 << 8) | v;
        b1 = (b1 << 8) | (in ? v : 0xFFu);
    }
    rc->base_Q32 = b0;
    rc->base1_Q32 = b1;
    rc->range_Q16 = 0x0000FFFF;
    rc->error = 0;
}
// the next byte into both bases (positions past the description: 00 / FF)
#define SX_RC2_FEED()                                                                         \
    {                                                                                         \
        base_Q32 <<= 8; base1_Q32 <<= 8;                                                      \
        if (bufferIx < rc->bufferLength) {                                                    \
            const i32 pos_ = 4 + bufferIx++;                                                  \
            const bool in_ = pos_ < rc->bufferLength;                                         \
            const u32 v_ = in_ ? (u32)rc->buf[pos_] : 0u;                                     \
            base_Q32 |= v_; base1_Q32 |= in_ ? v_ : 0xFFu;                                    \
        }                                                                                     \
    }
#define SX_RC2_NORMALISE()                                                                    \
    if (range_Q32 & 0xFF000000) {                                                             \
        range_Q16 = range_Q32 >> 16;                                                          \
    } else {                                                                                  \
        if (range_Q32 & 0xFFFF0000) {                                                         \
            range_Q16 = range_Q32 >> 8;                                                       \
            if (((base_Q32 >> 24) != 0) != ((base1_Q32 >> 24) != 0)) rc->ambiguous = 1;       \
            if (base_Q32 >> 24) { rc->error = SX_RC_NORMALIZATION_FAILED; return 0; }         \
        } else {                                                                              \
            range_Q16 = range_Q32;                                                            \
            if (((base_Q32 >> 16) != 0) != ((base1_Q32 >> 16) != 0)) rc->ambiguous = 1;       \
            if (base_Q32 >> 16) { rc->error = SX_RC_NORMALIZATION_FAILED; return 0; }         \
            SX_RC2_FEED()                                                                     \
        }                                                                                     \
        SX_RC2_FEED()                                                                         \
    }                                                                                         \
    if (range_Q16 == 0) { rc->error = SX_RC_ZERO_INTERVAL_WIDTH; return 0; }                  \
    rc->base_Q32 = base_Q32; rc->base1_Q32 = base1_Q32;                                       \
    rc->range_Q16 = range_Q16;                                                                \
    rc->bufferIx = bufferIx;

SX_HD i32 sx_rc_dec(SxRangeDec2* rc, const u16* prob, i32 probIx) {
    u32 low_Q16 = 0, high_Q16, range_Q32;
    u32 base_Q32 = rc->base_Q32, base1_Q32 = rc->base1_Q32, range_Q16 = rc->range_Q16;
    i32 bufferIx = rc->bufferIx;
    if (rc->error) return 0;
    high_Q16 = prob[probIx];
    if (range_Q16 * high_Q16 > base_Q32) {
        SX_PLAIN_LOOP
        for (;;) {
            probIx--;
            low_Q16 = prob[probIx];
            if (range_Q16 * low_Q16 <= base_Q32) break;
            high_Q16 = low_Q16;
            if (high_Q16 == 0) { rc->error = SX_RC_CDF_OUT_OF_RANGE; return 0; }
        }
    } else {
        SX_PLAIN_LOOP
        for (;;) {
            low_Q16 = high_Q16;
            high_Q16 = prob[probIx + 1];
            if (range_Q16 * high_Q16 > base_Q32) break;
            probIx++;
            if (high_Q16 == 0xFFFF) { rc->error = SX_RC_CDF_OUT_OF_RANGE; return 0; }      // (the upper end is past the table as well)
        }
    }
    if (!(range_Q16 * high_Q16 > base1_Q32)) rc->ambiguous = 1;        // the upper end of the interval decodes a later symbol
    base_Q32 -= range_Q16 * low_Q16;
    base1_Q32 -= range_Q16 * low_Q16;
    range_Q32 = range_Q16 * (high_Q16 - low_Q16);
    SX_RC2_NORMALISE()
    return probIx;
}
SX_HD i32 sx_rc_dec_bin(SxRangeDec2* rc, u32 p) {
    u32 base_Q32 = rc->base_Q32, base1_Q32 = rc->base1_Q32, range_Q16 = rc->range_Q16, range_Q32, low_Q16, high_Q16;
    i32 bufferIx = rc->bufferIx, sym;
    if (rc->error) return 0;
    if (range_Q16 * p > base_Q32) {
        sym = 0; low_Q16 = 0; high_Q16 = p;
    } else {
        if (!(range_Q16 * 0xFFFFu > base_Q32)) { rc->error = SX_RC_CDF_OUT_OF_RANGE; return 0; }
        sym = 1; low_Q16 = p; high_Q16 = 0xFFFFu;
    }
    if (!(range_Q16 * high_Q16 > base1_Q32)) rc->ambiguous = 1;
    base_Q32 -= range_Q16 * low_Q16;
    base1_Q32 -= range_Q16 * low_Q16;
    range_Q32 = range_Q16 * (high_Q16 - low_Q16);
    SX_RC2_NORMALISE()
    return sym;
}
SX_HD void sx_rc_check_after_decoding(SxRangeDec2* rc) {      // (reads a byte of the description itself)
    i32 nBytes;
    i32 bits = sx_rc_length_bits(rc->bufferIx, rc->range_Q16, &nBytes);
    if (nBytes - 1 >= rc->bufferLength) { rc->error = SX_RC_DECODER_CHECK_FAILED; return; }
    if (bits & 7) {
        i32 mask = 0xFF >> (bits & 7);
        if (((i32)rc->buf[nBytes - 1] & mask) != mask) { rc->error = SX_RC_DECODER_CHECK_FAILED; return; }
    }
}
#undef SX_RC2_NORMALISE
#undef SX_RC2_FEED

// ---------------------------------------------------------------------------------------------------
// encoder
// ---------------------------------------------------------------------------------------------------
struct SxRangeEnc {
    u8* buf;            // SX_MAX_ARITHM_BYTES bytes of workspace
    i32 bufferLength;
    i32 bufferIx;
    u32 base_Q32;
    u32 range_Q16;
    i32 error;
};

// SKP_Silk_range_enc_init, SKP_Silk_range_coder.c:249
SX_HD void sx_rc_enc_init(SxRangeEnc* rc, u8* buf) {
    rc->buf = buf;
    rc->bufferLength = SX_MAX_ARITHM_BYTES;
    rc->range_Q16 = 0x0000FFFF;
    rc->bufferIx = 0;
    rc->base_Q32 = 0;
    rc->error = 0;
}

// SKP_Silk_range_encoder, SKP_Silk_range_coder.c:31
SX_HD void sx_rc_enc(SxRangeEnc* rc, i32 data, const u16* prob) {
    if (rc->error) return;
    u32 base_Q32 = rc->base_Q32, range_Q16 = rc->range_Q16;
    i32 bufferIx = rc->bufferIx;
    u8* buffer = rc->buf;
    u32 low_Q16 = prob[data], high_Q16 = prob[data + 1];
    u32 base_tmp = base_Q32;
    base_Q32 += range_Q16 * low_Q16;
    u32 range_Q32 = range_Q16 * (high_Q16 - low_Q16);
    if (base_Q32 < base_tmp) {  // carry: propagate into the bytes already written
        i32 ix = bufferIx;
        while ((++buffer[--ix]) == 0) {}
    }
    if (range_Q32 & 0xFF000000) {
        range_Q16 = range_Q32 >> 16;
    } else {
        if (range_Q32 & 0xFFFF0000) {
            range_Q16 = range_Q32 >> 8;
        } else {
            range_Q16 = range_Q32;
            if (bufferIx >= rc->bufferLength) { rc->error = SX_RC_WRITE_BEYOND_BUFFER; return; }
            buffer[bufferIx++] = (u8)(base_Q32 >> 24);
            base_Q32 <<= 8;
        }
        if (bufferIx >= rc->bufferLength) { rc->error = SX_RC_WRITE_BEYOND_BUFFER; return; }
        buffer[bufferIx++] = (u8)(base_Q32 >> 24);
        base_Q32 <<= 8;
    }
    rc->base_Q32 = base_Q32;
    rc->range_Q16 = range_Q16;
    rc->bufferIx = bufferIx;
}

// sx_rc_enc for a two-symbol model {0, p, 65535} with the threshold in a register (signs, LSBs)
SX_HD void sx_rc_enc_bin(SxRangeEnc* rc, i32 data, u32 p) {
    if (rc->error) return;
    u32 base_Q32 = rc->base_Q32, range_Q16 = rc->range_Q16;
    i32 bufferIx = rc->bufferIx;
    u8* buffer = rc->buf;
    const u32 low_Q16 = data ? p : 0u, high_Q16 = data ? 0xFFFFu : p;
    u32 base_tmp = base_Q32;
    base_Q32 += range_Q16 * low_Q16;
    u32 range_Q32 = range_Q16 * (high_Q16 - low_Q16);
    if (base_Q32 < base_tmp) {
        i32 ix = bufferIx;
        while ((++buffer[--ix]) == 0) {}
    }
    if (range_Q32 & 0xFF000000) {
        range_Q16 = range_Q32 >> 16;
    } else {
        if (range_Q32 & 0xFFFF0000) {
            range_Q16 = range_Q32 >> 8;
        } else {
            range_Q16 = range_Q32;
            if (bufferIx >= rc->bufferLength) { rc->error = SX_RC_WRITE_BEYOND_BUFFER; return; }
            buffer[bufferIx++] = (u8)(base_Q32 >> 24);
            base_Q32 <<= 8;
        }
        if (bufferIx >= rc->bufferLength) { rc->error = SX_RC_WRITE_BEYOND_BUFFER; return; }
        buffer[bufferIx++] = (u8)(base_Q32 >> 24);
        base_Q32 <<= 8;
    }
    rc->base_Q32 = base_Q32;
    rc->range_Q16 = range_Q16;
    rc->bufferIx = bufferIx;
}

// SKP_Silk_range_enc_wrap_up, SKP_Silk_range_coder.c:305
SX_HD void sx_rc_enc_wrap_up(SxRangeEnc* rc) {
    i32 nBytes;
    u32 base_Q24 = rc->base_Q32 >> 8;
    i32 bits_in_stream = sx_rc_length_bits(rc->bufferIx, rc->range_Q16, &nBytes);
    i32 bits_to_store = bits_in_stream - (rc->bufferIx << 3);
    base_Q24 += 0x00800000u >> (bits_to_store - 1);
    base_Q24 &= 0xFFFFFFFFu << (24 - bits_to_store);
    if (base_Q24 & 0x01000000) {
        i32 ix = rc->bufferIx;
        while ((++(rc->buf[--ix])) == 0) {}
    }
    if (rc->bufferIx < rc->bufferLength) {
        rc->buf[rc->bufferIx++] = (u8)(base_Q24 >> 16);
        if (bits_to_store > 8) {
            if (rc->bufferIx < rc->bufferLength) rc->buf[rc->bufferIx++] = (u8)(base_Q24 >> 8);
        }
    }
    if (bits_in_stream & 7) {
        i32 mask = 0xFF >> (bits_in_stream & 7);
        if (nBytes - 1 < rc->bufferLength) rc->buf[nBytes - 1] |= (u8)mask;
    }
}
