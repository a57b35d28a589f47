/* solo_mi355x.h -- C ABI of libsolo_mi355x.so, the MI355X-native drop-in for the SOLO hot path.
 *
 * Part 1 re-declares, unchanged, the six entry points of the reference's public header
 *   JC1_SDK_SRC_ARM/interface/AGR_JC1_SDK_API.h:11-64   (identical in JC1_SDK_SRC_FLP/interface/)
 * so the reference's own callers (test/enc_main.c:190-274, test/dec_main.c:188-392, or a media
 * engine) link against this library instead of libJC1Codec.a without source changes.  Each handle is
 * a batch of ONE stream whose codec state lives in HBM; every call launches the same gfx950 kernels
 * as the batched API (there is no host-side codec arithmetic and no CPU fallback: if no HIP device is
 * usable, Init returns NULL and Encode/Decode return -1).
 * THESE SIX SYMBOLS ARE A CONFORMANCE AND MIGRATION PATH, NOT A REPLACEMENT FOR THE REFERENCE'S PER-CALL SPEED: a call is the
 * single-wavefront kernel chain of one packet end to end plus two small copies and one synchronisation -- Encode 1.06 ms (2.7 x the
 * reference's 0.39 ms on one host core), Decode 0.21 ms (8.8 x its 0.024 ms); tools/legacy_api_cost.py measures both.  One stream is
 * 1 / 4096 of what the device does in that time: throughput comes from Part 2.
 *
 * Part 2 is the additive batched API: N independent streams per handle, device pointers in,
 * device pointers out, one wavefront per stream.  This is what a server-side integration binds
 * (see INTEGRATION.md for the ctypes / cgo style stubs).
 */
#ifndef SOLO_MI355X_H
#define SOLO_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Part 1: reference-compatible surface (types from interface/SKP_Silk_typedef.h: SKP_int16=short,
 * SKP_int32=int, SKP_uint8=unsigned char)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {            /* AGR_JC1_SDK_API.h:11-21 */
    int32_t mode;           /* ignored by the library (as in the reference)            */
    int32_t targetRate_bps; /* <=0 -> 15600 (AGR_BWE_SDK_API.c:35-37); CLI default 13600 */
    int32_t samplerate;     /* 16000, or 32000 (1280-sample packets, SILK wide band; targetRate_bps >= 15600) */
    int32_t dtx_enable;     /* 0 / 1                                                    */
    int32_t framesize_ms;   /* 40, or 20 (one SILK frame + one 4-byte high-band frame per packet: half the samples; not with joint_mode 1) */
    int32_t joint_enable;   /* 0, or 1 with joint_mode 1 (one 40 ms high-band frame)   */
    int32_t joint_mode;
    int32_t useMDIndex;     /* 0/1: one extra description-index symbol per description  */
} USER_Ctrl_enc;

typedef struct {            /* AGR_JC1_SDK_API.h:23-31 */
    int32_t packetLoss_perc;
    int32_t samplerate;     /* 16000, or 32000 (1280-sample packets, SILK wide band at 16 kHz) */
    int32_t framesize_ms;   /* 40 or 20, like the encoder's */
    int32_t joint_enable;
    int32_t joint_mode;
    int32_t useMDIndex;
} USER_Ctrl_dec;

/* AGR_JC1_SDK_API.h:33  (impl. libBWE/AGR_BWE_SDK_API.c:11).  NULL if the configuration is not the
 * supported one (samplerate 16000 or 32000, framesize 40 or 20, joint off or joint_mode 1 with framesize 40) or no GPU is available. */
void *AGR_Sate_Encoder_Init(USER_Ctrl_enc *enc_Ctrl);
/* AGR_JC1_SDK_API.h:37  (impl. AGR_BWE_SDK_API.c:129).  pcm: 640 samples (1280 at 32 kHz; half of that with framesize_ms 20); returns total bytes,
 * nBytesOut[0] = total, nBytesOut[1] = len(MD2)+8 (+4 with framesize_ms 20 or joint_mode 1) (MD1 = first nBytesOut[0]-nBytesOut[1] bytes). */
int32_t AGR_Sate_Encoder_Encode(void *SATEEnc_State, const int16_t *AGR_Sate_PCM, uint8_t *AGR_Sate_Bit,
                                int32_t AGR_Sate_Buf_Size, int16_t *nBytesOut);
/* AGR_JC1_SDK_API.h:45 */
int AGR_Sate_Encoder_Uninit(void *SATEEnc_State);
/* AGR_JC1_SDK_API.h:49  (impl. AGR_BWE_SDK_API.c:166) */
void *AGR_Sate_Decoder_Init(USER_Ctrl_dec *dec_Ctrl);
/* AGR_JC1_SDK_API.h:53  (impl. AGR_BWE_SDK_API.c:249).  lostflag: 1 lost, 2 MD1 only, 3 MD2(+HB) only,
 * 4 both.  Like the reference: -1 for nBytes[0] <= 0 with nothing touched; otherwise nBytes[0..1] are overwritten with the
 * low-band description lengths and *nSamplesOut is always written.  Unlike the reference, lengths that do not fit the buffer
 * contract (nBytes[0] > 1088, nBytes[1] outside [0, nBytes[0]], a second description shorter than its high-band bytes) are
 * refused with SKP_SILK_DEC_PAYLOAD_TOO_LARGE (-11) / SKP_SILK_DEC_PAYLOAD_ERROR (-12) instead of read out of bounds; on a
 * decoder error the PCM buffer is left untouched. */
int32_t AGR_Sate_Decoder_Decode(void *SATEDec_State, int16_t *AGR_Sate_PCM, int16_t *nSamplesOut,
                                const uint8_t *AGR_Sate_Bit, int16_t nBytes[], int32_t lostflag);
/* AGR_JC1_SDK_API.h:62 */
int32_t AGR_Sate_Decoder_Uninit(void *SATEDec_State);

/* ------------------------------------------------------------------------------------------------
 * Part 2: batched device API.  All d_* pointers are DEVICE pointers (HBM) of the current HIP device.
 *
 *   d_pcm      int16  [n_streams][n_packets][640]      stream-major 16 kHz PCM (1280 samples per packet at 32 kHz; half as many with framesize_ms 20)
 *   d_bits     uint8  [n_streams][n_packets][slot]     one fixed-size slot per packet: MD1|MD2|HB
 *   d_nbytes   int16  [n_streams][n_packets][2]        {total, len(MD2)+8}  (the reference's nBytesOut[0..1])
 *   d_recv     uint8  [n_streams][n_packets]           bit0: MD1 arrived, bit1: MD2(+HB) arrived
 *                                                      (3 -> lostflag 4, 1 -> 2, 2 -> 3, 0 -> 1; the
 *                                                       pointer/length mapping of test/dec_main.c:255-378
 *                                                       is done on the device)
 *   d_status   int32  [n_streams]                      0, or the first negative SILK error of the call
 *
 * Length records are validated on the device (0 <= len(MD2)+HB <= total <= slot): a record that violates this is never
 * dereferenced, the packet is concealed as lost and d_status reports -11 / -12.  DELIBERATE CONVENTION (not reference
 * behaviour): an empty record (total <= 0, e.g. a DTX packet that was not sent) is concealed as a lost packet (lostflag 1);
 * the reference library returns -1 for it without touching its state (AGR_BWE_SDK_API.c:266) and its CLI repeats the previous
 * output buffer -- the legacy AGR_Sate_Decoder_Decode symbol above keeps that behaviour.
 *
 * Streams keep their codec state in HBM inside the handle between calls (a call with n_packets = P
 * is identical to P calls with n_packets = 1).  Work is enqueued on `hip_stream` (a hipStream_t, may
 * be NULL for the default stream) and is asynchronous; no host synchronisation is performed.
 * Return value: 0 or a negative hipError_t.  solo_batch_encode returns -1 for n_packets >= ~700 000 (16 kHz) / ~350 000 (32 kHz) per
 * call -- split longer (offline) inputs over several calls; state carries over.
 * Device memory a handle holds besides the stream states: encode -- the hand-over records of one call, 4.3 KB per packet of the call
 * (n_streams x n_packets), and 32 KB of quantiser ring per four streams; decode -- up to two buffers of extraction records, 2216 B per
 * packet of a CHUNK (16 kHz API rate; 2 x sizeof(SxExtracted) + two entries of the list of slots that carry bytes: solo_api.hip asserts the figure): a call is cut into chunks of
 * min(64, SOLO_DEC_SCRATCH_CAP / (n_streams x 2216)) packets but never less than ONE, so one buffer holds max(n_streams x 2216 B, at most
 * SOLO_DEC_SCRATCH_CAP bytes) (environment, read when the handle decodes for the first time; default -- also for 0 or an unparsable
 * value -- 1 GiB: 4096 streams x 64 packets are 581 MB, 8192 streams get chunks of 59 packets).
 * ---------------------------------------------------------------------------------------------- */
typedef struct solo_batch solo_batch_t;

#define SOLO_PACKET_SAMPLES 640
#define SOLO_DEFAULT_SLOT_BYTES 512

/* Creates encoder and/or decoder state for n_streams streams (pass NULL to skip one direction). */
solo_batch_t *solo_batch_create(int32_t n_streams, const USER_Ctrl_enc *enc, const USER_Ctrl_dec *dec,
                                int32_t slot_bytes);
void solo_batch_destroy(solo_batch_t *b);
/* Re-initialises all stream states (same as destroy + create) -- EXCEPT the receiver staging ring: descriptions filed with
 * solo_recv_insert, play-out positions and statistics survive a reset; call solo_recv_create again to start the ring afresh. */
int32_t solo_batch_reset(solo_batch_t *b, void *hip_stream);
int32_t solo_batch_encode(solo_batch_t *b, const int16_t *d_pcm, int32_t n_packets, uint8_t *d_bits,
                          int16_t *d_nbytes, int32_t *d_status, void *hip_stream);
int32_t solo_batch_decode(solo_batch_t *b, const uint8_t *d_bits, const int16_t *d_nbytes,
                          const uint8_t *d_recv, int32_t n_packets, int16_t *d_pcm, int32_t *d_status,
                          void *hip_stream);
/* Receiver front end: the two descriptions of every packet arrive separately (MD1, and MD2 || HB(8)), possibly only one,
 * possibly in either arrival slot.  d_descA / d_descB: uint8 [N][P][slot_bytes]; d_lenA / d_lenB: int16 [N][P], 0 = nothing
 * arrived.  With useMDIndex = 1 in the decoder control the kernel identifies the descriptions by the index they carry
 * (SKP_Silk_decode_parameters.c:55-57) and sorts them itself; with useMDIndex = 0 slot A is MD1 and slot B is MD2 || HB.
 * Builds the (ptr, nBytes, lostflag) call of test/dec_main.c:255-378 on the GPU and decodes.  Same outputs as
 * solo_batch_decode.  Packets above 252 bytes are rejected (status -11). */
int32_t solo_batch_decode_split(solo_batch_t *b, const uint8_t *d_descA, const int16_t *d_lenA, const uint8_t *d_descB,
                                const int16_t *d_lenB, int32_t slot_bytes, int32_t n_packets, int16_t *d_pcm,
                                int32_t *d_status, void *hip_stream);
/* Receiver staging ring: the "cache queue" of README.md:52-58 (imag/solo_neteq.png) kept in device memory for all streams of the
 * handle.  solo_recv_create: queue of `depth` sequence numbers per stream (1..4096), `slot_bytes` per description (<= 32767), every
 * stream's play-out position at `first_seq` (>= 0); calling it again empties the queue.  solo_recv_insert files n arrivals in any
 * order: stream, sequence number of the 40 ms packet, desc = 0 (MD1) / 1 (MD2 || HB) when the transport knows which description it
 * carries, -1 when it does not (the library reads the index the description carries as its first coded symbol: needs
 * useMDIndex = 1), payload = d_payload[offset .. offset + len).  An arrival for a packet that has been played, that lies `depth` or
 * more ahead, whose slot is taken (a second copy) or whose fields are out of range is dropped and counted.  solo_recv_decode decodes the
 * next n_packets (<= depth) sequence numbers of EVERY stream from what has arrived by then (the merge and the (ptr, nBytes, lostflag)
 * mapping of solo_batch_decode_split; nothing arrived = concealment), frees those entries and advances the play-out positions.
 * d_pcm int16 [N][n_packets][packet samples], d_status int32 [N] or NULL.  Calls on one handle must be ordered (same stream, or
 * events).  solo_recv_stats copies {inserted, late, ahead, duplicate, bad, 0, 0, 0} to HOST memory (synchronises the stream). */
typedef struct { int32_t stream, seq, desc, offset, len; } solo_arrival_t;
int32_t solo_recv_create(solo_batch_t *b, int32_t depth, int32_t slot_bytes, int32_t first_seq, void *hip_stream);
int32_t solo_recv_insert(solo_batch_t *b, const solo_arrival_t *d_arrivals, int32_t n_arrivals, const uint8_t *d_payload,
                         int64_t payload_bytes, void *hip_stream);
int32_t solo_recv_decode(solo_batch_t *b, int32_t n_packets, int16_t *d_pcm, int32_t *d_status, void *hip_stream);
int32_t solo_recv_stats(solo_batch_t *b, uint32_t *out8, void *hip_stream);
/* Pipelining consecutive encode calls: with on = 1 solo_batch_encode returns without making `hip_stream` wait for the handle's
 * internal streams, so the next encode call starts while the tail of this one still runs (the caller passes different output
 * buffers to calls in flight).  Before consuming the outputs of an encode call on some stream, call
 * solo_batch_wait_encode(b, stream, which): which = 0 the most recent encode call, 1 the one before; the INPUT buffers of a
 * call must stay valid and unmodified until then as well (a stream-ordered allocator does not know the handle's internal
 * streams).  Default: off (every
 * call is complete on its stream when the next operation of that stream runs). */
int32_t solo_batch_set_async_join(solo_batch_t *b, int32_t on);
int32_t solo_batch_wait_encode(solo_batch_t *b, void *hip_stream, int32_t which);
/* Geometry / introspection */
int32_t solo_batch_n_streams(const solo_batch_t *b);
int32_t solo_batch_slot_bytes(const solo_batch_t *b);
/* Name of the dominant kernel of the last encode / decode launch (for profiling tools). */
/* Names of the device kernels behind the four timed stages: 0 = quantiser, 1 = decoder proper (batch path; symbol extraction is
 * "solo_dec_extract_kernel"), 2 = encoder analysis, 3 = encoder high band + payload (the range coder is "solo_enc_rc_kernel"). */
const char *solo_kernel_name(int32_t which);
/* Benchmark aid: bracket every kernel of this handle with HIP events on its launch stream, and read the durations of
 * the most recent encode / decode call: ms4 = {analysis, quantiser, coding, decode} (-1 = not run yet).  solo_batch_encode
 * runs its four kernels (three timed stages: the range coder is timed with the coding stage) as a pipeline over chunks of the call's packets (several launches per kernel, overlapping in
 * time on internal streams): the encoder entries are the SUM over the launches of that kernel, and
 * solo_batch_last_encode_chunks() is the number of launches per kernel of that call. */
int32_t solo_batch_set_timing(solo_batch_t *b, int32_t on);
int32_t solo_batch_last_kernel_ms(solo_batch_t *b, float *ms4);
int32_t solo_batch_last_encode_chunks(const solo_batch_t *b);
/* Conformance probes (tests only; DEVICE pointers, default stream, synchronous).  solo_debug_l0: the fixed-point vocabulary of
 * solo_amd/csrc/solo_fix.h (the reference's SKP_SMULWB .. SKP_INVERSE32_varQ, SKP_Silk_macros.h:33-122 / SKP_Silk_Inlines.h:71-220)
 * as compiled for gfx950, out[i] = op(a[i], b[i], c[i]) with the op numbers of solo_amd/csrc/solo_l0_probe.h.
 * solo_debug_sum_sqr_shift: SKP_Silk_sum_sqr_shift (SKP_Silk_sum_sqr_shift.c:40) in its wave-cooperative form, one result pair
 * per row of `len` <= 1024 int16 samples (`stride` samples between rows). */
int32_t solo_debug_l0(int32_t op, int32_t n, const int32_t *d_a, const int32_t *d_b, const int32_t *d_c, int32_t *d_out);
int32_t solo_debug_sum_sqr_shift(const int16_t *d_x, int32_t rows, int32_t len, int32_t stride, int32_t odd_start,
                                 int32_t *d_energy, int32_t *d_shift);
/* solo_debug_rowops: the lane exchanges of the quantiser kernel (solo_amd/csrc/solo_enc_nsq_row.h: bank-masked DPP row shifts, row
 * rotations, quad permutes) applied to 64 input words, 14 rows of 64 results (tests/test_gpu_nsq_row.py).
 * solo_debug_clock: the effective shader clock in MHz while every SIMD runs vector instructions (~1 ms): lets benchmark lines from
 * different boxes of a pool be compared (bench.py records it). */
int32_t solo_debug_rowops(const int32_t *d_in, const int32_t *d_idx, int32_t *d_out, void *hip_stream);
/* solo_debug_nsq: the quantiser kernel ALONE on freshly initialised streams -- h_in: the arguments of the reference's
 * SKP_Silk_NSQ_del_dec calls (SKP_Silk_NSQ_del_dec.c:925) as hand-over records [n_streams][n_packets][2] of 660 bytes (16 kHz API rate),
 * h_out: the kernel's output records {int32 Seed; int32 r[160]; int8 q[2][164]}; HOST pointers; returns the output record size. */
int32_t solo_debug_nsq(int32_t n_streams, int32_t n_packets, const void *h_in, void *h_out);
int32_t solo_debug_clock(double *mhz_out);
/* Library version string. */
const char *solo_version(void);

#ifdef __cplusplus
}
#endif
#endif
