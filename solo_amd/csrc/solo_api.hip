// solo_api.hip -- gfx950 kernels + the C ABI of libsolo_mi355x.so (include/solo_mi355x.h).
//
// Execution model: one 64-lane wavefront (= one workgroup) owns one stream and walks its packets in
// order; thousands of streams run concurrently.  Persistent codec state is an array of per-stream
// structs in HBM; the per-packet working set lives in LDS.  No host-side codec arithmetic exists in
// this library: without a usable HIP device every entry point fails.
#include <hip/hip_runtime.h>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#include "../../include/solo_mi355x.h"
#include "solo_dec.h"

#define SOLO_CHECK(expr)                                        \
    do {                                                        \
        hipError_t e_ = (expr);                                 \
        if (e_ != hipSuccess) return -(int32_t)e_;              \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
#include "solo_dec_kernels.h"
#include "solo_l0_probe.h"

// conformance probe of the L0 fixed-point vocabulary as compiled for gfx950 (solo_debug_l0 below): out[i] = op(a[i], b[i], c[i])
__global__ void __launch_bounds__(64) solo_l0_probe_kernel(int op, int n, const i32* a, const i32* b, const i32* c, i32* out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) out[i] = sx_l0_probe(op, a[i], b[i], c[i]);
}
// SKP_Silk_sum_sqr_shift in its wave-cooperative form (solo_common.h): one wavefront per row of `len` samples
__global__ void __launch_bounds__(64) solo_sum_sqr_probe_kernel(const i16* x, int len, int stride, int odd_start, i32* energy, i32* shift) {
    __shared__ i16 buf[1024];
    const i16* row = x + (size_t)blockIdx.x * stride;
    SX_PAR(i, len) buf[i] = row[i];
    wv_sync();
    i32 e, s;
    sx_sum_sqr_shift_wv(&e, &s, buf, len, odd_start);
    if (SX_LANE == 0) { energy[blockIdx.x] = e; shift[blockIdx.x] = s; }
}

// the same decoder compiled for the 32 kHz API rate (SILK wide band, 16 kHz bands): solo_api_wb.hip
extern "C" {
size_t solo_wb_dec_state_bytes();
hipError_t solo_wb_dec_launch_init(void* states, int n_streams, int hb_joint, hipStream_t s);
hipError_t solo_wb_dec_launch(void* states, const uint8_t* bits, const int16_t* nbytes, const uint8_t* recv, int n_streams, int n_packets, int slot,
                              int useMDIndex, int16_t* pcm, int32_t* status, hipStream_t s);
hipError_t solo_wb_dec_launch_extract(const void* states, const uint8_t* bits, const int16_t* nbytes, const uint8_t* recv, int n_streams, int n_packets,
                                      int p0, int pc, int slot, int useMDIndex, void* recs, hipStream_t s);
hipError_t solo_wb_dec_launch_synth(void* states, const uint8_t* bits, const int16_t* nbytes, const uint8_t* recv, int n_streams, int n_packets, int p0,
                                    int pc, int slot, int useMDIndex, const void* recs, int16_t* pcm, int32_t* status, hipStream_t s);
size_t solo_wb_dec_extracted_bytes();
hipError_t solo_wb_dec_launch_split(void* states, const uint8_t* descA, const int16_t* lenA, const uint8_t* descB, const int16_t* lenB, int n_streams,
                                    int n_packets, int slot, int useMDIndex, int16_t* pcm, int32_t* status, hipStream_t s);
hipError_t solo_wb_dec_launch_raw(void* state, const uint8_t* bits, int n0, int n1, int lostflag, int useMDIndex, int16_t* pcm, int32_t* status,
                                  hipStream_t s);
hipError_t solo_wb_dec_launch_ring(void* states, const uint8_t* ring, uint32_t* lens, int32_t* play, int n_streams, int n_packets, int depth, int slot,
                                   int useMDIndex, int16_t* pcm, int32_t* status, hipStream_t s);
}

#ifdef SOLO_WITH_ENCODER
#include "solo_enc_ops.h"
extern "C" int solo_launch_gate(const unsigned int* flag, unsigned int target, void* hip_stream);      // solo_nsq_row.hip
extern "C" const solo_enc_ops* solo_nb_enc_ops();                                                      // solo_enc_k.hip
extern "C" const solo_enc_ops* solo_wb_enc_ops();                                                      // solo_enc_k_wb.hip
#endif

// ---------------------------------------------------------------------------------------------------
// host side: handle + C ABI
// ---------------------------------------------------------------------------------------------------
#define SOLO_MAX_CHUNKS 64
struct solo_batch {
    int32_t n_streams;
    int32_t slot;
    int have_enc, have_dec;
    USER_Ctrl_enc enc_ctrl;
    USER_Ctrl_dec dec_ctrl;
    void* d_enc_state;
#ifdef SOLO_WITH_ENCODER
    const solo_enc_ops* eops;        // launch table of the build that matches the encoder's rate (solo_enc_kernels.h)
#endif
    void* d_nsq_ring;                // emission-ring scratch of the quantiser launches (one launch group of streams; frame-local data)
    // decoder pipeline: symbol extraction (stream sP) one chunk of packets ahead of the decoder proper (stream sS)
    hipStream_t sP, sS;
    hipEvent_t evDFork, evDJoin, evDJoin2, evP[2], evS[2];
    void* d_parsed[2];               // extraction records of the chunk being extracted / being decoded
    size_t parsed_bytes;             // size of each
    int parsed_two;                  // both buffers have that size (a call of one chunk needs only the first)
    int dec_pipe_ready, dec_split, dec_chunk, dec_first;
    size_t dec_scratch_cap;          // env SOLO_DEC_SCRATCH_CAP, read when the handle's decode pipeline is set up
    unsigned int dec_calls;          // two-kernel decode calls so far (evDJoin / evDJoin2 are recorded once > 0)
    void* d_rc_scratch;              // range-coder byte buffers of one coding launch (the launches of a call run in order on sC)
    size_t rc_scratch_bytes;
    void* d_enc_work;                // hand-over records of one launch: SxNsqIn[N][P][2] | SxNsqOut[N][P][2] | SxCodeIn[N][P]
    int32_t enc_work_packets;        // P the hand-over area is sized for
    int timing;                      // solo_batch_set_timing: bracket every kernel with HIP events on its launch stream
    hipEvent_t ev[6];                // decode: 4|D|5  (0..3: unused since the encoder is pipelined)
    int ev_ready, ev_enc, ev_dec;
    // encoder: three internal streams (analysis / front kernel: sA, quantiser: sB, third stage of the launch-per-chunk schedule: sC)
    int pipe_ready, chunk_packets;   // chunk_packets: packets per chunk of the launch-per-chunk schedule (env SOLO_ENC_CHUNK, default 1; 0 = one chunk)
    hipStream_t sA, sB, sC;
    hipEvent_t evFork, evJoinA[2], evJoinC[2], evA[SOLO_MAX_CHUNKS], evB[SOLO_MAX_CHUNKS], evC[SOLO_MAX_CHUNKS];
    int async_join;                  // solo_batch_set_async_join: encode returns without joining its streams into the caller's
    unsigned int enc_seq;            // encode calls so far (selects the join-event set)
    int evC_valid;                   // chunks of the previous call whose coding-done events are recorded
    int last_np, last_cp;            // packets / packets per chunk of the previous call (layout of the hand-over records)
    hipEvent_t tev[3][SOLO_MAX_CHUNKS][2];   // timing brackets per kernel type / launch (created with set_timing)
    int tev_ready, last_chunks;
    unsigned int* d_started;         // per launch slot: workgroups of the quantiser launches that have started (running count)
    unsigned int started_target[SOLO_MAX_CHUNKS];
    int group_streams;               // streams per launch group (env SOLO_ENC_GROUP; default: launch per chunk 8192, persistent = what is resident at once)
    int gate;                        // env SOLO_ENC_GATE: the front / next analysis launch starts once the quantiser's workgroups are resident
    // persistent schedule (calls of two or more packets; env SOLO_ENC_PERSIST=0 turns it off): per-stream hand-over flags, tickets
    int persist;
    int persist_group;               // streams per launch group: all of a group's front workgroups are resident beside its quantiser's
    unsigned int* d_flags;           // ana[n_streams] | nsq[n_streams / 4 + 1] | prog[n_streams] | err[4]
    unsigned int ticket;             // packets encoded by persistent calls so far (mod 2^32): the flag words are never reset
    unsigned int final_wait_ticks;   // env SOLO_ENC_FINAL_WAIT_US: bound of the front kernel's wait for the quantiser's last packets (100 MHz ticks)
    int front_defer;                 // SOLO_ENC_FINAL_WAIT_US=-1
    void* d_nsq_stage;               // the persistent quantiser's staging records (one launch group)
    void* d_front_scratch;           // byte buffers of the front kernel's in-wave range coder (one launch group)
    void* d_dec_state;               // SxDecState[n_streams] of the build that matches `wb`
    // receiver staging ring (solo_recv.h): payload [N][D][2][slot] | length words [N][D] | play-out positions [N] | statistics
    uint8_t* d_recv_ring;
    uint32_t* d_recv_lens;
    int32_t* d_recv_play;
    uint32_t* d_recv_stats;
    int32_t recv_depth, recv_slot;
    int wb;                          // decoder control asked for samplerate 32000: 1280-sample packets, SILK at 16 kHz
};

// joint_enable = 0, or joint_mode 1 (one 40 ms high-band frame per packet, AGR_BWE_SDK_API.c:64-67); the other joint modes are
// "reserved" in the reference as well
static bool ctrl_enc_supported(const USER_Ctrl_enc* c) {
    // framesize_ms 40 (two SILK frames per packet) or 20 (one: AGR_BWE_SDK_API.c:100-115, test/enc_main.c:129); the 40 ms high-band frame of
    // joint_mode 1 needs the 40 ms packet
    if (!(c->framesize_ms == 40 || (c->framesize_ms == 20 && c->joint_enable == 0)) || !(c->joint_enable == 0 || c->joint_mode == 1)) return false;
    if (c->samplerate == 16000) return true;
    // 32 kHz input: SILK runs wide band.  Below WB2MB_BITRATE_BPS (14 kbps for SILK = 15.6 kbps total, 14.8 kbps with the 40 ms
    // high-band frame) the reference starts at, or switches down to, 12 / 8 kHz internally (SKP_Silk_control_audio_bandwidth.c:44-76):
    // those rates and the switching are not built, so such a configuration is refused instead of coded differently
    const int hb_bps = (c->joint_enable != 0 && c->joint_mode == 1) ? 800 : 1600;
    const int rate = c->targetRate_bps <= 0 ? 15600 : c->targetRate_bps;
    return c->samplerate == 32000 && rate - hb_bps >= 14000;
}
static bool ctrl_dec_supported(const USER_Ctrl_dec* c) {
    // 32000: the wide-band decoder (solo_api_wb.hip); a stream whose internal rate is not 16 kHz is rejected packet by packet
    return (c->samplerate == 16000 || c->samplerate == 32000) && (c->framesize_ms == 40 || (c->framesize_ms == 20 && c->joint_enable == 0)) &&
           (c->joint_enable == 0 || c->joint_mode == 1);
}
static int ctrl_hb_joint(int joint_enable, int joint_mode) { return joint_enable != 0 && joint_mode == 1; }
// bytes of high band per packet: (QMF_HB_FrameSize / BWE_FrameSize) * HB_BYTE
static int ctrl_hb_bytes(int joint_enable, int joint_mode, int framesize_ms) { return (ctrl_hb_joint(joint_enable, joint_mode) || framesize_ms == 20) ? SX_HB_BYTES / 2 : SX_HB_BYTES; }

#ifdef SOLO_WITH_ENCODER
static int32_t solo_enc_alloc(solo_batch* b) {
    SOLO_CHECK(hipMalloc(&b->d_enc_state, b->eops->state_bytes * (size_t)b->n_streams));
    return 0;
}
static int32_t solo_enc_reset(solo_batch* b, hipStream_t s) {
    // AGR_BWE_SDK_API.c:119: the SILK core gets the target rate minus the high-band share, 1600 * 20 / bwe_framesize_ms
    const int joint = ctrl_hb_joint(b->enc_ctrl.joint_enable, b->enc_ctrl.joint_mode);
    SOLO_CHECK(b->eops->init(b->d_enc_state, b->n_streams, b->enc_ctrl.targetRate_bps - (joint ? 800 : 1600), b->enc_ctrl.useMDIndex, joint,
                             b->enc_ctrl.dtx_enable ? 1 : 0, b->enc_ctrl.framesize_ms == 20 ? 1 : 2, s));
    return 0;
}
static void solo_enc_free(solo_batch* b) {
    if (b->d_enc_state) (void)hipFree(b->d_enc_state);
    if (b->d_enc_work) (void)hipFree(b->d_enc_work);
    if (b->d_nsq_ring) (void)hipFree(b->d_nsq_ring);
    b->d_nsq_ring = NULL;
    if (b->d_rc_scratch) (void)hipFree(b->d_rc_scratch);
    b->d_rc_scratch = NULL;
    if (b->d_flags) (void)hipFree(b->d_flags);
    b->d_flags = NULL;
    if (b->d_front_scratch) (void)hipFree(b->d_front_scratch);
    b->d_front_scratch = NULL;
    if (b->d_nsq_stage) (void)hipFree(b->d_nsq_stage);
    b->d_nsq_stage = NULL;
    b->d_enc_state = NULL;
    b->d_enc_work = NULL;
}
#endif

extern "C" {

const char* solo_version(void) { return "solo_mi355x 0.1 (gfx950)"; }

const char* solo_kernel_name(int32_t which) { return which == 0 ? "solo_nsq_kernel" : (which == 1 ? "solo_dec_synth_kernel" : (which == 2 ? "solo_enc_analysis_kernel" : "solo_enc_coding_kernel")); }

int32_t solo_batch_n_streams(const solo_batch_t* b) { return b ? b->n_streams : 0; }

// Kernel timing for benchmarks: when enabled every kernel launch of this handle is bracketed by HIP events on the launch
// stream; solo_batch_last_kernel_ms() synchronises on them and returns the durations of the most recent encode / decode call:
// ms[0] analysis, ms[1] quantiser, ms[2] coding, ms[3] decode (a field is -1 if that call has not happened yet).
int32_t solo_batch_set_timing(solo_batch_t* b, int32_t on) {
    if (!b) return -1;
    if (on && !b->ev_ready) {
        for (int i = 0; i < 6; i++) SOLO_CHECK(hipEventCreate(&b->ev[i]));
        b->ev_ready = 1;
        for (int k = 0; k < 3; k++) for (int c = 0; c < SOLO_MAX_CHUNKS; c++) for (int e = 0; e < 2; e++) SOLO_CHECK(hipEventCreate(&b->tev[k][c][e]));
        b->tev_ready = 1;
    }
    b->timing = on ? 1 : 0;
    return 0;
}
int32_t solo_batch_last_kernel_ms(solo_batch_t* b, float* ms4) {
    if (!b || !ms4 || !b->ev_ready) return -1;
    for (int i = 0; i < 4; i++) ms4[i] = -1.0f;
    if (b->ev_enc && b->tev_ready) {               // sum over the chunks (the three kernel types overlap in time)
        for (int k = 0; k < 3; k++) {
            float tot = 0.0f;
            for (int c = 0; c < b->last_chunks; c++) {
                float t = 0.0f;
                SOLO_CHECK(hipEventSynchronize(b->tev[k][c][1]));
                SOLO_CHECK(hipEventElapsedTime(&t, b->tev[k][c][0], b->tev[k][c][1]));
                tot += t;
            }
            ms4[k] = tot;
        }
    }
    if (b->ev_dec) {
        SOLO_CHECK(hipEventSynchronize(b->ev[5]));
        SOLO_CHECK(hipEventElapsedTime(&ms4[3], b->ev[4], b->ev[5]));
    }
    return 0;
}
int32_t solo_batch_slot_bytes(const solo_batch_t* b) { return b ? b->slot : 0; }
// Asynchronous joins: with on = 1 solo_batch_encode returns without making the caller's stream wait for its internal streams, so
// that the next encode call can start (its first analysis chunk) while the last quantiser / coding chunks of this one still run.
// Whoever consumes the outputs then calls solo_batch_wait_encode(b, stream, which) first: which = 0 the most recent encode call,
// 1 the one before.  The hand-over records are guarded inside the library; the OUTPUT buffers of two calls in flight must differ.
int32_t solo_batch_set_async_join(solo_batch_t* b, int32_t on) {
    if (!b) return -1;
    b->async_join = on ? 1 : 0;
    return 0;
}
int32_t solo_batch_wait_encode(solo_batch_t* b, void* hip_stream, int32_t which) {
    if (!b || !b->pipe_ready || which < 0 || which > 1 || b->enc_seq <= (unsigned int)which) return b ? 0 : -1;
    const int js = (int)((b->enc_seq - 1u - (unsigned int)which) & 1u);
    SOLO_CHECK(hipStreamWaitEvent((hipStream_t)hip_stream, b->evJoinA[js], 0));
    SOLO_CHECK(hipStreamWaitEvent((hipStream_t)hip_stream, b->evJoinC[js], 0));
    return 0;
}
int32_t solo_batch_last_encode_chunks(const solo_batch_t* b) { return b ? b->last_chunks : 0; }

int32_t solo_batch_reset(solo_batch_t* b, void* hip_stream) {
    if (!b) return -1;
    hipStream_t s = (hipStream_t)hip_stream;
    if (b->pipe_ready && b->enc_seq > 0) {
        // kernels of the most recent encode calls may still run on the internal streams (always so with async joins, and when
        // reset is issued on another stream than the encode was): the init kernels must not overtake them
        for (int i = 0; i < 2; i++) {
            if (b->enc_seq <= (unsigned int)i) break;
            const int js = (int)((b->enc_seq - 1u - (unsigned int)i) & 1u);
            SOLO_CHECK(hipStreamWaitEvent(s, b->evJoinA[js], 0));
            SOLO_CHECK(hipStreamWaitEvent(s, b->evJoinC[js], 0));
        }
    }
    if (b->have_dec) {
        if (b->dec_pipe_ready && b->dec_split && b->dec_calls > 0) {
            // like the encoder above: a decode call issued on another stream may still run on the internal streams
            SOLO_CHECK(hipStreamWaitEvent(s, b->evDJoin, 0));
            SOLO_CHECK(hipStreamWaitEvent(s, b->evDJoin2, 0));
        }
        const int hbj = ctrl_hb_joint(b->dec_ctrl.joint_enable, b->dec_ctrl.joint_mode) | (b->dec_ctrl.framesize_ms == 20 ? 2 : 0);      // (sx_dec_state_init: bit 1 = one frame per packet)
        SOLO_CHECK(b->wb ? solo_wb_dec_launch_init(b->d_dec_state, b->n_streams, hbj, s) : solo_dec_launch_init(b->d_dec_state, b->n_streams, hbj, s));
    }
#ifdef SOLO_WITH_ENCODER
    if (b->have_enc) {
        int32_t r = solo_enc_reset(b, s);
        if (r) return r;
    }
#endif
    return 0;
}

solo_batch_t* solo_batch_create(int32_t n_streams, const USER_Ctrl_enc* enc, const USER_Ctrl_dec* dec, int32_t slot_bytes) {
    if (n_streams <= 0) return NULL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "solo_mi355x: no HIP device available -- this library has no CPU path\n");
        return NULL;
    }
    if (enc && !ctrl_enc_supported(enc)) return NULL;
    if (dec && !ctrl_dec_supported(dec)) return NULL;
    solo_batch* b = new (std::nothrow) solo_batch();
    if (!b) return NULL;
    memset(b, 0, sizeof(*b));
    b->n_streams = n_streams;
    b->slot = slot_bytes > 0 ? slot_bytes : SOLO_DEFAULT_SLOT_BYTES;
    if (enc) {
#ifdef SOLO_WITH_ENCODER
        b->have_enc = 1;
        b->enc_ctrl = *enc;
        if (b->enc_ctrl.targetRate_bps <= 0) b->enc_ctrl.targetRate_bps = 15600;  // AGR_BWE_SDK_API.c:35
        b->eops = enc->samplerate == 32000 ? solo_wb_enc_ops() : solo_nb_enc_ops();
        if (solo_enc_alloc(b) != 0) { solo_batch_destroy(b); return NULL; }
#else
        delete b;
        return NULL;
#endif
    }
    if (dec) {
        b->have_dec = 1;
        b->dec_ctrl = *dec;
        b->wb = dec->samplerate == 32000;
        if (enc && enc->samplerate != dec->samplerate) { solo_batch_destroy(b); return NULL; }     // a handle has one rate
        if (hipMalloc(&b->d_dec_state, (b->wb ? solo_wb_dec_state_bytes() : solo_dec_state_bytes()) * (size_t)n_streams) != hipSuccess) { solo_batch_destroy(b); return NULL; }
    }
    if (solo_batch_reset(b, NULL) != 0 || hipDeviceSynchronize() != hipSuccess) { solo_batch_destroy(b); return NULL; }
    return b;
}

static void solo_recv_free(solo_batch_t* b);
void solo_batch_destroy(solo_batch_t* b) {
    if (!b) return;
    solo_recv_free(b);
    if (b->dec_pipe_ready && b->dec_split) {
        (void)hipStreamSynchronize(b->sP); (void)hipStreamSynchronize(b->sS);
        (void)hipStreamDestroy(b->sP); (void)hipStreamDestroy(b->sS);
        (void)hipEventDestroy(b->evDFork); (void)hipEventDestroy(b->evDJoin); (void)hipEventDestroy(b->evDJoin2);
        for (int i = 0; i < 2; i++) { (void)hipEventDestroy(b->evP[i]); (void)hipEventDestroy(b->evS[i]); }
    }
    for (int i = 0; i < 2; i++) {
        if (b->d_parsed[i]) (void)hipFree(b->d_parsed[i]);
        b->d_parsed[i] = NULL;
    }
    if (b->d_dec_state) (void)hipFree(b->d_dec_state);
    if (b->ev_ready) for (int i = 0; i < 6; i++) (void)hipEventDestroy(b->ev[i]);
    if (b->tev_ready) for (int k = 0; k < 3; k++) for (int c = 0; c < SOLO_MAX_CHUNKS; c++) for (int e = 0; e < 2; e++) (void)hipEventDestroy(b->tev[k][c][e]);
    if (b->pipe_ready) {
        (void)hipStreamSynchronize(b->sA); (void)hipStreamSynchronize(b->sB); (void)hipStreamSynchronize(b->sC);
        (void)hipStreamDestroy(b->sA); (void)hipStreamDestroy(b->sB); (void)hipStreamDestroy(b->sC);
        if (b->d_started) (void)hipFree(b->d_started);
        (void)hipEventDestroy(b->evFork);
        for (int i = 0; i < 2; i++) { (void)hipEventDestroy(b->evJoinA[i]); (void)hipEventDestroy(b->evJoinC[i]); }
        for (int c = 0; c < SOLO_MAX_CHUNKS; c++) (void)hipEventDestroy(b->evC[c]);
        for (int c = 0; c < SOLO_MAX_CHUNKS; c++) { (void)hipEventDestroy(b->evA[c]); (void)hipEventDestroy(b->evB[c]); }
    }
#ifdef SOLO_WITH_ENCODER
    solo_enc_free(b);
#endif
    delete b;
}

// Chunks of the two-kernel decoder.  The extraction of chunk c + 1 was meant to hide behind the synthesis of chunk c, but the two
// kernels never share a compute unit (the synthesis kernel's 16 workgroups take all of its LDS and registers), so every chunk
// boundary only costs: a synthesis launch reloads and stores 4096 stream states, and its last workgroups run in a thinly populated
// tail.  Measured (4096 streams x 50 packets): chunks of 24 packets after a first one of 4: 9.0 ms; 6 + 44: 7.7 ms; ONE chunk:
// 7.4 ms (8192 streams, 30 % description loss: 17.5 / 16.2 / 16.0 ms).  So a call is one chunk up to 64 packets (the extraction
// records of a chunk are 2216 B per packet: 581 MB for 4096 streams x 64 packets), longer calls are cut into chunks of 64.
#define SOLO_DEC_FIRST_CHUNK 64
#define SOLO_DEC_CHUNK_DEFAULT 64
static_assert(2 * sizeof(SxExtracted) + 8 == 2216, "include/solo_mi355x.h documents 2216 bytes of extraction records (2 x 1104 B + two list entries) per packet (16 kHz API rate)");
int32_t solo_batch_decode(solo_batch_t* b, const uint8_t* d_bits, const int16_t* d_nbytes, const uint8_t* d_recv,
                          int32_t n_packets, int16_t* d_pcm, int32_t* d_status, void* hip_stream) {
    if (!b || !b->have_dec || !d_bits || !d_nbytes || !d_pcm || n_packets <= 0) return -1;
    hipStream_t st = (hipStream_t)hip_stream;
    const bool tm = b->timing && b->ev_ready;
    if (!b->dec_pipe_ready) {
        const char* e = getenv("SOLO_DEC_SPLIT");
        b->dec_split = e ? atoi(e) : 1;
        if (b->dec_ctrl.framesize_ms == 20) b->dec_split = 0;      // (the read-ahead records describe two-frame packets)
        e = getenv("SOLO_DEC_CHUNK");
        b->dec_chunk = e ? atoi(e) : SOLO_DEC_CHUNK_DEFAULT;
        if (b->dec_chunk <= 0) b->dec_chunk = 1 << 30;
        e = getenv("SOLO_DEC_FIRST_CHUNK");
        b->dec_first = e ? atoi(e) : SOLO_DEC_FIRST_CHUNK;
        if (b->dec_first <= 0) b->dec_first = SOLO_DEC_FIRST_CHUNK;
        // (read per handle like the other SOLO_DEC_* knobs; a value that does not parse to a positive number means the default)
        e = getenv("SOLO_DEC_SCRATCH_CAP");
        b->dec_scratch_cap = e ? (size_t)strtoull(e, NULL, 10) : 0;
        if (b->dec_scratch_cap == 0) b->dec_scratch_cap = (size_t)1 << 30;
        if (b->dec_split) {
            SOLO_CHECK(hipStreamCreateWithFlags(&b->sP, hipStreamNonBlocking));
            SOLO_CHECK(hipStreamCreateWithFlags(&b->sS, hipStreamNonBlocking));
            SOLO_CHECK(hipEventCreateWithFlags(&b->evDFork, hipEventDisableTiming));
            SOLO_CHECK(hipEventCreateWithFlags(&b->evDJoin, hipEventDisableTiming));
            SOLO_CHECK(hipEventCreateWithFlags(&b->evDJoin2, hipEventDisableTiming));
            for (int i = 0; i < 2; i++) {
                SOLO_CHECK(hipEventCreateWithFlags(&b->evP[i], hipEventDisableTiming));
                SOLO_CHECK(hipEventCreateWithFlags(&b->evS[i], hipEventDisableTiming));
            }
        }
        b->dec_pipe_ready = 1;
    }
    if (tm) (void)hipEventRecord(b->ev[4], st);
    if (!b->dec_split) {
        // single kernel: one wavefront per stream parses (two lanes) and synthesises
        const hipError_t e = (b->wb ? solo_wb_dec_launch : solo_dec_launch)(b->d_dec_state, d_bits, d_nbytes, d_recv, b->n_streams, n_packets, b->slot,
                                                                             b->dec_ctrl.useMDIndex, d_pcm, d_status, st);
        if (tm) { (void)hipEventRecord(b->ev[5], st); b->ev_dec = 1; }
        SOLO_CHECK(e);
        return 0;
    }
    // Two kernels: the symbols of every description of a chunk of packets are read off the range coder at once, one lane each; the
    // decoder proper (one wavefront per stream, packets in order) follows a chunk behind on a second stream; the records go through
    // two alternating buffers.  X_c follows X_{c-1} and D_{c-2} (buffer free), D_c follows X_c and D_{c-1}.
    // (a short first chunk -- its extraction has nothing to hide behind -- then long ones: every decoder launch reloads the stream states)
    // packets per chunk: the knob, capped so that ONE buffer of extraction records stays below SOLO_DEC_SCRATCH_CAP bytes (default 1 GiB;
    // the records are 2216 B per packet at the 16 kHz API rate: 4096 streams x 64 packets = 581 MB, 65536 streams -> 7 packets per chunk).
    // A handle holds at most two such buffers (calls longer than one chunk); include/solo_mi355x.h states the footprint.
    const size_t rec_bytes = b->wb ? solo_wb_dec_extracted_bytes() : solo_dec_extracted_bytes();
    int cp = n_packets < b->dec_chunk ? n_packets : b->dec_chunk;
    {   // (floor: one packet per chunk -- a handle with more than cap / 2216 streams holds n_streams x 2216 B per buffer, see the header)
        const size_t fit = b->dec_scratch_cap / ((size_t)b->n_streams * rec_bytes);
        if ((size_t)cp > fit) cp = fit < 1 ? 1 : (int)fit;
    }
    const int c0 = (n_packets > 2 * b->dec_first && cp > b->dec_first) ? b->dec_first : cp;      // (never larger than the buffers: c0 <= cp)
    const int nchunks = 1 + (n_packets - c0 + cp - 1) / cp;
    const size_t need = (size_t)b->n_streams * (size_t)cp * rec_bytes + 256;      // (+ the count of the listed description slots)
    if (need > b->parsed_bytes || (nchunks > 1 && !b->parsed_two)) {
        SOLO_CHECK(hipStreamSynchronize(st));
        (void)hipStreamSynchronize(b->sP);
        (void)hipStreamSynchronize(b->sS);
        for (int i = 0; i < 2; i++) {
            if (b->d_parsed[i]) (void)hipFree(b->d_parsed[i]);
            b->d_parsed[i] = NULL;
        }
        b->parsed_bytes = 0;
        for (int i = 0; i < 2; i++) SOLO_CHECK(hipMalloc(&b->d_parsed[i], i == 0 || nchunks > 1 ? need : 256));     // (one chunk: one buffer)
        b->parsed_bytes = need;
        b->parsed_two = nchunks > 1;
    }
    if (b->dec_calls > 0) {
        // the previous decode call may have been issued on ANOTHER stream than this one: its kernels on sP / sS (and their use of
        // d_parsed[]) must be done before this call's fork lets the internal streams run on
        SOLO_CHECK(hipStreamWaitEvent(st, b->evDJoin, 0));
        SOLO_CHECK(hipStreamWaitEvent(st, b->evDJoin2, 0));
    }
    SOLO_CHECK(hipEventRecord(b->evDFork, st));
    SOLO_CHECK(hipStreamWaitEvent(b->sP, b->evDFork, 0));
    SOLO_CHECK(hipStreamWaitEvent(b->sS, b->evDFork, 0));
    b->dec_calls++;
    hipError_t lerr = hipSuccess;
    for (int c = 0; c < nchunks && lerr == hipSuccess; c++) {
        const int p0 = c == 0 ? 0 : c0 + (c - 1) * cp, pc = c == 0 ? c0 : ((p0 + cp <= n_packets) ? cp : n_packets - p0), k = c & 1;
        // (every failure inside the loop goes through the join block below: nothing of this call stays forked)
        if (c >= 2 && (lerr = hipStreamWaitEvent(b->sP, b->evS[k], 0)) != hipSuccess) break;
        lerr = (b->wb ? solo_wb_dec_launch_extract : solo_dec_launch_extract)(b->d_dec_state, d_bits, d_nbytes, d_recv, b->n_streams, n_packets, p0, pc,
                                                                              b->slot, b->dec_ctrl.useMDIndex, b->d_parsed[k], b->sP);
        if (lerr != hipSuccess) break;
        if ((lerr = hipEventRecord(b->evP[k], b->sP)) != hipSuccess) break;
        if ((lerr = hipStreamWaitEvent(b->sS, b->evP[k], 0)) != hipSuccess) break;
        lerr = (b->wb ? solo_wb_dec_launch_synth : solo_dec_launch_synth)(b->d_dec_state, d_bits, d_nbytes, d_recv, b->n_streams, n_packets, p0, pc, b->slot,
                                                                          b->dec_ctrl.useMDIndex, b->d_parsed[k], d_pcm, d_status, b->sS);
        if (lerr != hipSuccess) break;
        if ((lerr = hipEventRecord(b->evS[k], b->sS)) != hipSuccess) break;
    }
    // join both internal streams back into the caller's (also after a refused launch: nothing stays forked)
    (void)hipEventRecord(b->evDJoin, b->sP);
    (void)hipStreamWaitEvent(st, b->evDJoin, 0);
    (void)hipEventRecord(b->evDJoin2, b->sS);
    (void)hipStreamWaitEvent(st, b->evDJoin2, 0);
    if (tm) { (void)hipEventRecord(b->ev[5], st); b->ev_dec = 1; }
    SOLO_CHECK(lerr);
    return 0;
}

int32_t solo_batch_decode_split(solo_batch_t* b, const uint8_t* d_descA, const int16_t* d_lenA, const uint8_t* d_descB,
                                const int16_t* d_lenB, int32_t slot_bytes, int32_t n_packets, int16_t* d_pcm, int32_t* d_status,
                                void* hip_stream) {
    if (!b || !b->have_dec || !d_descA || !d_lenA || !d_descB || !d_lenB || !d_pcm || n_packets <= 0 || slot_bytes <= 0) return -1;
    SOLO_CHECK((b->wb ? solo_wb_dec_launch_split : solo_dec_launch_split)(b->d_dec_state, d_descA, d_lenA, d_descB, d_lenB, b->n_streams, n_packets,
                                                                          slot_bytes, b->dec_ctrl.useMDIndex, d_pcm, d_status, (hipStream_t)hip_stream));
    return 0;
}

// ---- receiver staging ring (solo_recv.h) ---------------------------------------------------------------------------------------
static void solo_recv_free(solo_batch_t* b) {
    if (b->d_recv_ring) (void)hipFree(b->d_recv_ring);
    if (b->d_recv_lens) (void)hipFree(b->d_recv_lens);
    if (b->d_recv_play) (void)hipFree(b->d_recv_play);
    if (b->d_recv_stats) (void)hipFree(b->d_recv_stats);
    b->d_recv_ring = NULL; b->d_recv_lens = NULL; b->d_recv_play = NULL; b->d_recv_stats = NULL;
    b->recv_depth = b->recv_slot = 0;
}
int32_t solo_recv_create(solo_batch_t* b, int32_t depth, int32_t slot_bytes, int32_t first_seq, void* hip_stream) {
    if (!b || !b->have_dec || depth <= 0 || depth > 4096 || slot_bytes <= 0 || slot_bytes > 0x7FFF || first_seq < 0) return -1;
    hipStream_t st = (hipStream_t)hip_stream;
    if (depth != b->recv_depth || slot_bytes != b->recv_slot) {
        SOLO_CHECK(hipStreamSynchronize(st));
        solo_recv_free(b);
        const size_t ne = (size_t)b->n_streams * (size_t)depth;
        hipError_t e = hipMalloc((void**)&b->d_recv_ring, ne * 2 * (size_t)slot_bytes);
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_recv_lens, ne * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_recv_play, (size_t)b->n_streams * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_recv_stats, SX_RECV_NSTATS * sizeof(uint32_t));
        if (e != hipSuccess) { solo_recv_free(b); return -(int32_t)e; }
        b->recv_depth = depth; b->recv_slot = slot_bytes;
    }
    SOLO_CHECK(solo_recv_launch_reset(b->d_recv_lens, b->d_recv_play, b->d_recv_stats, b->n_streams, depth, first_seq, st));
    return 0;
}
int32_t solo_recv_insert(solo_batch_t* b, const solo_arrival_t* d_arrivals, int32_t n_arrivals, const uint8_t* d_payload, int64_t payload_bytes,
                         void* hip_stream) {
    if (!b || !b->d_recv_ring || n_arrivals < 0 || (n_arrivals > 0 && (!d_arrivals || !d_payload)) || payload_bytes < 0) return -1;
    if (n_arrivals == 0) return 0;
    SOLO_CHECK(solo_recv_launch_insert(d_arrivals, n_arrivals, d_payload, (long long)payload_bytes, b->n_streams, b->recv_depth, b->recv_slot, b->dec_ctrl.useMDIndex,
                                       b->d_recv_ring, b->d_recv_lens, b->d_recv_play, b->d_recv_stats, (hipStream_t)hip_stream));
    return 0;
}
int32_t solo_recv_decode(solo_batch_t* b, int32_t n_packets, int16_t* d_pcm, int32_t* d_status, void* hip_stream) {
    if (!b || !b->d_recv_ring || !d_pcm || n_packets <= 0 || n_packets > b->recv_depth) return -1;
    SOLO_CHECK((b->wb ? solo_wb_dec_launch_ring : solo_dec_launch_ring)(b->d_dec_state, b->d_recv_ring, b->d_recv_lens, b->d_recv_play, b->n_streams, n_packets,
                                                                        b->recv_depth, b->recv_slot, b->dec_ctrl.useMDIndex, d_pcm, d_status,
                                                                        (hipStream_t)hip_stream));
    return 0;
}
int32_t solo_recv_stats(solo_batch_t* b, uint32_t* out8, void* hip_stream) {
    if (!b || !b->d_recv_ring || !out8) return -1;
    SOLO_CHECK(hipMemcpyAsync(out8, b->d_recv_stats, SX_RECV_NSTATS * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    SOLO_CHECK(hipStreamSynchronize((hipStream_t)hip_stream));
    return 0;
}

#ifdef SOLO_WITH_ENCODER
// one-time set-up of a handle's encoder pipeline: internal streams, events, knobs (documented in INTEGRATION.md section 5), scratch
static int32_t solo_enc_pipe_setup(solo_batch* b) {
    const solo_enc_ops* ops = b->eops;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);            // hi = numerically lowest = greatest priority
    SOLO_CHECK(hipStreamCreateWithPriority(&b->sA, hipStreamNonBlocking, lo));
    SOLO_CHECK(hipStreamCreateWithPriority(&b->sB, hipStreamNonBlocking, hi));
    SOLO_CHECK(hipStreamCreateWithPriority(&b->sC, hipStreamNonBlocking, lo));
    SOLO_CHECK(hipEventCreateWithFlags(&b->evFork, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) {
        SOLO_CHECK(hipEventCreateWithFlags(&b->evJoinA[i], hipEventDisableTiming));
        SOLO_CHECK(hipEventCreateWithFlags(&b->evJoinC[i], hipEventDisableTiming));
    }
    for (int c = 0; c < SOLO_MAX_CHUNKS; c++) {
        SOLO_CHECK(hipEventCreateWithFlags(&b->evA[c], hipEventDisableTiming));
        SOLO_CHECK(hipEventCreateWithFlags(&b->evB[c], hipEventDisableTiming));
        SOLO_CHECK(hipEventCreateWithFlags(&b->evC[c], hipEventDisableTiming));
    }
    int dev = 0, ncu = 0;
    SOLO_CHECK(hipGetDevice(&dev));
    SOLO_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    // SOLO_ENC_PERSIST=1: the persistent schedule (two kernels per call that hand packets over through flags) instead of one launch per
    // chunk and stage.  Bit-exact and deadlock-free by construction, but measured SLOWER (54 - 60 against 50.5 ms per 4096 x 50 packets:
    // DESIGN.md section 9 has the trace): it removes the launch tails, but it also fixes the SIMD's population at five dependent
    // chains, where the launch-per-chunk schedule reaches nine.  Off unless asked for.
    const char* e = getenv("SOLO_ENC_PERSIST");
    b->persist = e ? (atoi(e) != 0) : 0;
    e = getenv("SOLO_ENC_CHUNK");
    b->chunk_packets = e ? atoi(e) : 1;
    if (b->chunk_packets < 0) b->chunk_packets = 1;
    // launch per chunk: the residency gate (hold analysis chunk c + 1 until the quantiser launch of chunk c is resident) costs 3 % with 256
    // quantiser workgroups per chunk: off unless asked for.  Persistent: ONE gate per launch group, in front of the front kernel -- the
    // quantiser's wavefronts have to be resident before 4096 front workgroups take every register of the device: on unless turned off.
    e = getenv("SOLO_ENC_GATE");
    b->gate = e ? (atoi(e) != 0) : -1;                          // (-1: the schedule's default)
    // streams per launch group.  Launch per chunk: 8192 (one group of 8192 takes 120 ms per 50 packets, two of 4096 take 134).  Persistent:
    // what is resident at once -- front workgroups per compute unit (LDS-bound: 16 at the 16 kHz rate, 9 at 32 kHz) x compute units, a
    // multiple of the quantiser's four streams per wavefront: a larger group's last workgroups would only start when the first ones have
    // finished ALL their packets, while their quantiser wavefronts held registers from the start
    e = getenv("SOLO_ENC_GROUP");
    const int g_env = e ? atoi(e) : -1;
    b->group_streams = g_env >= 0 ? g_env : 8192;
    b->persist_group = g_env > 0 ? ((g_env + 3) & ~3) : ((ops->front_per_cu * (ncu > 0 ? ncu : 256)) & ~3);
    if (b->persist_group < 4) b->persist_group = 4;
    e = getenv("SOLO_ENC_FINAL_WAIT_US");
    {
        // (-1, tests: the front launch codes nothing -- no look at the quantiser's flags at all --, the second launch every packet)
        const long us = e ? atol(e) : 20000;
        b->front_defer = us < 0;
        b->final_wait_ticks = (unsigned int)((us < 0 ? 0 : (us > 10000000 ? 10000000 : us)) * 100);
    }
    {   // the quantiser launches of a call run one after the other on sB: one ring, sized for the largest launch group
        int gs = (b->group_streams > 0 && b->group_streams < b->n_streams) ? b->group_streams : b->n_streams;
        if (b->persist && b->persist_group > gs) gs = b->persist_group < b->n_streams ? b->persist_group : b->n_streams;
        SOLO_CHECK(hipMalloc(&b->d_nsq_ring, ops->nsq_ring_bytes(gs)));
    }
    SOLO_CHECK(hipMalloc((void**)&b->d_started, SOLO_MAX_CHUNKS * sizeof(unsigned int)));
    SOLO_CHECK(hipMemset(b->d_started, 0, SOLO_MAX_CHUNKS * sizeof(unsigned int)));
    memset(b->started_target, 0, sizeof(b->started_target));
    if (b->persist) {
        const size_t nflags = (size_t)b->n_streams * 2 + (size_t)b->n_streams / 4 + 1 + 4;
        SOLO_CHECK(hipMalloc((void**)&b->d_flags, nflags * sizeof(unsigned int)));
        SOLO_CHECK(hipMemset(b->d_flags, 0, nflags * sizeof(unsigned int)));
        const int gs = b->persist_group < b->n_streams ? b->persist_group : b->n_streams;
        SOLO_CHECK(hipMalloc(&b->d_front_scratch, ops->front_scratch_bytes(gs)));
        SOLO_CHECK(hipMalloc(&b->d_nsq_stage, ops->nsq_stage_bytes(gs)));
    }
    b->ticket = 0;
    b->pipe_ready = 1;
    return 0;
}

// Persistent schedule of one call: per launch group ONE quantiser launch (sB) and ONE front launch (sA) that hand packets to each other
// through flags while they run (solo_enc_kernels.h), then the front kernel once more behind the quantiser (mode 1: what a bounded wait
// left undone -- nothing, normally).
static int32_t solo_encode_persist(solo_batch* b, const int16_t* d_pcm, int32_t n_packets, uint8_t* d_bits, int16_t* d_nbytes, int32_t* d_status,
                                   hipStream_t st, void* nin, void* nout, void* cin) {
    const solo_enc_ops* ops = b->eops;
    const int G = b->persist_group, ngroups = (b->n_streams + G - 1) / G;
    const bool tm = b->timing && b->tev_ready && ngroups <= SOLO_MAX_CHUNKS;
    unsigned int* ana = b->d_flags;
    unsigned int* nsqf = ana + b->n_streams;
    unsigned int* prog = nsqf + (size_t)b->n_streams / 4 + 1;
    unsigned int* err = prog + b->n_streams;
    const unsigned int ticket0 = b->ticket;
    b->ticket += (unsigned int)n_packets;
    const size_t frame_samples = (size_t)(b->enc_ctrl.framesize_ms == 20 ? ops->packet_samples / 2 : ops->packet_samples);
    SOLO_CHECK(hipEventRecord(b->evFork, st));
    SOLO_CHECK(hipStreamWaitEvent(b->sA, b->evFork, 0));
    SOLO_CHECK(hipStreamWaitEvent(b->sB, b->evFork, 0));
    if (b->enc_seq > 0) {          // the previous call (of either schedule) has read its hand-over records to the end
        const int jp = (int)((b->enc_seq - 1u) & 1u);
        SOLO_CHECK(hipStreamWaitEvent(b->sA, b->evJoinC[jp], 0));
        SOLO_CHECK(hipStreamWaitEvent(b->sB, b->evJoinC[jp], 0));
        SOLO_CHECK(hipStreamWaitEvent(b->sB, b->evJoinA[jp], 0));
    }
    b->evC_valid = 0;
    b->last_np = n_packets;
    b->last_cp = 0;
    hipError_t lerr = hipSuccess;
    for (int g = 0; g < ngroups && lerr == hipSuccess; g++) {
        const int s0 = g * G, ns = (s0 + G <= b->n_streams) ? G : b->n_streams - s0, c = g % SOLO_MAX_CHUNKS;
        const size_t pk0 = (size_t)s0 * (size_t)n_packets;
        void* g_states = (char*)b->d_enc_state + (size_t)s0 * ops->state_bytes;
        void* g_nin = (char*)nin + pk0 * 2 * ops->nsq_in_bytes;
        void* g_nout = (char*)nout + pk0 * 2 * ops->nsq_out_bytes;
        void* g_cin = (char*)cin + pk0 * ops->code_in_bytes;
        const int16_t* g_pcm = d_pcm + pk0 * frame_samples;
        uint8_t* g_bits = d_bits + pk0 * (size_t)b->slot;
        int16_t* g_nbytes = d_nbytes + pk0 * 2;
        int32_t* g_status = d_status ? d_status + s0 : NULL;
        // the front kernel first; the quantiser's launch is held until the front workgroups have started (they fill every compute unit
        // up to one 128-register hole per SIMD, which is where the quantiser's wavefronts then go: solo_nsq_row.hip, "Residency")
        // (sB comes up to where sA stands -- behind the previous group's last launch -- before it starts counting)
        if ((lerr = hipEventRecord(b->evA[c], b->sA)) != hipSuccess) break;
        if ((lerr = hipStreamWaitEvent(b->sB, b->evA[c], 0)) != hipSuccess) break;
        if (tm) (void)hipEventRecord(b->tev[0][c][0], b->sA);
        lerr = ops->front(g_states, g_pcm, ns, n_packets, g_nin, g_cin, g_nout, ana + s0, nsqf + s0 / 4, prog + s0, ticket0, b->front_defer ? 2 : 0, b->final_wait_ticks, b->slot,
                          g_bits, g_nbytes, g_status, b->d_front_scratch, &b->d_started[c], b->sA);
        if (lerr != hipSuccess) break;
        b->started_target[c] += (unsigned int)((ns + ops->front_waves - 1) / ops->front_waves);
        if (tm) (void)hipEventRecord(b->tev[0][c][1], b->sA);
        if (b->gate != 0) (void)solo_launch_gate(&b->d_started[c], b->started_target[c], b->sB);
        if (tm) (void)hipEventRecord(b->tev[1][c][0], b->sB);
        lerr = (hipError_t)ops->nsq_persist(g_states, g_nin, g_nout, ns, n_packets, NULL, b->d_nsq_ring, ana + s0, nsqf + s0 / 4, ticket0, err, b->d_nsq_stage, b->sB);
        if (lerr != hipSuccess) break;
        if (tm) (void)hipEventRecord(b->tev[1][c][1], b->sB);
        if ((lerr = hipEventRecord(b->evB[c], b->sB)) != hipSuccess) break;
        if ((lerr = hipStreamWaitEvent(b->sA, b->evB[c], 0)) != hipSuccess) break;
        if (tm) (void)hipEventRecord(b->tev[2][c][0], b->sA);
        lerr = ops->front(g_states, g_pcm, ns, n_packets, g_nin, g_cin, g_nout, ana + s0, nsqf + s0 / 4, prog + s0, ticket0, 1, 0u, b->slot, g_bits, g_nbytes,
                          g_status, b->d_front_scratch, NULL, b->sA);
        if (tm) (void)hipEventRecord(b->tev[2][c][1], b->sA);
    }
    // join (also after a refused launch: whatever was enqueued still runs, nothing of this call stays forked)
    const int js = (int)(b->enc_seq & 1u);
    b->enc_seq++;
    (void)hipEventRecord(b->evJoinA[js], b->sA);
    (void)hipEventRecord(b->evJoinC[js], b->sB);
    if (!b->async_join || lerr != hipSuccess) {
        (void)hipStreamWaitEvent(st, b->evJoinA[js], 0);
        (void)hipStreamWaitEvent(st, b->evJoinC[js], 0);
    }
    b->last_chunks = (lerr == hipSuccess && (tm || ngroups <= SOLO_MAX_CHUNKS)) ? ngroups : 0;
    if (tm && lerr == hipSuccess) b->ev_enc = 1;
    SOLO_CHECK(lerr);
    SOLO_CHECK(hipGetLastError());
    return 0;
}

int32_t solo_batch_encode(solo_batch_t* b, const int16_t* d_pcm, int32_t n_packets, uint8_t* d_bits, int16_t* d_nbytes,
                          int32_t* d_status, void* hip_stream) {
    if (!b || !b->have_enc || !d_pcm || !d_bits || !d_nbytes || n_packets <= 0) return -1;
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t np = (size_t)b->n_streams * (size_t)n_packets;
    const solo_enc_ops* ops = b->eops;
    // the quantiser addresses the hand-over records of its wavefront's four streams with 32-bit offsets from a wave-uniform base
    // (solo_nsq_row.hip): the records of 3 streams x 2 n_packets, plus one more record for the offsets inside the last one, must stay
    // below 4 GiB (~700 k packets per call at the 16 kHz API rate)
    const unsigned long long per_wave = 64ull / (unsigned long long)ops->nsq_workgroups(64);
    if (((per_wave - 1ull) * 2ull * (unsigned long long)n_packets + 1ull) * (unsigned long long)ops->nsq_out_bytes >= (1ull << 32)) return -1;
    const size_t sz_in = np * 2 * ops->nsq_in_bytes, sz_out = np * 2 * ops->nsq_out_bytes, sz_code = np * ops->code_in_bytes;
    if (n_packets > b->enc_work_packets) {          // grow the hand-over area (synchronises; steady-state launches do not)
        SOLO_CHECK(hipStreamSynchronize(st));
        if (b->pipe_ready) { (void)hipStreamSynchronize(b->sA); (void)hipStreamSynchronize(b->sB); (void)hipStreamSynchronize(b->sC); }
        if (b->d_enc_work) (void)hipFree(b->d_enc_work);
        b->d_enc_work = NULL;
        SOLO_CHECK(hipMalloc(&b->d_enc_work, sz_in + sz_out + sz_code + 256));
        b->enc_work_packets = n_packets;
    }
    void* nin = b->d_enc_work;
    void* nout = (char*)b->d_enc_work + ((sz_in + 63) & ~(size_t)63);
    void* cin = (char*)nout + ((sz_out + 63) & ~(size_t)63);
    void* states = b->d_enc_state;
    if (!b->pipe_ready) {
        const int32_t r = solo_enc_pipe_setup(b);
        if (r) return r;
    }
    // SOLO_ENC_PERSIST=1: calls of two or more packets run the persistent schedule (a single packet has nothing to pipeline inside the call)
    if (b->persist && n_packets >= 2) return solo_encode_persist(b, d_pcm, n_packets, d_bits, d_nbytes, d_status, st, nin, nout, cin);

    // Launch per chunk: chunk c of the call's packets goes analysis (stream sA) -> quantiser (sB) -> high band, range coder (sC).  A_c
    // follows A_{c-1}, B_c follows A_c and B_{c-1}, C_c follows B_c and C_{c-1}; so the quantiser of chunk c (one wave per SIMD, latency
    // bound) runs next to the analysis of chunk c + 1 and the coding of chunk c - 1 (instruction bound): they share the SIMDs.  The
    // kernels of different types touch disjoint parts of the stream records.  The caller's stream is forked / joined by events.
    int cp = b->chunk_packets > 0 ? b->chunk_packets : n_packets;
    int nchunks = (n_packets + cp - 1) / cp;
    if (nchunks > SOLO_MAX_CHUNKS) { cp = (n_packets + SOLO_MAX_CHUNKS - 1) / SOLO_MAX_CHUNKS; nchunks = (n_packets + cp - 1) / cp; }
    {   // scratch of one coding launch: the byte buffers of its descriptions
        const int gs = (b->group_streams > 0 && b->group_streams < b->n_streams) ? b->group_streams : b->n_streams;
        const size_t need = ops->rc_scratch_bytes(gs, cp);
        if (need > b->rc_scratch_bytes) {
            if (b->d_rc_scratch) {
                SOLO_CHECK(hipStreamSynchronize(b->sC));             // (a coding launch of the previous call may still read the old one)
                (void)hipFree(b->d_rc_scratch);
                b->d_rc_scratch = NULL;
                b->rc_scratch_bytes = 0;
            }
            SOLO_CHECK(hipMalloc(&b->d_rc_scratch, need));
            b->rc_scratch_bytes = need;
        }
    }
    const bool tm_req = b->timing && b->tev_ready;
    if (b->enc_seq > 0 && (b->last_np != n_packets || b->last_cp != cp || b->evC_valid == 0)) {
        // the previous call laid its hand-over records out differently (or ran the persistent schedule): no chunk-wise reuse, wait for all of it
        SOLO_CHECK(hipStreamWaitEvent(b->sA, b->evJoinC[(b->enc_seq - 1u) & 1u], 0));
        SOLO_CHECK(hipStreamWaitEvent(b->sB, b->evJoinA[(b->enc_seq - 1u) & 1u], 0));
        b->evC_valid = 0;
    }
    b->last_np = n_packets;
    b->last_cp = cp;
    SOLO_CHECK(hipEventRecord(b->evFork, st));
    SOLO_CHECK(hipStreamWaitEvent(b->sA, b->evFork, 0));
    SOLO_CHECK(hipStreamWaitEvent(b->sB, b->evFork, 0));
    SOLO_CHECK(hipStreamWaitEvent(b->sC, b->evFork, 0));
    // Streams beyond the launch group (SOLO_ENC_GROUP) are processed group after group; all per-stream arrays are stream-major, so a
    // group is the same launch on offset pointers.
    const int G = b->group_streams > 0 ? b->group_streams : b->n_streams;
    const int ngroups = (b->n_streams + G - 1) / G;
    const bool tm = tm_req && (size_t)ngroups * (size_t)nchunks <= SOLO_MAX_CHUNKS;     // (per-launch timing brackets: one per event slot)
    const size_t frame_samples = (size_t)(b->enc_ctrl.framesize_ms == 20 ? ops->packet_samples / 2 : ops->packet_samples);
    int idx = 0;
    hipError_t lerr = hipSuccess;
    for (int g = 0; g < ngroups; g++) {
        const int s0 = g * G, ns = (s0 + G <= b->n_streams) ? G : b->n_streams - s0;
        const size_t pk0 = (size_t)s0 * (size_t)n_packets;                               // first packet record of the group
        void* g_states = (char*)states + (size_t)s0 * ops->state_bytes;
        const int16_t* g_pcm = d_pcm + pk0 * frame_samples;
        void* g_nin = (char*)nin + pk0 * 2 * ops->nsq_in_bytes;
        void* g_nout = (char*)nout + pk0 * 2 * ops->nsq_out_bytes;
        void* g_cin = (char*)cin + pk0 * ops->code_in_bytes;
        uint8_t* g_bits = d_bits + pk0 * (size_t)b->slot;
        int16_t* g_nbytes = d_nbytes + pk0 * 2;
        int32_t* g_status = d_status ? d_status + s0 : NULL;
        for (int cc = 0; cc < nchunks; cc++, idx++) {
            const int c = idx % SOLO_MAX_CHUNKS, cprev = (idx + SOLO_MAX_CHUNKS - 1) % SOLO_MAX_CHUNKS;     // event / counter slot
            const int p0 = cc * cp, pc = (p0 + cp <= n_packets) ? cp : n_packets - p0;
            if (ngroups == 1 && cc < b->evC_valid) SOLO_CHECK(hipStreamWaitEvent(b->sA, b->evC[c], 0));      // (previous call: its coding of this chunk's records is done)
            if (idx > 0 && b->gate > 0) (void)solo_launch_gate(&b->d_started[cprev], b->started_target[cprev], b->sA);
            if (tm) (void)hipEventRecord(b->tev[0][c][0], b->sA);
            if ((lerr = ops->analysis(g_states, g_pcm, ns, n_packets, p0, pc, g_nin, g_cin, b->sA)) != hipSuccess) goto launch_failed;
            if (tm) (void)hipEventRecord(b->tev[0][c][1], b->sA);
            SOLO_CHECK(hipEventRecord(b->evA[c], b->sA));
            SOLO_CHECK(hipStreamWaitEvent(b->sB, b->evA[c], 0));
            if (tm) (void)hipEventRecord(b->tev[1][c][0], b->sB);
#ifdef SX_EXPERIMENTS     // builds for the section tools only (tools/build_stops.sh): SOLO_EXP_SKIP bit 0 = no quantiser, bit 1 = no third stage -- wrong output
            static const int exp_skip = getenv("SOLO_EXP_SKIP") ? atoi(getenv("SOLO_EXP_SKIP")) : 0;
#else
            constexpr int exp_skip = 0;
#endif
            if (!(exp_skip & 1)) {
            if ((lerr = (hipError_t)ops->nsq(g_states, g_nin, g_nout, ns, n_packets, p0, pc, &b->d_started[c], b->d_nsq_ring, b->sB)) != hipSuccess) goto launch_failed;
            b->started_target[c] += (unsigned int)ops->nsq_workgroups(ns);     // workgroups of this launch, counted once it is enqueued
            }
            if (tm) (void)hipEventRecord(b->tev[1][c][1], b->sB);
            SOLO_CHECK(hipEventRecord(b->evB[c], b->sB));
            SOLO_CHECK(hipStreamWaitEvent(b->sC, b->evB[c], 0));
            if (tm) (void)hipEventRecord(b->tev[2][c][0], b->sC);
            if (!(exp_skip & 2))
            if ((lerr = ops->coding(g_states, g_cin, g_nout, ns, n_packets, p0, pc, b->slot, g_bits, g_nbytes, g_status, b->d_rc_scratch, b->sC)) != hipSuccess) goto launch_failed;
            if (tm) (void)hipEventRecord(b->tev[2][c][1], b->sC);
            SOLO_CHECK(hipEventRecord(b->evC[c], b->sC));
        }
    }
    if (0) {
launch_failed:
        // a kernel launch was refused: whatever was enqueued so far still runs; join the internal streams back into the
        // caller's stream so that nothing of this call is left forked, drop the chunk-wise guards, report the HIP error
        const int jf = (int)(b->enc_seq & 1u);
        b->enc_seq++;
        b->evC_valid = 0;
        b->last_chunks = 0;
        (void)hipEventRecord(b->evJoinA[jf], b->sA);
        (void)hipEventRecord(b->evJoinC[jf], b->sC);
        (void)hipStreamWaitEvent(st, b->evJoinA[jf], 0);
        (void)hipStreamWaitEvent(st, b->evJoinC[jf], 0);
        (void)hipEventRecord(b->evFork, b->sB);
        (void)hipStreamWaitEvent(st, b->evFork, 0);
        return -(int32_t)lerr;
    }
    b->evC_valid = ngroups == 1 ? nchunks : 0;       // chunk-wise hand-over guards only for single-group calls
    const int js = (int)(b->enc_seq & 1u);
    b->enc_seq++;
    SOLO_CHECK(hipEventRecord(b->evJoinA[js], b->sA));
    SOLO_CHECK(hipEventRecord(b->evJoinC[js], b->sC));      // (sC's last launch waits for sB's)
    if (!b->async_join) {
        SOLO_CHECK(hipStreamWaitEvent(st, b->evJoinA[js], 0));
        SOLO_CHECK(hipStreamWaitEvent(st, b->evJoinC[js], 0));
    }
    b->last_chunks = tm ? ngroups * nchunks : (ngroups == 1 ? nchunks : 0);
    if (tm) b->ev_enc = 1;
    SOLO_CHECK(hipGetLastError());
    return 0;
}
#else
int32_t solo_batch_encode(solo_batch_t*, const int16_t*, int32_t, uint8_t*, int16_t*, int32_t*, void*) { return -1; }
#endif

// ---- the reference's six entry points: a batch of one stream, staged through device buffers ----------
struct solo_single {
    solo_batch* b;
    // One device block and one pinned host block of the same layout: [status 16 B][pcm 2 x 1280 B][nbytes 16 B][bits: slot bytes].
    // A call is one host-to-device copy, the kernels, one device-to-host copy and ONE synchronisation (round 2 made three synchronous
    // copies per call: 0.26 ms per decoded packet against 0.024 ms of the reference on a host core, tools/legacy_api_cost.py).
    uint8_t* d_blk;
    uint8_t* h_blk;
    int16_t* d_pcm;
    uint8_t* d_bits;
    int16_t* d_nbytes;
    int32_t* d_status;
    int is_enc;
};
#define SOLO_SINGLE_PCM_OFF 16
#define SOLO_SINGLE_NB_OFF (16 + 2 * SX_PACKET * 2)
#define SOLO_SINGLE_BITS_OFF (SOLO_SINGLE_NB_OFF + 16)

static void single_free(solo_single* h) {
    if (!h) return;
    if (h->d_blk) (void)hipFree(h->d_blk);
    if (h->h_blk) (void)hipHostFree(h->h_blk);
    solo_batch_destroy(h->b);
    free(h);
}

static solo_single* single_new(const USER_Ctrl_enc* e, const USER_Ctrl_dec* d) {
    solo_single* h = (solo_single*)calloc(1, sizeof(solo_single));
    if (!h) return NULL;
    h->is_enc = e != NULL;
    h->b = solo_batch_create(1, e, d, 1024 + 64);   // MAX_FRAME_BYTES of the reference harness + slack
    const size_t blk = h->b ? (size_t)SOLO_SINGLE_BITS_OFF + (size_t)h->b->slot : 0;
    // A call moves ~1.4 KB each way through one pinned host block and one device block of the same layout.  (Letting the kernels address the
    // pinned block directly -- no staging copies -- measured the same 0.21 ms per decode call: the call is its single-wave kernel chain + one
    // synchronisation.)
    if (!h->b || hipHostMalloc((void**)&h->h_blk, blk, hipHostMallocDefault) != hipSuccess) { single_free(h); return NULL; }
    memset(h->h_blk, 0, blk);
    if (hipMalloc((void**)&h->d_blk, blk) != hipSuccess || hipMemset(h->d_blk, 0, blk) != hipSuccess) {
        single_free(h);
        return NULL;
    }
    h->d_status = (int32_t*)h->d_blk;
    h->d_pcm = (int16_t*)(h->d_blk + SOLO_SINGLE_PCM_OFF);
    h->d_nbytes = (int16_t*)(h->d_blk + SOLO_SINGLE_NB_OFF);
    h->d_bits = h->d_blk + SOLO_SINGLE_BITS_OFF;
    return h;
}

void* AGR_Sate_Encoder_Init(USER_Ctrl_enc* enc_Ctrl) {
    if (!enc_Ctrl) return NULL;
    if (enc_Ctrl->targetRate_bps <= 0) enc_Ctrl->targetRate_bps = 15600;   // the reference rewrites the caller's struct
    return single_new(enc_Ctrl, NULL);
}

int32_t AGR_Sate_Encoder_Encode(void* st, const int16_t* pcm, uint8_t* bits, int32_t bufSize, int16_t* nBytesOut) {
    solo_single* h = (solo_single*)st;
    if (!h || !h->is_enc) return -1;
    const size_t pcm_bytes = (size_t)(h->b->enc_ctrl.framesize_ms == 20 ? h->b->eops->packet_samples / 2 : h->b->eops->packet_samples) * 2;      // JC1_FrameSize samples
    memcpy(h->h_blk + SOLO_SINGLE_PCM_OFF, pcm, pcm_bytes);
    if (hipMemcpyAsync(h->d_pcm, h->h_blk + SOLO_SINGLE_PCM_OFF, pcm_bytes, hipMemcpyHostToDevice, (hipStream_t)0) != hipSuccess) return -1;
    if (solo_batch_encode(h->b, h->d_pcm, 1, h->d_bits, h->d_nbytes, h->d_status, NULL) != 0) return -1;
    // lengths + the whole payload slot in one copy (a payload is at most a few hundred bytes; the slot 1088)
    if (hipMemcpyAsync(h->h_blk + SOLO_SINGLE_NB_OFF, h->d_blk + SOLO_SINGLE_NB_OFF, 16 + (size_t)h->b->slot, hipMemcpyDeviceToHost, (hipStream_t)0) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)0) != hipSuccess) return -1;
    int16_t nb[2];
    memcpy(nb, h->h_blk + SOLO_SINGLE_NB_OFF, 4);
    int32_t n = nb[0];
    if (n == 0 && h->b->enc_ctrl.dtx_enable)                                 // DTX packet: the reference still returns the high-band bytes
        n = ctrl_hb_bytes(h->b->enc_ctrl.joint_enable, h->b->enc_ctrl.joint_mode, h->b->enc_ctrl.framesize_ms);
    if (n > bufSize) n = bufSize;                                            // AGR_Sate_bits_write truncates to max_nbytes
    if (n > h->b->slot) n = h->b->slot;
    if (n > 0) memcpy(bits, h->h_blk + SOLO_SINGLE_BITS_OFF, (size_t)n);
    nBytesOut[0] = nb[0];
    nBytesOut[1] = nb[1];
    return n;
}

int AGR_Sate_Encoder_Uninit(void* st) {
    if (!st) return -1;
    single_free((solo_single*)st);
    return 0;
}

void* AGR_Sate_Decoder_Init(USER_Ctrl_dec* dec_Ctrl) {
    if (!dec_Ctrl) return NULL;
    return single_new(NULL, dec_Ctrl);
}

int32_t AGR_Sate_Decoder_Decode(void* st, int16_t* pcm, int16_t* nSamplesOut, const uint8_t* bits, int16_t nBytes[], int32_t lostflag) {
    solo_single* h = (solo_single*)st;
    if (!h || h->is_enc) return -1;
    if (nBytes[0] <= 0) return -1;                                           // AGR_BWE_SDK_API.c:266 (state untouched, outputs unwritten)
    if (lostflag < 1 || lostflag > 4) return -1;
    const int ns = (h->b->wb ? 2 * SX_PACKET : SX_PACKET) / (h->b->dec_ctrl.framesize_ms == 20 ? 2 : 1);                      // JC1_FrameSize (AGR_BWE_SDK_API.c:277)
    const int32_t hbb = ctrl_hb_bytes(h->b->dec_ctrl.joint_enable, h->b->dec_ctrl.joint_mode, h->b->dec_ctrl.framesize_ms);
    int32_t n0 = nBytes[0], n1 = nBytes[1];
    if (lostflag != 1) {
        // lengths that do not describe bytes inside the caller's buffer are refused before anything is read (the reference would
        // read out of bounds): too long -> SKP_SILK_DEC_PAYLOAD_TOO_LARGE, inconsistent -> SKP_SILK_DEC_PAYLOAD_ERROR
        if (n0 > h->b->slot) { *nSamplesOut = (int16_t)ns; return -11; }
        if (n1 < 0 || n1 > n0 || (n1 > 0 && n1 < hbb) || (lostflag == 3 && n0 <= hbb) || (lostflag == 4 && n0 < hbb)) { *nSamplesOut = (int16_t)ns; return -12; }
        memcpy(h->h_blk + SOLO_SINGLE_BITS_OFF, bits, (size_t)n0);
        if (hipMemcpyAsync(h->d_bits, h->h_blk + SOLO_SINGLE_BITS_OFF, (size_t)n0, hipMemcpyHostToDevice, (hipStream_t)0) != hipSuccess) return -1;
    }
    if ((h->b->wb ? solo_wb_dec_launch_raw : solo_dec_launch_raw)(h->b->d_dec_state, h->d_bits, n0, n1, lostflag, h->b->dec_ctrl.useMDIndex, h->d_pcm,
                                                                  h->d_status, (hipStream_t)0) != hipSuccess) return -1;
    // status + decoded packet in one copy, one synchronisation
    if (hipMemcpyAsync(h->h_blk, h->d_blk, SOLO_SINGLE_PCM_OFF + (size_t)ns * 2, hipMemcpyDeviceToHost, (hipStream_t)0) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)0) != hipSuccess) return -1;
    int32_t ret = 0;
    memcpy(&ret, h->h_blk, 4);
    // the reference rewrites the caller's nBytes[] with the low-band lengths (AGR_BWE_decode_frame_FIX.c:150-169)
    int32_t nb0 = (lostflag == 2) ? n0 : n0 - hbb;
    int32_t nb1 = n1 ? n1 - hbb : 0;
    nBytes[0] = (int16_t)(nb0 - nb1);
    nBytes[1] = (int16_t)nb1;
    // like the reference, *nSamplesOut is always written (AGR_BWE_SDK_API.c:277); on a decoder error the PCM buffer is left
    // untouched (the reference leaves whatever its aborted synthesis produced there, which is not defined by its inputs)
    *nSamplesOut = (int16_t)ns;
    if (ret < 0) return ret;
    memcpy(pcm, h->h_blk + SOLO_SINGLE_PCM_OFF, (size_t)ns * 2);
    return 0;
}

int32_t AGR_Sate_Decoder_Uninit(void* st) {
    if (!st) return -1;
    single_free((solo_single*)st);
    return 0;
}

// Conformance probes (tests only; device pointers, default stream, synchronous): the L0 vocabulary of solo_fix.h evaluated by the
// gfx950 build, element-wise; and SKP_Silk_sum_sqr_shift in its wave-cooperative form over `rows` rows of `len` <= 1024 samples.
int32_t solo_debug_l0(int32_t op, int32_t n, const int32_t* d_a, const int32_t* d_b, const int32_t* d_c, int32_t* d_out) {
    if (n <= 0 || !d_a || !d_b || !d_c || !d_out) return -1;
    hipLaunchKernelGGL(solo_l0_probe_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)0, op, n, d_a, d_b, d_c, d_out);
    SOLO_CHECK(hipGetLastError());
    SOLO_CHECK(hipDeviceSynchronize());
    return 0;
}
int32_t solo_debug_sum_sqr_shift(const int16_t* d_x, int32_t rows, int32_t len, int32_t stride, int32_t odd_start, int32_t* d_energy, int32_t* d_shift) {
    if (rows <= 0 || len <= 0 || len > 1024 || !d_x || !d_energy || !d_shift) return -1;
    hipLaunchKernelGGL(solo_sum_sqr_probe_kernel, dim3(rows), dim3(64), 0, (hipStream_t)0, d_x, len, stride, odd_start, d_energy, d_shift);
    SOLO_CHECK(hipGetLastError());
    SOLO_CHECK(hipDeviceSynchronize());
    return 0;
}

#ifdef SOLO_WITH_ENCODER
// The quantiser kernel ALONE (tests/test_nsq_taps.py): h_in = SxNsqIn[n_streams][n_packets][2] as recorded from the reference's
// SKP_Silk_NSQ_del_dec calls, h_out = SxNsqOut[n_streams][n_packets][2]; host pointers; freshly initialised streams; synchronous.
int32_t solo_debug_nsq(int32_t n_streams, int32_t n_packets, const void* h_in, void* h_out) {
    if (n_streams <= 0 || n_packets <= 0 || !h_in || !h_out) return -1;
    const solo_enc_ops* ops = solo_nb_enc_ops();
    const size_t sz_in = (size_t)n_streams * n_packets * 2 * ops->nsq_in_bytes, sz_out = (size_t)n_streams * n_packets * 2 * ops->nsq_out_bytes;
    void *st = NULL, *d_in = NULL, *d_out = NULL, *ring = NULL;
    int32_t rc = -1;
    if (hipMalloc(&st, ops->state_bytes * (size_t)n_streams) == hipSuccess && hipMalloc(&d_in, sz_in) == hipSuccess && hipMalloc(&d_out, sz_out) == hipSuccess &&
        hipMalloc(&ring, ops->nsq_ring_bytes(n_streams)) == hipSuccess && hipMemset(d_out, 0, sz_out) == hipSuccess &&
        ops->init(st, n_streams, 12000, 0, 0, 0, 2, (hipStream_t)0) == hipSuccess && hipMemcpy(d_in, h_in, sz_in, hipMemcpyHostToDevice) == hipSuccess &&
        ops->nsq(st, d_in, d_out, n_streams, n_packets, 0, n_packets, NULL, ring, NULL) == 0 && hipDeviceSynchronize() == hipSuccess &&
        hipMemcpy(h_out, d_out, sz_out, hipMemcpyDeviceToHost) == hipSuccess)
        rc = (int32_t)ops->nsq_out_bytes;                       // (the caller checks its idea of the record size)
    (void)hipFree(st); (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(ring);
    return rc;
}
#endif

#if defined(SX_PROF)
// debug builds only: read (and clear) the per-section cycle counters of the decoder kernels (tools/prof_dec.py; the encoder kernels'
// counters live in their own translation unit: solo_debug_prof_enc, solo_enc_k.hip)
int32_t solo_debug_prof(unsigned long long* out64, int32_t reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_sx_prof), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sx_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#if defined(SX_STOPS)
// debug builds only (tools/debug/analysis_sections.py dec): where the decoder's waves of the next launches end (0: nowhere), and how often
// each site was passed by the launches that ran through
int32_t solo_debug_stop_dec(int32_t site_hit) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sx_stop), &site_hit, sizeof(site_hit)) == hipSuccess ? 0 : -1; }
int32_t solo_debug_site_hits_dec(unsigned long long* out64, int32_t reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_sx_site_hits), 128 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[128] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sx_site_hits), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

}  // extern "C"
