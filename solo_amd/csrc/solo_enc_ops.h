// solo_enc_ops.h -- what the host-side pipeline (solo_api.hip) needs of one build of the encoder kernels (solo_enc_k.hip: 16 kHz API
// rate, solo_enc_k_wb.hip: 32 kHz): record sizes and launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
// what the host-side pipeline (solo_api.hip) needs of one build: record sizes and launchers
#ifndef SOLO_ENC_OPS_DEFINED
#define SOLO_ENC_OPS_DEFINED
struct solo_enc_ops {
    size_t state_bytes, nsq_in_bytes, nsq_out_bytes, code_in_bytes;      // sizeof SxEncStream / SxNsqIn / SxNsqOut / SxCodeIn
    int packet_samples;
    hipError_t (*init)(void* states, int n_streams, int silk_rate_bps, int useMDIndex, int hb_joint, int useDTX, int frames_per_packet, hipStream_t s);
    hipError_t (*analysis)(void* states, const int16_t* pcm, int n_streams, int n_packets, int p0, int pc, void* nsq_in, void* code_in, hipStream_t s);
    int (*nsq)(void* states, const void* in, void* out, int n_streams, int n_packets, int p0, int pc, unsigned int* started, void* ring, void* hip_stream);
    // entropy coding of the descriptions (lane per description) into rc_scratch, then high band + payload assembly
    hipError_t (*coding)(void* states, const void* code_in, const void* nsq_out, int n_streams, int n_packets, int p0, int pc, int slot,
                         uint8_t* bits, int16_t* nbytes, int32_t* status, void* rc_scratch, int order, hipStream_t s);
    // (order 1: high band first, then the range coder, then a light assembly kernel -- the default of the batched pipeline; 0: range coder first)
    // the two halves of `coding` as launches of their own (SOLO_ENC_RC_STREAM=1: the range coder of chunk c + 1 on a fourth stream, next to
    // the high band of chunk c; they exchange rc_scratch, of which the pipeline then keeps two)
    hipError_t (*rc)(const void* states, const void* code_in, const void* nsq_out, int n_streams, int n_packets, int p0, int pc, void* rc_scratch, hipStream_t s);
    hipError_t (*hb_out)(void* states, const void* code_in, const void* nsq_out, int n_streams, int n_packets, int p0, int pc, int slot,
                         uint8_t* bits, int16_t* nbytes, int32_t* status, const void* rc_scratch, hipStream_t s);
    // the launches of order 1 one by one: high band only (its bytes into rc_scratch), payload assembly
    hipError_t (*hb)(void* states, const void* code_in, const void* nsq_out, int n_streams, int n_packets, int p0, int pc, void* rc_scratch, hipStream_t s);
    hipError_t (*out)(const void* states, const void* code_in, int n_streams, int n_packets, int p0, int pc, int slot, uint8_t* bits, int16_t* nbytes,
                      int32_t* status, const void* rc_scratch, hipStream_t s);
    size_t (*rc_scratch_bytes)(int n_streams, int pc);                   // scratch of one coding launch (pc packets per stream)
    int (*nsq_workgroups)(int n_streams);                                // workgroups of one quantiser launch (they count into the residency gate)
    size_t (*nsq_ring_bytes)(int n_streams);                             // emission-ring scratch of one quantiser launch
};
#endif
