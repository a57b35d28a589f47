// solo_enc_ops.h -- what the host-side pipeline (solo_api.hip) needs of one build of the encoder kernels (solo_enc_k.hip: 16 kHz API
// rate, solo_enc_k_wb.hip: 32 kHz): record sizes and launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#ifndef SOLO_ENC_OPS_DEFINED
#define SOLO_ENC_OPS_DEFINED
struct solo_enc_ops {
    size_t state_bytes, nsq_in_bytes, nsq_out_bytes, code_in_bytes;      // sizeof SxEncStream / SxNsqIn / SxNsqOut / SxCodeIn
    int packet_samples;
    hipError_t (*init)(void* states, int n_streams, int silk_rate_bps, int useMDIndex, int hb_joint, int useDTX, int frames_per_packet, hipStream_t s);
    // ---- launch per chunk ----
    hipError_t (*analysis)(void* states, const int16_t* pcm, int n_streams, int n_packets, int p0, int pc, void* nsq_in, void* code_in, hipStream_t s);
    int (*nsq)(void* states, const void* in, void* out, int n_streams, int n_packets, int p0, int pc, unsigned int* started, void* ring, void* hip_stream);
    // third stage of a chunk: high band (one wavefront per stream), then range coder + payload assembly (one lane per description)
    hipError_t (*coding)(void* states, const void* code_in, const void* nsq_out, int n_streams, int n_packets, int p0, int pc, int slot,
                         uint8_t* bits, int16_t* nbytes, int32_t* status, void* rc_scratch, hipStream_t s);
    size_t (*rc_scratch_bytes)(int n_streams, int pc);                   // scratch of one coding launch (pc packets per stream)
    int (*nsq_workgroups)(int n_streams);                                // workgroups of one quantiser launch (they count into the residency gate)
    size_t (*nsq_ring_bytes)(int n_streams);                             // emission-ring scratch of one quantiser launch
    // ---- persistent pipeline (solo_enc_kernels.h: solo_enc_front_kernel, solo_nsq_row.hip: solo_nsq_persist_kernel) ----
    hipError_t (*front)(void* states, const int16_t* pcm, int n_streams, int n_packets, void* nsq_in, void* code_in, const void* nsq_out,
                        unsigned int* ana_flag, const unsigned int* nsq_flag, unsigned int* prog, unsigned int ticket0, int mode,
                        unsigned int final_wait_ticks, int slot, uint8_t* bits, int16_t* nbytes, int32_t* status, void* scratch,
                        unsigned int* started, hipStream_t s);
    int (*nsq_persist)(void* states, const void* in, void* out, int n_streams, int n_packets, unsigned int* started, void* ring,
                       const unsigned int* ana_flag, unsigned int* nsq_flag, unsigned int ticket0, unsigned int* err, void* stage, void* hip_stream);
    size_t (*front_scratch_bytes)(int n_streams);                        // byte buffers of the front kernel's in-wave range coder
    int (*nsq_persist_workgroups)(int n_streams);                        // workgroups of the persistent quantiser's launch (residency gate)
    size_t (*nsq_stage_bytes)(int n_streams);                            // the persistent quantiser's private copies of the records it works on
    int front_waves;                                                     // streams (= wavefronts) per front workgroup
    int front_per_cu;                                                    // front wavefronts that a compute unit holds beside the quantiser's
};
#endif
