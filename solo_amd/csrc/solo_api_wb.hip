// solo_api_wb.hip -- the decoder and encoder (analysis / coding) kernels compiled for the 32 kHz API rate (`samplerate == 32000` in USER_Ctrl_dec,
// libBWE/AGR_BWE_SDK_API.c:197): 16 kHz bands, SILK running wide band (fs_kHz = 16, LPC order 16, order-16 NLSF codebooks,
// stage-3 pitch contours, 320-sample frames), 1280-sample packets.  Same source as the 16 kHz build (solo_dec.h), other
// compile-time constants; solo_api.hip dispatches here when a handle's decoder control asks for it.
#define SX_FS_KHZ 16
#include "solo_dec_kernels.h"

extern "C" {
size_t solo_wb_dec_state_bytes() { return solo_dec_state_bytes_wb(); }
hipError_t solo_wb_dec_launch_init(void* states, int n_streams, int hb_joint, hipStream_t s) {
    return solo_dec_launch_init_wb(states, n_streams, hb_joint, s);
}
hipError_t solo_wb_dec_launch(void* states, const uint8_t* bits, const int16_t* nbytes, const uint8_t* recv, int n_streams, int n_packets, int slot,
                              int useMDIndex, int16_t* pcm, int32_t* status, hipStream_t s) {
    return solo_dec_launch_wb(states, bits, nbytes, recv, n_streams, n_packets, slot, useMDIndex, pcm, status, s);
}
hipError_t solo_wb_dec_launch_extract(const void* states, const uint8_t* bits, const int16_t* nbytes, const uint8_t* recv, int n_streams, int n_packets,
                                      int p0, int pc, int slot, int useMDIndex, void* recs, hipStream_t s) {
    return solo_dec_launch_extract_wb(states, bits, nbytes, recv, n_streams, n_packets, p0, pc, slot, useMDIndex, recs, s);
}
hipError_t solo_wb_dec_launch_synth(void* states, const uint8_t* bits, const int16_t* nbytes, const uint8_t* recv, int n_streams, int n_packets, int p0,
                                    int pc, int slot, int useMDIndex, const void* recs, int16_t* pcm, int32_t* status, hipStream_t s) {
    return solo_dec_launch_synth_wb(states, bits, nbytes, recv, n_streams, n_packets, p0, pc, slot, useMDIndex, recs, pcm, status, s);
}
size_t solo_wb_dec_extracted_bytes() { return solo_dec_extracted_bytes_wb(); }
hipError_t solo_wb_dec_launch_split(void* states, const uint8_t* descA, const int16_t* lenA, const uint8_t* descB, const int16_t* lenB, int n_streams,
                                    int n_packets, int slot, int useMDIndex, int16_t* pcm, int32_t* status, hipStream_t s) {
    return solo_dec_launch_split_wb(states, descA, lenA, descB, lenB, n_streams, n_packets, slot, useMDIndex, pcm, status, s);
}
hipError_t solo_wb_dec_launch_ring(void* states, const uint8_t* ring, uint32_t* lens, int32_t* play, int n_streams, int n_packets, int depth, int slot,
                                   int useMDIndex, int16_t* pcm, int32_t* status, hipStream_t s) {
    return solo_dec_launch_ring_wb(states, ring, lens, play, n_streams, n_packets, depth, slot, useMDIndex, pcm, status, s);
}
hipError_t solo_wb_dec_launch_raw(void* state, const uint8_t* bits, int n0, int n1, int lostflag, int useMDIndex, int16_t* pcm, int32_t* status,
                                  hipStream_t s) {
    return solo_dec_launch_raw_wb(state, bits, n0, n1, lostflag, useMDIndex, pcm, status, s);
}
}
