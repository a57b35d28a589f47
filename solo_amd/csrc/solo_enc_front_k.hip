// solo_enc_front_k.hip -- the persistent schedule's front kernel (solo_enc_kernels.h) in a translation unit of its own.
//  * wv_sync() here is wave-local (solo_wave.h): a workgroup holds sixteen wavefronts that each follow their own stream and never meet at
//    a barrier;
//  * the stage functions the kernel calls are compiled once more here, with THIS kernel's attributes.  In one translation unit with the
//    analysis kernel their code changed for both (the backend derives a function's register budget and workgroup-size assumptions from
//    all its callers): sx_nlsf_msvq_encode grew from 11.8 to 13.7 KB, sx_pitch_analysis_core from 15.9 to 17.1 KB, the analysis kernel's
//    private segment from 200 to 272 bytes -- and the launch-per-chunk schedule, which never runs this kernel, lost 4 %.
#define SX_SYNC_WAVE_ONLY 1
#define SX_TU_FRONT 1
#include <hip/hip_runtime.h>
#include "solo_enc_kernels.h"

#if defined(SX_PIPE_TRACE) && SX_FS_KHZ == 8
extern "C" int32_t solo_debug_front_trace(unsigned long long* out, int32_t n_streams) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sx_front_trace), (size_t)n_streams * 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
