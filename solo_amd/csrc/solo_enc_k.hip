// solo_enc_k.hip -- the encoder's analysis / high-band / range-coding kernels and the launch table of the build
// (solo_enc_kernels.h).  A translation unit of their own, apart from the decoder's.
#include <hip/hip_runtime.h>
#include "solo_enc_kernels.h"
#if SX_FS_KHZ == 8
extern "C" const solo_enc_ops* solo_nb_enc_ops() { return &solo_enc_ops_table; }
#else
extern "C" const solo_enc_ops* solo_wb_enc_ops() { return &solo_enc_ops_table_wb; }
#endif

#if defined(SX_PROF) && SX_FS_KHZ == 8
// debug builds only: read (and clear) the per-section cycle counters of this translation unit's kernels (tools/prof_sections.py) and
// the histogram of the analysis waves' lifetimes (4 SIMDs x 64 bins of 50 us)
extern "C" int32_t solo_debug_prof_enc(unsigned long long* out64, int32_t reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_sx_prof), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sx_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int32_t solo_debug_hist(unsigned long long* out256, int32_t reset) {
    if (hipMemcpyFromSymbol(out256, HIP_SYMBOL(g_sx_hist), 256 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[256] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sx_hist), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#if defined(SX_STOPS) && SX_FS_KHZ == 8
// debug builds only (tools/debug/analysis_sections.py): where the analysis waves of the next launches end (0: nowhere), and how often each
// site was passed by the launches that ran through
extern "C" int32_t solo_debug_stop(int32_t site_hit) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sx_stop), &site_hit, sizeof(site_hit)) == hipSuccess ? 0 : -1; }
extern "C" int32_t solo_debug_site_hits(unsigned long long* out64, int32_t reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_sx_site_hits), 128 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[128] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sx_site_hits), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

