// solo_cdf.h -- LDS mirror of the entropy-coding tables.  The range coder is a serial chain of dependent table
// look-ups (two CDF entries per symbol); served from HBM/L2 each look-up costs a memory round trip, so the coding
// phase first copies the ~2.7 KB of tables it needs into LDS (lane-strided) and codes from there.
// Source of the values: solo_tables.inc (generated from the compiled reference, tools/gen_tables.py).
#pragma once
#include "solo_common.h"

#define SX_CDF_LIST(X)                                                                                                  \
    X(u16, cdf_gain, 130) X(u16, cdf_delta_gain, 46) X(u16, cdf_md_delta_gain, 9) X(u16, cdf_type_offset, 5)            \
    X(u16, cdf_type_offset_joint, 20) X(u16, cdf_fs, 5) X(u16, cdf_nlsf_interp, 6) X(u16, cdf_pitch_lag, SX_N_PITCH_LAG_CDF)        \
    X(u16, cdf_pitch_contour, SX_N_PITCH_CONTOUR_CDF) X(u16, cdf_ltp_per, 4) X(u16, cdf_ltp_gain0, 11) X(u16, cdf_ltp_gain1, 21)         \
    X(u16, cdf_ltp_gain2, 41) X(u16, cdf_ltpscale, 4) X(u16, cdf_seed, 5) X(u16, cdf_rate_levels, 20)                   \
    X(u16, cdf_pulses_per_block, 210) X(u16, cdf_shell0, 33) X(u16, cdf_shell1, 52) X(u16, cdf_shell2, 102)             \
    X(u16, cdf_shell3, 207) X(u16, shell_offsets, 19) X(u16, cdf_lsb, 3) X(u16, cdf_sign, 36) X(u16, cdf_vadflag, 3)    \
    X(u16, cdf_frame_term, 5) X(u16, cdf_mdindex, 3) X(u16, nlsf_cb0_cdf, SX_N_NLSF_CB0_CDF) X(u16, nlsf_cb1_cdf, SX_N_NLSF_CB1_CDF)
#define SX_CDF_LIST_ENC(X) X(i16, bits_rate_levels_Q6, 18) X(i16, bits_pulses_per_block_Q6, 180)    // rate estimation: encoder only

#define X(type, name, n) type name[((n) + 1) & ~1];
struct SxCdfDec { SX_CDF_LIST(X) };                 // what the decoder mirrors: a layout prefix of SxCdf
struct SxCdf { SX_CDF_LIST(X) SX_CDF_LIST_ENC(X) };
#undef X

// lane-strided copy HBM -> LDS; caller must wv_sync() before use
#define X(type, name, n) SX_PAR(i, n) c->name[i] = T_##name[i];
SX_HD void sx_cdf_load(SxCdf* c) { SX_CDF_LIST(X) SX_CDF_LIST_ENC(X) }
SX_HD void sx_cdf_load_dec(SxCdfDec* c) { SX_CDF_LIST(X) }
#undef X
