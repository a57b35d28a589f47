#!/usr/bin/env python3
"""Regenerate the numeric tables (entropy-coder CDFs, NLSF / LTP / high-band codebooks, QMF taps ...)
used by the oracle restatement and by the HIP kernels.

The VALUES are read out of the compiled reference (oracle/_ref/libsolo_ref_fix.so, built from
/root/reference by `make -C oracle ref`) through its symbol table, so no reference source text is
copied; only the live subset (8 kHz NB / LPC order 10 / high band order 8) is emitted, in this
repo's own naming and layout.  Run in the build container only:

    python tools/gen_tables.py

Writes  oracle/solo_oracle_tables.h   (plain C, `static const`)
        solo_amd/csrc/solo_tables.inc (qualified by SOLO_TAB, defined by the including .hip/.cpp)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libsolo_ref_fix.so")

I16, U16, I32 = ("int16_t", C.c_int16), ("uint16_t", C.c_uint16), ("int32_t", C.c_int32)

# (reference symbol, element type, our name, reference file the symbol is defined in)
TABLES = [
    # --- entropy coder CDFs (SKP_Silk_tables_*.c) ---
    ("SKP_Silk_gain_CDF", U16, "cdf_gain", "SKP_Silk_tables_gain.c:35"),
    ("SKP_Silk_delta_gain_CDF", U16, "cdf_delta_gain", "SKP_Silk_tables_gain.c:64"),
    ("SKP_Silk_md_delta_gain_CDF", U16, "cdf_md_delta_gain", "SKP_Silk_tables_gain.c:75"),
    ("SKP_Silk_type_offset_CDF", U16, "cdf_type_offset", "SKP_Silk_tables_type_offset.c"),
    ("SKP_Silk_type_offset_joint_CDF", U16, "cdf_type_offset_joint", "SKP_Silk_tables_type_offset.c"),
    ("SKP_Silk_SamplingRates_CDF", U16, "cdf_fs", "SKP_Silk_tables_other.c"),
    ("SKP_Silk_NLSF_interpolation_factor_CDF", U16, "cdf_nlsf_interp", "SKP_Silk_tables_other.c"),
    ("SKP_Silk_pitch_lag_NB_CDF", U16, "cdf_pitch_lag_nb", "SKP_Silk_tables_pitch_lag.c"),
    ("SKP_Silk_pitch_contour_NB_CDF", U16, "cdf_pitch_contour_nb", "SKP_Silk_tables_pitch_lag.c"),
    ("SKP_Silk_LTP_per_index_CDF", U16, "cdf_ltp_per", "SKP_Silk_tables_LTP.c:30"),
    ("SKP_Silk_LTP_gain_CDF_0", U16, "cdf_ltp_gain0", "SKP_Silk_tables_LTP.c:37"),
    ("SKP_Silk_LTP_gain_CDF_1", U16, "cdf_ltp_gain1", "SKP_Silk_tables_LTP.c:42"),
    ("SKP_Silk_LTP_gain_CDF_2", U16, "cdf_ltp_gain2", "SKP_Silk_tables_LTP.c:48"),
    ("SKP_Silk_LTPscale_CDF", U16, "cdf_ltpscale", "SKP_Silk_tables_other.c:87"),
    ("SKP_Silk_Seed_CDF", U16, "cdf_seed", "SKP_Silk_tables_other.c"),
    ("SKP_Silk_rate_levels_CDF", U16, "cdf_rate_levels", "SKP_Silk_tables_pulses_per_block.c"),
    ("SKP_Silk_pulses_per_block_CDF", U16, "cdf_pulses_per_block", "SKP_Silk_tables_pulses_per_block.c"),
    ("SKP_Silk_shell_code_table0", U16, "cdf_shell0", "SKP_Silk_tables_pulses_per_block.c"),
    ("SKP_Silk_shell_code_table1", U16, "cdf_shell1", "SKP_Silk_tables_pulses_per_block.c"),
    ("SKP_Silk_shell_code_table2", U16, "cdf_shell2", "SKP_Silk_tables_pulses_per_block.c"),
    ("SKP_Silk_shell_code_table3", U16, "cdf_shell3", "SKP_Silk_tables_pulses_per_block.c"),
    ("SKP_Silk_shell_code_table_offsets", U16, "shell_offsets", "SKP_Silk_tables_pulses_per_block.c"),
    ("SKP_Silk_lsb_CDF", U16, "cdf_lsb", "SKP_Silk_tables_other.c:84"),
    ("SKP_Silk_sign_CDF", U16, "cdf_sign", "SKP_Silk_tables_sign.c"),
    ("SKP_Silk_vadflag_CDF", U16, "cdf_vadflag", "SKP_Silk_tables_other.c:91"),
    ("SKP_Silk_FrameTermination_CDF", U16, "cdf_frame_term", "SKP_Silk_tables_other.c"),
    ("SKP_Silk_writeMDIndex_CDF", U16, "cdf_mdindex", "SKP_Silk_tables_other.c:95"),
    # bit-cost tables used by the encoder's rate estimates
    ("SKP_Silk_rate_levels_BITS_Q6", I16, "bits_rate_levels_Q6", "SKP_Silk_tables_pulses_per_block.c"),
    ("SKP_Silk_pulses_per_block_BITS_Q6", I16, "bits_pulses_per_block_Q6", "SKP_Silk_tables_pulses_per_block.c"),
    ("SKP_Silk_max_pulses_table", I32, "max_pulses", "SKP_Silk_tables_pulses_per_block.c"),
    ("SKP_Silk_LTP_gain_BITS_Q6_0", I16, "bits_ltp_gain0_Q6", "SKP_Silk_tables_LTP.c:63"),
    ("SKP_Silk_LTP_gain_BITS_Q6_1", I16, "bits_ltp_gain1_Q6", "SKP_Silk_tables_LTP.c:68"),
    ("SKP_Silk_LTP_gain_BITS_Q6_2", I16, "bits_ltp_gain2_Q6", "SKP_Silk_tables_LTP.c:74"),
    # --- codebooks ---
    ("SKP_Silk_LTP_gain_vq_0_Q14", I16, "ltp_vq0_Q14", "SKP_Silk_tables_LTP.c:94"),
    ("SKP_Silk_LTP_gain_vq_1_Q14", I16, "ltp_vq1_Q14", "SKP_Silk_tables_LTP.c:128"),
    ("SKP_Silk_LTP_gain_vq_2_Q14", I16, "ltp_vq2_Q14", "SKP_Silk_tables_LTP.c:192"),
    ("SKP_Silk_LTPScales_table_Q14", I16, "ltp_scales_Q14", "SKP_Silk_tables_other.c"),
    ("LTPScaleThresholds_Q15", I16, "ltp_scale_thresholds_Q15", "SKP_Silk_LTP_scale_ctrl_FIX.c:33"),
    ("SKP_Silk_NLSF_MSVQ_CB0_10_Q15", I16, "nlsf_cb0_Q15", "SKP_Silk_tables_NLSF_CB0_10.c:267"),
    ("SKP_Silk_NLSF_MSVQ_CB0_10_rates_Q5", I16, "nlsf_cb0_rates_Q5", "SKP_Silk_tables_NLSF_CB0_10.c:188"),
    ("SKP_Silk_NLSF_MSVQ_CB0_10_CDF", U16, "nlsf_cb0_cdf", "SKP_Silk_tables_NLSF_CB0_10.c:38"),
    ("SKP_Silk_NLSF_MSVQ_CB0_10_CDF_middle_idx", I32, "nlsf_cb0_cdf_mid", "SKP_Silk_tables_NLSF_CB0_10.c:178"),
    ("SKP_Silk_NLSF_MSVQ_CB0_10_ndelta_min_Q15", I32, "nlsf_cb0_ndelta_min_Q15", "SKP_Silk_tables_NLSF_CB0_10.c:252"),
    ("SKP_Silk_NLSF_MSVQ_CB1_10_Q15", I16, "nlsf_cb1_Q15", "SKP_Silk_tables_NLSF_CB1_10.c:195"),
    ("SKP_Silk_NLSF_MSVQ_CB1_10_rates_Q5", I16, "nlsf_cb1_rates_Q5", "SKP_Silk_tables_NLSF_CB1_10.c:140"),
    ("SKP_Silk_NLSF_MSVQ_CB1_10_CDF", U16, "nlsf_cb1_cdf", "SKP_Silk_tables_NLSF_CB1_10.c:38"),
    ("SKP_Silk_NLSF_MSVQ_CB1_10_CDF_middle_idx", I32, "nlsf_cb1_cdf_mid", "SKP_Silk_tables_NLSF_CB1_10.c:130"),
    ("SKP_Silk_NLSF_MSVQ_CB1_10_ndelta_min_Q15", I32, "nlsf_cb1_ndelta_min_Q15", "SKP_Silk_tables_NLSF_CB1_10.c:180"),
    ("SKP_Silk_LSFCosTab_FIX_Q12", I32, "lsf_cos_Q12", "SKP_Silk_LSF_cos_table.c:31"),
    ("SKP_Silk_CB_lags_stage2", I16, "pitch_cb_stage2", "SKP_Silk_pitch_est_tables.c:35"),
    ("SKP_Silk_CB_lags_stage3", I16, "pitch_cb_stage3", "SKP_Silk_pitch_est_tables.c:43"),
    ("SKP_Silk_Lag_range_stage3", I16, "pitch_lag_range_stage3", "SKP_Silk_pitch_est_tables.c:51"),
    ("SKP_Silk_cbk_sizes_stage3", I16, "pitch_cbk_sizes_stage3", "SKP_Silk_pitch_est_tables.c:76"),
    ("SKP_Silk_cbk_offsets_stage3", I16, "pitch_cbk_offsets_stage3", "SKP_Silk_pitch_est_tables.c:83"),
    ("SKP_Silk_Quantization_Offsets_Q10", I16, "quant_offsets_Q10", "SKP_Silk_tables_other.c:116"),
    # --- analysis-side constants ---
    ("TargetRate_table_NB", I32, "target_rate_nb", "SKP_Silk_tables_other.c:37"),
    ("SNR_table_Q1", I32, "snr_table_Q1", "SKP_Silk_tables_other.c:49"),
    ("SNR_table_one_bit_per_sample_Q7", I32, "snr_one_bit_Q7", "SKP_Silk_tables_other.c:53"),
    ("sigm_LUT_slope_Q10", I32, "sigm_slope_Q10", "SKP_Silk_sigm_Q15.c:42"),
    ("sigm_LUT_pos_Q15", I32, "sigm_pos_Q15", "SKP_Silk_sigm_Q15.c:46"),
    ("sigm_LUT_neg_Q15", I32, "sigm_neg_Q15", "SKP_Silk_sigm_Q15.c:50"),
    ("tiltWeights", I32, "vad_tilt_weights", "SKP_Silk_VAD.c:70"),
    ("freq_table_Q16", I16, "sine_win_freq_Q16", "SKP_Silk_apply_sine_window.c"),
    ("SKP_Silk_resampler_down2_0", I16, "down2_c0", "SKP_Silk_resampler_rom.c"),
    ("SKP_Silk_resampler_down2_1", I16, "down2_c1", "SKP_Silk_resampler_rom.c"),
    # --- high band / QMF (libBWE) ---
    ("AGR_Sate_qmf_coeffs_fix", I16, "qmf_taps", "AGR_BWE_tables_qmf.c:4"),
    ("AGR_Sate_highband_lsp_cdbk1_fix", I16, "hb_lsp_cb1", "AGR_BWE_tables_highband_coeff.c:5"),
    ("AGR_Sate_highband_lsp_cdbk2_fix", I16, "hb_lsp_cb2", "AGR_BWE_tables_highband_coeff.c:265"),
    ("AGR_Sate_highband_gain_cdbk_fix", I16, "hb_gain_cb", "AGR_BWE_tables_highband_coeff.c:285"),
]

# scalar ints of the reference ("..._offset" = start index of the decoder's CDF search)
SCALARS = [
    ("SKP_Silk_gain_CDF_offset", "CDF_MID_GAIN"), ("SKP_Silk_delta_gain_CDF_offset", "CDF_MID_DELTA_GAIN"),
    ("SKP_Silk_md_delta_gain_CDF_offset", "CDF_MID_MD_DELTA_GAIN"),
    ("SKP_Silk_type_offset_CDF_offset", "CDF_MID_TYPE_OFFSET"), ("SKP_Silk_SamplingRates_offset", "CDF_MID_FS"),
    ("SKP_Silk_NLSF_interpolation_factor_offset", "CDF_MID_NLSF_INTERP"),
    ("SKP_Silk_pitch_lag_NB_CDF_offset", "CDF_MID_PITCH_LAG_NB"),
    ("SKP_Silk_pitch_contour_NB_CDF_offset", "CDF_MID_PITCH_CONTOUR_NB"),
    ("SKP_Silk_LTP_per_index_CDF_offset", "CDF_MID_LTP_PER"), ("SKP_Silk_LTPscale_offset", "CDF_MID_LTPSCALE"),
    ("SKP_Silk_Seed_offset", "CDF_MID_SEED"), ("SKP_Silk_rate_levels_CDF_offset", "CDF_MID_RATE_LEVELS"),
    ("SKP_Silk_pulses_per_block_CDF_offset", "CDF_MID_PULSES_PER_BLOCK"),
    ("SKP_Silk_vadflag_offset", "CDF_MID_VADFLAG"), ("SKP_Silk_FrameTermination_offset", "CDF_MID_FRAME_TERM"),
    ("SKP_Silk_writeMDIndex_offset", "CDF_MID_MDINDEX"),
    ("SKP_Silk_LTP_gain_middle_avg_RD_Q14", "LTP_GAIN_MIDDLE_AVG_RD_Q14"),
]
SCALAR_ARRAYS = [("SKP_Silk_LTP_gain_CDF_offsets", I32, "cdf_mid_ltp_gain", "SKP_Silk_tables_LTP.c:57"),
                 ("SKP_Silk_LTP_vq_sizes", I32, "ltp_vq_sizes", "SKP_Silk_tables_LTP.c:322")]


# --- 32 kHz mode: SILK runs wide band (fs_kHz = 16, LPC order 16); the tables that depend on the internal rate ---
TABLES_WB = [
    ("SKP_Silk_pitch_lag_WB_CDF", U16, "cdf_pitch_lag_wb", "SKP_Silk_tables_pitch_lag.c"),
    ("SKP_Silk_pitch_contour_CDF", U16, "cdf_pitch_contour_wb", "SKP_Silk_tables_pitch_lag.c"),
    ("SKP_Silk_NLSF_MSVQ_CB0_16_Q15", I16, "nlsf16_cb0_Q15", "SKP_Silk_tables_NLSF_CB0_16.c"),
    ("SKP_Silk_NLSF_MSVQ_CB0_16_rates_Q5", I16, "nlsf16_cb0_rates_Q5", "SKP_Silk_tables_NLSF_CB0_16.c"),
    ("SKP_Silk_NLSF_MSVQ_CB0_16_CDF", U16, "nlsf16_cb0_cdf", "SKP_Silk_tables_NLSF_CB0_16.c"),
    ("SKP_Silk_NLSF_MSVQ_CB0_16_CDF_middle_idx", I32, "nlsf16_cb0_cdf_mid", "SKP_Silk_tables_NLSF_CB0_16.c"),
    ("SKP_Silk_NLSF_MSVQ_CB0_16_ndelta_min_Q15", I32, "nlsf16_cb0_ndelta_min_Q15", "SKP_Silk_tables_NLSF_CB0_16.c"),
    ("SKP_Silk_NLSF_MSVQ_CB1_16_Q15", I16, "nlsf16_cb1_Q15", "SKP_Silk_tables_NLSF_CB1_16.c"),
    ("SKP_Silk_NLSF_MSVQ_CB1_16_rates_Q5", I16, "nlsf16_cb1_rates_Q5", "SKP_Silk_tables_NLSF_CB1_16.c"),
    ("SKP_Silk_NLSF_MSVQ_CB1_16_CDF", U16, "nlsf16_cb1_cdf", "SKP_Silk_tables_NLSF_CB1_16.c"),
    ("SKP_Silk_NLSF_MSVQ_CB1_16_CDF_middle_idx", I32, "nlsf16_cb1_cdf_mid", "SKP_Silk_tables_NLSF_CB1_16.c"),
    ("SKP_Silk_NLSF_MSVQ_CB1_16_ndelta_min_Q15", I32, "nlsf16_cb1_ndelta_min_Q15", "SKP_Silk_tables_NLSF_CB1_16.c"),
    ("TargetRate_table_WB", I32, "target_rate_wb", "SKP_Silk_tables_other.c"),
]
SCALARS_WB = [("SKP_Silk_pitch_lag_WB_CDF_offset", "CDF_MID_PITCH_LAG_WB"), ("SKP_Silk_pitch_contour_CDF_offset", "CDF_MID_PITCH_CONTOUR_WB")]


def symtab():
    out = subprocess.check_output(["nm", "-S", "--defined-only", LIB], text=True)
    tab = {}
    for ln in out.splitlines():
        p = ln.split()
        if len(p) == 4:
            tab[p[3]] = (int(p[0], 16), int(p[1], 16))
    return tab


def main():
    if not os.path.exists(LIB):
        sys.exit("build oracle/_ref first (make -C oracle ref)")
    lib = C.CDLL(LIB)
    tab = symtab()
    # load base = address of an exported symbol - its nm value
    anchor = "SKP_Silk_gain_CDF"
    base = C.addressof(C.c_char.in_dll(lib, anchor)) - tab[anchor][0]

    def read(sym, ct):
        addr, size = tab[sym]
        n = size // C.sizeof(ct)
        return list((ct * n).from_address(base + addr))

    body = []
    for sym, (cname, ct), ours, where in TABLES + SCALAR_ARRAYS:
        vals = read(sym, ct)
        body.append("/* %s  (values of %s, %s) */" % (ours, sym, where))
        body.append("SOLO_TAB %s T_%s[%d] = {" % (cname, ours, len(vals)))
        for i in range(0, len(vals), 12):
            body.append("    " + ", ".join(str(v) for v in vals[i:i + 12]) + ",")
        body.append("};")
    body.append("")
    for sym, ours in SCALARS:
        v = read(sym, C.c_int32)[0]
        body.append("#define T_%s %d  /* %s */" % (ours, v, sym))
    # NLSF MSVQ stage structure: derive from the reference's stage-info structs
    # struct { int32 nVectors; (pad) ptr CB; ptr rates; } = 24 bytes on LP64
    for cb, n_st in (("CB0_10", 6), ("CB1_10", 6)):
        addr, size = tab["SKP_Silk_NLSF_%s_Stage_info" % cb]
        assert size == 24 * n_st
        nv = [C.c_int32.from_address(base + addr + 24 * s).value for s in range(n_st)]
        body.append("#define T_NLSF_%s_NVEC { %s }" % (cb[:3], ", ".join(map(str, nv))))
    text = "\n".join(body) + "\n"
    hdr = ("/* GENERATED by tools/gen_tables.py -- numeric tables read out of the compiled reference\n"
           "   (live 8 kHz NB / order-10 / high-band subset). Do not edit. */\n")
    with open(os.path.join(ROOT, "oracle", "solo_oracle_tables.h"), "w") as f:
        f.write(hdr + "#ifndef SOLO_ORACLE_TABLES_H\n#define SOLO_ORACLE_TABLES_H\n#include <stdint.h>\n"
                "#define SOLO_TAB static const\n" + text + "#endif\n")
    with open(os.path.join(ROOT, "solo_amd", "csrc", "solo_tables.inc"), "w") as f:
        f.write(hdr + "/* the including translation unit defines SOLO_TAB (e.g. `static __device__ const`) */\n" + text)
    print("wrote tables: %d arrays" % (len(TABLES) + len(SCALAR_ARRAYS)))
    body = []
    for sym, (cname, ct), ours, where in TABLES_WB:
        vals = read(sym, ct)
        body.append("/* %s  (values of %s, %s) */" % (ours, sym, where))
        body.append("SOLO_TAB %s T_%s[%d] = {" % (cname, ours, len(vals)))
        for i in range(0, len(vals), 12):
            body.append("    " + ", ".join(str(v) for v in vals[i:i + 12]) + ",")
        body.append("};")
    body.append("")
    for sym, ours in SCALARS_WB:
        body.append("#define T_%s %d  /* %s */" % (ours, read(sym, C.c_int32)[0], sym))
    for cb, n_st in (("CB0_16", 10), ("CB1_16", 10)):
        addr, size = tab["SKP_Silk_NLSF_%s_Stage_info" % cb]
        assert size == 24 * n_st
        nv = [C.c_int32.from_address(base + addr + 24 * s).value for s in range(n_st)]
        body.append("#define T_NLSF16_%s_NVEC { %s }" % (cb[:3], ", ".join(map(str, nv))))
    with open(os.path.join(ROOT, "solo_amd", "csrc", "solo_tables_wb.inc"), "w") as f:
        f.write("/* GENERATED by tools/gen_tables.py -- numeric tables read out of the compiled reference: the ones that\n"
                "   depend on SILK's internal rate, for the 32 kHz mode (fs_kHz = 16, order 16). Do not edit. */\n" + "\n".join(body) + "\n")
    print("wrote WB tables: %d arrays" % len(TABLES_WB))


if __name__ == "__main__":
    main()
