#!/usr/bin/env python3
"""Wave-instructions of the analysis kernel, section by section (debug build with -DSX_STOPS, solo_wave.h).

A launch whose waves END at site (id, hit) executes exactly the instructions that precede that site; the SQ instruction counters of
launches stopped at successive sites, subtracted, are the instructions of each section.  Every wave of a measurement launch encodes the
SAME packet of the SAME stream (so all of them take the same path and the counters are N x one wave's count); the result is averaged
over SAMPLES different (stream, packet) pairs.

  on the GPU box (tools/gpu_sections.sh):
    cd /tmp; SOLO_EXP_SKIP=3 SOLO_LIB_OVERRIDE=$ROOT/build/libsolo_stops.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
        SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $OUT/sections -o s --output-format csv -- \
        python $ROOT/tools/debug/analysis_sections.py run $OUT/sections_plan.json
    python tools/debug/analysis_sections.py report $OUT/sections_plan.json $OUT/sections > profiles/rNN_analysis_sections.txt
"""
import csv, ctypes, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# what ENDS at each site
NAMES = {0: "qmf split", 32: "vad", 1: "variable high-pass", 33: "pitch: window, autocorrelation, Schur, whitening filter",
         34: "pitch: decimation 8 -> 4 kHz, scaling", 35: "pitch: stage 1 (4 kHz correlations, candidates)",
         36: "pitch: stage 2 correlations (8 kHz)", 2: "pitch: stage 2 search, lags", 37: "noise shape: SNR, sparseness",
         38: "noise shape: window gains + windowing", 39: "noise shape: warped autocorrelation (4 subframes)",
         26: "noise shape: Schur .. coefficient limiting (rows)", 3: "noise shape: gains, tilt, harmonic shaping",
         40: "prefilter: warped analysis filter", 4: "prefilter: FIR, low-frequency + harmonic shaping, history",
         42: "pred: inverse gains + find_LTP", 43: "pred: quant_LTP_gains", 16: "pred: LTP scale + LTP analysis filter / unvoiced scaling",
         23: "burg: sum_sqr_shift", 24: "burg: first row of correlations", 25: "burg: recursion", 22: "find_LPC: (after second burg) expand",
         21: "find_LPC: A2NLSF of the second half", 53: "find_LPC interp: 4 x NLSF2A", 54: "find_LPC interp: 4 whitening filters",
         18: "find_LPC interp: energies, decision", 17: "find_LPC: final A2NLSF / exit", 46: "NLSF: weights (Laroia)",
         47: "msvq: stage head (first: codebook staging; later: survivor copy)", 52: "msvq: rate-distortion of the pairs",
         48: "msvq: sort network + selection rounds", 49: "msvq: survivor threshold", 51: "msvq: new residuals / paths of the last stage",
         50: "msvq: fluctuation reduction, winner, decode", 19: "NLSF2A x 2 (quantised)", 20: "residual energy", 5: "pred: tail",
         55: "(serial NLSF2A of the interpolation search is needed: up to here)", 56: "(serial NLSF2A of the quantised vector is needed: up to here)",
         6: "process gains", 7: "history + hand-over record", 8: "frame end"}
COUNTERS = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
NEVER = (63 << 8) | 1


def run(plan_path):
    import numpy as np, torch, solo_amd
    from solo_amd.synth import synth_stream
    N = int(os.environ.get("SECTIONS_N", "256"))
    samples = [(j, 3 + (j % 4)) for j in range(int(os.environ.get("SECTIONS_SAMPLES", "8")))]
    lib = solo_amd.load_library()
    lib.solo_debug_stop.argtypes = [ctypes.c_int32]
    lib.solo_debug_site_hits.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    b = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512)
    hits = (ctypes.c_ulonglong * 128)()
    plan, disp = [], 0          # disp = index of the next analysis dispatch (one per packet of every encode call)
    for j, W in samples:
        one = synth_stream(j, W + 1)
        x = torch.from_numpy(np.broadcast_to(one[None], (N, W + 1, 640)).copy()).cuda()
        warm, last = x[:, :W].contiguous(), x[:, W:].contiguous()
        b.reset(); lib.solo_debug_stop(0); b.encode(warm); torch.cuda.synchronize(); disp += W
        lib.solo_debug_site_hits(hits, 1)
        b.encode(last); torch.cuda.synchronize(); disp += 1
        lib.solo_debug_site_hits(hits, 1)
        per = {s: int(hits[s]) // N for s in range(64) if hits[s]}
        stops = [(s, h) for s, n in per.items() for h in range(1, n + 1)] + [(63, 1)]
        for s, h in stops:
            b.reset(); lib.solo_debug_stop(NEVER); b.encode(warm); disp += W
            lib.solo_debug_stop((s << 8) | h); b.encode(last); torch.cuda.synchronize()
            plan.append({"sample": j, "warm": W, "site": s, "hit": h, "dispatch": disp}); disp += 1
        lib.solo_debug_stop(0)
    json.dump({"n_streams": N, "samples": samples, "plan": plan}, open(plan_path, "w"))
    print("analysis_sections: %d measurement launches, %d analysis dispatches" % (len(plan), disp))


DEC_NAMES = {0: "symbols staged, de-quantisation (two lanes)", 1: "merge + inverse NSQ (+ frame-0 NLSF -> LPC)", 2: "decode_core (LTP / LPC synthesis)",
             3: "PLC update", 4: "outBuf + glue frames", 5: "CNG", 8: "high band: side information", 6: "high band: finish", 7: "QMF synthesis",
             9: "decode_core: excitation, subframe set-up", 10: "decode_core: LTP synthesis", 11: "decode_core: LPC synthesis", 12: "CNG: smoothing + excitation", 14: "(frame entry: glue between frames)", 13: "frame start .. shadow copy issued"}


COD_NAMES = {57: "high band: buffers staged, LPC blocks", 23: "burg: sum_sqr_shift", 24: "burg: first row of correlations", 25: "burg: recursion", 58: "high band: burg tail, expand",
             59: "high band: A2NLSF", 60: "high band: LSP weights + two-stage VQ", 61: "high band: NLSF -> LPC", 62: "high band: whitening filter",
             45: "high band: four gains", 9: "high band: history write-back", 11: "payload assembly"}


def run_cod(plan_path):
    """the same for the coding kernel (kernel tag 1 of the -DSX_STOPS encoder build): the whole encoder runs, only coding waves stop"""
    import numpy as np, torch, solo_amd
    from solo_amd.synth import synth_stream
    N = int(os.environ.get("SECTIONS_N", "256"))
    samples = [(j, 3 + (j % 4)) for j in range(int(os.environ.get("SECTIONS_SAMPLES", "8")))]
    lib = solo_amd.load_library()
    lib.solo_debug_stop.argtypes = [ctypes.c_int32]
    lib.solo_debug_site_hits.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    b = solo_amd.SoloBatch(N, encoder=True, decoder=False, slot_bytes=512)
    hits = (ctypes.c_ulonglong * 128)()
    plan, disp = [], 0
    never = (1 << 16) | NEVER
    for j, W in samples:
        one = synth_stream(j, W + 1)
        x = torch.from_numpy(np.broadcast_to(one[None], (N, W + 1, 640)).copy()).cuda()
        warm, last = x[:, :W].contiguous(), x[:, W:].contiguous()
        b.reset(); lib.solo_debug_stop(0); b.encode(warm); torch.cuda.synchronize(); disp += W
        lib.solo_debug_site_hits(hits, 1)
        b.encode(last); torch.cuda.synchronize(); disp += 1
        lib.solo_debug_site_hits(hits, 1)
        per = {s: int(hits[64 + s]) // N for s in range(64) if hits[64 + s]}
        stops = [(s, h) for s, n in per.items() for h in range(1, n + 1)] + [(62 + 0 * 1, 99)]
        for s, h in stops:
            b.reset(); lib.solo_debug_stop(never); b.encode(warm); disp += W
            lib.solo_debug_stop((1 << 16) | (s << 8) | h); b.encode(last); torch.cuda.synchronize()
            plan.append({"sample": j, "warm": W, "site": s if h != 99 else 63, "hit": h, "dispatch": disp}); disp += 1
        lib.solo_debug_stop(0)
    json.dump({"n_streams": N, "samples": samples, "plan": plan, "kernel": "enc_coding"}, open(plan_path, "w"))
    print("analysis_sections: %d measurement launches, %d coding dispatches" % (len(plan), disp))


def run_dec(plan_path):
    """the same for the decoder's synthesis kernel (-DSX_STOPS solo_api.hip): one decode call = one launch"""
    import numpy as np, torch, solo_amd
    from solo_amd.synth import synth_stream
    N = int(os.environ.get("SECTIONS_N", "256"))
    samples = [(j, 3 + (j % 4)) for j in range(int(os.environ.get("SECTIONS_SAMPLES", "8")))]
    lib = solo_amd.load_library()
    lib.solo_debug_stop_dec.argtypes = [ctypes.c_int32]
    lib.solo_debug_site_hits_dec.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.solo_debug_stop.argtypes = [ctypes.c_int32]
    lib.solo_debug_stop(0)
    b = solo_amd.SoloBatch(N, encoder=True, decoder=True, slot_bytes=512)
    hits = (ctypes.c_ulonglong * 128)()
    plan, disp = [], 0
    for j, W in samples:
        one = synth_stream(j, W + 1)
        x = torch.from_numpy(np.broadcast_to(one[None], (N, W + 1, 640)).copy()).cuda()
        b.reset(); lib.solo_debug_stop_dec(0)
        bits, nb, st = b.encode(x); torch.cuda.synchronize()
        wb, wn, lb, ln = bits[:, :W].contiguous(), nb[:, :W].contiguous(), bits[:, W:].contiguous(), nb[:, W:].contiguous()
        b.decode(wb, wn); torch.cuda.synchronize(); disp += 1
        lib.solo_debug_site_hits_dec(hits, 1)
        b.decode(lb, ln); torch.cuda.synchronize(); disp += 1
        lib.solo_debug_site_hits_dec(hits, 1)
        per = {s: int(hits[s]) // N for s in range(64) if hits[s]}
        stops = [(s, h) for s, n in per.items() for h in range(1, n + 1)] + [(63, 1)]
        for s, h in stops:
            b.reset(); lib.solo_debug_stop_dec(NEVER); b.decode(wb, wn); disp += 1
            lib.solo_debug_stop_dec((s << 8) | h); b.decode(lb, ln); torch.cuda.synchronize()
            plan.append({"sample": j, "warm": W, "site": s, "hit": h, "dispatch": disp}); disp += 1
        lib.solo_debug_stop_dec(0)
    json.dump({"n_streams": N, "samples": samples, "plan": plan, "kernel": "dec_synth"}, open(plan_path, "w"))
    print("analysis_sections: %d measurement launches, %d synthesis dispatches" % (len(plan), disp))


def report(plan_path, prof_dir):
    P = json.load(open(plan_path))
    N = P["n_streams"]
    KERNEL = P.get("kernel", "enc_analysis")
    global NAMES
    if KERNEL == "dec_synth": NAMES = DEC_NAMES
    if KERNEL == "enc_coding": NAMES = COD_NAMES
    rows = defaultdict(dict)          # dispatch id -> counter -> value (analysis kernel only)
    for f in glob.glob(os.path.join(prof_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    order = sorted(rows)
    per_site = defaultdict(lambda: defaultdict(float))
    totals = []
    for j, W in P["samples"]:
        ms = [m for m in P["plan"] if m["sample"] == j]
        pts = []
        for m in ms:
            c = rows[order[m["dispatch"]]]
            v = {k: c.get(k, 0.0) / N for k in COUNTERS}
            pts.append((sum(v.values()), m["site"], m["hit"], v))
        pts.sort(key=lambda t: t[0])
        prev = {k: 0.0 for k in COUNTERS}
        for tot, s, h, v in pts:
            for k in COUNTERS: per_site[s][k] += v[k] - prev[k]
            per_site[s]["hits"] += 1
            prev = v
        totals.append(pts[-1][0])
    n = len(P["samples"])
    print("%s kernel: wave-instructions per packet by section (mean of %d (stream, packet) samples x %d identical waves; the section" % (KERNEL, n, N))
    print("ENDS at the named site; SX_STOPS build: + ~8 instructions per site passed).  total %.0f (min %.0f, max %.0f)" % (sum(totals) / n, min(totals), max(totals)))
    print("%-66s %6s %8s %8s %7s %7s %7s" % ("section", "passes", "all", "VALU", "SALU", "LDS", "mem"))
    tot_all = sum(totals) / n
    items = sorted(per_site.items(), key=lambda kv: -sum(kv[1][k] for k in COUNTERS))
    for s, v in items:
        allv = sum(v[k] for k in COUNTERS) / n
        print("%-66s %6.1f %8.0f %8.0f %7.0f %7.0f %7.0f  %4.1f %%" % (NAMES.get(s, "end" if s == 63 else str(s)), v["hits"] / n, allv, v["SQ_INSTS_VALU"] / n,
              v["SQ_INSTS_SALU"] / n, v["SQ_INSTS_LDS"] / n, (v["SQ_INSTS_SMEM"] + v["SQ_INSTS_VMEM_RD"] + v["SQ_INSTS_VMEM_WR"]) / n, 100.0 * allv / tot_all))


def report_cycles(plan_path, prof_dir):
    """the same subtraction for a counter pass that carries cycle counters: wave cycles, cycles parked in s_waitcnt, issue stalls per section
    (256 identical waves, ONE per compute unit: what a lone wave's dependent chain and its LDS / memory round trips cost; quad-cycle units x 4)"""
    P = json.load(open(plan_path))
    N = P["n_streams"]
    KERNEL = P.get("kernel", "enc_analysis")
    global NAMES
    if KERNEL == "dec_synth": NAMES = DEC_NAMES
    if KERNEL == "enc_coding": NAMES = COD_NAMES
    CYC = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"]
    rows = defaultdict(dict)
    for f in glob.glob(os.path.join(prof_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    order = sorted(rows)
    per_site = defaultdict(lambda: defaultdict(float))
    n = len(P["samples"])
    tot = defaultdict(float)
    for j, W in P["samples"]:
        pts = []
        for m in [m for m in P["plan"] if m["sample"] == j]:
            c = rows[order[m["dispatch"]]]
            v = {k: c.get(k, 0.0) / N for k in CYC}
            pts.append((v["SQ_INSTS_VALU"] + v["SQ_INSTS_SALU"] + v["SQ_INSTS_LDS"], m["site"], v))
        pts.sort(key=lambda t: t[0])
        prev = {k: 0.0 for k in CYC}
        for _, s, v in pts:
            for k in CYC: per_site[s][k] += v[k] - prev[k]
            prev = v
        for k in CYC: tot[k] += pts[-1][2][k]
    print("%s kernel, one wave per compute unit: cycles per packet by section (mean of %d samples; SQ cycle counters x 4)" % (KERNEL, n))
    print("total: %.0f instructions (VALU + SALU + LDS), %.0f wave cycles = %.1f per instruction; parked in s_waitcnt %.0f %%, issue stall %.0f %%" % (
        (tot["SQ_INSTS_VALU"] + tot["SQ_INSTS_SALU"] + tot["SQ_INSTS_LDS"]) / n, 4 * tot["SQ_WAVE_CYCLES"] / n,
        4 * tot["SQ_WAVE_CYCLES"] / (tot["SQ_INSTS_VALU"] + tot["SQ_INSTS_SALU"] + tot["SQ_INSTS_LDS"]), 100 * tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 100 * tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"]))
    print("%-66s %8s %9s %7s %8s %8s %6s %6s" % ("section", "instr", "cycles", "cyc/ins", "waitcnt", "stall", "LDS", "share"))
    for s, v in sorted(per_site.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
        ins = (v["SQ_INSTS_VALU"] + v["SQ_INSTS_SALU"] + v["SQ_INSTS_LDS"]) / n
        cyc = 4 * v["SQ_WAVE_CYCLES"] / n
        print("%-66s %8.0f %9.0f %7.1f %8.0f %8.0f %6.0f %5.1f %%" % (NAMES.get(s, "end" if s == 63 else str(s)), ins, cyc, cyc / max(ins, 1.0), 4 * v["SQ_WAIT_ANY"] / n,
              4 * v["SQ_WAIT_INST_ANY"] / n, v["SQ_INSTS_LDS"] / n, 100.0 * v["SQ_WAVE_CYCLES"] / tot["SQ_WAVE_CYCLES"] * n / n))


if __name__ == "__main__":
    if sys.argv[1] == "run": run(sys.argv[2])
    elif sys.argv[1] == "rundec": run_dec(sys.argv[2])
    elif sys.argv[1] == "runcod": run_cod(sys.argv[2])
    elif sys.argv[1] == "reportcyc": report_cycles(sys.argv[2], sys.argv[3])
    else: report(sys.argv[2], sys.argv[3])
