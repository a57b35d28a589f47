#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs that a `gpurun` profiling call left under gpurun_out/prof/ into the small,
tracked summary under profiles/ (per-kernel trace stats + PMC counters averaged per launch).

  python tools/summarize_profile.py gpurun_out/prof profiles/r01_bench [packets_per_launch]
"""
import collections
import csv
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
packets_step = int(sys.argv[3]) if len(sys.argv) > 3 else 204800
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import solo_amd
try:
    git_head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
    git_dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "solo_amd/csrc"], text=True).strip())
except Exception:
    git_head, git_dirty = None, None
out = {"source": "rocprofv3 --kernel-trace --stats / --pmc (separate passes) over `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra`",
       "packets_per_step": packets_step, "kernel_source_sha16": None, "git_head": git_head, "git_kernel_sources_modified_since_head": git_dirty, "kernels": {}}
st = os.path.join(src, "trace", "r01_kernel_stats.csv")
lines = []
for r in csv.DictReader(open(st)):
    name = r["Name"].split("(")[0]
    if name.startswith("solo_"):
        out["kernels"].setdefault(name, {})["trace"] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                        "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6,
                                                        "percent": float(r["Percentage"])}
    lines.append(",".join([name[:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]]))
for d in ("fetch", "write", "sq", "inst", "lane", "lds"):
    f = os.path.join(src, d, "r01_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]),
                   "scratch_bytes_per_lane": int(r["Scratch_Size"]), "grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"])}
    for k, v in agg.items():
        if k.startswith("solo_") and "init" not in k:
            e = out["kernels"].setdefault(k, {})
            e.setdefault("resources", meta[k])
            e.setdefault("pmc_avg_per_launch", {}).update({c: sum(x) / len(x) for c, x in v.items()})
# the encoder kernels are launched once per chunk of a step's packets, the decoder once per step: packets per LAUNCH differ.
# The bench line (in the log of the trace pass) says how many launches per step each kernel has.
lps = {}
try:
    for line in open(os.path.join(src, "bench_trace.log")):
        if line.startswith("{") and '"kernels"' in line:
            j = json.loads(line)
            lps = {k: v.get("launches_per_step", 1) for k, v in j["kernels"].items()}
            lps.update(j.get("launches_per_step", {}))
            out["kernel_source_sha16"] = j.get("kernel_source_sha16")      # the build that was profiled (bench.py prints it)
            out["bench_line_of_the_trace_pass"] = {k: j.get(k) for k in ("value", "ms_per_step", "parity_checked")}
except Exception:
    pass
for k, e in out["kernels"].items():
    p = e.get("pmc_avg_per_launch", {})
    packets = packets_step / lps.get(k, 1)
    e["packets_per_launch"] = packets
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        # guide (MI355X_MICROARCH.md, HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the
        # bytes of wide coalesced reads -> doubled here; WRITE_SIZE is taken as is (uncalibrated)
        e["hbm_bytes_per_launch_corrected"] = (2.0 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
        e["hbm_bytes_per_packet_corrected"] = e["hbm_bytes_per_launch_corrected"] / packets
    if "SQ_INSTS_VALU" in p:
        e["wave_instructions_per_packet"] = {c[9:]: p[c] / packets for c in p if c.startswith("SQ_INSTS_")}
    if "SQ_THREAD_CYCLES_VALU" in p and p.get("SQ_ACTIVE_INST_VALU"):
        # lanes that are switched on while a VALU instruction executes / 64: SQ_THREAD_CYCLES_VALU counts active lanes x cycles,
        # SQ_ACTIVE_INST_VALU the cycles (quad-cycle units, the same for both)
        e["valu_lane_utilisation"] = p["SQ_THREAD_CYCLES_VALU"] / (64.0 * p["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAVE_CYCLES" in p:
        wc = p["SQ_WAVE_CYCLES"]
        e["wave_cycle_split"] = {"active_inst_any": p.get("SQ_ACTIVE_INST_ANY", 0) / wc, "wait_any(memory/barrier)": p.get("SQ_WAIT_ANY", 0) / wc,
                                 "wait_inst_any(issue stall)": p.get("SQ_WAIT_INST_ANY", 0) / wc}
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
json.dump(out, open(dst + "_summary.json", "w"), indent=1)
open(dst + "_kernel_stats.csv", "w").write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n" + "\n".join(lines) + "\n")
if out["kernel_source_sha16"] != solo_amd.kernel_source_hash():
    print("WARNING: the profiled build (%s) is not the kernel source of this tree (%s)" % (out["kernel_source_sha16"], solo_amd.kernel_source_hash()))
tr = {"what": "L2 memory-side (fabric) request bytes: TCC_EA0 read / write requests as rocprofv3 derives FETCH_SIZE / WRITE_SIZE from them. "
              "Infinity-Cache (256 MiB) hits are NOT excluded and this rocprofv3 exposes no counter behind the Infinity Cache, so this is an upper "
              "bound of the DRAM traffic (the quantiser's 32 MB emission ring and the stream states fit in the Infinity Cache)",
      "note": "rocprofv3 FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, KiB -> bytes, per 40 ms packet and kernel; see "
              + os.path.basename(dst) + "_summary.json", "kernel_source_sha16": out["kernel_source_sha16"], "git_head": git_head}
for k, e in out["kernels"].items():
    if "hbm_bytes_per_packet_corrected" in e:
        tr[k + "_bytes_per_packet"] = e["hbm_bytes_per_packet_corrected"]
json.dump(tr, open(os.path.join(os.path.dirname(dst) or ".", "hbm_traffic.json"), "w"), indent=1)
# VALU / all wave-instructions per packet (bench.py's valu_issue block reads this)
wi = {"source": os.path.basename(dst) + "_summary.json (rocprofv3 --pmc SQ_INSTS_* pass)", "kernel_source_sha16": out["kernel_source_sha16"], "git_head": git_head,
      "valu_per_packet": {}, "all_per_packet": {}, "valu_lane_utilisation": {}}
for k, e in out["kernels"].items():
    w = e.get("wave_instructions_per_packet")
    if w and "gate" not in k and "debug" not in k:
        wi["valu_per_packet"][k] = round(w.get("VALU", 0.0), 1)
        wi["all_per_packet"][k] = round(sum(w.get(c, 0.0) for c in ("VALU", "SALU", "LDS", "SMEM", "VMEM_RD", "VMEM_WR")), 1)
        if "valu_lane_utilisation" in e:
            wi["valu_lane_utilisation"][k] = round(e["valu_lane_utilisation"], 4)
if wi["valu_per_packet"]:
    wi["valu_per_packet_round_trip"] = round(sum(wi["valu_per_packet"].values()), 1)
    wi["all_per_packet_round_trip"] = round(sum(wi["all_per_packet"].values()), 1)
    json.dump(wi, open(os.path.join(os.path.dirname(dst) or ".", "wave_instructions.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
