#!/usr/bin/env python3
"""Wave-instructions per packet and kernel straight from a rocprofv3 `inst` counter pass (tools/profile_gpu.sh inst):
   python tools/inst_pass.py [gpurun_out/prof/inst/r01_counter_collection.csv] [streams per launch = 4096]"""
import csv, collections, sys
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof/inst/r01_counter_collection.csv"
streams = float(sys.argv[2]) if len(sys.argv) > 2 else 4096.0
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].split("(")[0]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
per = {"solo_dec_synth_kernel": 50.0, "solo_dec_extract_kernel": 50.0}      # one launch = 50 packets of every stream
for k in sorted(tot):
    if not n[k] or not k.startswith("solo_") or "init" in k or "debug" in k: continue
    d = {c[9:]: v / n[k] / streams / per.get(k, 1.0) for c, v in tot[k].items() if c != "SQ_WAVES"}
    print("%-28s %3d launches  %s  all %.1f" % (k, n[k], "  ".join("%s %.1f" % kv for kv in sorted(d.items())), sum(d.values())))
