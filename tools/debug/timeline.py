#!/usr/bin/env python3
"""Kernel timeline of interleaved encode / decode calls from a rocprofv3 --kernel-trace csv: per call, first start .. last end of each kernel type."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0]
    if n.startswith("solo_") and "init" not in n:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
t0 = rows[0][0]
# split into phases: a new phase starts when the kernel family (enc / dec) changes
fam = lambda n: "dec" if "dec" in n else "enc"
phases = []
for s, e, n in rows:
    f = fam(n)
    if not phases or phases[-1][0] != f:
        phases.append([f, s, e, collections.Counter()])
    phases[-1][2] = max(phases[-1][2], e)
    phases[-1][3][n] += e - s
prev_end = None
for f, s, e, c in phases[int(sys.argv[2]) if len(sys.argv) > 2 else 0:][:int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
    gap = (s - prev_end) / 1e6 if prev_end else 0.0
    print("%s  start %9.3f ms  span %8.3f ms  gap before %7.3f ms   busy: %s" % (f, (s - t0) / 1e6, (e - s) / 1e6, gap,
          ", ".join("%s %.1f" % (k.replace("solo_", "").replace("_kernel", ""), v / 1e6) for k, v in c.items())))
    prev_end = e
