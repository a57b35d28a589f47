#!/usr/bin/env python3
"""Static instruction mix of the row quantiser's sample step, phase by phase: compiles solo_nsq_row.hip with -DRW_MARKS (a scheduling
fence + an assembly comment at every phase boundary of the sample step) and counts the instructions between the marks, over the
two samples of the loop body.   python tools/debug/nsq_row_mix.py [-DFLAG ...]"""
import os, re, subprocess, sys, tempfile
from collections import Counter, OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(tempfile.gettempdir(), "nsq_row_marks.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-DSOLO_WITH_ENCODER", "-Wno-pass-failed", "-S", "--cuda-device-only", "-DRW_MARKS",
                       os.path.join(ROOT, "solo_amd/csrc/solo_nsq_row.hip"), "-o", out] + sys.argv[1:], stderr=subprocess.DEVNULL)
L = open(out).read().split("\n")
marks = [(i, l.split()[-1]) for i, l in enumerate(L) if "; RW_MARK" in l]
NAMES = dict(R="ring refill", A="A predict / shape / residual", B="B side candidates", C="C centre combinations + reorder", E="E joint decision",
             M="survivor move", D="D decoder simulation", F="F emission", G="G update + ring store", Z="(end)")
tot = OrderedDict()
for (a, n), (b, n2) in zip(marks, marks[1:]):
    if n == "Z":
        continue
    ops = [l.split()[0] for l in L[a + 1:b] if l.strip() and not l.strip().startswith((";", ".", "//"))]
    c = tot.setdefault(n, Counter())
    c.update(ops)
    c["__n"] += 1
print("%-34s %6s %6s %6s %6s %6s %6s %6s %6s" % ("phase (per sample)", "all", "valu", "salu", "lds", "mem", "dpp", "nop", "cnd"))
g = Counter()
for n, c in tot.items():
    k = c.pop("__n")
    f = lambda pred: sum(v for o, v in c.items() if pred(o)) / k
    row = (f(lambda o: True), f(lambda o: o.startswith("v_")), f(lambda o: o.startswith("s_")), f(lambda o: o.startswith("ds_")),
           f(lambda o: o.startswith(("global_", "flat_", "scratch_", "buffer_"))), f(lambda o: o.endswith("_dpp")), f(lambda o: o == "s_nop"), f(lambda o: "cndmask" in o))
    print("%-34s " % NAMES.get(n, n) + " ".join("%6.1f" % x for x in row))
    for i, x in enumerate(row):
        g[i] += x
print("%-34s " % "sum" + " ".join("%6.1f" % g[i] for i in range(8)))
