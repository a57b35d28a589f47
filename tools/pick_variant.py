#!/usr/bin/env python3
"""Reads the lines tools/quick_bench.py printed for several library builds (tools/gpu_round_end.sh) and names the build to keep:
the in-tree library unless a variant is bit-exact in every run and its mean encode time is at least 1.5 % lower.
Last line of the output = path of the chosen build without the .so suffix (relative to the repository root)."""
import collections
import os
import re
import sys

runs = collections.defaultdict(list)
for line in open(sys.argv[1]):
    m = re.match(r"(\S+) parity enc=(\w+) dec=(\w+) \| encode ([0-9.]+) ms .* decode ([0-9.]+) ms", line)
    if m:
        runs[m.group(1)].append((m.group(2) == "True" and m.group(3) == "True", float(m.group(4)), float(m.group(5))))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = None
table = []
for lib, rs in runs.items():
    rel = os.path.relpath(lib, root)[:-3] if lib != "default" else "solo_amd/libsolo_mi355x"
    ok = all(r[0] for r in rs)
    enc = sum(r[1] for r in rs) / len(rs)
    dec = sum(r[2] for r in rs) / len(rs)
    table.append((rel, ok, enc, dec, len(rs)))
    print("%-40s parity=%s encode %.3f ms decode %.3f ms (%d runs)" % (rel, ok, enc, dec, len(rs)))
    if rel == "solo_amd/libsolo_mi355x":
        base = enc
best = "solo_amd/libsolo_mi355x"
if base is not None:
    cand = [(enc, rel) for rel, ok, enc, dec, n in table if ok and rel != best and n >= 2 and enc < 0.985 * base]
    if cand:
        best = min(cand)[1]
print(best)
