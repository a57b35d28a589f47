#!/usr/bin/env python3
"""Loops of one function of a compiled kernel source (asm from tools/debug/asm_report.py in /tmp): first/last line, mix, nesting.
   python tools/debug/asm_func.py /tmp/solo_api.hip.s sx_vad [--dump a b]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
funcs = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^([A-Za-z_0-9]+):\s+; @", l)] if m] + [(len(lines), None)]
for (a, n), (b, _) in zip(funcs, funcs[1:]):
    if name in n:
        break
else:
    sys.exit("no such function")
if "--dump" in sys.argv:
    k = sys.argv.index("--dump"); x, y = int(sys.argv[k + 1]), int(sys.argv[k + 2])
    print("\n".join("%6d %s" % (i + 1, lines[i]) for i in range(x - 1, y)))
    sys.exit(0)
def mix(body):
    ops = [l.split()[0] for l in body if l.strip() and not l.strip().startswith((";", ".")) and not l.endswith(":")]
    return dict(v=sum(o.startswith("v_") for o in ops), s=sum(o.startswith("s_") for o in ops), ds=sum(o.startswith("ds_") for o in ops),
                mem=sum(o.startswith(("global_", "scratch_", "buffer_", "flat_")) for o in ops)), ops
c, ops = mix(lines[a:b])
print(n, "lines %d-%d" % (a + 1, b), c)
lab = {m.group(1): i for i, l in enumerate(lines[a:b], a) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = set()
for i in range(a, b):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
    if m and m.group(1) in lab and lab[m.group(1)] < i:
        loops.add((lab[m.group(1)], i))
for x, y in sorted(loops):
    depth = sum(1 for (p, q) in loops if p <= x and y <= q) - 1
    c, ops = mix(lines[x:y + 1])
    print("  " * depth + "loop %d-%d %s | %s" % (x + 1, y + 1, c, ", ".join("%s %d" % kv for kv in Counter(ops).most_common(7))))
