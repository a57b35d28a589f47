#!/usr/bin/env python3
"""Static look at the gfx950 code of a kernel source file: largest loops with their instruction mix, register / scratch use, and
the "tiny divergent regions" (an exec-mask region of <= 3 vector instructions and no memory access: usually a select that the
compiler turned into a branch -- chains of ?: on one index, short-circuit || / && on lane conditions).
    python tools/debug/asm_report.py solo_amd/csrc/solo_nsq_row.hip [-DFLAG ...]
Compiles with hipcc -S --cuda-device-only into /tmp (no GPU needed)."""
import os, re, subprocess, sys, tempfile
from collections import Counter

src = sys.argv[1]
out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-DSOLO_WITH_ENCODER", "-Wno-pass-failed", "-S", "--cuda-device-only",
                       src, "-o", out] + sys.argv[2:], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
funcs = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^([A-Za-z_0-9]+):\s+; @", l)] if m] + [(len(lines), None)]


def func_of(i):
    for (a, n), (b, _) in zip(funcs, funcs[1:]):
        if a <= i < b:
            return n


def mix(body):
    ops = [l.split()[0] for l in body if l.strip() and not l.strip().startswith((";", "."))]
    return dict(v=sum(o.startswith("v_") for o in ops), s=sum(o.startswith("s_") for o in ops), ds=sum(o.startswith("ds_") for o in ops),
                mem=sum(o.startswith(("global_", "scratch_", "buffer_", "flat_")) for o in ops)), ops


print("== resources")
for l in lines:
    if re.search(r"; (NumVgprs|TotalNumVgprs|ScratchSize|LDSByteSize|Occupancy):", l) or re.match(r"^[A-Za-z_0-9]+:\s+; @", l):
        print("  ", l.strip()[:110])
print("== largest loops (first line, last line, instructions)")
lab = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(lines):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < i:
        loops.append((lab[m.group(1)], i))
for a, b in sorted(set(loops), key=lambda x: x[0] - x[1])[:12]:
    c, ops = mix(lines[a:b + 1])
    top = ", ".join("%s %d" % kv for kv in Counter(ops).most_common(6))
    print("  %-40s %6d-%6d  %s  | %s" % ((func_of(a) or "")[:40], a + 1, b + 1, c, top))
print("== tiny divergent regions per function")
cnt = Counter()
for i, l in enumerate(lines):
    if re.match(r"\s*s_(and|andn2|or)_saveexec_b64", l):
        body, ok, j = [], False, i + 1
        while j < len(lines) and len(body) < 14:
            u = lines[j].strip()
            if u and not u.startswith(";"):
                if re.match(r"s_or_b64 exec, exec", u):
                    ok = True
                    break
                body.append(u)
            j += 1
        if ok and not any(re.match(r"(ds_|global_|scratch_|buffer_|flat_)", b) or "saveexec" in b for b in body) and sum(b.startswith("v_") for b in body) <= 3:
            cnt[func_of(i)] += 1
for f, c in cnt.most_common(15):
    print("  %5d  %s" % (c, f))
