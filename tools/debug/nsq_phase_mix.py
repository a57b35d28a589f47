#!/usr/bin/env python3
"""Static instruction mix of the quantiser's sample step, phase by phase: compiles solo_nsq16.hip with -DSX_PROF (the phase timer reads
the cycle counter at every phase boundary: s_memtime) and counts the instructions between consecutive reads of the FIRST of the two
samples of the loop body.  (The timer reads do not stop the instruction scheduler: arithmetic of one phase may sit on the other side of a
read -- the first line, "ring refill requests", is mostly filter arithmetic of phase A; the loop total is exact.)  With the cycles per phase of tools/prof_nsq.py (pass its output file) it prints cycles per instruction.
    python tools/debug/nsq_phase_mix.py [gpurun_out/prof_sections.log]           (no GPU needed for the counts)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(tempfile.gettempdir(), "nsq_prof.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-DSOLO_WITH_ENCODER", "-Wno-pass-failed", "-DSX_PROF", "-S", "--cuda-device-only",
                       os.path.join(ROOT, "solo_amd/csrc/solo_nsq16.hip"), "-o", out], stderr=subprocess.DEVNULL)
L = open(out).read().split("\n")
# the sample loop: the deepest loop that contains global_load ... nt (the ring prefetch); its header label
nt = [i for i, l in enumerate(L) if "global_load" in l and " nt" in l]
head = max(i for i, l in enumerate(L[:nt[0]]) if re.match(r"^\.LBB\d+_\d+:", l))
reads = [i for i, l in enumerate(L) if "s_memtime" in l and i > head]
PH = [("12", "ring refill requests"), ("2", "A predictions, shaping, residual (3 tracks)"), ("3", "B+C candidates"), ("4", "E judge"),
      ("5", "E replace-worst-by-best rounds + survivor move (one pass of its loop)"), ("6", "survivor gather"), ("10", "D undo"), ("11", "wait for the ring cells"),
      ("7", "F emit"), ("8", "G update")]
cyc = {}
if len(sys.argv) > 1:
    key = {"ring refill requests": "12", "A predict": "2", "B+C": "3", "E judge": "4", "E replace": "5", "survivor gather": "6", "D undo": "10",
           "wait for the ring": "11", "F emit": "7", "G update": "8"}
    for l in open(sys.argv[1]):
        m = re.match(r"(.+?)\s+(\d+) cycles per sample step", l)
        if m:
            for k, v in key.items():
                if m.group(1).strip().startswith(k):
                    cyc[v] = 2 * int(m.group(2))          # prof_nsq.py divides by 640 = 2 x the 320 sample steps of a packet
marks = [head] + reads[:len(PH)]
tot = 0
print("%-72s %5s %5s %5s %5s %5s   %s" % ("phase (first sample of the loop body)", "VALU", "SALU", "LDS", "VMEM", "all", "cycles -> per instruction"))
for (a, b), (k, name) in zip(zip(marks, marks[1:]), PH):
    ops = [l.split()[0] for l in L[a:b] if l.strip() and not l.strip().startswith((";", ".")) and not l.rstrip().endswith(":")]
    ops = [o for o in ops if not o.startswith(("s_memtime", "s_nop"))]
    v = sum(o.startswith("v_") for o in ops); s = sum(o.startswith("s_") and not o.startswith("s_waitcnt") for o in ops)
    d = sum(o.startswith("ds_") for o in ops); m = sum(o.startswith(("global_", "scratch_", "buffer_")) for o in ops)
    n = len(ops); tot += n
    print("%-72s %5d %5d %5d %5d %5d   %s" % (name, v, s, d, m, n, ("%5d -> %.1f" % (cyc[k], cyc[k] / max(n, 1))) if k in cyc else ""))
print("instructions per sample step (static, one pass of the replace loop): %d" % tot)
