#!/usr/bin/env python3
"""Per-chunk timeline of the encoder pipeline from a rocprofv3 kernel trace (csv): start / end of the analysis (A), quantiser (Q), high band or
coding (H), range coder (R) and assembly (O) launches relative to the analysis launch of the same ordinal.   python tools/debug/trace_steps.py trace.csv [first] [n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
K = collections.defaultdict(list)
for r in rows:
    K[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
def pick(prefix):
    return sorted(x for k, v in K.items() if k.startswith(prefix) for x in v)
A, Q, H, R, O = pick("solo_enc_analysis"), pick("solo_nsq"), pick("solo_enc_coding"), pick("solo_enc_rc"), pick("solo_enc_out")
print("launches: A %d Q %d H %d R %d O %d" % (len(A), len(Q), len(H), len(R), len(O)))
for i in range(first, min(first + n, len(A))):
    t0 = A[i][0]
    f = lambda L: "%6d..%-6d" % ((L[i][0] - t0) // 1000, (L[i][1] - t0) // 1000) if i < len(L) else "      -      "
    print("chunk %3d  A %s (%4d us)  Q %s  H %s  R %s  O %s" % (i, f(A), (A[i][1] - A[i][0]) // 1000, f(Q), f(H), f(R), f(O)))
