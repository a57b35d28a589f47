#!/bin/bash
# decode-leg A/B: in-tree library vs named builds, 8192 streams x 50 packets with 30 % description loss + the plain 4096 x 20 line
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
: > "$OUT/ab_dec.log"
for round in 1 2; do
  for lib in solo_amd/libsolo_mi355x "$@"; do
    f=$ROOT/$lib.so; [ -f "$f" ] || f=$ROOT/build/libsolo_$lib.so
    LOSS=0.3 SOLO_LIB_OVERRIDE=$f timeout 200 python tools/quick_bench.py 8192 50 2>&1 | grep -v amdgpu.ids | sed 's/^/L30 /' >> "$OUT/ab_dec.log"
    SOLO_LIB_OVERRIDE=$f timeout 150 python tools/quick_bench.py 4096 20 2>&1 | grep -v amdgpu.ids >> "$OUT/ab_dec.log"
  done
done
grep parity "$OUT/ab_dec.log" | sed "s#$ROOT/##"
