#!/bin/bash
# A/B of library builds on the GPU box: the in-tree library and every build/libsolo_<name>.so named on the command line, two rounds,
# parity against the goldens + encode / decode timing (tools/quick_bench.py).   bash tools/gpu_ab_quick.sh "4096 20" name1 name2 ...
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS=$1; shift
: > "$OUT/ab_quick.log"
for round in 1 2 3; do
  for lib in solo_amd/libsolo_mi355x "$@"; do
    f=$ROOT/$lib.so; [ -f "$f" ] || f=$ROOT/build/libsolo_$lib.so
    SOLO_LIB_OVERRIDE=$f timeout 150 python tools/quick_bench.py $ARGS 2>&1 | grep -v amdgpu.ids >> "$OUT/ab_quick.log"
  done
done
grep parity "$OUT/ab_quick.log" | sed "s#$ROOT/##"
