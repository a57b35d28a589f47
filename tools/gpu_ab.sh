#!/bin/bash
# A/B timing of the build-flag variants under build/ (libsolo_<name>.so + .flags) against the in-tree library, nothing else:
#   gpurun --timeout 400 -- 'bash tools/gpu_ab.sh [streams] [packets]'
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
N=${1:-4096}; P=${2:-10}
: > "$OUT/ab.log"
VARIANTS=$(ls build/libsolo_*.flags 2>/dev/null | sed 's/\.flags$//')
for round in 1 2; do
  for lib in solo_amd/libsolo_mi355x $VARIANTS; do
    SOLO_LIB_OVERRIDE=$ROOT/$lib.so timeout 150 python tools/quick_bench.py $N $P 2>&1 | grep -v amdgpu.ids >> "$OUT/ab.log"
  done
done
python tools/pick_variant.py "$OUT/ab.log" | tee "$OUT/ab_choice.txt"
