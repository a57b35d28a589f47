#!/bin/bash
# timing experiment: the encoder pipeline with stages left out (SX_EXPERIMENTS build of solo_api.hip: SOLO_EXP_SKIP bit 0 = no quantiser, bit 1 = no
# coding (range coder + high band), bit 2 = no range coder; wrong output).   bash tools/gpu_exp_skip.sh [list of SOLO_EXP_SKIP values]
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
: > "$OUT/exp_skip.log"
LIST=${*:-0 2 1 3}
for rep in 1 2; do
for sk in $LIST; do
  echo "== SOLO_EXP_SKIP=$sk" >> "$OUT/exp_skip.log"
  SOLO_EXP_SKIP=$sk SOLO_LIB_OVERRIDE=$ROOT/build/libsolo_exp.so timeout 120 python tools/quick_bench.py 4096 20 2>&1 | grep -v amdgpu.ids >> "$OUT/exp_skip.log"
done; done
cat "$OUT/exp_skip.log"
