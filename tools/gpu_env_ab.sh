#!/bin/bash
# A/B of runtime-knob settings (environment variables of the library) with tools/quick_bench.py, two rounds:
#   gpurun --timeout 300 -- 'bash tools/gpu_env_ab.sh "SOLO_ENC_GATE=0" "SOLO_ENC_GATE=1" ...'     (each argument: one or more VAR=value, space separated)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
: > "$OUT/env_ab.log"
for round in 1 2; do
  for cfg in "$@"; do
    echo "== $cfg" >> "$OUT/env_ab.log"
    env $cfg timeout 150 python tools/quick_bench.py ${QB_STREAMS:-4096} ${QB_PACKETS:-10} 2>&1 | grep -v amdgpu.ids >> "$OUT/env_ab.log"
  done
done
cat "$OUT/env_ab.log"
