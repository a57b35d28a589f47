#!/bin/bash
# the persistent encoder pipeline on the GPU: encoder parity suite, then A/B timing against the launch-per-chunk schedule
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q > "$OUT/p1_enc.log" 2>&1; echo "gpu_encoder (persistent) rc=$? $(tail -1 $OUT/p1_enc.log)"
SOLO_ENC_FINAL_WAIT_US=-1 timeout 400 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q > "$OUT/p1_enc_fw0.log" 2>&1; echo "gpu_encoder (deferred coding) rc=$? $(tail -1 $OUT/p1_enc_fw0.log)"
: > "$OUT/p1_ab.log"
for round in 1 2; do
  for cfg in "SOLO_ENC_PERSIST=1" "SOLO_ENC_PERSIST=1 SOLO_ENC_FINAL_WAIT_US=-1" "SOLO_ENC_PERSIST=0"; do
    echo "$cfg" >> "$OUT/p1_ab.log"
    env $cfg timeout 200 python tools/quick_bench.py ${1:-4096} ${2:-50} 2>&1 | grep -v amdgpu.ids >> "$OUT/p1_ab.log"
  done
done
cat "$OUT/p1_ab.log"
