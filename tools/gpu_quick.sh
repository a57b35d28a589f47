#!/bin/bash
# parity + timing of the in-tree library (pipelined and stage by stage), the encoder GPU tests
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 150 python tools/quick_bench.py ${1:-4096} ${2:-10} 2>&1 | grep -v amdgpu.ids | tee "$OUT/qb.log"
SOLO_ENC_CHUNK=0 timeout 150 python tools/quick_bench.py ${1:-4096} ${2:-10} 2>&1 | grep -v amdgpu.ids | sed 's/^/  (stage by stage) /' | tee -a "$OUT/qb.log"
timeout 150 python tools/quick_bench.py 8192 10 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/qb.log"
timeout 300 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q > "$OUT/gpu_enc.log" 2>&1; echo "gpu_encoder rc=$? $(tail -1 $OUT/gpu_enc.log)"
