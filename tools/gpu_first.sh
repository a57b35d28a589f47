#!/bin/bash
# first look at a new quantiser build on the GPU: unit check of the row exchanges, parity + timing of the in-tree library and the variants under
# build/ -- pipelined (the product's schedule) and stage by stage (SOLO_ENC_CHUNK=0: every kernel alone on the chip)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_nsq_row.py -m gpu -x -q > "$OUT/rowops.log" 2>&1; echo "rowops rc=$? $(tail -1 $OUT/rowops.log)"
: > "$OUT/ab.log"
VARIANTS=$(ls build/libsolo_*.flags 2>/dev/null | sed 's/\.flags$//')
for round in 1 2; do
  for lib in solo_amd/libsolo_mi355x $VARIANTS; do
    SOLO_LIB_OVERRIDE=$ROOT/$lib.so timeout 150 python tools/quick_bench.py ${1:-4096} ${2:-10} 2>&1 | grep -v amdgpu.ids >> "$OUT/ab.log"
    if [ $round = 1 ]; then
      echo "  (stage by stage)" >> "$OUT/ab.log"
      SOLO_ENC_CHUNK=0 SOLO_LIB_OVERRIDE=$ROOT/$lib.so timeout 150 python tools/quick_bench.py ${1:-4096} ${2:-10} 2>&1 | grep -v amdgpu.ids | sed 's/^/  /' >> "$OUT/ab.log"
    fi
  done
done
cat "$OUT/ab.log"
timeout 300 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q > "$OUT/gpu_enc.log" 2>&1; echo "gpu_encoder rc=$? $(tail -1 $OUT/gpu_enc.log)"
