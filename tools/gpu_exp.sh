#!/bin/bash
# experiments of the moment (scratch; results under gpurun_out/)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
: > "$OUT/exp.log"
for skip in 0 1 2 3; do
  echo "SOLO_EXP_SKIP=$skip (bit 0: no quantiser, bit 1: no coding)" >> "$OUT/exp.log"
  SOLO_EXP_SKIP=$skip SOLO_LIB_OVERRIDE=$ROOT/build/libsolo_exp.so timeout 150 python tools/quick_bench.py 4096 10 2>&1 | grep -v amdgpu.ids >> "$OUT/exp.log"
done
cat "$OUT/exp.log"
cd /tmp
rm -rf "$OUT/pcs"
SOLO_ENC_CHUNK=0 timeout -k 5 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 2 --kernel-trace -d "$OUT/pcs" -o pcs --output-format csv -- python $ROOT/tools/quick_bench.py 4096 10 > "$OUT/pcs.log" 2>&1
echo "pc sampling rc=$?"; tail -3 "$OUT/pcs.log"; ls -la "$OUT/pcs" 2>/dev/null | head; find "$OUT/pcs" -name "*.csv" | head
for f in $(find "$OUT/pcs" -name "*pc_sampling*.csv" | head -1); do head -5 $f; wc -l $f; done
