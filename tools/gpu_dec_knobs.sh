#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
: > "$OUT/dec_knobs.log"
for rep in 1 2; do
for cfg in "64 64" "25 25" "25 5" "17 17" "10 10" "50 8"; do
  set -- $cfg
  echo "== SOLO_DEC_CHUNK=$1 SOLO_DEC_FIRST_CHUNK=$2" >> "$OUT/dec_knobs.log"
  LOSS=0.3 SOLO_DEC_CHUNK=$1 SOLO_DEC_FIRST_CHUNK=$2 timeout 200 python tools/quick_bench.py 8192 50 2>&1 | grep parity >> "$OUT/dec_knobs.log"
done; done
cat "$OUT/dec_knobs.log" | cut -c1-200
