#!/bin/bash
# timing experiments on the persistent pipeline: library variants x knobs (tools/quick_bench.py; wrong output where stages are left out)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
: > "$OUT/p2.log"
while read -r lib cfg; do
  [ -z "$lib" ] && continue
  f=$ROOT/solo_amd/libsolo_mi355x.so; [ "$lib" != "default" ] && f=$ROOT/build/libsolo_$lib.so
  echo "== $lib $cfg" >> "$OUT/p2.log"
  env SOLO_LIB_OVERRIDE=$f $cfg timeout 200 python tools/quick_bench.py ${N:-4096} ${P:-50} ${FS:-16000} 2>&1 | grep -v amdgpu.ids | sed "s#$ROOT/##" >> "$OUT/p2.log"
done < "$1"
cat "$OUT/p2.log"
