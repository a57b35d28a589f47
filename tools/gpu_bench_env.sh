#!/bin/bash
# the headline round trip (bench.py, main leg only) under environment variants:  bash tools/gpu_bench_env.sh "A=1" "B=2 C=3" ...
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
: > "$OUT/bench_env.log"
for round in 1 2; do
  for e in "NONE=0" "$@"; do
    echo "== $e" >> "$OUT/bench_env.log"
    env $e timeout 300 python bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline --no-hash 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('value %.0f  ms_per_step %.3f  kernels %s' % (j['value'], j['ms_per_step'], {k: v['avg_launch_ms'] for k, v in j['kernels'].items()}))
" >> "$OUT/bench_env.log"
  done
done
cat "$OUT/bench_env.log"
