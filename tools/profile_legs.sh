#!/bin/bash
# rocprofv3 --kernel-trace --stats over the legs that the default bench command does not profile:
#   dec8192   BASELINE configs[3]: 8192 streams, decode only, Bernoulli 0.3 description loss (tools/quick_bench.py, LOSS=0.3)
#   rt8192    the configs[4] share of one GPU: 8192 streams, round trip (bench.py --streams 8192)
#   wb32k     the 32 kHz build (tools/quick_bench.py ... 32000)
# gpurun --timeout 600 -- 'bash tools/profile_legs.sh'; the per-kernel statistics land in gpurun_out/prof_legs/<leg>_kernel_stats.csv
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_legs; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() {   # leg, command...
  local leg=$1; shift
  rm -rf "$OUT/$leg"
  timeout -k 5 200 rocprofv3 --kernel-trace --stats -d "$OUT/$leg" -o r --output-format csv -- "$@" > "$OUT/$leg.log" 2>&1
  echo "leg $leg: rc=$? $(grep -v amdgpu.ids "$OUT/$leg.log" | grep -i "parity\|metric" | tail -1 | cut -c1-220)"
  f=$(find "$OUT/$leg" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${leg}_kernel_stats.csv"
  rm -rf "$OUT/$leg"
}
LOSS=0.3 run dec8192 python $ROOT/tools/quick_bench.py 8192 50
run rt8192 python $ROOT/bench.py --streams 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-extra
run wb32k python $ROOT/tools/quick_bench.py 4096 25 32000
ls -la "$OUT"
