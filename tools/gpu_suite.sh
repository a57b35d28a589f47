#!/bin/bash
# the whole GPU suite + quick timings of both builds (16 kHz / 32 kHz) + the driver's bench command
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/gputest.log" 2>&1; echo "GPU suite rc=$? $(tail -1 $OUT/gputest.log)"
timeout 150 python tools/quick_bench.py 4096 10 2>&1 | grep -v amdgpu.ids | tee "$OUT/qb.log"
timeout 150 python tools/quick_bench.py 8192 10 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/qb.log"
timeout 150 python tools/quick_bench.py 4096 10 32000 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/qb.log"
timeout 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-400 "$OUT/bench.json" | tail -1
