#!/bin/bash
# round 5, first GPU call: the new GPU tests, the counter list of this rocprofv3, a lane-utilisation pass, a baseline A/B line
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
T0=$(date +%s); log() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a "$OUT/run.log"; }
: > "$OUT/run.log"
python -c "import torch; print(torch.cuda.get_device_name(0))" > "$OUT/dev.log" 2>&1
log "torch: $(tail -1 "$OUT/dev.log")"
timeout 600 python -m pytest tests/test_gpu_dist_nccl.py tests/test_harness.py -m gpu -x -q > "$OUT/gputest_new.log" 2>&1
log "new tests rc=$? $(tail -1 "$OUT/gputest_new.log")"
(cd /tmp && timeout 60 rocprofv3 -L > "$OUT/counters.txt" 2>&1); log "counters: $(wc -l < "$OUT/counters.txt") lines"
timeout 150 python tools/quick_bench.py 4096 10 > "$OUT/qb.log" 2>&1; log "qb: $(grep parity "$OUT/qb.log" | cut -c1-200)"
bash tools/profile_gpu.sh lane > "$OUT/profile_lane.log" 2>&1; log "lane pass: $(tr '\n' ' ' < "$OUT/profile_lane.log" | cut -c1-200)"
