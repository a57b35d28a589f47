#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
for zc in 0 1; do echo "== SOLO_LEGACY_ZEROCOPY=$zc"; SOLO_LEGACY_ZEROCOPY=$zc timeout 200 python tools/legacy_api_cost.py 2>&1 | tail -1; done | tee "$OUT/legacy_cost.txt"
timeout 300 python -m pytest tests/test_dropin_link.py tests/test_pinned_corners.py tests/test_gpu_encoder.py tests/test_framesize20.py -m gpu -x -q 2>&1 | tail -2
