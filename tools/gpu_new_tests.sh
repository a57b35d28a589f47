#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nsq_taps.py tests/test_gpu_nsq_row.py "tests/test_gpu_decoder.py::test_stated_tolerance_against_the_floating_point_tree_on_the_gpu" tests/test_abi.py -m gpu -x -q --durations=5 > "$OUT/gputest_new.log" 2>&1; echo "new tests rc=$? $(tail -1 $OUT/gputest_new.log)"
timeout 900 python -m pytest "tests/test_gpu_fullsize.py::test_config4_all_blocks_on_one_gpu" -m gpu -x -q --durations=3 > "$OUT/gputest_c4.log" 2>&1; echo "config4 rc=$? $(tail -1 $OUT/gputest_c4.log)"
timeout 500 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; python - <<'PY'
import json
try:
    r=json.loads(open("gpurun_out/bench.json").read().strip().splitlines()[-1])
    print("value", r["value"], "ms/step", r["ms_per_step"], "clock", r.get("shader_clock_mhz_under_vector_load"), "parity", r["parity_checked"])
    for k,v in r.get("extra",{}).items(): print(" ", k, v.get("value"), v.get("parity_checked"))
    print(" cpu", r.get("cpu_baseline",{}).get("value"))
except Exception as e: print("no bench line", e)
PY
tail -3 "$OUT/bench.err"
