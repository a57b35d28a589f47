#!/bin/bash
# cycles (wave cycles, cycles parked in s_waitcnt, issue stalls) of a kernel section by section: the SX_STOPS launches of tools/gpu_sections.sh under a
# counter pass with the SQ cycle counters.   gpurun --timeout 600 -- 'bash tools/gpu_sections_cycles.sh [run|rundec|runcod]'
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
MODE=${1:-run}
rm -rf "$OUT/sections_cyc"
cd /tmp
[ "$MODE" = run ] && export SOLO_EXP_SKIP=3
SOLO_LIB_OVERRIDE=$ROOT/build/libsolo_stops.so timeout -k 5 ${SECTIONS_TIMEOUT:-400} rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD \
    -d "$OUT/sections_cyc" -o s --output-format csv -- python $ROOT/tools/debug/analysis_sections.py $MODE "$OUT/sections_cyc_plan.json" > "$OUT/sections_cyc_run.log" 2>&1
echo "rc=$?"; tail -3 "$OUT/sections_cyc_run.log"
cd $ROOT
python tools/debug/analysis_sections.py reportcyc "$OUT/sections_cyc_plan.json" "$OUT/sections_cyc" > "$OUT/${MODE}_sections_cycles.txt" 2>&1
cat "$OUT/${MODE}_sections_cycles.txt"
find "$OUT/sections_cyc" -name "*.csv" -size +20M -delete
