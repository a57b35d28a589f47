#!/bin/bash
# wave-instructions of the analysis kernel by section (tools/debug/analysis_sections.py; build/libsolo_stops.so = -DSX_STOPS encoder kernels
# + -DSX_EXPERIMENTS api, tools/build_stops.sh):   gpurun --timeout 600 -- 'bash tools/gpu_sections.sh'
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
MODE=${1:-run}            # run: the encoder's analysis kernel; rundec: the decoder's synthesis kernel
rm -rf "$OUT/sections"
cd /tmp
[ "$MODE" = run ] && export SOLO_EXP_SKIP=3        # (the analysis kernel alone: the stopped launches leave the later stages nothing to work on)
SOLO_LIB_OVERRIDE=$ROOT/build/libsolo_stops.so timeout -k 5 ${SECTIONS_TIMEOUT:-400} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES \
    -d "$OUT/sections" -o s --output-format csv -- python $ROOT/tools/debug/analysis_sections.py $MODE "$OUT/sections_plan.json" > "$OUT/sections_run.log" 2>&1
echo "rc=$?"; tail -3 "$OUT/sections_run.log"
cd $ROOT
python tools/debug/analysis_sections.py report "$OUT/sections_plan.json" "$OUT/sections" > "$OUT/analysis_sections.txt" 2>&1
cat "$OUT/analysis_sections.txt"
# keep the merged output small
find "$OUT/sections" -name "*.csv" -size +20M -delete
