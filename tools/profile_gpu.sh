#!/bin/bash
# rocprofv3 passes over the default bench command; run on the GPU box through gpurun:
#   gpurun --timeout 900 -- 'bash tools/profile_gpu.sh [trace fetch write sq inst lane]'
# then, back in the container:   python tools/summarize_profile.py gpurun_out/prof profiles/r01_bench_v2
# The counter passes are separate from the trace pass and from each other (the TCC block cannot hold FETCH_SIZE and
# WRITE_SIZE at once; the SQ block holds 8 counters).  Every pass runs under its own timeout: a rejected counter set makes
# rocprofv3 abort and then hang.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
PASSES=${*:-trace fetch write sq inst lane}
cd /tmp
for p in $PASSES; do
  rm -rf "$OUT/$p"
  case $p in
    trace) ARGS="--kernel-trace --stats" ;;
    fetch) ARGS="--pmc FETCH_SIZE" ;;
    write) ARGS="--pmc WRITE_SIZE" ;;
    sq)    ARGS="--pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" ;;
    inst)  ARGS="--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" ;;
    lane)  ARGS="--pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" ;;
    icache) ARGS="--pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH" ;;
    lds)   ARGS="--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 GRBM_GUI_ACTIVE" ;;
    scalar) ARGS="--pmc SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_SMEM SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAVE_CYCLES" ;;
    *) echo "unknown pass $p"; continue ;;
  esac
  timeout -k 5 ${PASS_TIMEOUT:-300} rocprofv3 $ARGS -d "$OUT/$p" -o r01 --output-format csv -- $CMD > "$OUT/bench_$p.log" 2>&1
  echo "pass $p: rc=$?"
done
grep -h '"metric"' "$OUT/bench_trace.log" 2>/dev/null | tail -1 | cut -c1-300
