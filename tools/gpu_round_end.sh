#!/bin/bash
# The one gpurun call that closes a round (GPU minutes are scarce, so the steps are ordered by what only this call can produce):
#   gpurun --timeout 900 -- 'bash tools/gpu_round_end.sh'
# 1. A/B of build-flag variants under build/ against the in-tree library (tools/quick_bench.py: parity against the goldens + timing);
#    a variant that is bit-exact and >= 1.5 % faster on the encoder is adopted for the rest of the call: its flags go to
#    solo_amd/build_flags.txt (part of the build's identity, solo_amd.kernel_source_hash()) and to gpurun_out/build_flags.txt, from
#    where the container copies them and rebuilds the same library.
# 2. the GPU tests of what is new this round; 3. rocprofv3 trace + instruction + traffic passes; 4. the driver's bench command;
# 5. the remaining counter passes; 6. the whole GPU suite.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$(date +%s)
log() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a "$OUT/run.log"; }
: > "$OUT/run.log"
python -c "import torch; print(torch.cuda.get_device_name(0))" > "$OUT/dev.log" 2>&1
log "torch imported: $(tail -1 "$OUT/dev.log")"

# ---- 1. build-flag variants:  build/libsolo_<name>.so built with the flags in build/libsolo_<name>.flags
: > "$OUT/ab.log"
VARIANTS=$(ls build/libsolo_*.flags 2>/dev/null | sed 's/\.flags$//')
if [ -n "$VARIANTS" ]; then
  for round in 1 2; do
    for lib in solo_amd/libsolo_mi355x $VARIANTS; do
      SOLO_LIB_OVERRIDE=$ROOT/$lib.so timeout 150 python tools/quick_bench.py 4096 10 2>&1 | grep -v amdgpu.ids >> "$OUT/ab.log"
    done
  done
  python tools/pick_variant.py "$OUT/ab.log" > "$OUT/ab_choice.txt" 2>&1
  CH=$(tail -1 "$OUT/ab_choice.txt")
  log "A/B: $CH"
  if [ -f "$ROOT/$CH.flags" ]; then
    cp "$ROOT/$CH.flags" solo_amd/build_flags.txt
    cp "$ROOT/$CH.flags" "$OUT/build_flags.txt"
    cp "$ROOT/$CH.so" solo_amd/libsolo_mi355x.so
  fi
fi

# ---- 2. what is new this round
timeout 300 python -m pytest tests/test_gpu_nsq_row.py tests/test_nsq_taps.py tests/test_l0_primitives.py tests/test_gpu_encoder.py -m gpu -x -q > "$OUT/gputest_new.log" 2>&1
log "new GPU tests: rc=$? $(tail -1 "$OUT/gputest_new.log")"

# ---- 3. profile, first half
bash tools/profile_gpu.sh trace inst fetch write > "$OUT/profile1.log" 2>&1
log "profile passes 1: $(tr '\n' ' ' < "$OUT/profile1.log" | cut -c1-200)"

# ---- 4. the driver's bench command
timeout 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
log "bench: rc=$? $(cut -c1-160 "$OUT/bench.json" | tail -1)"

# ---- 5. profile, second half
bash tools/profile_gpu.sh sq lane lds > "$OUT/profile2.log" 2>&1
log "profile passes 2: $(tr '\n' ' ' < "$OUT/profile2.log" | cut -c1-200)"

# ---- 6. the whole GPU suite
timeout 600 python -m pytest tests -m gpu -x -q > "$OUT/gputest.log" 2>&1
log "GPU suite: rc=$? $(tail -1 "$OUT/gputest.log")"

# ---- 7. section timers of the profiling build (build/libsolo_prof.so = the same sources with -DSX_PROF, tools/build_prof.sh), the analysis
# kernel's wave-instructions by section (build/libsolo_stops.so, tools/build_stops.sh), the legs the default command does not profile, the
# issue-rate microbenchmark
if [ -f build/libsolo_prof.so ]; then
  { SOLO_ENC_CHUNK=0 SOLO_LIB_OVERRIDE=$ROOT/build/libsolo_prof.so timeout 120 python tools/prof_sections.py 4096 10
    SOLO_LIB_OVERRIDE=$ROOT/build/libsolo_prof.so timeout 120 python tools/prof_dec.py 4096 10; } > "$OUT/prof_sections.log" 2>&1
  log "section timers: $(grep -c cycles "$OUT/prof_sections.log") lines"
fi
if [ -f build/libsolo_stops.so ]; then
  SECTIONS_SAMPLES=8 bash tools/gpu_sections.sh > "$OUT/sections.log" 2>&1
  log "analysis sections: $(grep -c "%" "$OUT/analysis_sections.txt") lines"
fi
bash tools/profile_legs.sh > "$OUT/profile_legs.log" 2>&1
log "legs: $(grep "^leg" "$OUT/profile_legs.log" | cut -c1-120 | tr '\n' ' ')"
if [ -x build/mb_valu ]; then timeout 120 build/mb_valu > "$OUT/mb_valu.txt" 2>&1; log "microbenchmark: $(wc -l < "$OUT/mb_valu.txt") lines"; fi
