#!/bin/bash
# A/B/A/B timing of library builds on the GPU box:  bash tools/ab.sh "<quick_bench args>" libA.so libB.so [libC.so ...]   (two rounds)
ARGS=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    SOLO_LIB_OVERRIDE=$lib python tools/quick_bench.py $ARGS 2>&1 | grep -v amdgpu.ids
  done
done
