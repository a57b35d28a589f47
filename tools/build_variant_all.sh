#!/bin/bash
# build/libsolo_<name>.so = the whole library compiled with extra / other flags (A/B timing of compiler options on the GPU):
#   tools/build_variant_all.sh o3 -O3        (the flags are appended to the standard ones: a later -O wins)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/var_$name
for f in solo_api solo_api_wb solo_enc_k solo_enc_k_wb solo_enc_front_k solo_enc_front_k_wb solo_nsq_row solo_nsq_row_wb; do
  hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-pass-failed -DSOLO_WITH_ENCODER "$@" -c solo_amd/csrc/$f.hip -o build/var_$name/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC build/var_$name/*.o -o build/libsolo_$name.so
echo "$@" > build/libsolo_$name.flags
ls -la build/libsolo_$name.so
