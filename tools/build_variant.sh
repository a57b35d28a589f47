#!/bin/bash
# build/libsolo_<name>.so = the in-tree library with the QUANTISER objects compiled with extra flags (the analysis / coding / decoder objects
# are taken from build/obj as __graft_entry__.build() left them), for A/B timing on the GPU (tools/gpu_ab.sh, tools/gpu_first.sh):
#   tools/build_variant.sh r160 -DSX_NSQ_VGPR_CAP=80
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/var_$name
for f in solo_nsq_row solo_nsq_row_wb; do
  hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-pass-failed -DSOLO_WITH_ENCODER "$@" -c solo_amd/csrc/$f.hip -o build/var_$name/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC build/obj/solo_api.o build/obj/solo_api_wb.o build/obj/solo_enc_k.o build/obj/solo_enc_k_wb.o build/obj/solo_enc_front_k.o build/obj/solo_enc_front_k_wb.o build/var_$name/solo_nsq_row.o build/var_$name/solo_nsq_row_wb.o -o build/libsolo_$name.so
echo "$@" > build/libsolo_$name.flags
ls -la build/libsolo_$name.so
