#!/bin/bash
# build/libsolo_<name>.so = the in-tree library with the QUANTISER objects (solo_nsq_row*.hip) compiled with extra flags; the other
# objects are taken from build/obj as __graft_entry__.build() left them (A/B timing on the GPU: tools/gpu_ab_quick.sh):
#   tools/build_variant_nsq.sh prio1 -DSX_NSQ_PRIO=1
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/var_$name
hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-pass-failed -DSOLO_WITH_ENCODER "$@" -c solo_amd/csrc/solo_nsq_row.hip -o build/var_$name/solo_nsq_row.o &
hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-pass-failed -DSOLO_WITH_ENCODER "$@" -c solo_amd/csrc/solo_nsq_row_wb.hip -o build/var_$name/solo_nsq_row_wb.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC build/obj/solo_api.o build/obj/solo_api_wb.o build/obj/solo_enc_k.o build/obj/solo_enc_k_wb.o build/obj/solo_enc_front_k.o build/obj/solo_enc_front_k_wb.o build/var_$name/solo_nsq_row.o build/var_$name/solo_nsq_row_wb.o -o build/libsolo_$name.so
echo "$@" > build/libsolo_$name.flags
ls -la build/libsolo_$name.so
