#!/bin/bash
# build/libsolo_stops.so: the in-tree library with the encoder kernels compiled with -DSX_STOPS (waves end at a chosen site, solo_wave.h) and the
# API with -DSX_EXPERIMENTS (SOLO_EXP_SKIP); the other objects are taken from build/obj as __graft_entry__.build() left them
set -e
cd "$(dirname "$0")/.."
mkdir -p build/var_stops
F="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-pass-failed -DSOLO_WITH_ENCODER"
hipcc $F -DSX_EXPERIMENTS -DSX_STOPS -c solo_amd/csrc/solo_api.hip -o build/var_stops/solo_api.o &
hipcc $F -DSX_STOPS "$@" -c solo_amd/csrc/solo_enc_k.hip -o build/var_stops/solo_enc_k.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC build/var_stops/solo_api.o build/obj/solo_api_wb.o build/var_stops/solo_enc_k.o build/obj/solo_enc_k_wb.o build/obj/solo_enc_front_k.o build/obj/solo_enc_front_k_wb.o build/obj/solo_nsq_row.o build/obj/solo_nsq_row_wb.o -o build/libsolo_stops.so
ls -la build/libsolo_stops.so
