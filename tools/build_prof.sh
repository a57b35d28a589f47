#!/bin/bash
# build/libsolo_prof.so: the in-tree library with the decoder (solo_api.hip) and the encoder's analysis / coding kernels (solo_enc_k.hip) compiled with
# -DSX_PROF (cycle counters per section, solo_wave.h); the other objects are taken from build/obj as __graft_entry__.build() left them
set -e
cd "$(dirname "$0")/.."
mkdir -p build/var_prof
F="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-pass-failed -DSOLO_WITH_ENCODER -DSX_PROF"
for f in solo_api solo_enc_k; do hipcc $F -c solo_amd/csrc/$f.hip -o build/var_prof/$f.o & done
wait
hipcc --offload-arch=gfx950 -shared -fPIC build/var_prof/solo_api.o build/obj/solo_api_wb.o build/var_prof/solo_enc_k.o build/obj/solo_enc_k_wb.o build/obj/solo_enc_front_k.o build/obj/solo_enc_front_k_wb.o build/obj/solo_nsq_row.o build/obj/solo_nsq_row_wb.o -o build/libsolo_prof.so
ls -la build/libsolo_prof.so
