# the analysis kernel alone (SOLO_ENC_CHUNK=50: one launch of 50 packets before anything else runs) for several library builds
for lib in "$@"; do for r in 1 2; do echo -n "$lib: "; SOLO_ENC_CHUNK=50 SOLO_LIB_OVERRIDE=$lib timeout 120 python tools/quick_bench.py 4096 50 2>&1 | grep kernels; done; done
