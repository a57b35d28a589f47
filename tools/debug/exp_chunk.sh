L=${1:-build/libsolo_exp.so}
for c in 1 10 50; do for sk in 0 1 3; do echo "== SOLO_ENC_CHUNK=$c SOLO_EXP_SKIP=$sk"; SOLO_EXP_SKIP=$sk SOLO_ENC_CHUNK=$c SOLO_LIB_OVERRIDE=$L timeout 120 python tools/quick_bench.py 4096 50 2>&1 | grep -v amdgpu.ids; done; done
