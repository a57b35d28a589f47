for c in 1 2 3 5; do echo "CHUNK=$c"; SOLO_ENC_CHUNK=$c SOLO_LIB_OVERRIDE=build/v5.so python tools/quick_bench.py 4096 50 2>&1 | grep -v amdgpu; done
for k in 0 1 2 3; do echo "EXP_SKIP=$k"; SOLO_EXP_SKIP=$k SOLO_LIB_OVERRIDE=build/v5x.so python tools/quick_bench.py 4096 50 2>&1 | grep -v amdgpu; done
