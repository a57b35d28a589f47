L=build/libsolo_exp.so
echo "== base"; SOLO_LIB_OVERRIDE=$L timeout 120 python tools/quick_bench.py 4096 50 2>&1 | grep -v amdgpu.ids
echo "== no coding (SOLO_EXP_SKIP=2)"; SOLO_EXP_SKIP=2 SOLO_LIB_OVERRIDE=$L timeout 120 python tools/quick_bench.py 4096 50 2>&1 | grep -v amdgpu.ids
echo "== no quantiser (SOLO_EXP_SKIP=1)"; SOLO_EXP_SKIP=1 SOLO_LIB_OVERRIDE=$L timeout 120 python tools/quick_bench.py 4096 50 2>&1 | grep -v amdgpu.ids
echo "== neither (SOLO_EXP_SKIP=3)"; SOLO_EXP_SKIP=3 SOLO_LIB_OVERRIDE=$L timeout 120 python tools/quick_bench.py 4096 50 2>&1 | grep -v amdgpu.ids
for lag in 1 2 3; do echo "== coding on the analysis stream, lag $lag"; SOLO_EXP_CODE_LAG=$lag SOLO_LIB_OVERRIDE=$L timeout 120 python tools/quick_bench.py 4096 50 2>&1 | grep -v amdgpu.ids; done
echo "== base"; SOLO_LIB_OVERRIDE=$L timeout 120 python tools/quick_bench.py 4096 50 2>&1 | grep -v amdgpu.ids
