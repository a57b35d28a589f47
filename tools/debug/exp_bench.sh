# bench.py main leg under library / environment variants (GPU box):  bash tools/debug/exp_bench.sh
run() { echo "== $*"; env "$@" timeout 200 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['parity_checked'])"; }
run SOLO_LIB_OVERRIDE=build/libsolo_base3.so
run SOLO_LIB_OVERRIDE=build/libsolo_pa3.so
run SOLO_LIB_OVERRIDE=build/libsolo_base3.so SOLO_ENC_CHUNK=2
run SOLO_LIB_OVERRIDE=build/libsolo_pa3.so SOLO_ENC_CHUNK=2
run SOLO_LIB_OVERRIDE=build/libsolo_base3.so
run SOLO_LIB_OVERRIDE=build/libsolo_pa3.so
