# co-run and quantiser-alone timing of what-if builds:  bash tools/debug/exp_whatif.sh lib...
for lib in "$@"; do
  echo -n "$lib co-run: "; SOLO_LIB_OVERRIDE=$lib timeout 120 python tools/quick_bench.py 4096 50 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' | sed 's/parity.*| encode/encode/; s/decode.*kernels/kernels/'; echo
  echo -n "$lib alone : "; SOLO_ENC_CHUNK=0 SOLO_LIB_OVERRIDE=$lib timeout 120 python tools/quick_bench.py 4096 50 2>&1 | grep kernels
done
