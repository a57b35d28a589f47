echo "WB default"; python tools/quick_bench.py 4096 10 32000 2>&1 | grep -v amdgpu
echo "WB CHUNK=0"; SOLO_ENC_CHUNK=0 python tools/quick_bench.py 4096 10 32000 2>&1 | grep -v amdgpu
echo "WB GROUP=1792"; SOLO_ENC_GROUP=1792 python tools/quick_bench.py 4096 10 32000 2>&1 | grep -v amdgpu
echo "WB GROUP=2048"; SOLO_ENC_GROUP=2048 python tools/quick_bench.py 4096 10 32000 2>&1 | grep -v amdgpu
echo "WB N=2048"; python tools/quick_bench.py 2048 10 32000 2>&1 | grep -v amdgpu
echo "WB N=1024"; python tools/quick_bench.py 1024 10 32000 2>&1 | grep -v amdgpu
