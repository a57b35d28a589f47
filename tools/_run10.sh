cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_receiver.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/quick_bench.py 4096 10 32000 2>&1 | tail -2
timeout 300 python tools/quick_bench.py 2048 10 32000 2>&1 | tail -2
