cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "linear_act or bitwise" 2>&1 | tail -3
timeout 600 python tools/bench_kernels.py 2>&1 | grep -E "linear_act_split" | cut -c1-220
