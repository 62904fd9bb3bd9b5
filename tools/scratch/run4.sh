cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "nt3" 2>&1 | tail -15
timeout 600 python tools/bench_kernels.py 2>&1 | grep -E "linear_act_split|nt3" | cut -c1-220
