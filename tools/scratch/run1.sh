cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_backward_gpu.py -q -x -k "wgrad" 2>&1 | tail -3
timeout 300 python tools/bench_kernels.py 2>&1 | grep -i -E "wgrad" | cut -c1-200
bash tools/step_breakdown.sh train_n100 --mode train --objects 100 2>&1 | cut -c1-250
