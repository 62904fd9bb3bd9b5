cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_backward_gpu.py -q -x 2>&1 | tail -3
for o in 100; do
timeout 600 python bench.py --mode train --objects $o --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['loss']); print(d['kernel_ms_per_step']); print(d['roofline'])"
done
