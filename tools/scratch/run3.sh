cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backward_gpu.py -q -x -k "bf16_mode" 2>&1 | tail -15
for m in fp32 bf16; do
timeout 600 python bench.py --mode train --objects 100 --steps 10 --warmup 3 --mlp-math $m 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['dtype'], d['ms_per_step'], d['value'], d['loss']); print(d['kernel_ms_per_step'])"
done
