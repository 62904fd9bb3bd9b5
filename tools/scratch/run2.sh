cd $GRAFT_REPO_ROOT
bash tools/step_breakdown.sh train_n100 --mode train --objects 100 > gpurun_out/train_breakdown.md 2>&1
head -30 gpurun_out/train_breakdown.md | cut -c1-200; tail -1 gpurun_out/train_breakdown.md
timeout 600 python bench.py --mode train --objects 100 --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
