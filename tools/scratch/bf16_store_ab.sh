#!/bin/bash
# A/B of the bf16 mode's activation storage at N = 100 (bench.py --mode train): bf16 storage (default), fp32 storage (DFOL_BF16_STORE=0), fp32 mode
mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["ms_per_step"], d["value"], {k: d.get(k) for k in ("peak_memory_gb","step_path","dtype")})'
for s in 1 0; do
  DFOL_BF16_STORE=$s timeout 300 python bench.py --mode train --mlp-math bf16 --objects ${1:-100} --steps 20 --warmup 5 --cpu-sample 0 2>gpurun_out/ab_err_$s.txt | tail -1 > gpurun_out/ab_bf16_store$s.json
  python -c "$P" "bf16 store=$s" < gpurun_out/ab_bf16_store$s.json || tail -5 gpurun_out/ab_err_$s.txt
done
timeout 300 python bench.py --mode train --objects ${1:-100} --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/ab_fp32.json
python -c "$P" "fp32" < gpurun_out/ab_fp32.json
