#!/bin/bash
# Round-end measurement pass (runs on the GPU box): rocprofv3 stats + HBM counters, both bench lines, kernel / operator / training /
# calibrated-forward benchmarks and the per-step launch breakdowns.  Copy what should be judged from gpurun_out/ into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_bench.sh r01 > gpurun_out/prof_r01.log 2>&1
cp gpurun_out/prof_r01/traffic.json profiles/traffic.json
python bench.py > gpurun_out/bench_n100.json 2> gpurun_out/bench_n100.err
python bench.py --objects 36 > gpurun_out/bench_n36.json 2> gpurun_out/bench_n36.err
python tools/bench_kernels.py > gpurun_out/kb.jsonl 2>&1
python tools/bench_ops.py > gpurun_out/ops.jsonl 2>&1
python tools/bench_train.py --objects 36 > gpurun_out/train.jsonl 2>&1
python tools/bench_train.py --objects 100 >> gpurun_out/train.jsonl 2>&1
python tools/bench_train.py --objects 100 --ragged 10 >> gpurun_out/train.jsonl 2>&1
python tools/bench_train.py --objects 100 --calibrator 1 >> gpurun_out/train.jsonl 2>&1
python tools/bench_calibrated.py > gpurun_out/calibrated.txt 2>&1
bash tools/step_breakdown.sh n100 > gpurun_out/steps_n100.md 2>&1
bash tools/step_breakdown.sh n36 --objects 36 > gpurun_out/steps_n36.md 2>&1
tail -1 gpurun_out/bench_n100.json | cut -c1-300
tail -1 gpurun_out/bench_n36.json | cut -c1-200
