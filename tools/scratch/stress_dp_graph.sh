#!/bin/bash
# the graphed data-parallel train step in a one-rank RCCL group, repeatedly (capture races with the process group's watchdog would abort)
mkdir -p gpurun_out/stress
for i in 1 2 3 4; do
  for a in "--objects 100 --calibrator 1" "--objects 100" "--objects 36" "--objects 100 --mlp-math bf16"; do
    DFOL_BENCH_FORCE_PG=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29700 + RANDOM % 200)) timeout 300 python bench.py --gpus 1 --steps 10 --mode train $a > gpurun_out/stress/out.json 2> gpurun_out/stress/err.txt
    rc=$?
    python - "$rc" "$a" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/stress/out.json").read().strip().splitlines()[-1])
    print("rc", sys.argv[1], sys.argv[2], "|", d["config"]["launch"][:20], round(d["ms_per_step"], 2), d["replicas_equal"], d["ranks"]["backend"])
except Exception as e:
    print("rc", sys.argv[1], sys.argv[2], "| FAILED", e)
PY
  done
done
