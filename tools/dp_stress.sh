#!/bin/bash
# N consecutive FRESH-PROCESS runs of the data-parallel train bench over RCCL (one-rank nccl group: what a one-GPU box can run of it):
# the default launch form (two step graphs, the bucket's all-reduce issued eagerly between the replays).  Every run must exit 0 with
# replicas_equal true.  The log is the evidence VERDICT r3 #1 asks for (profiles/r04_dp_stress.txt).
# usage: tools/dp_stress.sh [runs] [extra bench args...]      e.g. tools/dp_stress.sh 50     tools/dp_stress.sh 10 --graph-collective 1
N=${1:-50}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/dp_stress
LOG=gpurun_out/dp_stress/log${DFOL_STRESS_TAG:-}.txt
echo "# dp_stress: $N fresh-process runs of: DFOL_BENCH_FORCE_PG=1 python bench.py --gpus 1 --mode train --steps 6 --warmup 2 --batch 32 --objects 36 --cpu-sample 0 --stress-preds 0 --sustain 0 --fresh-batches 0 $*" > $LOG
echo "# $(date -u +%FT%TZ)  $(python -c 'import torch;print(torch.__version__, torch.cuda.get_device_name(0))' 2>/dev/null)" >> $LOG
ok=0; bad=0
for i in $(seq 1 $N); do
  port=$((29700 + i))
  DFOL_BENCH_FORCE_PG=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port timeout 300 python bench.py --gpus 1 --mode train --steps 6 --warmup 2 --batch 32 --objects 36 \
      --cpu-sample 0 --stress-preds 0 --sustain 0 --fresh-batches 0 "$@" > gpurun_out/dp_stress/run.out 2> gpurun_out/dp_stress/run.err
  rc=$?
  line=$(grep '^{' gpurun_out/dp_stress/run.out | tail -1 | python -c '
import json,sys
try:
    d=json.loads(sys.stdin.read())
    print("backend=%s replicas_equal=%s ms_per_step=%.3f launch=%r" % (d["ranks"]["backend"], d["replicas_equal"], d["ms_per_step"], d["config"]["launch"][:60]))
except Exception as e:
    print("no json line (%r)" % (e,))')
  echo "run $i rc=$rc $line" >> $LOG
  if [ $rc -eq 0 ] && echo "$line" | grep -q "replicas_equal=True"; then ok=$((ok+1)); else bad=$((bad+1)); tail -5 gpurun_out/dp_stress/run.err >> $LOG; fi
done
echo "# total: $ok ok, $bad failed of $N" >> $LOG
tail -1 $LOG
