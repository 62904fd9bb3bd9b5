#!/bin/bash
# Runs on the GPU box: kernel-trace of bench.py without the stress/CPU legs, then per-step GPU-busy vs wall.
# usage: tools/step_breakdown.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/steps_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 $R/bench.py --steps 20 --warmup 2 --cpu-sample 0 --stress-preds 0 "$@" > $OUT/bench.log 2>&1
cd $R
tail -1 $OUT/bench.log | cut -c1-400
python3 - "$OUT" <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
db = sqlite3.connect(glob.glob(os.path.join(out, "stats", "**", "*.db"), recursive=True)[0])
rows = list(db.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc"))
steps = max([n for name, n, _, _ in rows if "box_positions" in name] or [1])     # one launch per step
tot = 0.0
print("| kernel | calls/step | us/step |")
for name, n, s, a in rows:
    if n % steps:
        continue
    tot += s / steps / 1e3
    print("| %s | %d | %.1f |" % (name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70], n // steps, s / steps / 1e3))
print("GPU-busy per step: %.1f us over %d launches" % (tot, sum(n // steps for _, n, _, _ in rows if n % steps == 0)))
PY
find $OUT -name "*.db" -delete
