#!/bin/bash
# Runs on the GPU box: kernel-trace of bench.py without the stress/CPU legs, then per-step GPU-busy vs wall.
# (inference steps one replay at a time - `--pipeline 0` - so that a kernel's duration is its own: the replay lanes of the default loop run two steps
# side by side and every launch then shares the device)
# usage: tools/step_breakdown.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/steps_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 $R/bench.py --steps 20 --warmup 2 --cpu-sample 0 --stress-preds 0 --fresh-batches 0 --sustain 0 --pipeline 0 "$@" > $OUT/bench.log 2>&1
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
# idle gaps of the last step (from its box_positions launch, the first kernel of a forward, to the end of the trace)
t0 = list(db.execute("select max(start) from kernels where name like '%box_positions%'"))[0][0]
last = list(db.execute("select name, start, end from kernels where start >= ? order by start", (t0,)))
gaps = [((b[1] - a[2]) / 1e3, a[0], b[0]) for a, b in zip(last, last[1:])]
short = lambda s: s.replace("void ", "").replace("at::native::", "").split("(")[0].split("<")[0][-32:]
print("last step: %d launches, %.0f us from first launch to last end, %.0f us idle in %d gaps of more than 2 us; largest:" % (
    len(last), (last[-1][2] - last[0][1]) / 1e3, sum(g[0] for g in gaps if g[0] > 0), sum(1 for g in gaps if g[0] > 2)))
for g, a, b in sorted(gaps, reverse=True)[:10]:
    print("  %.0f us between %s and %s" % (g, short(a), short(b)))
# the individual launches of the three largest kernels in the last step (which call is the expensive one)
for name, n, _, _ in rows[:3]:
    per = n // steps
    if n % steps or per < 2:
        continue
    d = [r[0] / 1e3 for r in db.execute("select duration from kernels where name = ? and start >= ? order by start", (name, t0))]
    print("%s, last step, us per launch: %s" % (name.split("(")[0][-40:], " ".join("%.0f" % x for x in d)))
PY
find $OUT -name "*.db" -delete
