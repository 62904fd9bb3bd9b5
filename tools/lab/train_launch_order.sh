#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of the train bench; prints the launches of the LAST step in order (start offset, duration, gap before).
# usage: tools/lab/train_launch_order.sh <tag> [bench args...]     -> gpurun_out/order_<tag>.txt
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/order_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o run -- python3 $R/bench.py --mode train --steps 6 --warmup 2 --cpu-sample 0 --stress-preds 0 --sustain 0 "$@" > $OUT/bench.log 2>&1
cd $R
tail -1 $OUT/bench.log | cut -c1-300
python3 - "$OUT" > $R/gpurun_out/order_$TAG.txt <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
db = sqlite3.connect(glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
rows = list(db.execute("select name, start, end from %s order by start" % kt))
firsts = [i for i, r in enumerate(rows) if "box_positions" in r[0]]
i0 = firsts[-1]
# a step starts with the zeroing of the bucket, a few launches before box_positions: take from the end of the previous step's last Adam kernel
prev_end = firsts[-2] if len(firsts) > 1 else 0
per = i0 - prev_end
last = rows[i0 - 4:]
print("launches from the last box_positions - 4 to the end of the trace: %d (launches between the last two box_positions: %d)" % (len(last), per))
t0 = last[0][1]
short = lambda s: s.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")[:110]
tot = 0
for k, (n, s, e) in enumerate(last):
    gap = (s - last[k - 1][2]) / 1e3 if k else 0.0
    tot += (e - s) / 1e3
    print("%4d %9.1f us  dur %8.1f  gap %6.1f  %s" % (k, (s - t0) / 1e3, (e - s) / 1e3, gap, short(n)))
print("busy %.1f us, span %.1f us" % (tot, (last[-1][2] - t0) / 1e3))
PY
rm -rf $OUT/trace
tail -3 $R/gpurun_out/order_$TAG.txt
