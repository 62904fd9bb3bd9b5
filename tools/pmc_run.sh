#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<COUNTER1 COUNTER2 ...>" <python script or binary (path from the repo root) and args...>   (runs on the GPU box)
set -u
TAG=$1; PMC=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [[ $1 == *.py ]]; then
  rocprofv3 --pmc $PMC --kernel-trace -d $OUT/run -o run -- python3 $R/$1 "${@:2}" > $OUT/log.txt 2>&1
else
  rocprofv3 --pmc $PMC --kernel-trace -d $OUT/run -o run -- $R/$1 "${@:2}" > $OUT/log.txt 2>&1
fi
cd $R
python3 - "$OUT" <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
db = sqlite3.connect(glob.glob(os.path.join(out, "run", "**", "*.db"), recursive=True)[0])
q = ("select kernel_name, grid_size_x, counter_name, count(*), avg(value) from counters_collection "
     "group by kernel_name, grid_size_x, counter_name order by kernel_name, grid_size_x, counter_name")
rows = list(db.execute(q))
dur = {(n, g): (c, a) for n, g, c, a in db.execute("select name, grid_x, count(*), avg(duration) from kernels group by name, grid_x")}
last = None
for name, gx, cn, n, v in rows:
    key = (name, gx)
    if key != last:
        d = dur.get(key, (0, 0))
        print("\n%s grid=%d  dispatches=%d avg_us=%.1f" % (name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:80], gx, d[0], d[1] / 1e3))
        last = key
    print("    %-32s %.4g" % (cn, v))
PY
find $OUT -name "*.db" -delete
