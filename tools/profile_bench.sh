#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats + HBM traffic counters (separate passes) of bench.py.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --cpu-sample 0 --fresh-batches 0 --sustain 0.5 $*"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o run -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o run -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
cd $R
python3 tools/summarize_profile.py $OUT > $OUT/summary.md 2>&1   # also writes $OUT/traffic.json and $OUT/roofline_rocprof.json
python3 - $OUT/roofline_rocprof.json profiles/roofline_rocprof.json <<'PY'      # merge this shape's entry into the committed file
import json, os, sys
src, dst = sys.argv[1:3]
if os.path.exists(src):
    cur = json.load(open(dst)) if os.path.exists(dst) else {}
    cur.update(json.load(open(src)))
    json.dump(cur, open(dst, "w"), indent=1, sort_keys=True)
PY
cp profiles/roofline_rocprof.json $OUT/roofline_rocprof_merged.json 2>/dev/null
find $OUT -name "*.db" -delete               # the summary has what matters; keep gpurun_out small
cat $OUT/summary.md
