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
# the same command with the steps one replay at a time (--pipeline 0): in the default loop two steps run side by side on two streams and every launch of
# the timed region shares the device, so the `--stats` average of a kernel is no longer its duration alone; in this pass it is
mkdir -p $OUT/serial
rocprofv3 --kernel-trace --stats -d $OUT/serial/stats -o run -- python3 $R/bench.py $ARGS --pipeline 0 > $OUT/serial/stats.log 2>&1
cd $R
python3 tools/summarize_profile.py $OUT/serial > $OUT/summary_serial.md 2>&1
python3 tools/summarize_profile.py $OUT > $OUT/summary.md 2>&1   # also writes $OUT/traffic.json and $OUT/roofline_rocprof.json
python3 - $OUT/roofline_rocprof.json profiles/roofline_rocprof.json <<'PY'      # merge this shape's entry into the committed file
import json, os, sys
src, dst = sys.argv[1:3]
if os.path.exists(src):
    cur = json.load(open(dst)) if os.path.exists(dst) else {}
    new = json.load(open(src))
    ser = os.path.join(os.path.dirname(src), "serial", "roofline_rocprof.json")
    if os.path.exists(ser):                  # the `--pipeline 0` pass of the same command: the `--stats` row with every launch alone on the device
        for k, v in json.load(open(ser)).items():
            d = v.get("dominant")
            if k in new and d and new[k].get("dominant"):
                new[k]["dominant"]["serial_run"] = {"command": v.get("command"), "all_launches": d.get("all_launches"),
                                                    "all_launches_avg_us": d.get("all_launches_avg_us"), "last_launches_avg_us": d.get("avg_us")}
    cur.update(new)
    json.dump(cur, open(dst, "w"), indent=1, sort_keys=True)
PY
cp profiles/roofline_rocprof.json $OUT/roofline_rocprof_merged.json 2>/dev/null
find $OUT -name "*.db" -delete               # the summary has what matters; keep gpurun_out small
cat $OUT/summary.md
