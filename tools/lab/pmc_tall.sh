#!/bin/bash
# Counter passes (rocprofv3 --pmc, one group per pass) over tools/lab/time_tall.py: where the tall products' cycles go.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_tall
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/p$i -o run --output-format csv -- python3 $R/tools/lab/time_tall2.py > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
agg = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        for key in ("linear_act_split_kernel", "pair_wgrad_fused_kernel", "tall_h2_kernel"):
            if key in name:
                name = name[name.index(key):].split("(")[0][:60]
        if "linear_act_split" not in name and "pair_wgrad_fused" not in name and "tall_h2" not in name:
            continue
        key = (name, row["Counter_Name"])
        a = agg.setdefault(key, [0.0, 0])
        a[0] += float(row["Counter_Value"]); a[1] += 1
for (name, c), (v, n) in sorted(agg.items()):
    print("%-62s %-24s %14.0f per launch (%d)" % (name, c, v / n, n))
PY
