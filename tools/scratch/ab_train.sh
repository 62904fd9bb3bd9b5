#!/bin/bash
# A/B of two library builds on the train step: tools/scratch/ab_train.sh <lib A or ""> <lib B> [bench args]
A=$1; B=$2; shift 2
for r in 1 2; do for L in "$A" "$B"; do
  DFOL_LIB=$L timeout 300 python bench.py --mode train --steps 20 --warmup 5 --cpu-sample 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib', sys.argv[1] or 'default', d['ms_per_step'])" "$L"
done; done
