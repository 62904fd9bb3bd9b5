#!/bin/bash
# A/B of two library builds on the inference step: tools/scratch/ab_infer.sh <lib A or ""> <lib B> [bench args]
A=$1; B=$2; shift 2
for r in 1 2 3; do for L in "$A" "$B"; do
  DFOL_LIB=$L timeout 300 python bench.py --cpu-sample 0 --stress-preds 0 --steps 100 --streamed 0 --sustain 0 --parity-all 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib', sys.argv[1] or 'default', round(d['value']), round(d['ms_per_step'],4), d['kernel_ms_per_step'].get('dfol_pair_ll_split_f32'))" "$L"
done; done
