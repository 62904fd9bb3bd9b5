for rep in 1 2; do
for cfg in "1 1" "0 0"; do set -- $cfg
for mm in fp32 bf16; do
DFOL_FUSED_ADAM=$1 DFOL_DIRECT_GRAD=$2 python bench.py --mode train --mlp-math $mm --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_adam=$1 direct_grad=$2 $mm', round(d['value']), round(d['ms_per_step'],3))"
done; done; done
