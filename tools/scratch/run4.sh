cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "persistent" 2>&1 | tail -3
for sk in 0 1 2 4; do
  (cd dfol_vqa_amd/csrc && rm -f dfol_dense_split.o && make EXTRA=-DLS_SKEW=$sk 2>&1 | grep -E "error")
  echo "== skew $sk"; timeout 600 python tools/bench_kernels.py 2>&1 | grep -E '"linear_act_split"' | grep 2534400 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['M'], d['N'], d['K'], round(d['ms'], 4))"
done
