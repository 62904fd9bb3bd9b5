#!/usr/bin/env python3
"""cProfile of the eager inference step (host side): where the ~15 us per launch go.  usage: python tools/scratch/profile_host.py [bench args]"""
import cProfile, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
args = bench.parse(sys.argv[1:])
dev = torch.device("cuda", 0)
model, ontology, paths, names = bench.build_model(args, dev)
qs, pbs = bench.build_batch(args, 0, ontology, names, dev)
with torch.no_grad():
    for _ in range(20):
        model(pbs, False)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        model(pbs, False)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
