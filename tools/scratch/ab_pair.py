#!/usr/bin/env python3
"""Times the fused pair kernel (256 questions x N objects, one relation column per image) for several library builds, interleaved.
usage: python tools/scratch/ab_pair.py libA.so libB.so ...   ("" = the default library); env LAB_N (default 100)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
import bench
args = bench.parse(["--objects", os.environ.get("LAB_N", "100")])
dev = torch.device("cuda", 0)
model, ontology, paths, names = bench.build_model(args, dev)
qs, pbs = bench.build_batch(args, 0, ontology, names, dev)
from dfol_vqa_amd import _lib as L
with torch.no_grad():
    for _ in range(5): model(pbs, False)
    torch.cuda.synchronize()
    L.enable_kernel_timing(list(L.SIGNATURES))
    for _ in range(30): model(pbs, False)
    torch.cuda.synchronize()
    t = L.disable_kernel_timing()
n, s = t["dfol_pair_ll_split_f32"]
print("%%.2f" %% (s / n * 1e6))
''' % ROOT
libs = sys.argv[1:] or [""]
res = {l: [] for l in libs}
for rep in range(3):
    for l in libs:
        env = dict(os.environ, DFOL_LIB=l) if l else {k: v for k, v in os.environ.items() if k != "DFOL_LIB"}
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, cwd=ROOT)
        res[l].append(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-200:])
for l in libs:
    print("%-32s us per launch: %s" % (l or "default", " ".join(res[l])))
