#!/usr/bin/env python3
"""Times the fused pair kernel inside the north-star step (256 questions x N objects, one relation column per image) for several library
builds and arithmetic modes, interleaved, each in a fresh process.
usage: python tools/lab/ab_pair.py spec ...    spec = [lib.so][@math[+form]]   ("" = the default library; math = f16x2 | bf16x3 | f32; form = pingpong:
round 4's schedule of the f16x2 kernel); env LAB_N (default 100)"""
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
for k in ("dfol_pair_ll_h2_f32", "dfol_pair_ll_split_f32", "dfol_pair_ll_packed_f32"):
    if k in t and t[k][0]:
        n, s = t[k]
        print("%%s %%.2f" %% (k[10:], s / n * 1e6))
''' % ROOT
specs = sys.argv[1:] or [""]
res = {s: [] for s in specs}
for rep in range(3):
    for s in specs:
        lib, _, math = s.partition("@")
        form = None
        if "+" in math:
            math, _, form = math.partition("+")
        env = {k: v for k, v in os.environ.items() if k not in ("DFOL_LIB", "DFOL_PAIR_MATH", "DFOL_PAIR_H2_FORM")}
        if lib:
            env["DFOL_LIB"] = lib
        if math:
            env["DFOL_PAIR_MATH"] = math
        if form:
            env["DFOL_PAIR_H2_FORM"] = form
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, cwd=ROOT)
        lines = [l for l in out.stdout.strip().splitlines() if l.startswith("ll_")]
        res[s].append(lines[-1] if lines else "ERR " + out.stderr[-300:])
for s in specs:
    print("%-40s us per launch: %s" % (s or "default", " | ".join(res[s])))
