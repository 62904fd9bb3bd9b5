#!/usr/bin/env python3
"""Which source lines call the small tensor constructors / selects (fill, where, cat kernels) during one eager inference step."""
import os, sys, collections, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
args = bench.parse(sys.argv[1:])
dev = torch.device("cuda", 0)
model, ontology, paths, names = bench.build_model(args, dev)
qs, pbs = bench.build_batch(args, 0, ontology, names, dev)
with torch.no_grad():
    for _ in range(5):
        model(pbs, False)
    torch.cuda.synchronize()
    seen = collections.Counter()
    def mk(orig, fn):
        def w(*a, **k):
            fr = traceback.extract_stack(limit=2)[0]
            seen[(fn, os.path.basename(fr.filename), fr.lineno, str(a[0])[:40] if a else "")] += 1
            return orig(*a, **k)
        return w
    for fn in ("full", "zeros", "ones", "where", "zeros_like", "ones_like", "cat", "stack", "full_like"):
        setattr(torch, fn, mk(getattr(torch, fn), fn))
    model(pbs, False)
    torch.cuda.synchronize()
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, k)
