"""Lab: the headline step (256 questions x N objects, one resident batch) in four loop forms - graph replay one at a time / two in flight, and the native
executor's eager launches one at a time / two in flight (forward_async) - same process, interleaved rounds, ms per step."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from dfol_vqa_amd import _lib as L
from dfol_vqa_amd.interpreter import GraphedForward
args = bench.parse(["--objects", sys.argv[1] if len(sys.argv) > 1 else "100"])
rank, world, device, td, share = bench.setup(args)
model, ontology, paths, names = bench.build_model(args, device)
qs, pbs = bench.build_batch(args, rank, ontology, names, device, world)
g = GraphedForward(model, pbs)
def graph_serial(n):
    for _ in range(n): g()
def graph_pipe(n):
    p = None
    for _ in range(n):
        t = g.submit()
        if p is not None: g.collect(p)
        p = t
    g.collect(p)
def eager_serial(n):
    for _ in range(n): model(pbs, False)
def eager_pipe(n):
    p = None
    for _ in range(n):
        t = model.forward_async(pbs, False)
        if p is not None: p.result()
        p = t
    p.result()
qs2, pbs2 = bench.build_batch(args, rank, ontology, names, device, world)
g2 = GraphedForward(model, pbs2)
streams = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
def graph_two_streams(n):
    gs, p = [g, g2], []
    for i in range(n):
        with torch.cuda.stream(streams[i % 2]):
            t = gs[i % 2].submit()
        p.append((gs[i % 2], t))
        if len(p) > 2:
            a, b = p.pop(0); a.collect(b)
    for a, b in p: a.collect(b)
forms = [("two graphs on two streams", graph_two_streams), ("graph replay, one at a time", graph_serial), ("graph replay, two in flight", graph_pipe), ("executor launches, one at a time", eager_serial),
         ("executor launches, two in flight", eager_pipe)]
with torch.no_grad():
    for _, f in forms: f(10)
    L.PATH_COUNTS.clear()
    for rnd in range(3):
        for name, f in forms:
            torch.cuda.synchronize(); t0 = time.perf_counter(); f(200); torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 200 * 1e3
            print("round %d  %-34s %.3f ms/step  %.0f q/s" % (rnd, name, ms, args.batch / ms * 1e3))
print({k: v for k, v in L.PATH_COUNTS.items() if "program" in k})
