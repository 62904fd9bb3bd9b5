"""Lab: soak of interpreter.ReplayLanes - N steps of the headline batch on two lanes, every step's log-probabilities and answers compared with the
first step's (same inputs: they must be bit-identical; a race between the lanes would show as a difference)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from dfol_vqa_amd.interpreter import ReplayLanes
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
args = bench.parse(["--objects", "100"])
rank, world, device, td, share = bench.setup(args)
model, ontology, paths, names = bench.build_model(args, device)
pbs = [bench.build_batch(args, rank, ontology, names, device, world)[1] for _ in range(2)]
with torch.no_grad():
    ref = model(pbs[0], False)
    want_lp, want_ans = ref["log_probability"].cpu(), ref["answer"]
    lanes = ReplayLanes(model, pbs)
    pending, bad = [], 0
    t0 = time.perf_counter()
    for i in range(steps):
        pending.append(lanes.submit())
        if len(pending) > 1:
            r = lanes.collect(pending.pop(0))
            bad += int(not (torch.equal(r["log_probability"], want_lp) and r["answer"] == want_ans))
    for t in pending:
        r = lanes.collect(t)
        bad += int(not (torch.equal(r["log_probability"], want_lp) and r["answer"] == want_ans))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("soak: %d steps on %d lanes in %.1f s (%.3f ms per step, %.0f questions/s); steps whose results differ from the eager forward's: %d"
      % (steps, len(lanes), dt, dt / steps * 1e3, steps * args.batch / dt, bad))
