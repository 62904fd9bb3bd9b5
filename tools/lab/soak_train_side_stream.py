"""Lab: N graph-replayed train steps (256 questions x 36 objects, the bench's program) with the pair branch on its side stream and without it, from the
same initial state: the final weights must be bit-identical (a race between the two streams would show as a difference)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from dfol_vqa_amd import training, parallel
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
finals = {}
for flag in ("1", "0"):
    os.environ["DFOL_TRAIN_PAIR_STREAM"] = flag
    args = bench.parse(["--mode", "train", "--objects", "36"])
    rank, world, device, td, share = bench.setup(args)
    torch.manual_seed(3)
    model, ontology, paths, names = bench.build_model(args, device, train=True)
    qs, pbs = bench.build_batch(args, rank, ontology, names, device, world)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4, capturable=True)
    bucket = parallel.GradBucket(params)
    step = training.GraphedTrainStep(model, opt, pbs, 0.65, bucket=bucket, warmup=1)
    t0 = time.perf_counter()
    losses = [float(step()[0]) for _ in range(steps)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    finals[flag] = ({k: v.detach().clone() for k, v in model.state_dict().items()}, losses)
    print("side stream %s: %d steps, %.3f ms per step, loss %.6f -> %.6f" % (flag, steps, dt / steps * 1e3, losses[0], losses[-1]))
same = all(torch.equal(finals["1"][0][k], finals["0"][0][k]) for k in finals["1"][0]) and finals["1"][1] == finals["0"][1]
print("final weights and every step's loss bit-identical with and without the side stream:", same)
