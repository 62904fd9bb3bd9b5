"""ATen / custom ops of one eager train step by device time (torch.profiler), with input shapes: where the small launches come from."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from dfol_vqa_amd import training, parallel  # noqa: E402

args = bench.parse(["--mode", "train", "--objects", "100"] + sys.argv[1:])
dev = torch.device("cuda", 0)
model, ontology, paths, names = bench.build_model(args, dev, train=True)
_, pbs = bench.build_batch(args, 0, ontology, names, dev)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
bucket = parallel.GradBucket(params)
for _ in range(2):
    training.train_batch(model, opt, pbs, 0.65, bucket=bucket, sync_loss=False)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    training.train_batch(model, opt, pbs, 0.65, bucket=bucket, sync_loss=False)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
rows = sorted(rows, key=lambda r: -getattr(r, "self_device_time_total", getattr(r, "self_cuda_time_total", 0)))
rows = [r for r in rows if r.key.startswith("aten::") or r.key.startswith("dfol")]
for r in rows[:60]:
    t = getattr(r, "self_device_time_total", getattr(r, "self_cuda_time_total", 0))
    if t > 0:
        print("%8.1f us  x%-3d %-32s %s" % (t, r.count, r.key[:32], str(r.input_shapes)[:150]))
