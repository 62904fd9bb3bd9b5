"""Which ATen ops fill large tensors in a train step (torch.profiler, eager)."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    training.train_batch(model, opt, pbs, 0.65, bucket=bucket, sync_loss=False)
    torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::full", "aten::new_zeros") and e.input_shapes and any(len(s) and (s[0] if isinstance(s[0], int) else 0) > 100000 for s in e.input_shapes if isinstance(s, (list, tuple))):
        print(e.name, e.input_shapes, [str(f)[:100] for f in (e.stack or [])[:6]])
