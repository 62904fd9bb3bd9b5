#!/usr/bin/env python3
"""torch.profiler view of one eager train step (which aten ops the non-library launches of a step belong to).
usage: python tools/scratch/profile_train_ops.py [bench.py train-mode args]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from dfol_vqa_amd import training

args = bench.parse(["--mode", "train"] + sys.argv[1:])
dev = torch.device("cuda", 0)
model, ontology, paths, names = bench.build_model(args, dev, train=True)
_, pbs = bench.build_batch(args, 0, ontology, names, dev)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
for _ in range(3):
    training.train_batch(model, opt, pbs, clip_norm=0.65, sync_loss=False)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        training.train_batch(model, opt, pbs, clip_norm=0.65, sync_loss=False)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=50, max_shapes_column_width=60))
