"""The featurizer's dense product ([M, 2054-column rows] x [512, 2048], ELU... as the north-star step runs it) at several row counts: how much of
its time is the last, partly filled round of 128 x 128 tiles.  usage: python tools/lab/time_featurizer.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dfol_vqa_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
w = torch.randn(512, 2048, device=dev, generator=g) / 32
b = torch.randn(512, device=dev, generator=g)
for M in (8192, 16384, 24576, 25600, 32768, 49152):
    x = torch.rand(M, 2054, device=dev, generator=g)[:, :2048]
    for _ in range(3):
        _lib.linear_act(x, w, b, _lib.ACT_ELU)
    best = 1e9
    for _ in range(5):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            _lib.linear_act(x, w, b, _lib.ACT_ELU)
        e.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) / 10)
    tiles = ((M + 127) // 128) * 4
    print("M=%6d  %4d tiles (%.2f rounds of 512)  %7.1f us  %6.1f TFLOP/s algorithmic  %5.1f ns per tile-round-slot" %
          (M, tiles, tiles / 512.0, best * 1e3, 2.0 * M * 2048 * 512 / best / 1e9, best * 1e6 / tiles))
