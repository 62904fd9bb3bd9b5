#!/usr/bin/env python3
"""Train-step throughput (BASELINE configs[3]'s step): a thin front of `bench.py --mode train`, kept for the old command line.

usage: python tools/bench_train.py [--objects 36] [--ragged 0] [--calibrator 0] [--batch 256] [--steps 10] [--warmup 3] [--gpus 1]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--objects" not in argv:
        argv += ["--objects", "36"]
    bench.main(["--mode", "train"] + argv)
