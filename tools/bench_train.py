#!/usr/bin/env python3
"""Train-step throughput (BASELINE configs[3] shape on one GPU): forward + loss + backward + clip + Adam on one resident batch.

usage: python tools/bench_train.py [--objects 36] [--batch 256] [--steps 5] [--warmup 2]
Prints one JSON line (questions/s of training).  With more than one rank (torchrun) the gradients are summed with one
all-reduce of the flat bucket per step (dfol_vqa_amd.parallel), as trainer.py:429-442 does through nn.DataParallel.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=36)
    ap.add_argument("--ragged", type=int, default=0, help="if > 0: object counts ~ U{ragged..objects} instead of a fixed count")
    ap.add_argument("--calibrator", type=int, default=0, help="1: the calibrator phases (cur6-7): oracle frozen, only the attention networks train")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("DFOL_BENCH_SHARE_GPU") == "1"    # debugging aid for one-GPU boxes: every rank on cuda:0, gradients over gloo
    local = 0 if share else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    group = None
    if world > 1:
        import torch.distributed as td
        td.init_process_group("gloo") if share else td.init_process_group("nccl", device_id=device)
        group = td.group.WORLD
    from dfol_vqa_amd import experiment, training
    from dfol_vqa_amd import synthetic as syn
    tmp = tempfile.mkdtemp(prefix="dfol_train_")
    paths, names = syn.write_synthetic_ontology(tmp)
    if args.calibrator:
        cfg = syn.reference_config(paths, dropout=0.0, activate_attention_transfer=True)
    else:
        cfg = syn.reference_config(paths, dropout=0.0, freeze_featurizer=False, freeze_attribute_network=False, freeze_relation_network=False,
                                   freeze_embedding_network=False)     # the oracle-training phases (cur1-5) of the curriculum
    ontology = experiment.build_ontology(cfg)
    model = experiment.build_model(cfg, ontology)
    bench.init_weights(model)
    model = model.to(device).train()
    if args.ragged > 0:                                      # ragged scenes (BASELINE configs[2]/[3] shape)
        import numpy as np
        rng = np.random.RandomState(rank)
        orig = syn.feature_scene
        syn.feature_scene = lambda qid, n, dim: orig(qid, int(rng.randint(args.ragged, args.objects + 1)), dim)
    _, pbs = bench.build_batch(args, rank, ontology, names, device)
    if args.calibrator:                                      # the LSTM inputs need token embeddings (random here, GloVe in the reference)
        voc = list(ontology._vocabulary["idx_to_arg"])
        emb = (torch.randn(len(voc), 300) * 0.1).to(device)
        for pb in pbs:
            pb._meta_data = {"index": {t: i for i, t in enumerate(voc)}, "embedding": emb}
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    torch.cuda.reset_peak_memory_stats()
    for _ in range(args.warmup):
        loss, _ = training.train_batch(model, opt, pbs, 0.65, global_batch_size=args.batch * world, group=group)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = training.train_batch(model, opt, pbs, 0.65, global_batch_size=args.batch * world, group=group)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if rank == 0:
        print(json.dumps({"metric": "training questions/s (forward + backward + Adam)", "value": args.batch * world / dt, "ms_per_step": dt * 1e3,
                          "n_gpus": world, "phase": "calibrator" if args.calibrator else "oracle", "objects": args.objects if not args.ragged else [args.ragged, args.objects], "batch_per_gpu": args.batch, "loss": loss,
                          "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == "__main__":
    main()
