#!/usr/bin/env python3
"""Headline benchmark: questions/s of the ∇-FOL interpreter forward on synthetic GQA-style scenes.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--objects 100] [--batch 256]

One step = one pass of the hot path (build_scene: featurizer + oracle MLPs -> likelihoods; then the
program select -> filter -> relate -> exist for every question) over one batch of `--batch` questions
whose object features and lowered program are already resident in HBM.  Workload = BASELINE.json
configs[1] (3-hop Filter->Relate->Exist programs, fp32, batch=256) on N-object scenes; N defaults to
100, the size BASELINE.json's metric is quoted on (configs[1] itself says 36: pass --objects 36).

For --gpus N > 1 the driver launches one rank per GPU (torch.distributed.run); questions are
independent, so each rank runs its own shard with no data-path collective (weak scaling) and the
ranks only meet at the timing barriers.

Prints ONE JSON line on rank 0.  `roofline` is the dominant kernel of the step; `kernels` adds the
HBM roofline of the Relate/Filter logic kernels on >= 65536 resident predicates;
`cpu_baseline` is the CPU oracle (a port of the reference's algorithm) timed on this host.
"""

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12            # B/s, MI355X_MICROARCH.md
F32_MFMA_PEAK = 157.3e12     # FLOP/s, dense f32-input MFMA
BF16_MFMA_PEAK = 2.5e15      # FLOP/s, dense bf16 MFMA (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--objects", type=int, default=100)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--cpu-sample", type=int, default=None, help="questions in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--stress-preds", type=int, default=65536)
    ap.add_argument("--graph", type=int, default=1, help="1: replay the step as a captured HIP graph (interpreter.GraphedForward); 0: eager launches")
    return ap.parse_args()


def build_batch(args, rank, ontology, names, device):
    import dfol_vqa_amd as D
    from dfol_vqa_amd import synthetic as syn

    class Collater(D.ProgramCollaterBase):
        def __init__(self):
            super(Collater, self).__init__("select", "relate", "filter", 1, ontology=ontology)

        def collate_object_features(self, questions):
            feats = torch.cat([torch.from_numpy(q["scene"]["X"]) for q in questions], 0)
            bi = torch.cat([torch.full((q["scene"]["n"],), i, dtype=torch.int64) for i, q in enumerate(questions)])
            return feats, bi

        def collate_meta_data(self, questions):
            return {"index": {}, "embedding": torch.zeros(1, 1)}

    nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
    qs = []
    for i in range(args.batch):
        qid = rank * args.batch + i                        # scenes are keyed by question id: sharding never changes inputs
        br, last = syn.three_hop_program(qid, nouns, attrs, rels)
        qs.append(syn.question(qid, br, last, "yes", syn.feature_scene(qid, args.objects, 2048)))
    pbs = Collater().collate(qs)
    for pb in pbs:
        pb.create_sparse_tensors()
    return qs, [pb.to_cuda(device) for pb in pbs]


def init_weights(model):
    """Random weights of the reference architecture; the embedding rows get GloVe-like magnitudes so that the
    concept probabilities are sparse instead of saturated."""
    torch.manual_seed(0)
    lin = model._oracle._embedding_network.linear
    with torch.no_grad():
        lin.weight.normal_(0.0, 0.1)
        lin.bias.fill_(-2.0)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1
    # DFOL_BENCH_SHARE_GPU=1 is a debugging aid for boxes with one GPU: every rank uses cuda:0 and the ranks meet over gloo
    share = os.environ.get("DFOL_BENCH_SHARE_GPU") == "1"
    local = 0 if share else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if dist:
        import torch.distributed as td
        if share:
            td.init_process_group("gloo")
        else:
            td.init_process_group("nccl", device_id=device)      # RCCL over xGMI

    import dfol_vqa_amd as D
    from dfol_vqa_amd import _lib as L
    from dfol_vqa_amd import experiment
    from dfol_vqa_amd import synthetic as syn
    L.load()

    tmp = tempfile.mkdtemp(prefix="dfol_bench_")
    paths, names = syn.write_synthetic_ontology(tmp)
    cfg = syn.reference_config(paths)
    ontology = experiment.build_ontology(cfg)
    model = experiment.build_model(cfg, ontology)
    init_weights(model)
    model = model.to(device).eval()
    qs, pbs = build_batch(args, rank, ontology, names, device)

    eager = lambda: model(pbs, False)
    step = eager
    graphed = False
    if args.graph:
        from dfol_vqa_amd.interpreter import GraphedForward
        try:
            step = GraphedForward(model, pbs)
            graphed = True
        except Exception as e:                              # capture is an optimisation of the host side only: fall back to eager launches
            sys.stderr.write("graph capture failed (%s); running eager\n" % e)
            torch.cuda.synchronize()

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            res = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
        barrier()
        elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device="cpu" if share else device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
    total_q = args.batch * world * args.steps
    out = {"metric": "questions/sec (GQA programs, N=%d objects)" % args.objects, "value": total_q / elapsed, "unit": "questions/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE configs[1]: select->filter->relate->exist (3-hop), fp32, %d questions/GPU/step, "
                                  "%d-object synthetic scenes, full-size oracle (2048->512, 516/1036->256->300->2335)"
                                  % (args.batch, args.objects),
                      "global_batch": args.batch * world, "objects_per_scene": args.objects, "parallelism": "dp%d" % world,
                      "launch": "hip graph replay" if graphed else "eager",
                      "contraction_math": "fp32 matrix pipe" if os.environ.get("DFOL_PAIR_MATH") == "f32" else
                      "fp32 results from the bf16 matrix pipe: exact 3-way bf16 operand split, 6 of 9 piece products, fp32 accumulate (DESIGN 3.3)"}}

    if rank == 0:
        # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream ------
        names_timed = list(L.SIGNATURES)
        with torch.no_grad():
            L.enable_kernel_timing(names_timed)
            for _ in range(3):
                eager()
            torch.cuda.synchronize()
            timing = L.disable_kernel_timing()
        per_step = {k: (n / 3.0, t / 3.0) for k, (n, t) in timing.items() if n}
        dom = max(per_step, key=lambda k: per_step[k][1])
        out["kernel_ms_per_step"] = {k: round(v[1] * 1e3, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1][1])}
        O = args.batch * args.objects
        pairs = args.batch * args.objects * (args.objects - 1)
        launches, secs = per_step[dom]
        if dom == "dfol_pair_ll_split_f32":
            # The same algorithmic flops, executed on the bf16 matrix pipe as six piece products per fp32 product (three exact
            # bf16 pieces per operand, fp32 accumulate: fp32 results, csrc/dfol_pair_split.hip).  `achieved` / `frac` follow the
            # contract (ALGORITHMIC flops against the peak of the pipe that executes them); the pipe itself does 6x that work.
            flops = 2.0 * pairs * (4 * 256 + 256 * 300 + 300 * 1)
            ach = flops / secs
            out["roofline"] = {"kernel": "pair_ll32s_kernel<19> (fused pair MLP -> requested relation tiles, bf16x3 split)", "bound": "mfma",
                               "achieved": ach / 1e12, "peak": BF16_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / BF16_MFMA_PEAK,
                               "traffic": None, "launches_per_step": launches, "us_per_launch": secs / launches * 1e6,
                               "flops_per_pair": 2 * (4 * 256 + 256 * 300 + 300),
                               "executed": {"mfma_flops_per_algorithmic_flop": 6, "achieved": 6 * ach / 1e12, "frac": 6 * ach / BF16_MFMA_PEAK},
                               "vs_f32_mfma_peak": {"peak": F32_MFMA_PEAK / 1e12, "frac": ach / F32_MFMA_PEAK}}
        elif dom in ("dfol_pair_ll_f32", "dfol_pair_ll_packed_f32"):
            # reduced-form algorithmic flops per ordered pair: geometry term, 256->300 layer, and the
            # K requested embedding columns (K = 1 relation per question in this workload)
            flops = 2.0 * pairs * (4 * 256 + 256 * 300 + 300 * 1)
            ach = flops / secs
            out["roofline"] = {"kernel": "pair_ll32b_kernel<19> (fused pair MLP -> requested relation tiles)", "bound": "mfma",
                               "achieved": ach / 1e12, "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / F32_MFMA_PEAK,
                               "traffic": None, "launches_per_step": launches, "us_per_launch": secs / launches * 1e6,
                               "flops_per_pair": 2 * (4 * 256 + 256 * 300 + 300)}
        elif dom == "dfol_linear_act_f32":
            if getattr(model._oracle, "_needed_columns", False) and model._oracle.supports_needed_columns():
                flops = 2.0 * O * (2048 * 512 + 516 * 256 + 256 * 300 + 516 * 512)
            else:   # full cached tables (only the 333 relation columns of the pair embedding are computed)
                flops = 2.0 * (O * 2048 * 512 + O * (516 * 256 + 256 * 300 + 300 * 2335) + pairs * (1036 * 256 + 256 * 300 + 300 * 333))
            ach = flops / secs
            out["roofline"] = {"kernel": "linear_act_kernel (all GEMM launches of one step)", "bound": "mfma", "achieved": ach / 1e12,
                               "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / F32_MFMA_PEAK, "traffic": None}
        else:
            N = args.objects
            nbytes = args.batch * (4 * N * N + 16 * N)
            ach = nbytes / (secs / max(launches, 1))
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                               "frac": ach / HBM_PEAK, "traffic": None}
        # the logic kernels' roofline stress and the host baseline belong to the single-GPU run; with more ranks the others would
        # only wait for rank 0 at the final barrier
        out["kernels"] = stress_kernels(L, device, args.stress_preds, 100) if (args.stress_preds > 0 and world == 1) else []
        attach_traffic(out, args)
        sample = args.cpu_sample if args.cpu_sample is not None else (64 if args.objects > 64 else 256)     # about 10 s of host work
        out["cpu_baseline"] = None
        if sample > 0 and world == 1:
            out["cpu_baseline"], out["parity"] = cpu_baseline(model, paths, qs[:sample], res, sample)
        print(json.dumps(out))
    if dist:
        td.barrier()
        td.destroy_process_group()


def attach_traffic(out, args):
    """HBM traffic per launch from the committed rocprofv3 counter passes of THIS command (profiles/traffic.json, written by
    tools/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc passes, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950).  Left null when no profile of the same shape is on disk."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return
    with open(path) as f:
        t = json.load(f)

    def total(prefix, grid=None):
        for key, v in t.items():
            name, gx = key.rsplit("@", 1)
            if name.startswith(prefix) and (grid is None or int(gx) == grid) and "fetch_bytes_x2" in v and "write_bytes" in v:
                return v["fetch_bytes_x2"] + v["write_bytes"]
        return None

    if "pair_ll" in out["roofline"]["kernel"] and args.objects == 100 and args.batch == 256:
        out["roofline"]["traffic"] = total("pair_ll32s_kernel" if "32s" in out["roofline"]["kernel"] else "pair_ll32b_kernel") or total("pair_ll16")
        out["roofline"]["traffic_unit"] = "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/traffic.json)"
    P = args.stress_preds
    for k in out["kernels"]:
        pref = "relate_one_fwd_kernel" if k["kernel"].startswith("relate_one") else "relate_fwd_kernel" if k["kernel"].startswith("relate_fwd") \
            else "filter_fwd_kernel"
        k["traffic"] = total(pref, (P // 4) * 256 if "relate" in pref else P * 64) if P == 65536 else None


def stress_kernels(L, device, P, N):
    """HBM roofline of the logic kernels on P resident predicates of N objects (>= 2.6 GB of relation tiles)."""
    NS = (N + 3) // 4 * 4
    g = torch.Generator(device=device).manual_seed(1)
    u = torch.rand(P, NS, NS, device=device, generator=g)
    tile = torch.log(torch.where(torch.rand(P, NS, NS, device=device, generator=g) < 0.1, 0.5 + 0.5 * u, 0.05 * u).clamp_min(1e-5))
    del u
    tile.diagonal(dim1=1, dim2=2).fill_(-30.0)              # self-relations hold the absent value, as the oracle writes them
    prior = torch.log(torch.rand(P, NS, device=device, generator=g).clamp_min(1e-3)) * 0.3
    pq = torch.arange(P, dtype=torch.int32, device=device)
    n_obj = torch.full((P,), N, dtype=torch.int32, device=device)
    ones = torch.ones(P, device=device)
    res = []

    def timed(name, fn, nbytes, iters=20):
        """Average launch duration from ONE HIP-event pair around `iters` back-to-back launches on the launch stream
        (bracketing every single launch with its own event pair adds marker/fence latency to sub-millisecond kernels)."""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) * 1e-3 / iters
        ach = nbytes / t
        return {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                "predicates": P, "objects": N, "us_per_launch": t * 1e6}

    r = timed("dfol_relate_one_fwd_f32", lambda: L.relate_one_fwd(prior, prior, tile, pq, n_obj, ones), P * (4 * N * N + 12 * N))
    res.append(dict(r, kernel="relate_one_fwd (fused single-posterior Relate, the interpreter's path)", bytes_per_predicate=4 * N * N + 12 * N))
    r = timed("dfol_relate_fwd_f32", lambda: L.relate_fwd(prior, prior, tile, pq, n_obj, ones, ones, need_s=False, diag_absent=True), P * (4 * N * N + 12 * N))
    res.append(dict(r, kernel="relate_fwd (generic cell, one posterior wanted)", bytes_per_predicate=4 * N * N + 12 * N))
    r = timed("dfol_relate_fwd_f32", lambda: L.relate_fwd(prior, prior, tile, pq, n_obj, ones, ones, diag_absent=True), P * (4 * N * N + 16 * N))
    res.append(dict(r, kernel="relate_fwd (both posteriors, RelateBatch API)", bytes_per_predicate=4 * N * N + 16 * N))
    ll = tile[:, 0, :].contiguous()
    del tile
    r = timed("dfol_filter_fwd_f32", lambda: L.filter_fwd(prior, ll, pq, n_obj), P * 12 * N)
    res.append(dict(r, kernel="filter_fwd", bytes_per_predicate=12 * N))
    return res


def cpu_baseline(model, paths, questions, gpu_result, sample):
    """The CPU oracle (numpy port of the reference's flat-layout algorithm, full-size tables) on a bounded sample of
    the same workload, timed on this host; its log-probabilities double as an in-run parity check."""
    from oracle import dfol_oracle as orc
    ont = orc.Ontology(paths["attribute_file"], paths["class_file"], paths["vocabulary_file"], paths["relation_file"])
    weights = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if k.startswith("_featurizer.") or k.startswith("_oracle.")}
    scenes = [q["scene"] for q in questions]
    best = None
    for pb_size in (1, 4):                                  # the reference's cost is super-linear in the ProgramBatch size
        chunks = max(1, -(-sample // pb_size))
        t0 = time.perf_counter()
        r = orc.run_questions(ont, questions, scenes, np.float32, split=chunks, weights=weights)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, r, pb_size)
    dt, r, split = best
    lp_gpu = gpu_result["log_probability"][:sample].detach().cpu().numpy()
    lp_cpu = r["log_probability"]
    agree = sum(1 for a, b in zip(gpu_result["answer"][:sample], r["answer"]) if a == b)
    try:                                                    # threads numpy's BLAS actually runs the MLP layers on
        from threadpoolctl import threadpool_info
        cores = max([int(i.get("num_threads", 1)) for i in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count()
    base = {"value": sample / dt, "unit": "questions/s", "cores": cores, "kind": "port",
            "sample": "%d questions of the same workload (N=%d), numpy fp32 oracle incl. full [pairs,2335] tables, "
                      "ProgramBatch size %d, %.1f s" % (sample, questions[0]["scene"]["n"], split, dt)}
    parity = {"max_abs_dp": float(np.abs(np.exp(lp_gpu) - np.exp(lp_cpu)).max()), "max_abs_dlp": float(np.abs(lp_gpu - lp_cpu).max()),
              "answers_agree": "%d/%d" % (agree, sample)}
    return base, parity


if __name__ == "__main__":
    main()
