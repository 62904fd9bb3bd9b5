#!/usr/bin/env python3
"""Headline benchmark: questions/s of the ∇-FOL interpreter forward on synthetic GQA-style scenes.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--objects 100] [--batch 256]

One step = one pass of the hot path (build_scene: featurizer + oracle MLPs -> likelihoods; then the
program select -> filter -> relate -> exist for every question) over one batch of `--batch` questions
whose object features and lowered program are already resident in HBM.  Workload = BASELINE.json
configs[1] (3-hop Filter->Relate->Exist programs, fp32, batch=256) on N-object scenes; N defaults to
100, the size BASELINE.json's metric is quoted on (configs[1] itself says 36: pass --objects 36).

--gpus N > 1: one process per GPU.  Either the driver starts the ranks (torch.distributed.run sets
RANK / LOCAL_RANK / WORLD_SIZE) or, when those are absent, this script starts them itself - as child
processes, BEFORE anything here touches the GPU - and rank 0's JSON line is the output.  Questions are
independent, so each rank runs its own shard with no data-path collective (weak scaling) and the ranks
only meet at the timing barriers.  --mode train times the train step instead (forward + loss + backward
+ ONE RCCL all-reduce of the flat gradient bucket + clip + Adam; reference trainer.py:429-442), the
only place a collective sits on the path.

Prints ONE JSON line on rank 0.  `roofline` is the dominant kernel of the step; `kernels` adds the
HBM roofline of the Relate/Filter logic kernels on >= 65536 resident predicates;
`cpu_baseline` is the CPU oracle (a port of the reference's algorithm) timed on this host.
"""

import argparse
import json
import contextlib
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)            # (a 20-step timed region at N = 100 is 47 ms: too short for the driver to sample)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=("infer", "train"), default="infer")
    ap.add_argument("--workload", choices=("north_star", "c1", "c3", "c4"), default="north_star",
                    help="north_star: configs[1]'s program on 100-object scenes (the size the metric is quoted on); c1: configs[1] verbatim "
                         "(36 objects); c3: configs[2]'s shape - ragged scenes of 10..100 objects, several questions per image sharing one scene; "
                         "c4: configs[4] (256 objects, 8-hop open programs, bf16 relation tiles)")
    ap.add_argument("--data", default=None, help="--workload c3 on FILES in the reference's formats instead of synthetic scenes: a directory with "
                    "metadata/{attribute,class,relation,vocab}.json [+ glove.txt], programs/*.h5 (GQAH5Encoder's program bytecode, one file per terminal "
                    "operator: gqa_preprocess.py:51-94) and objects/<prefix>_<i>.h5 + objects/<prefix>_info.json (feature chunks: "
                    "batch_gqa_boxfeatures_pipeline.py:29-55); e.g. GQA testdev-balanced when it is on the box")
    ap.add_argument("--questions-per-image", type=int, default=None,
                    help="questions that share one image (and, with sharing, one featurizer pass and one set of relation tiles); default 8 for c3, else 1")
    ap.add_argument("--share-scenes", type=int, default=1, help="0: collate one copy of the scene per question even when questions share an image (the reference's layout)")
    ap.add_argument("--objects", type=int, default=None)
    ap.add_argument("--ragged", type=int, default=0, help="train mode: object counts ~ U{ragged..objects}")
    ap.add_argument("--calibrator", type=int, default=0, help="train mode: 1 = calibrator phases (cur6-7): oracle frozen, the attention networks train")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--overlap-allreduce", type=int, default=0,
                    help="train mode, N > 1: 0 (default) = ONE all-reduce of the flat bucket after the backward; 1 = EXPERIMENTAL: three ranges of the bucket, each "
                         "all-reduced as soon as the backward has produced it (exercised over gloo only so far; 9.2 MB over xGMI is ~0.1 ms of a 10 ms step)")
    ap.add_argument("--mlp-math", choices=("fp32", "bf16"), default="fp32",
                    help="train mode: bf16 = configs[3]'s 'bf16 fwd / fp32 logic' (config key mlp_math): the large dense products on bf16-rounded "
                         "operands, fp32 accumulation; reported as dtype bf16, never the default")
    ap.add_argument("--cpu-sample", type=int, default=None, help="questions in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--parity-all", type=int, default=1, help="1: check questions beyond the timed CPU sample against the oracle too (up to --parity-questions)")
    ap.add_argument("--parity-questions", type=int, default=128, help="questions of the timed batch the in-bench parity leg checks (the whole batch of 256 is "
                    "checked by tests/test_interpreter_gpu.py::test_north_star_batch_parity_all_questions; 0 = all)")
    ap.add_argument("--stress-preds", type=int, default=65536)
    ap.add_argument("--fresh-batches", type=int, default=56, help="batches of the `value_fresh_programs` leg: every step a different ProgramBatch through "
                    "collate -> lower -> eager launches (0 = skip; north_star / c1 workloads, one process)")
    ap.add_argument("--hops", choices=["aligned", "ragged"], default="aligned", help="ragged: every question its own program length (select -> 1..3 filter / relate "
                    "hops -> exist; collate pads with no-op tokens) instead of the headline's select -> filter -> relate -> exist for all - what files of GQA "
                    "programs look like; reported as config.hops")
    ap.add_argument("--fresh-streams", type=int, default=2, help="`value_fresh_programs`: HIP streams the unseen batches alternate on (2: the end of one batch "
                    "overlaps with the start of the next on the device, like the replay lanes of `value`; 1: one stream)")
    ap.add_argument("--fresh-depth", type=int, default=2, help="batches queued on the device before the oldest one's answers are waited for (`value_fresh_programs`)")
    ap.add_argument("--fresh-workers", type=int, default=6, help="collate worker PROCESSES of the `value_fresh_programs` leg (the reference's DataLoader "
                    "workers, data_pipeline.py:893-898): they collate and lower, the launching process unpickles, uploads and launches; 0 = collate on "
                    "the launching thread under the batch before")
    ap.add_argument("--streamed", type=int, default=1, help="1: also measure the rate with object features streamed from pinned host memory")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the step as a captured HIP graph (interpreter.GraphedForward); 0: eager launches")
    ap.add_argument("--pipeline", type=int, default=1, help="graph replays two in flight (replay i + 1 is launched before replay i's answers are read back and "
                    "decoded; every step's answers are decoded inside the timed region); 0: one replay at a time")
    ap.add_argument("--setup-replays", type=int, default=64, help="replays run as part of setting the graph lanes up, before the --warmup steps (the first few "
                    "dozen replays after the captures run 5 - 10 %% below the steady rate - clocks and caches; --steps 20 would otherwise time mostly those); "
                    "reported as config.setup_replays; 0: none")
    ap.add_argument("--lanes", type=int, default=2, help="with --pipeline: captured forwards replayed round-robin on as many HIP streams, each over its own copy of "
                    "the batch (interpreter.ReplayLanes: the small logic launches at the end of a batch overlap with the featurizer / pair kernel of the "
                    "next); 1: two replays of ONE graph in flight on one stream")
    ap.add_argument("--graph-collective", type=int, default=0, help="train mode over RCCL: 1 = ONE step graph with the all-reduce captured inside (opt-in: a "
                    "failure of that path - capture error or an abort from the process group's watchdog - ends THIS process with a non-zero exit, "
                    "there is no in-process fallback); 0 (default) = two graphs with the all-reduce issued eagerly between the replays")
    ap.add_argument("--calibrated-leg", type=int, default=1, help="1: also time the CALIBRATED forward (activate_attention_transfer: True, the reference's "
                    "default config) - graph replay of a resident batch and a stream of unseen batches on the native executor / the Python loop "
                    "(north_star workload, one GPU): `calibrated` and config.legs.calibrated_*")
    ap.add_argument("--sustain", type=float, default=5.0, help="seconds of back-to-back steps AFTER the --steps region for `value_sustained` (0 = skip)")
    ap.add_argument("--roofline-reps", type=int, default=20, help="eager steps of the roofline leg (per-launch HIP events); they are the LAST launches "
                                                                  "of the dominant kernel in the process, so a rocprofv3 trace of the same command can average the same launches")
    ap.add_argument("--parity-fp64", type=int, default=1, help="1: the parity leg also runs the oracle in float64 (the tolerance policy's yardstick)")
    ap.add_argument("--cpu-budget", type=float, default=60.0, help="seconds of host time the cpu_baseline + parity legs may take in total: the number of "
                                                                    "questions checked beyond the timed sample shrinks to fit (the default run must finish within minutes)")
    args = ap.parse_args(argv)
    if args.objects is None:
        args.objects = {"north_star": 100, "c1": 36, "c3": 100, "c4": 256}[args.workload]
    if args.questions_per_image is None:
        args.questions_per_image = 8 if args.workload == "c3" else 1
    if args.workload == "c3" and not args.ragged:
        args.ragged = 10
    return args


def launch_ranks(args, argv):
    """`--gpus N` without a launcher: start N ranks as CHILD processes through torch.distributed.run and pass rank 0's line through.
    Nothing in this process has touched the GPU yet (torch.cuda.device_count() does not initialise it on this image)."""
    import socket
    import subprocess
    share = os.environ.get("DFOL_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not share:
        sys.exit("bench.py: --gpus %d but only %d GPU(s) visible (DFOL_BENCH_SHARE_GPU=1 puts every rank on cuda:0 over gloo, for debugging)"
                 % (args.gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.call(cmd, env=env))


def build_batch(args, rank, ontology, names, device, world=1):
    """This rank's questions and ProgramBatches.  Fixed-size scenes (north_star, c1, c4): `--batch` questions per rank, ids rank * batch ...
    (weak scaling; scenes are keyed by question id, so sharding never changes anyone's inputs).  Ragged scenes (`--workload c3`, world > 1):
    the GLOBAL list of world * batch questions is cut into contiguous shards of whole images balanced by the sum of n^2 over a shard's images
    (parallel.shard_bounds: the oracle's cost is the object pairs; data_parallel.py:59-80 cuts by question count) - `args._shard` then holds
    every rank's question count and cost and the imbalance."""
    import dfol_vqa_amd as D
    from dfol_vqa_amd import parallel
    from dfol_vqa_amd import synthetic as syn

    class Collater(D.ProgramCollaterBase):
        def __init__(self):
            super(Collater, self).__init__("select", "relate", "filter", 1, ontology=ontology,
                                           share_scenes=getattr(args, "questions_per_image", 1) > 1 and bool(getattr(args, "share_scenes", 1)))

        def collate_object_features(self, questions):
            feats = torch.cat([torch.from_numpy(q["scene"]["X"]) for q in questions], 0)
            bi = torch.cat([torch.full((q["scene"]["n"],), i, dtype=torch.int64) for i, q in enumerate(questions)])
            return feats, bi

        def collate_meta_data(self, questions):
            return {"index": {}, "embedding": torch.zeros(1, 1)}

    nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
    rng = np.random.RandomState(1000 + rank)
    G = max(1, getattr(args, "questions_per_image", 1))
    qs, scene_of = [], {}
    ids = range(rank * args.batch, (rank + 1) * args.batch)
    n_of = None
    if world > 1 and getattr(args, "ragged", 0) and getattr(args, "workload", "north_star") == "c3":
        total = world * args.batch
        n_img = -(-total // G)
        grng = np.random.RandomState(1000)                  # every rank draws the same global scene sizes
        n_of = [int(grng.randint(args.ragged, args.objects + 1)) for _ in range(n_img)]
        bounds = parallel.shard_bounds([float(n) ** 2 for n in n_of], world)
        cost = [sum(float(n) ** 2 for n in n_of[a:b]) for a, b in bounds]
        count = [min(total, b * G) - min(total, a * G) for a, b in bounds]
        args._shard = {"by": "contiguous shards of whole images, balanced by sum n^2 (parallel.shard_bounds)", "questions_per_rank": count,
                       "cost_per_rank": cost, "imbalance_max_over_mean": max(cost) / (sum(cost) / world)}
        a, b = bounds[rank]
        ids = range(min(total, a * G), min(total, b * G))
    for qid in ids:                                        # scenes are keyed by question / image id: sharding never changes inputs
        img = qid // G                                     # G consecutive questions look at the same image
        if img not in scene_of:
            n = args.objects if not getattr(args, "ragged", 0) else (n_of[img] if n_of is not None else int(rng.randint(args.ragged, args.objects + 1)))
            scene_of[img] = syn.feature_scene(img if G > 1 else qid, n, 2048)
        if getattr(args, "workload", "north_star") == "c4":
            br, last = syn.open_program(qid, nouns, attrs, rels, names["categories"], hops=4)
            q = syn.question(qid, br, last, attrs[qid % len(attrs)], scene_of[img])
        elif getattr(args, "hops", "aligned") == "ragged":
            br, last = syn.ragged_hop_program(qid, nouns, attrs, rels)
            q = syn.question(qid, br, last, "yes", scene_of[img])
        else:
            br, last = syn.three_hop_program(qid, nouns, attrs, rels)
            q = syn.question(qid, br, last, "yes", scene_of[img])
        if G > 1:
            q["image_id"] = "img%d" % img
        else:
            q["image_id"] = "q%d" % qid                    # (synthetic.question folds ids modulo 64: every question its own image here)
        qs.append(q)
    pbs = Collater().collate(qs)
    for pb in pbs:
        pb.create_sparse_tensors()
    return qs, [pb.to_cuda(device) for pb in pbs]


def init_weights(model):
    """Random weights of the reference architecture; the embedding rows get GloVe-like magnitudes so that the
    concept probabilities are sparse instead of saturated.  Seeded: every rank builds the same replica (and
    parallel.broadcast_parameters makes that a guarantee instead of a convention)."""
    torch.manual_seed(0)
    lin = model._oracle._embedding_network.linear
    with torch.no_grad():
        lin.weight.normal_(0.0, 0.1)
        lin.bias.fill_(-2.0)


WORKLOADS = {
    "north_star": "BASELINE configs[1]'s program: select->filter->relate->exist (3-hop), fp32, %d questions/GPU/step, %d-object synthetic "
                  "scenes (the size BASELINE.json's metric is quoted on), full-size oracle (2048->512, 516/1036->256->300->2335)",
    "c1": "BASELINE configs[1] verbatim: select->filter->relate->exist (3-hop), fp32, %d questions/GPU/step, %d-object synthetic scenes, "
          "full-size oracle (2048->512, 516/1036->256->300->2335)",
    "c3": "BASELINE configs[2]'s shape with synthetic features: select->filter->relate->exist, fp32, %d questions/GPU/step on ragged scenes of "
          "10..%d objects, QPI questions per image, full-size oracle (2048->512, 516/1036->256->300->2335)",
    "c4": "BASELINE configs[4]: select->(filter->relate)x4->query_attr (8-hop open programs), %d questions/GPU/step, %d-object synthetic "
          "scenes, bf16 relation tiles (fp32 logic arithmetic), full-size oracle",
}


_JSON_OUT = None


def emit(line):
    out = _JSON_OUT if _JSON_OUT is not None else sys.stdout
    out.write(line + "\n")
    out.flush()


def setup(args):
    """Rank / device / process group; fails loudly when the launch does not match --gpus."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # DFOL_BENCH_SHARE_GPU=1 is a debugging aid for boxes with one GPU: every rank uses cuda:0 and the ranks meet over gloo
    share = os.environ.get("DFOL_BENCH_SHARE_GPU") == "1"
    local = 0 if share else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    td = None
    # DFOL_BENCH_FORCE_PG=1: a process group even at world size 1, so that a ONE-GPU box runs every collective of the N > 1 path through RCCL
    # itself (communicator setup, barrier, all-reduce of the gradient bucket incl. the overlap hooks, all-gather of the rank report)
    force = os.environ.get("DFOL_BENCH_FORCE_PG") == "1"
    if world > 1 or force:
        # the JSON line must be the only thing on stdout: native libraries print there too (RCCL: "Librccl path : ..." at exit, from C stdio),
        # so from here on file descriptor 1 is stderr for everyone else and the line goes to the saved stdout (emit())
        global _JSON_OUT
        if _JSON_OUT is None:
            sys.stdout.flush()
            _JSON_OUT = os.fdopen(os.dup(1), "w")
            os.dup2(2, 1)
        import torch.distributed as td
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if share:
            td.init_process_group("gloo", timeout=__import__("datetime").timedelta(seconds=600))
        else:
            td.init_process_group("nccl", device_id=device)      # RCCL over xGMI
        assert td.get_world_size() == args.gpus
    return rank, world, device, td, share


def host_threads():
    """What the host side of a rank runs on (N Python hosts share one CPU: where data-parallel inference stops scaling first)."""
    try:
        aff = sorted(os.sched_getaffinity(0))
    except Exception:
        aff = []
    return {"OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"), "torch_num_threads": torch.get_num_threads(), "cpu_affinity_count": len(aff),
            "cpu_affinity": "%d-%d" % (aff[0], aff[-1]) if aff and aff[-1] - aff[0] + 1 == len(aff) else ",".join(map(str, aff[:64])),
            "host_cpus": os.cpu_count()}


def per_rank_rates(td, world, rank, leg):
    """Every rank's own rate of a fresh-programs / end-to-end leg (the whole-job value is the slowest rank's): [{rank, questions_per_s,
    ms_per_batch, host_collate_ms_per_batch}]."""
    mine = {"rank": rank, "questions_per_s": leg["questions_per_s"], "ms_per_batch": leg["ms_per_batch"],
            "host_collate_ms_per_batch": leg.get("host_collate_ms_per_batch")}
    if td is None:
        return [mine]
    parts = [None] * world
    td.all_gather_object(parts, mine)
    return parts


def rank_report(td, share, device, rank, world, elapsed, steps):
    """Proof of what ran where, for the N > 1 lines: every rank's device (index, name, PCI bus id) and its own elapsed time, all-gathered."""
    prop = torch.cuda.get_device_properties(device)
    mine = {"rank": rank, "device_index": device.index, "name": prop.name, "pci_bus_id": getattr(prop, "pci_bus_id", None),
            "uuid": str(getattr(prop, "uuid", "")), "ms_per_step": elapsed / max(1, steps) * 1e3, "pid": os.getpid(), "host": host_threads()}
    if td is None:
        return {"ranks_seen": [mine], "distinct_devices": 1, "rank_ms_per_step_min": mine["ms_per_step"], "rank_ms_per_step_max": mine["ms_per_step"],
                "backend": None}
    parts = [None] * world
    td.all_gather_object(parts, mine)
    ms = [p["ms_per_step"] for p in parts]
    rep = {"ranks_seen": parts, "distinct_devices": len({(p["device_index"], p["uuid"], p["pci_bus_id"]) for p in parts}),
           "rank_ms_per_step_min": min(ms), "rank_ms_per_step_max": max(ms), "backend": td.get_backend()}
    # an N-GPU line must come from N GPUs: with RCCL every rank has its own device (the gloo / shared-GPU debugging mode says so in `backend`)
    if rep["backend"] == "nccl" and not share and rep["distinct_devices"] != world:
        raise SystemExit("bench.py: %d ranks over RCCL but %d distinct devices in the rank report: %r" % (world, rep["distinct_devices"], parts))
    return rep


def build_model(args, device, train=False):
    from dfol_vqa_amd import _lib as L
    from dfol_vqa_amd import experiment
    from dfol_vqa_amd import synthetic as syn
    L.load()
    tmp = tempfile.mkdtemp(prefix="dfol_bench_")
    paths, names = syn.write_synthetic_ontology(tmp)
    if not train:
        cfg = syn.reference_config(paths)
    elif args.calibrator:
        cfg = syn.reference_config(paths, dropout=0.0, activate_attention_transfer=True)
    else:                                                        # the oracle-training phases (cur1-5) of the curriculum
        cfg = syn.reference_config(paths, dropout=0.0, freeze_featurizer=False, freeze_attribute_network=False, freeze_relation_network=False,
                                   freeze_embedding_network=False)
    if args.workload == "c4":
        cfg["relation_tile_dtype"] = "bf16"
    if train and getattr(args, "mlp_math", "fp32") == "bf16":
        cfg["mlp_math"] = "bf16"
    ontology = experiment.build_ontology(cfg)
    model = experiment.build_model(cfg, ontology)
    init_weights(model)
    model = model.to(device)
    return (model.train() if train else model.eval()), ontology, paths, names


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, argv)                                 # does not return
    rank, world, device, td, share = setup(args)
    if args.mode == "train":
        return train_main(args, rank, world, device, td, share)
    if args.data:
        return data_main(args, rank, world, device, td, share)

    from dfol_vqa_amd import _lib as L
    model, ontology, paths, names = build_model(args, device)
    if td is not None:
        from dfol_vqa_amd import parallel
        parallel.broadcast_parameters(model, 0)
    qs, pbs = build_batch(args, rank, ontology, names, device, world)
    args._scene_ns = [int(n) for pb in pbs for n in pb._object_nums]
    if args.questions_per_image > 1 and args.share_scenes:
        seen = {}
        for q in qs:
            rel = [o for o in q["program"]["branches"][0] if o["operator"] == "relate"][0]["arguments"]
            seen.setdefault(q["image_id"], set()).add((rel[0], bool(rel[1])))
        args._tiles_per_scene = float(np.mean([len(v) for v in seen.values()]))

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
        if td is not None:
            td.barrier()
        torch.cuda.synchronize()

    pipelined = bool(graphed and args.pipeline)
    lanes = None
    if pipelined and args.lanes > 1:
        from dfol_vqa_amd.interpreter import ReplayLanes
        try:
            # every further lane replays its OWN upload of the same batch (own feature tensors, own intermediates, own status word)
            lanes = ReplayLanes(model, [pbs] + [build_batch(args, rank, ontology, names, device, world)[1] for _ in range(args.lanes - 1)])
        except Exception as e:
            sys.stderr.write("replay lanes unavailable (%s); two replays of one graph in flight\n" % e)
            torch.cuda.synchronize()

    def run_steps(n, serial=False):
        """n steps, the answers of every one of them read back and decoded before this returns.  Graph replays are software-pipelined two deep
        (replay i + 1 is launched, then replay i's answers are decoded from its own pinned copies while it runs) - the loop shape of the
        fresh-programs leg; serial=True: one replay at a time, each waited for."""
        res = None
        if not pipelined or serial:
            for _ in range(n):
                res = step()
            return res
        src = lanes if lanes is not None else step
        pending = []
        for _ in range(n):
            pending.append(src.submit())
            if len(pending) > 1:                                 # (one ticket per lane / two per graph outstanding at most)
                res = src.collect(pending.pop(0))
        for ticket in pending:
            res = src.collect(ticket)
        return res

    with torch.no_grad():
        if pipelined and args.setup_replays > 0:
            run_steps(args.setup_replays)
        res = run_steps(args.warmup)
        barrier()
        t0 = time.perf_counter()
        res = run_steps(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        serial_elapsed = None
        if pipelined:                                            # the same steps one at a time (what `value` was before round 6's pipelining), for the record
            barrier()
            t0 = time.perf_counter()
            run_steps(args.steps, serial=True)
            barrier()
            serial_elapsed = time.perf_counter() - t0
    ranks = rank_report(td, share, device, rank, world, elapsed, args.steps)
    if td is not None:
        t = torch.tensor([elapsed], device="cpu" if share else device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
    total_q = args.batch * world * args.steps
    metric = "questions/sec (GQA programs, N=%d objects)" % args.objects
    if args.workload == "c3":
        metric = "questions/sec (GQA programs, ragged scenes of %d..%d objects, %d questions per image%s)" % (
            args.ragged, args.objects, args.questions_per_image, "" if args.share_scenes else ", scenes NOT shared")
    out = {"argv": list(argv), "metric": metric, "value": total_q / elapsed, "unit": "questions/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": (WORKLOADS[args.workload] % (args.batch, args.objects)).replace("QPI", str(args.questions_per_image)).replace(
               " (the size BASELINE.json's metric is quoted on)", " (the size BASELINE.json's metric is quoted on)" if args.objects == 100 else ""),
                      "workload_id": args.workload, "batch_per_gpu": args.batch, "questions_per_image": args.questions_per_image,
                      "scenes_shared": bool(args.questions_per_image > 1 and args.share_scenes), "distinct_scenes_per_gpu": len(args._scene_ns),
                      "objects_total_per_gpu": int(sum(args._scene_ns)),
                      "global_batch": args.batch * world, "objects_per_scene": args.objects, "parallelism": "dp%d" % world,
                      "launch": ("hip graph replay, %d lanes: consecutive steps replay captured forwards of the same batch on %d HIP streams and overlap on the "
                                 "device; the answers of every step are read back and decoded inside the timed region (`one_replay_at_a_time` in legs: the "
                                 "same steps serialised)" % (len(lanes), len(lanes)) if lanes is not None
                                 else "hip graph replay, two replays in flight (answers of every step decoded inside the timed region)" if pipelined
                                 else "hip graph replay" if graphed else "eager"),
                      "lanes": len(lanes) if lanes is not None else 1, "setup_replays": int(args.setup_replays) if pipelined else 0,
                      "contraction_math": {"f32": "fp32 matrix pipe",
                                           "bf16x3": "fp32 results from the bf16 matrix pipe: exact 3-way bf16 operand split, 6 of 9 piece products, fp32 accumulate (DESIGN 3.3)",
                                           "f16x2": "fp32 results from the fp16 matrix pipe: 2 fp16 pieces per operand (weight rows scaled by powers of two), 3 of 4 piece "
                                                    "products, fp32 accumulate (DESIGN 3.4)"}[os.environ.get("DFOL_PAIR_MATH", "f16x2")]}}

    out["ranks"] = ranks
    if serial_elapsed is not None:
        if td is not None:
            t = torch.tensor([serial_elapsed], device="cpu" if share else device, dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            serial_elapsed = float(t.item())
        out["value_one_replay_at_a_time"] = total_q / serial_elapsed
    if getattr(args, "_shard", None):
        out["shard"] = args._shard
    if args.sustain > 0:
        # the headline rate again over >= --sustain seconds of back-to-back steps (the --steps region at N = 100 lasts 47 ms when the
        # driver passes --steps 20: too short for an outside sampler to see)
        # (the step count comes from the max-over-ranks time of the --steps region, so every rank runs the same number of steps)
        n_sus = max(args.steps, int(np.ceil(args.sustain / (elapsed / args.steps) * 1.05)))
        with torch.no_grad():
            barrier()
            t0 = time.perf_counter()
            res = run_steps(n_sus)
            barrier()
            sus = time.perf_counter() - t0
        if td is not None:
            t = torch.tensor([sus], device="cpu" if share else device, dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            sus = float(t.item())
        out["value_sustained"] = args.batch * world * n_sus / sus
        out["sustained"] = {"steps": n_sus, "seconds": sus, "ms_per_step": sus / n_sus * 1e3}
    if args.streamed:
        # every rank streams its own batches (the ranks share the host's PCIe root complexes, so this is measured with all of them live)
        sv = streamed_rate(args, step if graphed else eager, pbs, td, share, device)
        out["value_streamed"] = sv["questions_per_s"] * world
        out["streamed"] = sv
    if args.fresh_batches > 0 and args.workload in ("north_star", "c1"):
        # every rank runs its own stream of unseen batches (max over ranks of the elapsed time: the whole-job rate)
        fp = fresh_programs_rate(args, model, ontology, names, paths, device, rank, n_batches=args.fresh_batches)
        fp["per_rank"] = per_rank_rates(td, world, rank, fp)
        if td is not None:
            t = torch.tensor([fp["ms_per_batch"]], device="cpu" if share else device, dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            fp["ms_per_batch"] = float(t.item())
            fp["questions_per_s"] = args.batch / (fp["ms_per_batch"] * 1e-3)
        out["value_fresh_programs"] = fp["questions_per_s"] * world
        fp["vs_value"] = "%.2f x the replayed-batch rate `value`" % (out["value_fresh_programs"] / out["value"])
        out["fresh_programs"] = fp
        # `value_end_to_end`: new programs AND new features every batch - the reference's test() loop as it is (trainer.py:685-720)
        ee = fresh_programs_rate(args, model, ontology, names, paths, device, rank, n_batches=args.fresh_batches, stream_features=True)
        ee["per_rank"] = per_rank_rates(td, world, rank, ee)
        if td is not None:
            t = torch.tensor([ee["ms_per_batch"]], device="cpu" if share else device, dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            ee["ms_per_batch"] = float(t.item())
            ee["questions_per_s"] = args.batch / (ee["ms_per_batch"] * 1e-3)
        out["value_end_to_end"] = ee["questions_per_s"] * world
        ee["vs_value"] = "%.2f x the replayed-batch rate `value`" % (out["value_end_to_end"] / out["value"])
        out["end_to_end"] = ee
    if args.calibrated_leg and world == 1 and args.workload == "north_star" and args.fresh_batches > 0:
        out["calibrated"] = calibrated_leg(args, device)          # (before the roofline leg: that one's launches must be the process's last)
    if rank == 0:
        # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream ------
        # --roofline-reps EAGER steps, one event pair per launch.  They come after every other leg that launches these kernels, so
        # they are the last launches of the dominant kernel in the process: tools/summarize_profile.py averages exactly those launches
        # of a rocprofv3 trace of this command (profiles/roofline_rocprof.json, attached below as `roofline.rocprofv3`).
        names_timed = list(L.SIGNATURES)
        reps = max(1, args.roofline_reps)
        with torch.no_grad():
            for _ in range(5):
                eager()
            torch.cuda.synchronize()
            L.enable_kernel_timing(names_timed)
            for _ in range(reps):
                eager()
            torch.cuda.synchronize()
            timing = L.disable_kernel_timing()
        per_step = {k: (n / float(reps), t / float(reps)) for k, (n, t) in timing.items() if n}
        dom = max(per_step, key=lambda k: per_step[k][1])
        out["kernel_ms_per_step"] = {k: round(v[1] * 1e3, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1][1])}
        out["kernel_ms_per_step_note"] = ("isolated timings: HIP events around every launch of %d eager steps; the timed region replays the same "
                                          "launches as one captured graph, so their sum need not equal ms_per_step" % reps)
        out["roofline"] = dominant_roofline(args, model, dom, per_step)
        out["roofline"]["reps"] = reps
        # the logic kernels' roofline stress and the host baseline belong to the single-GPU run; with more ranks the others would
        # only wait for rank 0 at the final barrier
        out["kernels"] = stress_kernels(L, device, args.stress_preds, 100) if (args.stress_preds > 0 and world == 1) else []
        attach_traffic(out, args)
        attach_rocprof(out, args)
        c4 = args.workload == "c4"
        sample = args.cpu_sample if args.cpu_sample is not None else (2 if c4 else 64 if args.objects > 64 else 256)     # about 10 s of host work
        sample = min(sample, args.batch)
        out["cpu_baseline"] = None
        if sample > 0 and world == 1:
            # configs[4] (256 objects, 26 options per question): two questions through the oracle are the parity sample; the relation
            # tiles are bf16 there, so the probabilities differ from the fp32 reference by the rounding of the stored likelihoods
            out["cpu_baseline"], out["parity"] = cpu_baseline(model, paths, qs, res, sample, args.parity_all and not c4, fp64=bool(args.parity_fp64),
                                                              budget=args.cpu_budget, max_questions=args.parity_questions)
            if c4:
                # bf16 STORAGE of the relation tiles (configs[4]): the fp32 policy's 1e-4 does not apply; the bound that does is the one golden g20's test
                # derives (tests/test_interpreter_gpu.py::test_g20_configs4_open_programs_against_the_reference): a stored log-likelihood l becomes
                # l (1 + d), |d| <= 2^-9, and a relate hop moves its log-domain aggregate by at most 2^-9 max|l|: after the program's four relates
                # |lp - lp64| <= 4 x 2^-9 x L + 2 x the oracle's own fp32-vs-fp64 noise + 1e-4, with L = 12 >= the largest |log-likelihood| of these
                # weights (LogSigmoid of logits -2 +- a few tenths x sqrt(300))
                par = out["parity"]
                own = ((par.get("oracle_fp32_own_noise") or {}).get("max_abs_dlp_vs_fp64_where_lp_ge_-5")) or 0.0
                got = par.get("max_abs_dlp_vs_fp64_where_lp_ge_-5")
                bound = 4 * 2.0 ** -9 * 12.0 + 2 * own + 1e-4
                agree = par.get("answers_agree", "0/1").split("/")
                par["note"] = "bf16 relation tiles (configs[4]); against the REFERENCE itself: golden g20 under -m gpu"
                par["policy"] = {"applies": True, "rule": "|lp - lp64| <= 4 hops x 2^-9 x L (= 12) + 2 x the oracle's own fp32 noise + 1e-4 where lp64 >= -5; answers equal",
                                 "bound": bound, "max_abs_dlp_vs_fp64_where_lp_ge_-5": got, "pass": bool(got is None or got <= bound) and agree[0] == agree[1]}
                out["dtype"] = "f32 logic arithmetic on bf16 relation tiles"
        out["config"]["legs"] = legs_summary(out)
        emit(json.dumps(out))
        sys.stdout.flush()
    if td is not None:
        td.barrier()
        td.destroy_process_group()


def calibrated_leg(args, device, n_batches=24):
    """The forward with the attention calibrator ON - config/sample_config.yaml's default and what the reference's final (cur6-7) model runs:
    the LSTM walks over the aligned program + apply_modulations around every operator (batch_base_interpreter.py:87-140).  Same workload as
    `value` (256 questions x N objects, select -> filter -> relate -> exist, full-size model + LSTMCell(318 -> 50) x 2 + Linear(100 -> 4)):
    (a) one resident batch replayed as a HIP graph, (b) a stream of UNSEEN batches - plans lowered at collate time, two batches in flight -
    on the native executor (round 6: the calibration passes are part of the plan) and (c) the same stream on the Python operator loop."""
    import gc
    import dfol_vqa_amd as D
    from dfol_vqa_amd import _lib as L
    from dfol_vqa_amd import experiment, native_exec
    from dfol_vqa_amd import synthetic as syn
    from dfol_vqa_amd.interpreter import GraphedForward
    tmp = tempfile.mkdtemp(prefix="dfol_bench_calib_")
    paths, names = syn.write_synthetic_ontology(tmp)
    cfg = syn.reference_config(paths, activate_attention_transfer=True)
    ontology = experiment.build_ontology(cfg)
    model = experiment.build_model(cfg, ontology)
    init_weights(model)
    with torch.no_grad():                                        # (the reference zero-initialises the output layer: every modulation would be Sigmoid(bias))
        o = model._ops['filter']._filter._attention_output_network[0]
        o.weight.normal_(0.0, 0.5)
        o.bias.normal_(0.0, 0.5)
    model = model.to(device).eval()
    voc = list(ontology._vocabulary["idx_to_arg"])
    emb = torch.randn(len(voc), 300, generator=torch.Generator().manual_seed(3)) * 0.1
    index = {t: i for i, t in enumerate(voc)}
    N, B = args.objects, args.batch

    class Collater(D.ProgramCollaterBase):
        def __init__(self, spec=None):
            super(Collater, self).__init__("select", "relate", "filter", 1, ontology=ontology, native_spec=spec)

        def collate_object_features(self, qs):
            return torch.cat([torch.from_numpy(q["scene"]["X"]) for q in qs], 0), torch.cat([torch.full((q["scene"]["n"],), i, dtype=torch.int64) for i, q in enumerate(qs)])

        def collate_meta_data(self, qs):                        # (batch_gqa_boxfeatures_pipeline.py:88-92: token -> embedding row)
            return {"index": index, "embedding": emb}

    nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
    scenes = [syn.feature_scene(i, N, 2048) for i in range(B)]

    def batch(k, spec=None):
        qs = []
        for i in range(B):
            br, last = syn.three_hop_program(100000 * k + i, nouns, attrs, rels)
            qs.append(syn.question(100000 * k + i, br, last, "yes", scenes[i]))
        pb = Collater(spec).collate(qs)[0]
        pb.create_sparse_tensors()
        return pb

    base = batch(0).to_cuda(device)
    res = {"questions_per_batch": B, "objects": N}
    with torch.no_grad():
        eager = model([base], False)
        g = GraphedForward(model, [base])
        r = g()
        res["graph_equals_eager"] = bool(torch.equal(r["log_probability"], eager["log_probability"]) and r["answer"] == eager["answer"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        src = g
        if args.pipeline and args.lanes > 1:                     # (the loop shape of `value`: replay lanes on as many streams)
            from dfol_vqa_amd.interpreter import ReplayLanes
            src = ReplayLanes(model, [[base]] + [[batch(0).to_cuda(device)] for _ in range(args.lanes - 1)])
            for t in [src.submit() for _ in range(len(src))]:
                src.collect(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        pending, rp = [], None
        for _ in range(20):
            if not args.pipeline:
                g()
                continue
            pending.append(src.submit())
            if len(pending) > 1:
                rp = src.collect(pending.pop(0))
        for ticket in pending:
            rp = src.collect(ticket)
        if rp is not None:
            res["pipelined_replay_equals_eager"] = bool(torch.equal(rp["log_probability"], eager["log_probability"].cpu()) and rp["answer"] == eager["answer"])
        torch.cuda.synchronize()
        res["replay_ms_per_step"] = (time.perf_counter() - t0) / 20 * 1e3
        res["replay_lanes"] = len(src) if src is not g else 1
        spec = native_exec.model_spec(model, calibrate=True)
        saved = os.environ.get("DFOL_NATIVE")
        try:
            for route, nb in (("1", n_batches), ("0", max(8, n_batches // 3))):
                os.environ["DFOL_NATIVE"] = route
                fresh = []
                for k in range(1, nb + 5):
                    pb = batch(k, spec if route == "1" else None).to_cuda(device)
                    pb._object_features = base._object_features       # (resident features: the leg times programs, not uploads)
                    fresh.append([pb])
                L.PATH_COUNTS.clear()
                for pbk in fresh[:4]:
                    model(pbk, False)
                torch.cuda.synchronize()
                # (the prepared batches are tens of thousands of long-lived Python objects: a generation-2 collection in the middle of the loop
                # walks all of them - 2 ms pauses, seen as a 256-element list comprehension taking 2 ms; a serving process freezes what it keeps)
                gc.collect()
                gc.freeze()
                # (the executor's batches are self-contained - blob, arena, read-back buffer, range word: they alternate two streams like the
                # `value_fresh_programs` loop; the Python loop's cached side arrays are uploaded on the stream that first needs them: one stream)
                lanes_ = [torch.cuda.Stream(device=device) for _ in range(2)] if (route == "1" and getattr(args, "fresh_streams", 1) > 1) else None
                if lanes_:
                    for s_ in lanes_:
                        s_.wait_stream(torch.cuda.current_stream(device))
                t0 = time.perf_counter()
                pend = []
                for j, pbk in enumerate(fresh[4:]):
                    with (torch.cuda.stream(lanes_[j % 2]) if lanes_ else contextlib.nullcontext()):
                        pend.append(model.forward_async(pbk, False))
                    if len(pend) > 2:
                        pend.pop(0).result()
                for x in pend:
                    x.result()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / nb * 1e3
                gc.unfreeze()
                key = "fresh_native" if route == "1" else "fresh_python_loop"
                res[key + "_ms_per_batch"] = ms
                res[key + "_routes"] = {k: v for k, v in L.PATH_COUNTS.items() if k in ("native_program", "python_program")}
        finally:
            if saved is None:
                os.environ.pop("DFOL_NATIVE", None)
            else:
                os.environ["DFOL_NATIVE"] = saved
    res["replay_questions_per_s"] = B / (res["replay_ms_per_step"] * 1e-3)
    res["fresh_native_questions_per_s"] = B / (res["fresh_native_ms_per_batch"] * 1e-3)
    res["fresh_python_loop_questions_per_s"] = B / (res["fresh_python_loop_ms_per_batch"] * 1e-3)
    res["fresh_native_over_replay"] = res["fresh_native_questions_per_s"] / res["replay_questions_per_s"]
    return res


def legs_summary(out):
    """The honest loop rates, NUMBERS ONLY, inside a key the driver keeps (BENCH_rNN.json stores `config`, `roofline` and
    `cpu_baseline` in full but only the NAMES of other top-level keys): questions/s of every leg, its ratio to `value`, the executor's route
    counts of the fresh-programs loop and the logic kernels' HBM fractions (the north star's >= 0.5 bar on Relate / Filter)."""
    v = float(out["value"])
    legs = {"value": v}
    for key, short in (("value_one_replay_at_a_time", "one_replay_at_a_time"), ("value_sustained", "sustained"), ("value_streamed", "streamed"), ("value_fresh_programs", "fresh"), ("value_end_to_end", "end_to_end")):
        if out.get(key):
            legs[short] = float(out[key])
            legs[short + "_over_value"] = float(out[key]) / v
    fp = out.get("fresh_programs") or {}
    route = fp.get("executor") or {}
    if fp:
        legs["native_batches"] = int(route.get("native_program", 0))
        legs["python_batches"] = int(route.get("python_program", 0))
        if fp.get("device_ms_per_batch"):
            legs["fresh_device_ms_per_batch"] = float(fp["device_ms_per_batch"])
        legs["fresh_ms_per_batch"] = float(fp.get("ms_per_batch", 0.0))
    cal = out.get("calibrated") or {}
    if cal:
        legs["calibrated_replay"] = float(cal["replay_questions_per_s"])
        legs["calibrated_fresh"] = float(cal["fresh_native_questions_per_s"])
        legs["calibrated_fresh_over_replay"] = float(cal["fresh_native_over_replay"])
        legs["calibrated_fresh_python_loop"] = float(cal["fresh_python_loop_questions_per_s"])
        legs["calibrated_native_batches"] = int((cal.get("fresh_native_routes") or {}).get("native_program", 0))
    ee = out.get("end_to_end") or {}
    if ee.get("h2d_GBps"):
        legs["end_to_end_h2d_GBps"] = float(ee["h2d_GBps"])
    for row in out.get("kernels") or []:
        name = str(row.get("kernel", ""))
        for tag, short in (("relate_one_fwd (fused single-posterior", "hbm_frac_relate_one"), ("relate_fwd (both posteriors, RelateBatch", "hbm_frac_relate_both"),
                           ("filter_fwd", "hbm_frac_filter"), ("quantify", "hbm_frac_quantify")):
            if name.startswith(tag) and short not in legs and row.get("frac") is not None:
                legs[short] = float(row["frac"])
    par = out.get("parity") or {}
    for k in ("questions_checked", "max_abs_dlp_well_conditioned", "max_abs_dp_vs_fp64", "max_abs_dp"):
        if isinstance(par.get(k), (int, float)) and not isinstance(par.get(k), bool):
            legs["parity_" + k] = par[k]
    if isinstance(par.get("answers_agree"), str) and "/" in par["answers_agree"]:
        a, b = par["answers_agree"].split("/")
        legs["parity_answers_agree"], legs["parity_answers_of"] = int(a), int(b)
    if isinstance(par.get("policy"), dict) and "pass" in par["policy"]:
        legs["parity_policy_pass"] = int(bool(par["policy"]["pass"]))
    return legs


def data_main(args, rank, world, device, td, share):
    """`--workload c3 --data <dir>`: BASELINE configs[2] from FILES in the reference's formats: programs/*.h5 through
    data.ProgramDataset (data_pipeline.py:328-367, 391-453), objects/*.h5 through data.BatchGQABoxFeaturesCollator
    (batch_gqa_boxfeatures_pipeline.py:29-92), the HIP interpreter, answers and error rate (trainer.py:264-318) per batch.  One step = one
    ProgramBatch of up to --batch questions of ONE program file (the reference's sampler batches by terminal operator), launched eagerly -
    every step another batch; the batches of a rank are its contiguous share of the global batch list.  Weights are random (no checkpoint
    travels), so the error rate is chance: the line is about rate and parity, which the oracle checks on the first batch of every file."""
    import glob
    from dfol_vqa_amd import _lib as L
    from dfol_vqa_amd import data, experiment, training
    from dfol_vqa_amd import synthetic as syn
    d = os.path.abspath(args.data)
    md = os.path.join(d, "metadata")
    paths = {"attribute_file": os.path.join(md, "attribute.json"), "class_file": os.path.join(md, "class.json"),
             "relation_file": os.path.join(md, "relation.json"), "vocabulary_file": os.path.join(md, "vocab.json"),
             "word_embedding_file": os.path.join(md, "glove.txt") if os.path.exists(os.path.join(md, "glove.txt")) else None}
    infos = sorted(glob.glob(os.path.join(d, "objects", "*_info.json")))
    progs = sorted(glob.glob(os.path.join(d, "programs", "*.h5")) + glob.glob(os.path.join(d, "programs", "*.npz")))
    if not infos or not progs or not all(os.path.exists(paths[k]) for k in ("attribute_file", "class_file", "relation_file", "vocabulary_file")):
        sys.exit("bench.py --data %s: expected metadata/{attribute,class,relation,vocab}.json, programs/*.h5 and objects/<prefix>_<i>.h5 + <prefix>_info.json" % d)
    prefix = os.path.basename(infos[0])[:-len("_info.json")]
    chunks = len(glob.glob(os.path.join(d, "objects", prefix + "_[0-9]*.h5")) + glob.glob(os.path.join(d, "objects", prefix + "_[0-9]*.npz")))
    L.load()
    cfg = syn.reference_config(paths)
    ontology = experiment.build_ontology(cfg)
    coll = data.BatchGQABoxFeaturesCollator(os.path.join(d, "objects"), prefix, chunks, infos[0], ontology, 1)
    cfg["box_features_dim"] = int(coll._feature_dim)          # 2048 for GQA's bottom-up features; the model's first layer follows the files
    if getattr(ontology, "_embedding_file", None) is not None:
        cfg["word_embedding_dim"] = int(ontology._embedding_dim)
    torch.manual_seed(0)
    model = experiment.build_model(cfg, ontology)
    init_weights(model)
    model = model.to(device).eval()
    if td is not None:
        from dfol_vqa_amd import parallel
        parallel.broadcast_parameters(model, 0)
    # the reference's epoch driver (data_pipeline.py:787-900): one ProgramDataset per program file, batches that never mix files, here in the
    # sequential order its test() loop uses (MultiSetSequencialSampler)
    datasets = [data.ProgramDataset(path, ontology, in_memory=True, shuffle_options=False) for path in progs]
    concat = torch.utils.data.ConcatDataset(datasets)
    bounds = np.cumsum([len(ds) for ds in datasets])
    per_file = {os.path.basename(path): len(ds) for path, ds in zip(progs, datasets)}
    batches = []
    for idx in data.MultiSetSequencialSampler(datasets, args.batch, drop_last=False):
        k = int(np.searchsorted(bounds, idx[0], side="right"))
        batches.append((os.path.basename(progs[k]), [concat[i] for i in idx]))
    if len(batches) < world:
        sys.exit("bench.py --data: %d batches for %d ranks" % (len(batches), world))
    mine = batches[rank * len(batches) // world:(rank + 1) * len(batches) // world]

    def prepare(items):
        pbs = coll.collate(items)
        for pb in pbs:
            pb.create_sparse_tensors()
        return [pb.to_cuda(device) for pb in pbs]

    def barrier():
        if td is not None:
            td.barrier()
        torch.cuda.synchronize()

    err = np.zeros(training.ERROR_DIM)
    cnt = np.zeros(training.ERROR_DIM)
    with torch.no_grad():
        for i in range(max(1, args.warmup)):
            model(prepare(mine[i % len(mine)][1]), False)
        barrier()
        t0 = time.perf_counter()
        nq = 0
        for i in range(args.steps):
            pbs = prepare(mine[i % len(mine)][1])
            res = model(pbs, False)
            training.accumulate_test_batch(err, cnt, pbs, res)
            nq += sum(pb.batch_size() for pb in pbs)
        barrier()
        elapsed = time.perf_counter() - t0
        # per-kernel device time of one pass over this rank's batches (HIP events on the launch stream)
        if rank == 0:
            L.enable_kernel_timing(list(L.SIGNATURES))
            for _, items in mine:
                model(prepare(items), False)
            torch.cuda.synchronize()
            timing = {k: v for k, v in L.disable_kernel_timing().items() if v[0]}
    ranks = rank_report(td, share, device, rank, world, elapsed, args.steps)
    tot = torch.tensor([float(nq), elapsed], dtype=torch.float64, device="cpu" if share or td is None else device)
    if td is not None:
        both = [torch.zeros_like(tot) for _ in range(world)]
        td.all_gather(both, tot)
        nq, elapsed = int(sum(float(b[0]) for b in both)), max(float(b[1]) for b in both)
    if rank == 0:
        # parity: the oracle on the first batch of every program file of this rank (fp32 and float64), under the tolerance policy
        parity = None
        if args.cpu_sample != 0:
            from oracle import dfol_oracle as orc
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import golden_util as gu
            oont = orc.Ontology(paths["attribute_file"], paths["class_file"], paths["vocabulary_file"], paths["relation_file"])
            weights = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if k.startswith("_featurizer.") or k.startswith("_oracle.")}
            checked, worst, agree, total, t_cpu = 0, 0.0, 0, 0, 0.0
            done = set()
            for fname, items in mine:
                if fname in done or checked >= (args.cpu_sample or 64):
                    continue
                done.add(fname)
                items = items[:16]
                pbs = prepare(items)
                with torch.no_grad():
                    res = model(pbs, False)
                feats = pbs[0]._object_features.cpu().numpy()
                ns = [int(n) for n in pbs[0]._object_nums]
                off = np.concatenate([[0], np.cumsum(ns)])
                scenes = [{"n": n, "X": feats[off[i]:off[i + 1]]} for i, n in enumerate(ns)]
                qs = [{"program": it["program"], "answer": it["answer"], "question_id": i, "image_id": it["image_id"]} for i, it in enumerate(items)]
                h0 = time.perf_counter()
                r32 = orc.run_questions(oont, qs, scenes, np.float32, weights=weights)
                t_cpu += time.perf_counter() - h0
                r64 = orc.run_questions(oont, qs, scenes, np.float64, weights=weights)
                lp = res["log_probability"].cpu().numpy()
                gu.check_logprob(lp, r32["log_probability"], r64["log_probability"], fname)       # raises on a policy violation
                worst = max(worst, float(np.abs(np.exp(lp) - np.exp(r64["log_probability"])).max()))
                agree += sum(1 for x, y in zip(res["answer"], r64["answer"]) if x == y)
                total += len(items)
                checked += len(items)
            parity = {"questions_checked": checked, "program_files_checked": sorted(done), "max_abs_dp_vs_fp64": worst, "answers_agree": "%d/%d" % (agree, total),
                      "policy": {"K": 2.0, "p_tol": 1e-6, "lp_tol": 1e-4, "pass": True}}
            cpu = {"value": checked / t_cpu if t_cpu else None, "unit": "questions/s", "cores": 1, "kind": "port",
                   "sample": "%d questions (the first <= 16 of every program file), numpy fp32 oracle incl. the full tables, %.1f s" % (checked, t_cpu)}
        dom = max(timing, key=lambda k: timing[k][1])
        out = {"argv": sys.argv[1:], "metric": "questions/sec (GQA programs from .h5 files: %s)" % d, "value": nq / elapsed, "unit": "questions/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "files: %s" % d,
               "config": {"workload": "BASELINE configs[2] on files in the reference's formats under %s: %d program files (%d questions: %s), object features "
                                      "%s_<i>.h5 x %d (%d features per object, up to %d objects per image), full op set, fp32, <= %d questions per ProgramBatch, "
                                      "random-init weights" % (d, len(progs), sum(per_file.values()), ", ".join("%s %d" % kv for kv in sorted(per_file.items())),
                                                               prefix, chunks, coll._feature_dim, coll._max_object_per_image, args.batch),
                          "workload_id": "c3-files", "batches": len(batches), "batches_per_rank": len(mine), "parallelism": "dp%d" % world,
                          "launch": "eager, every step another ProgramBatch (collate -> lower -> launch -> answers)",
                          "model": "%d -> %d, %d / %d -> %s -> %d -> %d concepts" % (
                              cfg["box_features_dim"], cfg["oracle_input_dim"], cfg["oracle_input_dim"] + 4, 2 * (cfg["oracle_input_dim"] + 4) + 4,
                              cfg["attribute_network_layers_config"], cfg["word_embedding_dim"], len(ontology._vocabulary["idx_to_arg"]))},
               "error_rate": training.metric_dict(err / np.maximum(cnt, 1)), "ranks": ranks,
               "kernel_ms_per_pass": {k: round(v[1] * 1e3, 4) for k, v in sorted(timing.items(), key=lambda kv: -kv[1][1])},
               "roofline": {"kernel": dom, "bound": "mfma" if ("pair_ll" in dom or "linear_act" in dom) else "hbm", "achieved": None, "peak": None, "unit": None,
                            "frac": None, "traffic": None, "us_per_launch": timing[dom][1] / timing[dom][0] * 1e6,
                            "note": "the dominant entry point of a pass over the files; its roofline at the benchmark shape is in the north-star line"},
               "cpu_baseline": cpu if args.cpu_sample != 0 else None, "parity": parity}
        emit(json.dumps(out))
        sys.stdout.flush()
    if td is not None:
        td.barrier()
        td.destroy_process_group()


_FW = {}                                                         # state of a collate worker process of the fresh-programs leg


def _fresh_worker_init(paths, names, kinds, B, N, seeds, spec=None, drop_dir=None):
    """A collate worker: its own ontology (no GPU work in this process), and every batch's question dicts generated up front from their
    seeds - like the launching process, which generates its dicts before the clock starts (a DataLoader worker reads decoded questions)."""
    import dfol_vqa_amd as D
    from dfol_vqa_amd import experiment, synthetic as syn
    # a collate worker runs index bookkeeping on tiny tensors: ONE thread (torch's default is every hardware thread of the box - 256 here -
    # per process, and a pool of workers then fights over the cores with its own idle OpenMP teams)
    torch.set_num_threads(1)
    ontology = experiment.build_ontology(syn.reference_config(paths))
    with open(paths["attribute_file"]) as f:
        cats = json.load(f)
    # (spec: the model's widths, native_exec.model_spec - the worker then also lowers every batch to the native executor's instruction table)
    _FW["coll"] = D.ProgramCollaterBase("select", "relate", "filter", 1, ontology=ontology, native_spec=spec)
    _FW["N"] = N
    _FW["drop_dir"] = drop_dir
    _FW["qs"] = {b: syn.full_size_questions(kinds[b % len(kinds)], B, N, N, names, cats, seed, with_scene=False) for b, seed in seeds.items()}


def _fresh_worker_batch(b):
    coll = _FW["coll"]
    qs = _FW["qs"][b]
    coll.collate_object_features = lambda questions: (None, np.repeat(np.arange(len(questions)), _FW["N"]))      # (object counts: the plan's geometry)
    pbs = coll.collate(qs)                                        # collate -> lower -> plan
    for pb in pbs:
        pb.create_sparse_tensors()
    # (pickled HERE, handed over as bytes: the executor's result thread in the launching process then moves one bytes object instead of
    # rebuilding 10 k small objects under the interpreter lock while the main thread launches; OperatorBatch.__getstate__ sends the small
    # tensors as numpy arrays)
    import pickle
    blob = pickle.dumps(pbs, protocol=pickle.HIGHEST_PROTOCOL)
    if _FW.get("drop_dir"):
        # ~330 KB per batch: through the executor's result pipe they are read by a thread of the launching process that needs the interpreter
        # lock the launching thread holds most of the time - the writers then block on a full pipe, and MORE workers made the leg SLOWER
        # (12 workers: 88 - 114 k questions/s against 125 - 147 k with 6).  A file in memory instead; the path goes through the pipe.
        path = os.path.join(_FW["drop_dir"], "batch_%d.bin" % b)
        with open(path, "wb") as f:
            f.write(blob)
        return path
    return blob


def fresh_programs_rate(args, model, ontology, names, paths, device, rank, n_batches=56, pool=4, stream_features=False):
    """`value_fresh_programs`: the reference's test() loop (trainer.py:685-720) - every step a DIFFERENT ProgramBatch: new programs of mixed
    shapes (eight terminal operators in rotation, 1..3 filter / relate hops, negations, second branches; dfol_vqa_amd.synthetic.full_size_questions)
    on new scenes, through collate (data_pipeline.py:647-783) -> create_sparse_tensors -> lower -> eager launches -> answers read back, per
    batch, like the reference's loop.  The headline `value` replays ONE captured batch; this is what a stream of unseen batches costs.
    Object features are device-resident (a pool of `pool` distinct feature sets, rotated: the boundary takes device pointers; `value_streamed`
    prices the host->device link), program dicts are generated before the clock starts (the reference's DataLoader workers hand over decoded
    questions).  Launch form: eager - a captured graph bakes in the addresses of a batch's uploaded token / index arrays, so a graph cache
    keyed by batch structure would hit only on a repeated batch (hit rate reported as 0 of n)."""
    import dfol_vqa_amd as D
    from dfol_vqa_amd import synthetic as syn
    from dfol_vqa_amd import training
    N, B = args.objects, args.batch
    kinds = ["exist", "verify_rel", "choose_attr", "and", "query_attr", "verify_attrs", "or", "choose_rel"]
    with open(paths["attribute_file"]) as f:
        cats = json.load(f)                                  # category -> its attribute names
    g = torch.Generator(device="cpu").manual_seed(77 + rank)
    feats = [torch.rand(B * N, 2054, generator=g).to(device) for _ in range(pool)]       # (U(0,1) features and box columns, as feature_scene draws them)
    for f in feats:
        f[:, 2048:2050] *= 400.0
        f[:, 2050:2052] = f[:, 2050:2052] * 100.0 + 5.0
        f[:, 2052], f[:, 2053] = 640.0, 480.0
    bindex = torch.arange(B, dtype=torch.int64).repeat_interleave(N)
    state = {"k": 0}

    from dfol_vqa_amd import native_exec
    spec = native_exec.model_spec(model) if native_exec.enabled() else None

    class Collater(D.ProgramCollaterBase):
        def __init__(self):
            super(Collater, self).__init__("select", "relate", "filter", 1, ontology=ontology, native_spec=spec)

        def collate_object_features(self, questions):
            return feats[state["k"] % pool], bindex

    # stream_features (`value_end_to_end`): the features of every batch ALSO arrive from the host - a pool of pinned feature sets (the
    # reference's pin_memory DataLoader, data_pipeline.py:893-898), copied on a copy stream into one of two device buffers while the batch
    # before runs; the batch's launches wait for its copy, the copy waits for the batch that last read its buffer
    if stream_features:
        host_feats = [f.cpu().pin_memory() for f in feats]
        stage = feats[:2]
        copy_stream = torch.cuda.Stream(device=device)
        ready = [torch.cuda.Event() for _ in range(2)]
        consumed = [torch.cuda.Event() for _ in range(2)]
        for e in consumed:
            e.record()
        h2d = {"bytes": 0}

        def stage_features(i):
            """Queue the upload of batch i's features; -> (device tensor, event to wait for)."""
            b = i & 1
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(consumed[b])
                stage[b].copy_(host_feats[i % pool], non_blocking=True)
                ready[b].record(copy_stream)
            h2d["bytes"] += stage[b].numel() * 4
            return stage[b], ready[b], consumed[b]

        def collate_meta_data(self, questions):
            return {"index": {}, "embedding": torch.zeros(1, 1)}

    batches = [syn.full_size_questions(kinds[b % len(kinds)], B, N, N, names, cats, 5000 + 97 * rank + b, with_scene=False) for b in range(n_batches + 3)]
    coll = Collater()

    def one(qs):
        pbs = coll.collate(qs)
        for pb in pbs:
            pb.create_sparse_tensors()
        pbs = [pb.to_cuda(device) for pb in pbs]
        res = model(pbs, False)
        state["k"] += 1
        return training.compute_evaluation_metrics(pbs, res), res

    with torch.no_grad():
        for qs in batches[:2]:                               # warm-up: two batches (allocator, weight images)
            one(qs)
        torch.cuda.synchronize()
        # One thread, software-pipelined: batch i's launches are enqueued (BatchInterpreterBase.forward_async: no read-back yet), batch
        # i + 1 is collated and uploaded while the device runs batch i, then batch i's answers are read back and scored.  (The reference
        # collates in DataLoader worker processes, data_pipeline.py:893-898.  Collating on a worker THREAD was the first form of this leg:
        # 5.2 ms per batch single-threaded became 6.1 - 8.4 - two Python threads take turns on the interpreter lock, and the launching
        # thread waits for it with the device idle.  --fresh-workers N > 0 moves collate and lowering into N spawned worker processes)
        host_s = [0.0]
        workers = max(0, int(getattr(args, "fresh_workers", 0)))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if workers and world > 1:
            # N ranks share the host: every rank takes its share of the cores (one stays with the launching thread), so 8 ranks x 6 workers do not
            # become 48 collate processes on a 32-core host
            workers = max(1, min(workers, (os.cpu_count() or 8) // world - 1))
        # under a profiler (rocprofv3 preloads its tool library into every child too: the workers would become profiled GPU processes
        # writing their own traces into the same directory) this process collates itself
        if workers and (any(k.startswith(("ROCPROFILER", "ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")):
            workers = 0
        executor, futures = None, {}
        if workers:
            # collate + lower in worker processes (spawned: no GPU state is inherited), several batches ahead; this process unpickles
            import multiprocessing
            import pickle
            from concurrent.futures import ProcessPoolExecutor
            seeds = {b: 5000 + 97 * rank + b for b in range(2, n_batches + 3)}
            import tempfile
            drop_dir = tempfile.mkdtemp(prefix="dfol_fresh_%d_" % rank, dir="/dev/shm") if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
            try:
                executor = ProcessPoolExecutor(workers, mp_context=multiprocessing.get_context("spawn"), initializer=_fresh_worker_init,
                                               initargs=(paths, names, kinds, B, N, seeds, spec, drop_dir))
                list(executor.map(int, range(workers * 2)))      # the workers are up (and have generated their dicts) before the clock starts
            except Exception as exc:                             # (a host that cannot start them: this process collates, and the line says so)
                print("bench: collate workers unavailable (%s); collating on the launching thread" % exc, file=sys.stderr)
                if executor is not None:
                    executor.shutdown(wait=False, cancel_futures=True)
                executor, workers = None, 0
            meta = {"index": {}, "embedding": torch.zeros(1, 1)}

            def request(b):
                if 2 <= b < n_batches + 2 and b not in futures:
                    futures[b] = executor.submit(_fresh_worker_batch, b)

        lost = []                                                # (a worker that died mid-run: this process collates from there on, and the line says so)

        def prepare(b):
            blob = None
            if workers and not lost:
                try:
                    for ahead in range(b, b + 2 * workers + 1):
                        request(ahead)
                    blob = futures.pop(b).result()               # (not counted as host time of this process while it waits)
                    if isinstance(blob, str):                    # (a path under /dev/shm: see _fresh_worker_batch)
                        with open(blob, "rb") as f:
                            data = f.read()
                        os.unlink(blob)
                        blob = data
                except Exception as exc:
                    print("bench: collate worker lost (%s); collating on the launching thread from batch %d" % (exc, b), file=sys.stderr)
                    lost.append(b)
            h0 = time.perf_counter()
            if blob is not None:
                pbs = pickle.loads(blob)
                for pb in pbs:                                   # the scenes never left this process
                    pb._object_features, pb._object_batch_index = coll.collate_object_features(None)
                    pb._object_nums, pb._meta_data = [N] * B, meta
            else:
                pbs = coll.collate(batches[b])
                for pb in pbs:
                    pb.create_sparse_tensors()
            pbs = [pb.to_cuda(device) for pb in pbs]
            sync = None
            if stream_features:
                f, rdy, cons = stage_features(b)
                for pb in pbs:
                    pb._object_features = f
                sync = (rdy, cons)
            host_s[0] += time.perf_counter() - h0
            return pbs, sync

        marks = []                                               # (first launch, last launch) of every batch on the device's clock
        n_streams = 1 if stream_features else max(1, int(getattr(args, "fresh_streams", 1)))
        lane_streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)] if n_streams > 1 else None
        if lane_streams:
            for s_ in lane_streams:
                s_.wait_stream(torch.cuda.current_stream(device))

        def launch(pbs, sync):
            # (the executor's batch is self-contained - its blob, its arena, its read-back buffer and event, its own fp16-range word - so
            # consecutive batches may run on different streams; the uploads of to_cuda in prepare() ran on the default stream: wait for them)
            lane = lane_streams[len(marks) % n_streams] if lane_streams else None
            if lane is not None:
                lane.wait_stream(torch.cuda.current_stream(device))
            with (torch.cuda.stream(lane) if lane is not None else contextlib.nullcontext()):
                if sync is not None:
                    torch.cuda.current_stream().wait_event(sync[0])  # this batch's features have landed
                m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                m0.record()
                pending = model.forward_async(pbs, False)
                m1.record()
                marks.append((m0, m1))
                if sync is not None:
                    sync[1].record()                             # its buffer may be overwritten once these launches are done
            return pending

        _route_counts()
        kept, device_ms = [], [None]
        phases = {"prepare_incl_wait_for_worker": 0.0, "launch": 0.0, "wait_for_answers": 0.0, "score": 0.0}
        try:
            # two batches in flight: batch i + 1 is prepared AND launched before batch i's answers are read back, so the device never waits
            # for the host's read-back / scoring / next launch (with one in flight it idled ~0.2 ms of every 2.3 ms batch)
            import gc
            gc.collect()
            gc.freeze()                                          # (what the process keeps - model, ontology, question dicts - leaves the collector's
            t0 = time.perf_counter()                             # generations: a full collection inside the loop is a 2 ms pause, see calibrated_leg)
            import collections
            depth = 2 if stream_features else max(2, int(getattr(args, "fresh_depth", 2)))       # (the streamed form owns two feature buffers)
            inflight, launched = collections.deque(), 0
            for i in range(n_batches):
                while len(inflight) < depth and launched < n_batches:      # keep `depth` batches queued on the device before any answer is waited for
                    h0 = time.perf_counter()
                    p, sy = prepare(2 + launched)
                    h1 = time.perf_counter()
                    inflight.append((p, launch(p, sy)))
                    phases["prepare_incl_wait_for_worker"] += h1 - h0
                    phases["launch"] += time.perf_counter() - h1
                    launched += 1
                    state["k"] += 1
                pbs, pending = inflight.popleft()
                h3 = time.perf_counter()
                res = pending.result()
                h4 = time.perf_counter()
                training.compute_evaluation_metrics(pbs, res)
                phases["wait_for_answers"] += h4 - h3
                phases["score"] += time.perf_counter() - h4
                if len(kept) < 16:
                    kept.append(pbs)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            gc.unfreeze()
            # the device's own clock over the loop: time inside the batches (first to last launch of each) and idle time between them
            inside = sum(a.elapsed_time(b) for a, b in marks)
            between = sum(max(0.0, marks[i][1].elapsed_time(marks[i + 1][0])) for i in range(len(marks) - 1))
            loop_clock = {"inside_batches_ms_per_batch": inside / len(marks), "idle_between_batches_ms_per_batch": between / len(marks)}
            # what the DEVICE needs for batches of this mix (more operators per question than the north-star program, two or three relation
            # columns per image): the first 16 batches of the leg again, launched back to back with nothing read in between - their plans and
            # side arrays are resident by now - between two HIP events
            if kept and not stream_features:
                for x in [model.forward_async(p, False) for p in kept[:2]]:
                    x.result()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                again = [model.forward_async(p, False) for p in kept]
                e1.record()
                torch.cuda.synchronize()
                device_ms[0] = e0.elapsed_time(e1) / len(kept)
                for x in again:
                    x.result()
        finally:
            if executor is not None:
                executor.shutdown(wait=True, cancel_futures=True)
            if workers and drop_dir:
                import shutil
                shutil.rmtree(drop_dir, ignore_errors=True)
        host = host_s[0]
    route = _route_counts()
    extra = {}
    if stream_features:
        gbps = h2d["bytes"] / dt / 1e9
        extra = {"h2d_GBps": gbps, "h2d_bytes_per_question": h2d["bytes"] / (n_batches * B), "pcie_link_GBps": 63.0, "h2d_frac_of_link": gbps / 63.0,
                 "bound": "the host->device link: %.0f KB of raw object features per question at %.1f GB/s of a 63 GB/s PCIe Gen5 x16 link" % (
                     h2d["bytes"] / (n_batches * B) / 1e3, gbps) if gbps > 0.6 * 63.0 else "see host_collate_ms_per_batch / ms_per_batch",
                 "features": "every batch's object features uploaded from pinned host memory on a copy stream (two device buffers), under the batch before"}
    extra = dict(extra, launching_thread_ms_per_batch={k: v / n_batches * 1e3 for k, v in phases.items()}, device_clock_in_the_loop=loop_clock)
    if device_ms[0]:
        extra = dict(extra, device_ms_per_batch=device_ms[0], vs_device_bound="%.2f x what the device alone needs for these batches (%.3f ms each, 16 of them "
                     "launched back to back from resident plans)" % (device_ms[0] / (dt / n_batches * 1e3), device_ms[0]))
    extra = dict(extra, collate_workers=int(workers), streams=int(n_streams))
    return {"questions_per_s": n_batches * B / dt, "ms_per_batch": dt / n_batches * 1e3, "batches": n_batches, "questions_per_batch": B, "executor": route, **extra,
            "terminal_operators": kinds, "host_collate_ms_per_batch": host / n_batches * 1e3, "collate": ("in %d worker processes, batches ahead%s" % (workers, " (lost at batch %d: this process from there)" % lost[0] if lost else "")) if workers else "on the launching thread, while the device runs the batch before (forward_async)",
            "launch": "native executor: one dfol_run_program call per ProgramBatch, lowered at collate time" if route.get("native_program") else "eager (Python operator loop)",
            "graph_cache": {"hits": 0, "of": n_batches},
            "how": "every batch new programs (1..3 hops, mixed terminal operators) and another scene set; collate -> create_sparse_tensors -> lower -> "
                   "eager launches -> answers and error rate read back per batch; object features device-resident"}


def _route_counts():
    from dfol_vqa_amd import _lib as L
    c = {k: v for k, v in L.PATH_COUNTS.items() if k in ("native_program", "python_program") or k.startswith("fallback:")}
    L.PATH_COUNTS.clear()
    return c


def dominant_roofline(args, model, dom, per_step):
    ns = np.asarray(getattr(args, "_scene_ns", None) or [args.objects] * args.batch, np.float64)      # objects of every DISTINCT scene of the batch
    O = float(ns.sum())
    pairs = float((ns * (ns - 1)).sum())                # (with n*n slots per image the kernel also computes the diagonal; not counted)
    launches, secs = per_step[dom]
    KR = 4 if args.workload == "c4" else 1              # relation columns requested per image (one per relate hop of the program)
    if getattr(args, "_tiles_per_scene", None):
        KR = args._tiles_per_scene                       # shared scenes: distinct (relation, orientation) requests per image
    if dom in ("dfol_pair_ll_split_f32", "dfol_pair_ll_h2_f32"):
        # The same algorithmic flops, executed on a 16-bit matrix pipe as piece products per fp32 product: three (two fp16 pieces per
        # operand, csrc/dfol_pair_h2.hip, the default) or six (three exact bf16 pieces, csrc/dfol_pair_split.hip), fp32 accumulate, fp32
        # results.  `achieved` / `frac` follow the contract (ALGORITHMIC flops against the peak of the pipe that executes them - the fp16
        # and bf16 dense peaks are the same 2.5 PFLOP/s); the pipe itself does 3x / 6x that work (`executed`).
        h2 = dom == "dfol_pair_ll_h2_f32"
        pieces = 3 if h2 else 6
        flops = 2.0 * pairs * (4 * 256 + 256 * 300 + 300 * KR)
        ach = flops / secs
        return {"kernel": "pair_ll32h_kernel<19> (fused pair MLP -> requested relation tiles, fp16x2 split)" if h2 else
                          "pair_ll32s_kernel<19> (fused pair MLP -> requested relation tiles, bf16x3 split)",
                "trace_name": "pair_ll32h_kernel" if h2 else "pair_ll32s_kernel", "bound": "mfma",
                "achieved": ach / 1e12, "peak": BF16_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / BF16_MFMA_PEAK,
                "traffic": None, "launches_per_step": launches, "us_per_launch": secs / launches * 1e6,
                "flops_per_pair": 2 * (4 * 256 + 256 * 300 + 300 * KR),
                "executed": {"mfma_flops_per_algorithmic_flop": pieces, "achieved": pieces * ach / 1e12, "frac": pieces * ach / BF16_MFMA_PEAK},
                "vs_f32_mfma_peak": {"peak": F32_MFMA_PEAK / 1e12, "frac": ach / F32_MFMA_PEAK}}
    if dom in ("dfol_pair_ll_f32", "dfol_pair_ll_packed_f32"):
        # reduced-form algorithmic flops per ordered pair: geometry term, 256->300 layer, and the
        # K requested embedding columns (K = 1 relation per question in this workload)
        flops = 2.0 * pairs * (4 * 256 + 256 * 300 + 300 * 1)
        ach = flops / secs
        return {"kernel": "pair_ll32b_kernel<19> (fused pair MLP -> requested relation tiles)", "trace_name": "pair_ll32b_kernel", "bound": "mfma",
                "achieved": ach / 1e12, "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / F32_MFMA_PEAK,
                "traffic": None, "launches_per_step": launches, "us_per_launch": secs / launches * 1e6,
                "flops_per_pair": 2 * (4 * 256 + 256 * 300 + 300)}
    if dom in ("dfol_linear_act_f32", "dfol_linear_act_split_f32", "dfol_linear_act_h2_f32"):
        if getattr(model._oracle, "_needed_columns", False) and model._oracle.supports_needed_columns():
            flops = 2.0 * O * (2048 * 512 + 516 * 256 + 256 * 300 + 516 * 512)
        else:   # full cached tables (only the 333 relation columns of the pair embedding are computed)
            flops = 2.0 * (O * 2048 * 512 + O * (516 * 256 + 256 * 300 + 300 * 2335) + pairs * (1036 * 256 + 256 * 300 + 300 * 333))
        ach = flops / secs
        peak = F32_MFMA_PEAK if dom == "dfol_linear_act_f32" else BF16_MFMA_PEAK
        return {"kernel": "%s (all GEMM launches of one step)" % dom, "bound": "mfma", "achieved": ach / 1e12,
                "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None}
    N = args.objects
    nbytes = args.batch * (4 * N * N + 16 * N)
    ach = nbytes / (secs / max(launches, 1))
    return {"kernel": dom, "bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": ach / HBM_PEAK, "traffic": None}


def streamed_rate(args, step, pbs, td, share, device, batches=4):
    """The same step with the object features of every batch arriving from PINNED host memory (the reference's pin_memory DataLoader,
    data_pipeline.py:893-898): upload of batch i+1 on a copy stream, double-buffered, under the replay of batch i.  The boundary takes
    device pointers, so this is the caller's side of it; `value` stays the resident-input rate, as the bench contract asks."""
    pb = pbs[0]
    feats = pb._object_features
    host = [feats.detach().cpu().pin_memory() for _ in range(2)]
    stage = [torch.empty_like(feats) for _ in range(2)]
    copy_stream = torch.cuda.Stream(device=device)
    main_stream = torch.cuda.current_stream()
    ready = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]

    def upload(i):
        b = i & 1
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[b])              # the step that read stage[b] has finished
            stage[b].copy_(host[b], non_blocking=True)
            ready[b].record(copy_stream)

    def run(n):
        for b in range(2):
            consumed[b].record(main_stream)
        upload(0)
        for i in range(n):
            b = i & 1
            if i + 1 < n:
                upload(i + 1)
            main_stream.wait_event(ready[b])
            feats.copy_(stage[b])                            # device-to-device into the tensor the captured graph reads (0.21 GB at 8 TB/s)
            consumed[b].record(main_stream)
            step()

    with torch.no_grad():
        run(2)
        if td is not None:
            td.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = max(4, args.steps)
        run(n)
        if td is not None:
            td.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if td is not None:
        t = torch.tensor([dt], device="cpu" if share else device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
    nbytes = feats.numel() * 4
    return {"questions_per_s": args.batch * n / dt, "ms_per_step": dt / n * 1e3, "h2d_bytes_per_step": nbytes, "h2d_bytes_per_question": nbytes / args.batch,
            "h2d_GBps": nbytes * n / dt / 1e9, "how": "pinned host -> device on a copy stream, double-buffered, overlapped with the step"}


def train_main(args, rank, world, device, td, share):
    """One train step = zero grads -> forward (is_training) -> loss / B_global -> backward -> ONE all-reduce of the flat fp32 gradient
    bucket (RCCL) -> clip_grad_norm_ -> Adam, on a resident batch of `--batch` questions per GPU (BASELINE configs[3]'s step)."""
    from dfol_vqa_amd import _lib as L
    from dfol_vqa_amd import parallel, training
    model, ontology, paths, names = build_model(args, device, train=True)
    group = None
    if td is not None:
        group = td.group.WORLD
        parallel.broadcast_parameters(model, 0, group)
    _, pbs = build_batch(args, rank, ontology, names, device)
    if args.calibrator:                                      # the LSTM inputs need token embeddings (random here, GloVe in the reference)
        voc = list(ontology._vocabulary["idx_to_arg"])
        g = torch.Generator().manual_seed(5)
        emb = (torch.randn(len(voc), 300, generator=g) * 0.1).to(device)
        for pb in pbs:
            pb._meta_data = {"index": {t: i for i, t in enumerate(voc)}, "embedding": emb}
    params = [p for p in model.parameters() if p.requires_grad]
    # one process: the whole step as ONE captured HIP graph.  Data parallel (any backend): TWO graphs - zero + forward + loss + backward, and
    # clip + Adam - with the bucket's all-reduce issued eagerly between the two replays (training.GraphedTrainStep; no collective is ever
    # captured).  --graph-collective 1 (RCCL only, opt-in): one graph with the all-reduce as a node.  The overlap hooks issue collectives
    # from autograd threads: eager only.
    use_graph = bool(args.graph) and not (td is not None and args.overlap_allreduce)
    graph_collective = bool(args.graph_collective) and td is not None
    if graph_collective and (td.get_backend() != "nccl" or not use_graph):
        raise SystemExit("--graph-collective 1 needs --graph 1, --overlap-allreduce 0 and the nccl (RCCL) backend")
    opt = torch.optim.Adam(params, lr=1e-4, capturable=use_graph)
    bucket = parallel.GradBucket(params)
    if td is not None and args.overlap_allreduce:
        bucket.enable_overlap(group, segments=3)             # ranges of the bucket are all-reduced while the backward still runs
    gb = args.batch * world
    # (the loss stays on the GPU and is read once after the timed steps: no host wait between steps)
    eager_step = lambda: training.train_batch(model, opt, pbs, 0.65, global_batch_size=gb, group=group, bucket=bucket, sync_loss=False)
    step, graphed = eager_step, False
    if use_graph:
        def capture():
            return training.GraphedTrainStep(model, opt, pbs, 0.65, bucket=bucket, group=group, global_batch_size=gb, graph_collective=graph_collective)
        if graph_collective:                                 # opt-in path: no fallback - a failure is this process's non-zero exit
            step = capture()
            graphed = True
        else:
            try:
                step = capture()
                graphed = True
            except Exception as e:
                if td is not None:                           # ranks must not diverge (one eager, one replaying): fail the run instead
                    raise
                sys.stderr.write("train-step graph capture failed (%r); running eager\n" % (e,))
                torch.cuda.synchronize()
        if graphed:
            # a GPU-bound step (the oracle phases at N = 100: 12.9 ms of kernels in a 13.5 ms step) gains nothing from the replay and
            # pays for the graph's private memory pool; the host-bound ones (calibrator phases, small scenes) gain 15-35 %: keep the faster
            def clock(fn, n=4):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n
            t_graph, t_eager = clock(step), clock(eager_step)
            if td is not None:                               # every rank must take the same path: decide on the slowest rank's clocks
                tt = torch.tensor([t_graph, t_eager], device="cpu" if share else device, dtype=torch.float64)
                td.all_reduce(tt, op=td.ReduceOp.MAX)
                t_graph, t_eager = float(tt[0]), float(tt[1])
            if t_eager < 0.98 * t_graph:
                step, graphed = eager_step, False
                torch.cuda.empty_cache()

    def barrier():
        if td is not None:
            td.barrier()
        torch.cuda.synchronize()

    torch.cuda.reset_peak_memory_stats()
    for _ in range(args.warmup):
        loss, _ = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    barrier()
    elapsed = time.perf_counter() - t0
    loss = float(loss)
    equal = True
    ranks = rank_report(td, share, device, rank, world, elapsed, args.steps)
    allreduce_ms = None
    if td is not None:
        # the collective alone: 10 all-reduces of a bucket-sized buffer, barrier + synchronize on both sides, MAX over ranks
        scratch = torch.zeros_like(bucket.flat)
        for _ in range(3):
            td.all_reduce(scratch, group=group)
        barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            td.all_reduce(scratch, group=group)
        barrier()
        ta = torch.tensor([(time.perf_counter() - t1) / 10 * 1e3], device="cpu" if share else device, dtype=torch.float64)
        td.all_reduce(ta, op=td.ReduceOp.MAX)
        allreduce_ms = float(ta.item())
        del scratch
    if td is not None:
        t = torch.tensor([elapsed], device="cpu" if share else device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
        dg = parallel.parameters_digest(model).to("cpu" if share else device)
        both = [torch.zeros_like(dg) for _ in range(world)]
        td.all_gather(both, dg)
        equal = all(torch.equal(both[0], b) for b in both)
        lt = torch.tensor([loss], device="cpu" if share else device, dtype=torch.float64)
        td.all_reduce(lt)
        loss = float(lt.item())
    # per-kernel times of two more steps (every rank runs them: the step contains the collective; rank 0 keeps the numbers)
    if rank == 0:
        L.enable_kernel_timing(list(L.SIGNATURES))
    for _ in range(2):
        eager_step()
    torch.cuda.synchronize()
    if rank == 0:
        timing = L.disable_kernel_timing()
        per_step = {k: (n / 2.0, t / 2.0) for k, (n, t) in timing.items() if n}
        out = {"metric": "training questions/sec (GQA programs, N=%s objects; forward + backward + all-reduce + clip + Adam)"
                         % (args.objects if not args.ragged else "U{%d..%d}" % (args.ragged, args.objects)),
               "value": gb * args.steps / elapsed, "unit": "questions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16" if args.mlp_math == "bf16" else "f32", "data": "synthetic",
               "config": {"workload": "BASELINE configs[3]'s step on synthetic scenes: %s, BCE loss, %s phase, "
                                      "%d questions/GPU/step%s" % ("select -> 1..3 filter / relate hops -> exist, a program length per question (no-op tokens after collate)"
                                                                   if args.hops == "ragged" else "select->filter->relate->exist",
                                                                   "calibrator (cur6-7)" if args.calibrator else "oracle (cur1-5)", args.batch,
                                                                   ", mlp_math bf16 (dense products on bf16 operands, fp32 accumulation and logic)"
                                                                   if args.mlp_math == "bf16" else ""),
                          "hops": args.hops, "global_batch": gb, "parallelism": "dp%d" % world, "gradient_bucket_bytes": bucket.nbytes(),
                          "launch": ("eager" if not graphed else "hip graph replay of the whole step (training.GraphedTrainStep)" if td is None else
                                     "hip graph replay of the whole step, the all-reduce captured inside (training.GraphedTrainStep, graph_collective)"
                                     if graph_collective else
                                     "two hip graph replays per step (zero + forward + loss + backward | clip + Adam) with the all-reduce issued "
                                     "eagerly between them (training.GraphedTrainStep)"),
                          "collective": "%s of the flat fp32 bucket per step (%s)" % (
                              "three all-reduces(sum) of contiguous ranges, issued during the backward," if (td is not None and args.overlap_allreduce)
                              else "one all-reduce(sum)", "gloo, shared GPU" if share else "RCCL")},
               "loss": loss, "replicas_equal": bool(equal), "ranks": ranks, "allreduce_ms": allreduce_ms,
               "allreduce_note": None if allreduce_ms is None else "standalone all-reduce(sum) of a %d-byte fp32 buffer (the gradient bucket's size), mean of 10, max over ranks; "
                                 "inside the step it is %s" % (bucket.nbytes(), "issued per range during the backward" if args.overlap_allreduce else "one call after the backward"),
               "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9,
               "kernel_ms_per_step": {k: round(v[1] * 1e3, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1][1])},
               "kernels": train_kernel_rooflines(args, per_step), "roofline": None, "cpu_baseline": None}
        # the roofline object: the entry point the step spends most time in (the split-kernel GEMMs)
        out["roofline"] = max(out["kernels"], key=lambda k: k["ms_per_step"]) if out["kernels"] else None
        emit(json.dumps(out))
        sys.stdout.flush()
    if td is not None:
        td.barrier()
        td.destroy_process_group()


def train_kernel_rooflines(args, per_step):
    """Algorithmic bytes / flops of the training kernels of one step (fixed-size scenes only) against their measured time: the four
    streams around the pair MLP's tall GEMMs (csrc/dfol_pair_train.hip), the logic backward kernels, the weight-gradient kernel."""
    if args.ragged or args.calibrator:
        return []
    Q, N, H1, H2 = args.batch, args.objects, 256, 300
    pairs, O = Q * N * (N - 1), Q * N
    rows = []
    traffic = {}                                             # HBM bytes per step and entry point from the committed counter passes of this shape
    tpath = os.path.join(ROOT, "profiles", "train_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f).get("bytes_per_step", {})

    def add(entry, what, bound, work, note, peak=None, extra=None):
        if entry not in per_step:
            return
        launches, secs = per_step[entry]
        peak = peak or (HBM_PEAK if bound == "hbm" else F32_MFMA_PEAK)
        rows.append({"entry": entry, "kernel": what, "bound": bound, "achieved": work / secs / (1e9 if bound == "hbm" else 1e12), "peak": peak / (1e9 if bound == "hbm" else 1e12),
                     "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": work / secs / peak, "launches_per_step": launches, "ms_per_step": secs * 1e3,
                     "algorithmic": note})
        if extra:
            rows[-1].update(extra(work / secs))
        rows[-1]["traffic"] = traffic.get(entry) if (N == 100 and Q == 256 and getattr(args, "mlp_math", "fp32") == "fp32") else None

    add("dfol_pair_hidden1_fwd_f32", "pair_hidden1_fwd (Z = ELU(U[s] + V[o] + Wg geo) written once)", "hbm", pairs * (4.0 * H1 + 16), "pairs x (4 HID1 + 16) B written")
    add("dfol_pair_hidden1_bwd_f32", "pair_hidden1_bwd (dU, dV, dWg reduced per image, no atomics)", "hbm", pairs * (8.0 * H1 + 16), "pairs x (8 HID1 + 16) B read")
    add("dfol_pair_hidden1_bwd_recompute_f32", "pair_hidden1_bwd, Z rebuilt from U, V, Wg and the geometry instead of read (dU, dV, dWg reduced per image, no atomics)", "hbm",
        pairs * (4.0 * H1 + 16), "pairs x (4 HID1 + 16) B read")
    add("dfol_pair_logit_fwd_f32", "pair_logit_fwd (Sigmoid . embedding row -> logit)", "hbm", pairs * (4.0 * H2 + 4), "pairs x (4 HID2 + 4) B")
    add("dfol_pair_logit_bwd_f32", "pair_logit_bwd (dP2, dE, db in one pass)", "hbm", pairs * (8.0 * H2 + 4), "pairs x (8 HID2 + 4) B")
    # round 4: the head's backward without dpre2 in memory, and the tall products as one persistent workgroup per CU - all four stream the
    # per-pair activations once; the two with MFMAs execute three fp16 products per algorithmic product on top
    mm = lambda rate_bytes, bytes_per_pair: {"executed": {"mfma_flops_per_algorithmic_flop": 3, "achieved": 3 * 2.0 * H2 * H1 * (rate_bytes / bytes_per_pair) / 1e12,
                                                          "frac": 3 * 2.0 * H2 * H1 * (rate_bytes / bytes_per_pair) / BF16_MFMA_PEAK}}
    add("dfol_linear_tall_h2_f32", "tall_h2_kernel<5, 2> (pre2 = Z W2^T + b2 written once, Z read once, the logit partial sums from the epilogue)", "hbm",
        pairs * (4.0 * H1 + 4 * H2 + 16), "pairs x (4 HID1 + 4 HID2 + 16) B", extra=lambda r: mm(r, 4.0 * H1 + 4 * H2 + 16))
    add("dfol_pair_dz_tall_f32", "tall_h2_kernel<4, 1> (dZ = dpre2 W2, dpre2 rebuilt from pre2; + the row-scale pass)", "hbm",
        pairs * (4.0 * H2 + 4 * H1 + 24), "pairs x (4 HID2 + 4 HID1 + 24) B", extra=lambda r: mm(r, 4.0 * H2 + 4 * H1 + 24))
    add("dfol_pair_dz_fused_f32", "linear_act_split_kernel<.., 1> (dZ = dpre2 W2, dpre2 rebuilt from pre2; tiled form)", "hbm",
        pairs * (4.0 * H2 + 4 * H1 + 8), "pairs x (4 HID2 + 4 HID1 + 8) B", extra=lambda r: mm(r, 4.0 * H2 + 4 * H1 + 8))
    add("dfol_pair_wgrad_fused_f32", "pair_wgrad_fused_kernel<3> (dW2 = dpre2^T Z, pre2 and Z read once) + reduce", "hbm",
        pairs * (4.0 * H2 + 4 * H1 + 8), "pairs x (4 HID2 + 4 HID1 + 8) B", extra=lambda r: mm(r, 4.0 * H2 + 4 * H1 + 8))
    add("dfol_pair_wgrad_fused_sums_f32", "pair_wgrad_fused_kernel<3, true> (dW2 = dpre2^T Z and the logit layer's sums dE, dbe, db2; pre2 and Z read once) + reduces", "hbm",
        pairs * (4.0 * H2 + 4 * H1 + 8), "pairs x (4 HID2 + 4 HID1 + 8) B", extra=lambda r: mm(r, 4.0 * H2 + 4 * H1 + 8))
    add("dfol_pair_logit_bwd_sums_f32", "pair_logit_bwd4_kernel (dE, dbe, db2 sums; no dpre2 written)", "hbm", pairs * (4.0 * H2 + 4), "pairs x (4 HID2 + 4) B")
    add("dfol_relate_bwd_f32", "relate_bwd (tile read twice, gradient tile written)", "hbm", Q * (12.0 * N * N + 24 * N), "P x (12 N^2 + 24 N) B")
    add("dfol_filter_bwd_f32", "filter_bwd", "hbm", per_step.get("dfol_filter_bwd_f32", (1, 1))[0] * Q * 16.0 * N, "launches x P x 16 N B")
    add("dfol_quantify_bwd_f32", "quantify_bwd", "hbm", Q * (8.0 * N + 4), "P x (8 N + 4) B")
    # (round 4: the pair layer's three products have entry points of their own - rows above - when the deferred head runs; the rows below
    # then count the per-object layers only)
    head = any(k in per_step for k in ("dfol_pair_wgrad_fused_sums_f32", "dfol_pair_wgrad_fused_f32"))
    tall_fwd = "dfol_linear_tall_h2_f32" in per_step
    wflops = 2.0 * ((0 if head else pairs * H2 * H1) + O * (512 * 2048 + 2 * H1 * 516 + 256 * 516 + H2 * 256))
    # fp32 results from the bf16 pipe: six bf16 MFMA flops are executed per algorithmic flop (three-way operand split), so `frac` against the
    # bf16 dense peak is bounded by 1/6; `executed` is the pipe-side rate (as for the pair kernel of the inference line)
    for entry, pieces in (("dfol_linear_wgrad_bias_f32", 6), ("dfol_linear_wgrad_bias_bf16", 1)):
        add(entry, "wgrad_tn3_kernel (dW = dY^T X and db: %sthe five per-object layers)" % ("" if head else "pair layer 300 x 256 over all pairs + "), "mfma", wflops,
            "2 (%sO (512 2048 + 2 256 516 + 256 516 + 300 256)) flops" % ("" if head else "pairs HID2 HID1 + "), peak=BF16_MFMA_PEAK,
            extra=lambda rate, pieces=pieces: {"executed": {"mfma_flops_per_algorithmic_flop": pieces, "achieved": pieces * rate / 1e12,
                                                            "frac": pieces * rate / BF16_MFMA_PEAK},
                                               "vs_f32_mfma_peak": {"peak": F32_MFMA_PEAK / 1e12, "frac": rate / F32_MFMA_PEAK}})
    # the split-kernel GEMMs of the step: forward of the four per-object layers and of the pair layer (two fp16 pieces, three products - in the
    # bf16 mode one bf16 piece), input gradients of all but the featurizer (three bf16 pieces, six products: operands of any magnitude)
    fwd = 2.0 * ((0 if tall_fwd else pairs * H2 * H1) + O * (512 * 2048 + 2 * H1 * 516 + 256 * 516 + H2 * 256))
    bwd = 2.0 * ((0 if head else pairs * H2 * H1) + O * (2 * H1 * 516 + 256 * 516 + H2 * 256))
    h2_runs = "dfol_linear_act_h2_f32" in per_step
    for entry, what, work, note in (
            ("dfol_linear_act_h2_f32", "linear_act_split_kernel, two fp16 pieces (the forward products: %sthe per-object layers)" % ("" if tall_fwd else "pair layer [pairs,256]->300, "), fwd,
             "2 (%sO (512 2048 + 2 256 516 + 256 516 + 300 256)) flops" % ("" if tall_fwd else "pairs HID2 HID1 + ")),
            ("dfol_linear_act_split_f32", "linear_act_split_kernel, three bf16 pieces (the input-gradient products%s)" % ("" if h2_runs else " and the forward products"),
             bwd if h2_runs else fwd + bwd, "2 (%sO (2 256 516 + 256 516 + 300 256)) flops" % ("" if head else "pairs HID2 HID1 + ") + ("" if h2_runs else " + the forward products'")),
            ("dfol_linear_act_bf16_f32", "linear_act_split_kernel, one bf16 piece (forward and input-gradient products)", fwd + bwd,
             "2 (2 pairs HID2 HID1 + O (512 2048 + 2 (2 256 516 + 256 516 + 300 256))) flops")):
        pieces = {"dfol_linear_act_split_f32": 6, "dfol_linear_act_h2_f32": 3}.get(entry, 1)
        add(entry, what, "mfma", work, note, peak=BF16_MFMA_PEAK,
            extra=lambda rate, pieces=pieces: {"executed": {"mfma_flops_per_algorithmic_flop": pieces, "achieved": pieces * rate / 1e12,
                                                            "frac": pieces * rate / BF16_MFMA_PEAK},
                                               "vs_f32_mfma_peak": {"peak": F32_MFMA_PEAK / 1e12, "frac": rate / F32_MFMA_PEAK}})
    return rows


def attach_traffic(out, args):
    """HBM traffic per launch from the committed rocprofv3 counter passes of THIS command (profiles/traffic.json, written by
    tools/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc passes, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950).  Entries are matched by kernel name AND launch grid (total work-items), which every `kernels` row computes
    from its own launch geometry.  Left null when no profile of the same shape is on disk."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return
    with open(path) as f:
        t = json.load(f)

    def total(prefix, grid=None):
        for key, v in t.items():
            name, gx = key.rsplit("@", 1)
            if name.startswith(prefix) and (grid is None or int(gx) == grid) and "fetch_bytes_x2" in v:
                return v["fetch_bytes_x2"] + v.get("write_bytes", 0.0)      # (a kernel whose WRITE_SIZE pass recorded nothing: < 1 KiB per XCD counter tick)
        return None

    same_shape = args.workload == "north_star" and args.objects == 100 and args.batch == 256 and not args.ragged and args.questions_per_image == 1
    if "pair_ll" in out["roofline"]["kernel"] and same_shape:              # (the counters were collected on the default command only)
        # (the kernel of this run by its trace name: pair_ll32h_kernel on two fp16 pieces, pair_ll32s / pair_ll32b / pair_ll16 under the A/B switches)
        out["roofline"]["traffic"] = total(out["roofline"].get("trace_name") or "pair_ll32h_kernel")
        out["roofline"]["traffic_unit"] = "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/traffic.json)"
    for k in out["kernels"]:
        tr = k.get("trace")
        k["traffic"] = total(tr["kernel"], tr["grid_x"]) if tr else None
        if k["traffic"] is not None and tr["kernel"].startswith("relate"):
            k["traffic_note"] = "counter average over every launch of this kernel and grid in the profiled run (all predicate kinds)"


def attach_rocprof(out, args):
    """rocprofv3 kernel durations of THE SAME launches, from a committed trace of this command (profiles/roofline_rocprof.json, written by
    tools/profile_bench.sh + tools/summarize_profile.py on the GPU box): the roofline leg's launches are the last --roofline-reps
    launches of the dominant kernel in the process, the `kernels` rows are launch groups in a fixed order.  Printed BESIDE the live
    HIP-event numbers so that the line can be re-derived from profiles/; null when no trace of this shape is on disk."""
    path = os.path.join(ROOT, "profiles", "roofline_rocprof.json")
    out["roofline"]["rocprofv3"] = None
    if not os.path.exists(path):
        return
    with open(path) as f:
        prof = json.load(f)
    key = "%s:n%d:b%d" % (args.workload, args.objects, args.batch)
    entry = prof.get(key)
    if not entry:
        return
    r = out["roofline"]
    dom = entry.get("dominant")
    if dom and dom.get("trace_name") == r.get("trace_name") and r.get("unit") == "TFLOP/s":
        work = r["achieved"] * 1e12 * r["us_per_launch"] * 1e-6             # algorithmic flops per launch, as priced above
        ach = work / (dom["avg_us"] * 1e-6)
        r["rocprofv3"] = {"us_per_launch": dom["avg_us"], "launches_averaged": dom["launches"], "achieved": ach / 1e12, "frac": ach / (r["peak"] * 1e12),
                          "hip_events_in_that_run_us": dom.get("hip_event_us_per_launch_same_run"),
                          "source": "profiles/roofline_rocprof.json (%s; the last %d launches of %s in a rocprofv3 --kernel-trace of this command)"
                                    % (entry.get("head", "?"), dom["launches"], dom["trace_name"])}
        if dom.get("all_launches_avg_us"):                                  # the `--stats` row of the same trace: every launch of the run
            r["rocprofv3"]["stats_avg_us"] = dom["all_launches_avg_us"]
            r["rocprofv3"]["stats_launches"] = dom["all_launches"]
            r["rocprofv3"]["frac_from_stats_avg"] = work / (dom["all_launches_avg_us"] * 1e-6) / (r["peak"] * 1e12)
        if dom.get("alone_avg_us"):
            # ... of which the launches that had the device to themselves (the replay lanes of the timed region run two steps side by side: those
            # launches share the CUs with the other lane's kernels and last longer while the steps get shorter - not what a roofline fraction prices)
            r["rocprofv3"]["stats_alone_avg_us"] = dom["alone_avg_us"]
            r["rocprofv3"]["stats_alone_launches"] = dom["alone_launches"]
            r["rocprofv3"]["stats_overlapped_avg_us"] = dom.get("overlapped_avg_us")
            r["rocprofv3"]["stats_overlapped_launches"] = dom.get("overlapped_launches")
            r["rocprofv3"]["frac_from_stats_alone_avg"] = work / (dom["alone_avg_us"] * 1e-6) / (r["peak"] * 1e12)
        ser = dom.get("serial_run")
        if ser and ser.get("all_launches_avg_us"):
            # ... and the `--stats` row of the same command run with --pipeline 0 (profiles/r06_rocprof_summary_serial.md): every launch alone on the device
            r["rocprofv3"]["stats_serial_run_avg_us"] = ser["all_launches_avg_us"]
            r["rocprofv3"]["stats_serial_run_launches"] = ser["all_launches"]
            r["rocprofv3"]["frac_from_stats_serial_run_avg"] = work / (ser["all_launches_avg_us"] * 1e-6) / (r["peak"] * 1e12)
    groups = entry.get("kernels", {})
    for k in out["kernels"]:
        tr = k.get("trace")
        g = groups.get("%s@%d#%d" % (tr["kernel"], tr["grid_x"], tr["group"])) if tr else None
        if g:
            nbytes = k["achieved"] * 1e9 * k["us_per_launch"] * 1e-6
            k["rocprofv3"] = {"us_per_launch": g["avg_us"], "launches_averaged": g["launches"], "frac": nbytes / (g["avg_us"] * 1e-6) / HBM_PEAK}


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
    RG = (P + 3) // 4 * 256                                  # launch grid (work-items) of the Relate kernels: one wavefront per predicate

    groups = {}                                              # (kernel name in the trace, grid) -> launch groups so far, in order

    def trace_id(kernel, grid_x):
        g = groups.get((kernel, grid_x), 0)
        groups[(kernel, grid_x)] = g + 1
        return {"kernel": kernel, "grid_x": int(grid_x), "group": g, "launches": WARM + ITERS, "timed": ITERS}

    WARM, ITERS = 10, 20                                     # (3 warm-up launches left the first rows 8 % slow: the clocks had not ramped yet)

    def timed(name, fn, nbytes, iters=ITERS):
        """Average launch duration from ONE HIP-event pair around `iters` back-to-back launches on the launch stream
        (bracketing every single launch with its own event pair adds marker/fence latency to sub-millisecond kernels)."""
        for _ in range(WARM):
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
    res.append(dict(r, kernel="relate_one_fwd (fused single-posterior Relate, the interpreter's path)", bytes_per_predicate=4 * N * N + 12 * N,
                    trace=trace_id("relate_one_fwd_kernel", RG)))
    r = timed("dfol_relate_fwd_f32", lambda: L.relate_fwd(prior, prior, tile, pq, n_obj, ones, ones, need_s=False, diag_absent=True), P * (4 * N * N + 12 * N))
    res.append(dict(r, kernel="relate_fwd (generic cell, one posterior wanted)", bytes_per_predicate=4 * N * N + 12 * N, trace=trace_id("relate_fwd_kernel", RG)))
    r = timed("dfol_relate_fwd_f32", lambda: L.relate_fwd(prior, prior, tile, pq, n_obj, ones, ones, diag_absent=True), P * (4 * N * N + 16 * N))
    res.append(dict(r, kernel="relate_fwd (both posteriors, RelateBatch API)", bytes_per_predicate=4 * N * N + 16 * N, trace=trace_id("relate_fwd_kernel", RG)))
    # the other predicate kinds (negated / FOR_ALL): transcendental-poor forms with a clamping fallback (csrc/dfol_logic.hip)
    zeros, negs = torch.zeros(P, device=device), torch.ones(P, dtype=torch.uint8, device=device)
    r = timed("dfol_relate_one_fwd_f32", lambda: L.relate_one_fwd(prior, prior, tile, pq, n_obj, ones, negs), P * (4 * N * N + 12 * N))
    res.append(dict(r, kernel="relate_one_fwd (negated EXISTS predicates)", bytes_per_predicate=4 * N * N + 12 * N, trace=trace_id("relate_one_fwd_kernel", RG)))
    r = timed("dfol_relate_one_fwd_f32", lambda: L.relate_one_fwd(prior, prior, tile, pq, n_obj, zeros), P * (4 * N * N + 12 * N))
    res.append(dict(r, kernel="relate_one_fwd (FOR_ALL predicates)", bytes_per_predicate=4 * N * N + 12 * N, trace=trace_id("relate_one_fwd_kernel", RG)))
    r = timed("dfol_relate_fwd_f32", lambda: L.relate_fwd(prior, prior, tile, pq, n_obj, ones, ones, negs, diag_absent=True), P * (4 * N * N + 16 * N))
    res.append(dict(r, kernel="relate_fwd (both posteriors, negated EXISTS predicates)", bytes_per_predicate=4 * N * N + 16 * N, trace=trace_id("relate_fwd_kernel", RG)))
    r = timed("dfol_relate_fwd_f32", lambda: L.relate_fwd(prior, prior, tile, pq, n_obj, zeros, zeros, diag_absent=True), P * (4 * N * N + 16 * N))
    res.append(dict(r, kernel="relate_fwd (both posteriors, FOR_ALL predicates)", bytes_per_predicate=4 * N * N + 16 * N, trace=trace_id("relate_fwd_kernel", RG)))
    del tile
    # Filter / quantify on HBM-sized inputs: 65 536 predicates are 79 MB of blocks, which sit in the 256 MiB Infinity Cache
    PF = max(P, 1 << 19)
    llf = torch.log(torch.rand(PF, NS, device=device, generator=g).clamp_min(1e-5))
    prf = torch.log(torch.rand(PF, NS, device=device, generator=g).clamp_min(1e-3)) * 0.3
    pqf = torch.arange(PF, dtype=torch.int32, device=device)
    nof = torch.full((PF,), N, dtype=torch.int32, device=device)
    r = timed("dfol_filter_fwd_f32", lambda: L.filter_fwd(prf, llf, pqf, nof), PF * 12 * N)
    res.append(dict(r, kernel="filter_fwd (%.0f MB of blocks)" % (PF * 12 * N / 1e6), bytes_per_predicate=12 * N, predicates=PF,
                    trace=trace_id("filter_fwd_kernel", (PF * (NS // 4) + 1023) // 1024 * 256)))
    del llf
    PQ = max(P, 1 << 21)
    prq = torch.log(torch.rand(PQ, NS, device=device, generator=g).clamp_min(1e-3)) * 0.3
    pqq = torch.arange(PQ, dtype=torch.int32, device=device)
    noq = torch.full((PQ,), N, dtype=torch.int32, device=device)
    onq = torch.ones(PQ, device=device)
    r = timed("dfol_quantify_fwd_f32", lambda: L.quantify_fwd(prq, onq, pqq, noq), PQ * (4 * N + 4))
    lpp = 1
    while lpp < NS // 4:
        lpp *= 2
    res.append(dict(r, kernel="quantify_fwd (%.0f MB of blocks)" % (PQ * (4 * N + 4) / 1e6), bytes_per_predicate=4 * N + 4, predicates=PQ,
                    trace=trace_id("quantify_fwd4_kernel", -(-PQ // (16 * (64 // lpp))) * 256)))
    return res


def cpu_baseline(model, paths, questions, gpu_result, sample, parity_all=True, fp64=True, budget=150.0, max_questions=0):
    """The reference's algorithm on the GPU box's host cores, on a bounded sample of the same workload, and the in-run parity check.

    `cpu_baseline` = the torch-CPU restatement of the reference's flat-layout forward (oracle/dfol_oracle_torch.py: the reference's own
    operator sequence, bit-identical outputs, within 5-15 % of its wall time on equal cores - tools/time_reference.py), timed on `sample`
    questions with the thread count and ProgramBatch size that serve it best here (the reference sets torch's thread count from its
    config, trainer.py:57-62; on this 256-thread host 16-32 threads are fastest, 128+ are 2-3x slower).  For programs outside its scope
    (configs[4]) the numpy port (oracle/dfol_oracle.py) is the baseline, kind "port".
    `parity` = the GPU's log-probabilities against the numpy oracle's fp32 run and - `fp64` - its float64 run, the yardstick of the
    tolerance policy (tests/golden_util.py), on as many questions of the timed batch - at most `max_questions`, 0 = all - as fit
    the host-time `budget`."""
    from oracle import dfol_oracle as orc
    t_begin = time.perf_counter()
    ont = orc.Ontology(paths["attribute_file"], paths["class_file"], paths["vocabulary_file"], paths["relation_file"])
    weights = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if k.startswith("_featurizer.") or k.startswith("_oracle.")}
    head = questions[:sample]
    big = max(q["scene"]["n"] for q in head) > 64
    best = None
    for pb_size in (1, 4):                                  # the reference's cost is super-linear in the ProgramBatch size
        chunks = max(1, -(-sample // pb_size))
        t0 = time.perf_counter()
        r = orc.run_questions(ont, head, [q["scene"] for q in head], np.float32, split=chunks, weights=weights)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, r, pb_size)
    dt, r, split = best
    try:                                                    # threads numpy's BLAS actually runs the MLP layers on
        from threadpoolctl import threadpool_info
        cores = max([int(i.get("num_threads", 1)) for i in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count()
    base = {"value": sample / dt, "unit": "questions/s", "cores": cores, "kind": "port",
            "sample": "%d questions of the same workload (N<=%d), numpy fp32 oracle incl. full [pairs,2335] tables, "
                      "ProgramBatch size %d, %.1f s" % (sample, max(q["scene"]["n"] for q in head), split, dt),
            "vs_reference": reference_ratio_note()}
    lp_head = np.asarray(r["log_probability"], np.float64)
    try:
        from oracle import dfol_oracle_torch as orct
        ncpu = os.cpu_count() or 1
        probe = head[:min(8, sample)]
        cand = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} | {min(ncpu, 8)})
        timing = {}
        for th in cand:                                     # a short calibration of the thread count (the reference's `cpu_cores_num`)
            torch.set_num_threads(th)
            t0 = time.perf_counter()
            orct.run_questions(ont, probe, [q["scene"] for q in probe], weights, split=max(1, len(probe) // (2 if big else 5)))
            timing[th] = time.perf_counter() - t0
        th = min(timing, key=timing.get)
        torch.set_num_threads(th)
        best_t = None
        for pb_size in ((2, 4) if big else (5, 10)):
            chunks = max(1, -(-sample // pb_size))
            t0 = time.perf_counter()
            rt = orct.run_questions(ont, head, [q["scene"] for q in head], weights, split=chunks)
            dtt = time.perf_counter() - t0
            if best_t is None or dtt < best_t[0]:
                best_t = (dtt, rt, pb_size)
        dtt, rt, pbs_t = best_t
        base = {"value": sample / dtt, "unit": "questions/s", "cores": th, "kind": "port",
                "sample": "%d questions of the same workload (N<=%d), torch-CPU restatement of the reference's flat-layout forward incl. full "
                          "[pairs,2335] tables (oracle/dfol_oracle_torch.py), %d threads (best of %s on this %d-thread host), ProgramBatch size %d, %.1f s"
                          % (sample, max(q["scene"]["n"] for q in head), th, sorted(timing), ncpu, pbs_t, dtt),
                "vs_reference": restatement_ratio_note(),
                "max_abs_dlp_vs_numpy_port": float(np.abs(rt["log_probability"].astype(np.float64) - lp_head).max()),
                "numpy_port": {"value": sample / dt, "cores": cores, "program_batch_size": split}}
    except NotImplementedError:
        pass
    # ---- parity: as many questions as the host-time budget allows (fp32 on the rest, float64 on all of them)
    per_q = dt / sample
    checked = sample
    if max_questions and max_questions >= sample:
        questions = questions[:max_questions]
    if parity_all and len(questions) > sample:
        left = budget - (time.perf_counter() - t_begin)
        cost = lambda n: (n - sample) * per_q + (2.5 * n * per_q if fp64 else 0.0)
        checked = len(questions)
        while checked > sample and cost(checked) > left:
            checked -= max(1, (checked - sample) // 4)
        checked = max(sample, checked)
    lp_cpu, ans_cpu = [lp_head], list(r["answer"])
    if checked > sample:
        rest = questions[sample:checked]
        r2 = orc.run_questions(ont, rest, [q["scene"] for q in rest], np.float32, split=max(1, -(-len(rest) // split)), weights=weights)
        lp_cpu.append(np.asarray(r2["log_probability"], np.float64))
        ans_cpu += list(r2["answer"])
    lp_cpu = np.concatenate(lp_cpu)
    lp_gpu = gpu_result["log_probability"][:len(lp_cpu)].detach().cpu().numpy().astype(np.float64)      # (QUERY programs: several predicates per question)
    agree = sum(1 for a, b in zip(gpu_result["answer"][:checked], ans_cpu) if a == b)
    well = lp_cpu >= -5.0
    parity = {"questions_checked": checked, "max_abs_dp": float(np.abs(np.exp(lp_gpu) - np.exp(lp_cpu)).max()),
              "max_abs_dlp": float(np.abs(lp_gpu - lp_cpu).max()),
              "max_abs_dlp_where_lp_ge_-5": float(np.abs(lp_gpu - lp_cpu)[well].max()) if well.any() else None,
              "answers_agree": "%d/%d" % (agree, checked)}
    if fp64:
        qs_c = questions[:checked]
        r64 = orc.run_questions(ont, qs_c, [q["scene"] for q in qs_c], np.float64, split=max(1, -(-checked // split)), weights=weights)
        lp64 = np.asarray(r64["log_probability"], np.float64)
        K, p_tol, lp_tol = 2.0, 1e-6, 1e-4                       # tests/golden_util.check_logprob defaults
        region = lp64 >= -5.0
        own_lp = np.abs(lp_cpu - lp64)
        good = region & (own_lp <= lp_tol / 4)                   # demonstrably well-conditioned: the oracle's own fp32 run is within 2.5e-5 of its fp64 run
        own_p = float(np.abs(np.exp(lp_cpu) - np.exp(lp64)).max())
        got_p = float(np.abs(np.exp(lp_gpu) - np.exp(lp64)).max())
        d_good = float(np.abs(lp_gpu - lp_cpu)[good].max()) if good.any() else None
        got_lp = float(np.abs(lp_gpu - lp64)[region].max()) if region.any() else None
        own_lp_max = float(own_lp[region].max()) if region.any() else None
        parity.update({
            "yardstick": "the oracle's float64 run of the same %d questions" % checked,
            "well_conditioned_checked": int(good.sum()), "max_abs_dlp_well_conditioned": d_good,
            "max_abs_dp_vs_fp64": got_p, "max_abs_dlp_vs_fp64_where_lp_ge_-5": got_lp,
            "oracle_fp32_own_noise": {"max_abs_dp_vs_fp64": own_p, "max_abs_dlp_vs_fp64_where_lp_ge_-5": own_lp_max},
            "policy": {"K": K, "p_tol": p_tol, "lp_tol": lp_tol,
                       "rule_i_dlp_le_1e-4_on_well_conditioned": bool(d_good is None or d_good <= lp_tol),
                       "rule_ii_dp_vs_fp64_within_K_times_oracle_noise": bool(got_p <= K * own_p + p_tol),
                       "rule_iii_dlp_vs_fp64_within_K_times_oracle_noise": bool(got_lp is None or got_lp <= K * own_lp_max + lp_tol)}})
        parity["policy"]["pass"] = all(v for k, v in parity["policy"].items() if k.startswith("rule_"))
    parity["host_seconds"] = time.perf_counter() - t_begin
    return base, parity


def restatement_ratio_note():
    """How the torch-CPU restatement's wall time relates to the reference's own Python on identical inputs and cores (measured in the
    build container by tools/time_reference.py; the reference itself never travels to the GPU box)."""
    path = os.path.join(ROOT, "profiles", "reference_timing.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("restatement_summary")


def reference_ratio_note():
    """How this port's wall time relates to the reference's own Python on identical inputs (measured in the build container by
    tools/time_reference.py, recorded in profiles/reference_timing.json; the reference itself never travels to the GPU box)."""
    path = os.path.join(ROOT, "profiles", "reference_timing.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    return t.get("summary")


if __name__ == "__main__":
    main()
