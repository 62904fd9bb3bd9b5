"""One rank of the world_size-2 GPU tests (started by tests/test_multirank_gpu.py as a fresh child process; not a test module).

Every rank uses cuda:0 (the GPU box has one GPU) and the ranks meet over gloo; on an 8-GPU node the same code runs with
backend "nccl" (= RCCL) and one GPU per rank.  Writes its findings as JSON to the path in argv[2]."""

import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def infer_case(rank, world, out):
    """Sharded inference == single-process inference, bit for bit, on the same question ids (SURVEY.md 8(e) split invariance)."""
    import bench
    from dfol_vqa_amd import parallel
    args = bench.parse(["--batch", "12", "--objects", "20", "--gpus", str(world)])
    dev = torch.device("cuda", 0)
    model, ontology, paths, names = bench.build_model(args, dev)
    if rank == 1:                                           # replicas that start different must be made equal by the broadcast
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01)
    parallel.broadcast_parameters(model, 0)
    qs_mine, pbs = bench.build_batch(args, rank, ontology, names, dev)     # question ids rank*12 .. rank*12+11
    with torch.no_grad():
        res = model(pbs, False)
    allres = parallel.gather_results(res)
    out["gathered"] = len(allres["answer"])
    if rank == 0:
        args1 = bench.parse(["--batch", str(12 * world), "--objects", "20"])
        _, pbs1 = bench.build_batch(args1, 0, ontology, names, dev)         # the same ids 0 .. 12*world-1 in one process
        with torch.no_grad():
            one = model(pbs1, False)
        a, b = allres["log_probability"].numpy(), one["log_probability"].cpu().numpy()
        out["bit_equal"] = bool(np.array_equal(a, b))
        out["max_abs_diff"] = float(np.abs(a - b).max())
        out["answers_equal"] = allres["answer"] == one["answer"]


def train_case(rank, world, out):
    """A 2-rank train step on golden g12's questions: all-reduced weight gradients == the reference's own autograd (g12), and the
    replicas hold identical parameters after clip + Adam."""
    import golden_util as gu
    import dfol_vqa_amd as D
    from dfol_vqa_amd import parallel, training
    from test_interpreter_gpu import DEV, TableCollater, neural_model
    from test_backward_gpu import grad_close
    from conftest import GOLDEN
    d = os.path.join(GOLDEN, "mini_ontology")
    ontology = D.GQAOntology(os.path.join(d, "attribute.json"), os.path.join(d, "class.json"), os.path.join(d, "vocab.json"),
                             os.path.join(d, "glove.txt"), relation_json_path=os.path.join(d, "relation.json"))
    a, meta = gu.load("g12_weight_gradients")
    weights = {k[2:]: a[k] for k in a.files if k.startswith("w:")}
    checked, failures = {}, []
    for name in sorted(meta["sets"]):
        model = neural_model(ontology, meta["config"], weights).train()
        parallel.broadcast_parameters(model, 0)
        qs = [{"program": q["program"], "answer": q["answer"], "question_id": q["question_id"], "image_id": "img000", "tokens": [],
               "original_dict": None, "question": None, "scene": {"n": q["n"], "X": a["%s:X_%d" % (name, i)]}}
              for i, q in enumerate(meta["sets"][name]["questions"])]
        s, e = parallel.shard_bounds([1.0] * len(qs), world)[rank]
        mine = qs[s:e]
        pbs = [pb.to_cuda(DEV) for pb in TableCollater(1, ontology, "X").collate(mine)]
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.Adam(params, lr=1e-3)
        bucket = parallel.GradBucket(params)
        if os.environ.get("DFOL_TEST_OVERLAP", "1") == "1":
            bucket.enable_overlap(dist.group.WORLD, segments=3)              # ranges all-reduced during the backward
        bucket.zero_()
        if os.environ.get("DFOL_TEST_SIDE_STREAM") == "1":
            # forward + backward on a SIDE stream: the hooks then issue the range collectives from the autograd thread with that stream
            # current, and the bucket must order them after it (GradBucket._launch_ready) and the caller's stream after them
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                res = model(pbs, True)
                loss = training.compute_loss(pbs, res) / len(qs)
                loss.backward()
                bucket.allreduce(dist.group.WORLD)
            torch.cuda.current_stream().wait_stream(side)
        else:
            res = model(pbs, True)
            loss = training.compute_loss(pbs, res) / len(qs)                   # sum / B_global (trainer.py:433-436)
            loss.backward()
            bucket.allreduce(dist.group.WORLD)
        lt = torch.tensor([float(loss.detach())], dtype=torch.float64, device=DEV if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(lt)
        if rank == 0:                                        # (failures are collected, not raised: the peer is waiting in a collective)
            l32, l64 = float(a[name + ":loss_f32"]), float(a[name + ":loss_f64"])
            if not abs(lt.item() - l64) <= 8 * abs(l32 - l64) + 2e-5 * max(1.0, abs(l64)):
                failures.append("%s loss %r vs %r / %r" % (name, lt.item(), l32, l64))
            n = 0
            for pname, prm in model.named_parameters():
                key = "%s:g:%s:f64" % (name, pname)
                if key in a.files:
                    try:
                        grad_close(prm.grad.detach().cpu().numpy(), a[key[:-3] + "f32"], a[key], "%s d%s (2 ranks)" % (name, pname))
                    except AssertionError as exc:
                        failures.append(str(exc))
                    n += 1
            checked[name] = n
        # the rest of the step through the product's train_batch, then compare replicas
        loss2, _ = training.train_batch(model, opt, pbs, 0.65, global_batch_size=len(qs), group=dist.group.WORLD, bucket=bucket, l1_lambda=1e-3)
        dg = parallel.parameters_digest(model)
        dg = dg if dist.get_backend() == "nccl" else dg.cpu()
        both = [torch.zeros_like(dg) for _ in range(world)]
        dist.all_gather(both, dg)
        if not all(torch.equal(both[0], b) for b in both):
            failures.append("%s: replicas differ after the step: %r" % (name, both))
    out["g12_checked"] = checked
    out["replicas_equal"] = not any("replicas" in f for f in failures)
    assert not failures, failures


def graph_case(rank, world, out):
    """training.GraphedTrainStep with a process group: by default two graphs with the bucket's all-reduce issued eagerly between the replays
    (any backend), with DFOL_TEST_GRAPH_COLLECTIVE=1 one graph with RCCL's all-reduce captured inside; three replays == three eager
    train_batch steps with the same group (losses and parameters bit for bit on the full-size model)."""
    import bench
    from dfol_vqa_amd import parallel, training
    dev = torch.device("cuda", 0)
    finals = []
    for graphed in (False, True):
        args = bench.parse(["--mode", "train", "--objects", "16", "--batch", "8", "--gpus", str(world)])
        torch.manual_seed(11)
        model, ontology, paths, names = bench.build_model(args, dev, train=True)
        parallel.broadcast_parameters(model, 0)
        _, pbs = bench.build_batch(args, rank, ontology, names, dev)
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
        bucket = parallel.GradBucket(params)
        gb = 8 * world
        if graphed:
            step = training.GraphedTrainStep(model, opt, pbs, 0.65, bucket=bucket, warmup=1, group=dist.group.WORLD, global_batch_size=gb,
                                             graph_collective=os.environ.get("DFOL_TEST_GRAPH_COLLECTIVE") == "1")
            out["graphs"] = 1 if step._graph_b is None else 2
            losses = [float(step()[0]) for _ in range(3)]
        else:
            losses = [float(training.train_batch(model, opt, pbs, 0.65, global_batch_size=gb, group=dist.group.WORLD, bucket=bucket, sync_loss=False)[0])
                      for _ in range(4)][1:]
        finals.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
        dg = parallel.parameters_digest(model)
        dg = dg if dist.get_backend() == "nccl" else dg.cpu()
        both = [torch.zeros_like(dg) for _ in range(world)]
        dist.all_gather(both, dg)
        assert all(torch.equal(both[0], b) for b in both), "replicas differ"
    (l0, s0), (l1, s1) = finals
    out["losses"] = [l0, l1]
    assert l0 == l1, (l0, l1)
    bad = [k for k in s0 if not torch.equal(s0[k], s1[k])]
    assert not bad, bad
    out["graph_equals_eager"] = True


def main():
    case, path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    import datetime
    if os.environ.get("DFOL_TEST_BACKEND") == "nccl":        # world size 1 on the one-GPU box: the collectives run through RCCL itself
        assert world == 1
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=180))
    else:
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=180))     # a rank that died must not hang its peer for 30 minutes
    out_backend = dist.get_backend()
    out = {"rank": rank, "ok": False, "backend": out_backend}
    try:
        {"infer": infer_case, "train": train_case, "graph": graph_case}[case](rank, world, out)
        out["ok"] = True
    except Exception as exc:  # pragma: no cover
        import traceback
        out["error"] = "%r\n%s" % (exc, traceback.format_exc())
    finally:
        with open(path, "w") as f:
            json.dump(out, f)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
