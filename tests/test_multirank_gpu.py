"""world_size-2 tests of the data-parallel path on the GPU: two fresh child processes share cuda:0 and meet over gloo
(reference: data_parallel.py:54-83 scatter / replicate / gather, trainer.py:429-442 the train step).  Also drives bench.py's own
launcher, which must start the ranks itself when no launcher did."""

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(case, tmp_path, world=2, timeout=420, extra_env=None):
    port = _free_port()
    procs, paths = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", **(extra_env or {}))
        path = str(tmp_path / ("%s_rank%d.json" % (case, r)))
        paths.append(path)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_rank_worker.py"), case, path], env=env, cwd=ROOT))
    try:
        for p in procs:
            p.wait(timeout=timeout)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    outs = []
    for path in paths:
        with open(path) as f:
            outs.append(json.load(f))
    for o in outs:
        assert o["ok"], o.get("error")
    return outs


def test_two_rank_inference_equals_one_rank(tmp_path):
    outs = _run_ranks("infer", tmp_path)
    r0 = [o for o in outs if o["rank"] == 0][0]
    assert r0["gathered"] == 24
    assert r0["bit_equal"], r0["max_abs_diff"]
    assert r0["answers_equal"]


@pytest.mark.parametrize("overlap,side_stream", [(1, 0), (0, 0), (1, 1)])
def test_two_rank_train_step_matches_g12_and_replicas_stay_equal(tmp_path, overlap, side_stream):
    """overlap = 1: ranges of the gradient bucket all-reduced from autograd hooks during the backward; 0: one all-reduce after it;
    side_stream = 1: forward and backward run on a non-default stream (the collectives must be ordered after THAT stream)."""
    outs = _run_ranks("train", tmp_path, extra_env={"DFOL_TEST_OVERLAP": str(overlap), "DFOL_TEST_SIDE_STREAM": str(side_stream)})
    r0 = [o for o in outs if o["rank"] == 0][0]
    assert r0["g12_checked"] and all(n == 12 for n in r0["g12_checked"].values()), r0["g12_checked"]
    assert all(o["replicas_equal"] for o in outs)


@pytest.mark.parametrize("mode", ["infer", "train"])
def test_bench_launches_its_own_ranks(mode):
    """`python bench.py --gpus 2` with no launcher around it: the script starts two ranks (DFOL_BENCH_SHARE_GPU=1: both on cuda:0, gloo)."""
    env = dict(os.environ, DFOL_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16", "--objects", "20",
           "--cpu-sample", "0", "--stress-preds", "0", "--sustain", "0.2", "--fresh-batches", "3", "--mode", mode]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 32 and out["value"] > 0
    assert out["ranks"]["backend"] == "gloo" and [r["rank"] for r in out["ranks"]["ranks_seen"]] == [0, 1]
    assert out["ranks"]["rank_ms_per_step_min"] > 0
    # every rank reports the threads / cores its Python host runs on, and (inference) its own rate over unseen batches
    assert all("OMP_NUM_THREADS" in r["host"] and r["host"]["cpu_affinity_count"] >= 1 for r in out["ranks"]["ranks_seen"])
    if mode == "infer":
        for leg in ("fresh_programs", "end_to_end"):
            per = out[leg]["per_rank"]
            assert [p["rank"] for p in per] == [0, 1] and all(p["questions_per_s"] > 0 for p in per), (leg, per)
        assert out["value_fresh_programs"] > 0 and out["value_end_to_end"] > 0
    if mode == "train":
        assert out["replicas_equal"] is True
        assert out["allreduce_ms"] > 0


@pytest.mark.parametrize("case,overlap", [("infer", 0), ("train", 0), ("train", 1)])
def test_rccl_world_size_one(tmp_path, case, overlap):
    """The same rank worker over backend "nccl" (= RCCL) with ONE rank: what a one-GPU box can run of the RCCL path - communicator
    setup, broadcast of the parameters, the bucket all-reduce (after the backward, and as three ranges issued from the autograd hooks
    on RCCL's own stream), all-gather of the results and of the parameter digests - with the same g12 / bit-equality checks."""
    outs = _run_ranks(case, tmp_path, world=1, extra_env={"DFOL_TEST_BACKEND": "nccl", "DFOL_TEST_OVERLAP": str(overlap)})
    assert outs[0]["backend"] == "nccl"
    if case == "infer":
        assert outs[0]["gathered"] == 12 and outs[0]["bit_equal"] and outs[0]["answers_equal"]
    else:
        assert all(n == 12 for n in outs[0]["g12_checked"].values()) and outs[0]["replicas_equal"]


@pytest.mark.parametrize("mode,overlap", [("infer", 1), ("train", 1), ("train", 0)])
def test_bench_over_rccl_world_size_one(mode, overlap):
    """bench.py with DFOL_BENCH_FORCE_PG=1: the N > 1 code path (barriers, max over ranks, rank report, bucket all-reduce, replica check)
    over RCCL with one rank."""
    env = dict(os.environ, DFOL_BENCH_FORCE_PG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "16", "--objects", "20",
           "--cpu-sample", "0", "--stress-preds", "0", "--sustain", "0.2", "--fresh-batches", "3", "--mode", mode, "--overlap-allreduce", str(overlap)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-500:]       # stdout carries the JSON line only (RCCL's own prints go to stderr)
    out = json.loads(lines[0])
    assert out["ranks"]["backend"] == "nccl" and out["value"] > 0
    if mode == "train":
        assert out["replicas_equal"] is True and out["allreduce_ms"] > 0
        # without the overlap hooks the step is two HIP graphs with RCCL's all-reduce issued eagerly between the replays (small scenes: the
        # replay is the faster); never the captured-collective form unless asked for
        assert ("graph" in out["config"]["launch"]) == (overlap == 0), out["config"]["launch"]
        assert "captured inside" not in out["config"]["launch"]


@pytest.mark.parametrize("backend,world", [("nccl", 1), ("gloo", 2)])
def test_graphed_train_step_with_process_group(tmp_path, backend, world):
    """The step graphs with data parallelism, robust form (the default): graph A = zero + forward + loss + backward, graph B = clip + Adam,
    the gradient bucket's all-reduce issued EAGERLY between the two replays - one rank over RCCL and two ranks over gloo (sharing
    cuda:0): three replays == three eager steps with the same group, bit for bit, replicas equal."""
    outs = _run_ranks("graph", tmp_path, world=world, extra_env={"DFOL_TEST_BACKEND": backend})
    assert all(o["backend"] == backend and o["graph_equals_eager"] and o["graphs"] == 2 for o in outs)


def test_graphed_train_step_with_captured_collective(tmp_path):
    """The opt-in single-graph form (graph_collective=True: RCCL's all-reduce is a node of the step graph).  It ran green on some boxes and
    was aborted by the process group's watchdog thread (SIGABRT - nothing in-process can catch it) on another in round 3, which is why it
    is opt-in and why this test runs it in a child process: an abort of the child is reported as an expected failure of the opt-in
    path; a child that completes must equal the eager steps bit for bit."""
    try:
        outs = _run_ranks("graph", tmp_path, world=1, extra_env={"DFOL_TEST_BACKEND": "nccl", "DFOL_TEST_GRAPH_COLLECTIVE": "1"})
    except (FileNotFoundError, ValueError):                      # the child died before it could write its findings
        pytest.xfail("the captured-collective step graph took its process down on this box (opt-in path; the default is the two-graph form)")
    assert outs[0]["backend"] == "nccl" and outs[0]["graph_equals_eager"] and outs[0]["graphs"] == 1


def test_bench_shards_ragged_scenes_by_cost():
    """`--workload c3` over two ranks (both on cuda:0, gloo): the global question list is cut into contiguous shards of whole images balanced
    by the sum of n^2 (parallel.shard_bounds), the line reports every rank's question count, cost and the imbalance, and the total is still
    world x batch questions per step."""
    env = dict(os.environ, DFOL_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c3", "--steps", "2", "--warmup", "1", "--batch", "32", "--objects", "40",
           "--questions-per-image", "4", "--cpu-sample", "0", "--stress-preds", "0", "--sustain", "0.2", "--streamed", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    sh = out["shard"]
    assert sum(sh["questions_per_rank"]) == 64 and len(sh["questions_per_rank"]) == 2 and all(q % 4 == 0 for q in sh["questions_per_rank"])
    assert 1.0 <= sh["imbalance_max_over_mean"] < 1.35, sh
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and out["value"] > 0


def _gpu_count():
    import torch
    return torch.cuda.device_count()          # (counting devices does not initialise the GPU in this process)


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: runs wherever a node has them (the one-GPU test box skips it)")
@pytest.mark.parametrize("mode", ["infer", "train"])
def test_bench_over_rccl_two_gpus(mode):
    """`python bench.py --gpus 2` over RCCL, one GPU per rank (the driver's N > 1 launch path, self-launched here): the rank report must show two
    distinct devices (bench.py refuses the line otherwise), the train replicas must stay equal after the all-reduced steps, and - train - the
    step is the two-graph form with the all-reduce issued eagerly between the replays."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DFOL_BENCH_SHARE_GPU", "DFOL_BENCH_FORCE_PG"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "16", "--objects", "20",
           "--cpu-sample", "0", "--stress-preds", "0", "--sustain", "0.2", "--fresh-batches", "2", "--mode", mode]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["ranks"]["backend"] == "nccl" and out["ranks"]["distinct_devices"] == 2
    assert out["config"]["global_batch"] == 32 and out["value"] > 0
    if mode == "train":
        assert out["replicas_equal"] is True and out["allreduce_ms"] > 0
        assert "two hip graph replays" in out["config"]["launch"] or out["config"]["launch"] == "eager"


def test_bench_on_files_in_the_reference_formats(tmp_path):
    """`bench.py --workload c3 --data <dir>`: BASELINE configs[2] from files on disk - here the g18 fixtures (program bytecode .h5 files the
    reference's encoder wrote, object-feature chunk .h5 files) laid out the way the flag expects; real GQA testdev files take the same path
    when they are on the box.  The line says where its data came from, and its parity leg checked the oracle on every program file."""
    import shutil
    golden = os.path.join(HERE, "golden")
    d = tmp_path / "gqa"
    for sub in ("metadata", "programs", "objects"):
        (d / sub).mkdir(parents=True)
    for f in ("attribute.json", "class.json", "relation.json", "vocab.json", "glove.txt"):
        shutil.copy(os.path.join(golden, "mini_ontology", f), str(d / "metadata" / f))
    for f in os.listdir(os.path.join(golden, "h5")):
        if f.startswith("g18_objects"):
            shutil.copy(os.path.join(golden, "h5", f), str(d / "objects" / f))
        elif f.startswith("g18_"):
            shutil.copy(os.path.join(golden, "h5", f), str(d / "programs" / f))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3", "--data", str(d), "--batch", "4", "--steps", "16", "--warmup", "2"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["data"].startswith("files: ") and "g18_exist.h5 6" in out["config"]["workload"] and out["config"]["batches"] == 16
    assert out["value"] > 0 and out["parity"]["policy"]["pass"] and len(out["parity"]["program_files_checked"]) == 8
    assert out["parity"]["questions_checked"] == 32 and out["cpu_baseline"]["value"] > 0


@pytest.mark.parametrize("mode", ["infer", "train", "c3"])
def test_bench_eight_ranks_share_one_gpu(mode):
    """8-GPU readiness without 8 GPUs (no scaling curve exists - SCALE was skipped every round): `bench.py --gpus 8` with its
    own launcher, eight ranks on cuda:0 over gloo - inference, the train step and the ragged c3 workload.  Eight rank reports, every rank its
    own OMP_NUM_THREADS share and a capped number of collate workers (8 x 6 worker processes would oversubscribe the host), replicas equal
    after the all-reduced train steps, question shards balanced by the sum of n^2 within 5 %."""
    env = dict(os.environ, DFOL_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMP_NUM_THREADS"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--stress-preds", "0",
           "--sustain", "0.2"]
    if mode == "c3":
        cmd += ["--workload", "c3", "--batch", "64", "--objects", "40", "--questions-per-image", "1", "--streamed", "0"]
    else:
        cmd += ["--batch", "16", "--objects", "20", "--fresh-batches", "3", "--mode", mode]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    out = json.loads(lines[0])
    ranks = out["ranks"]["ranks_seen"]
    assert out["n_gpus"] == 8 and out["ranks"]["backend"] == "gloo" and [x["rank"] for x in ranks] == list(range(8)) and out["value"] > 0
    cores = os.cpu_count() or 8
    for x in ranks:
        assert int(x["host"]["OMP_NUM_THREADS"]) == max(1, cores // 8), x["host"]
    if mode == "infer":
        assert out["config"]["global_batch"] == 128
        per = out["fresh_programs"]["per_rank"]
        assert [p["rank"] for p in per] == list(range(8)) and all(p["questions_per_s"] > 0 for p in per)
        assert out["fresh_programs"]["collate_workers"] <= max(1, cores // 8 - 1)
        assert out["config"]["legs"]["fresh"] > 0 and out["config"]["legs"]["native_batches"] > 0
    elif mode == "train":
        assert out["replicas_equal"] is True and out["allreduce_ms"] > 0 and out["config"]["global_batch"] == 128
    else:
        sh = out["shard"]
        assert len(sh["questions_per_rank"]) == 8 and sum(sh["questions_per_rank"]) == 512
        assert 1.0 <= sh["imbalance_max_over_mean"] <= 1.05, sh
