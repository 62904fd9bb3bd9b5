"""Lab: a train step on batches whose programs differ in length (select -> 1..3 filter / relate hops -> terminal: what collate pads with no-op tokens,
the normal case of GQA program files) against the bench's uniform three-hop program, eager, 256 questions x N objects; routes taken and ms per step."""
import os, sys, time, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from dfol_vqa_amd import experiment, training, synthetic as syn, _lib
from test_interpreter_gpu import TableCollater
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = 256
paths, names = syn.write_synthetic_ontology(tempfile.mkdtemp())
cfg = syn.reference_config(paths, freeze_featurizer=False, freeze_attribute_network=False, freeze_relation_network=False, freeze_embedding_network=False)
ont = experiment.build_ontology(cfg)
model = experiment.build_model(cfg, ont)
syn.load_seeded_weights(model, 19)
model = model.to("cuda").train()
cats = json.load(open(paths["attribute_file"]))
nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
def uniform():
    qs = []
    for i in range(B):
        br, last = syn.three_hop_program(i, nouns, attrs, rels)
        qs.append(syn.question(i, br, last, "yes" if i % 2 else "no", syn.feature_scene(i, N, 2048)))
    return qs
def ragged(kind, seed):
    qs = syn.full_size_questions(kind, B, N, N, names, cats, seed)
    for i, q in enumerate(qs):
        q["answer"] = "yes" if i % 2 else "no"
    return qs
for name, qs in (("uniform three-hop exist", uniform()), ("ragged 1..3 hops exist", ragged("exist", 71)), ("ragged 1..3 hops verify_rel", ragged("verify_rel", 72))):
    pbs = [pb.to_cuda("cuda") for pb in TableCollater(1, ont, "X").collate(qs)]
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    for _ in range(2):
        training.train_batch(model, opt, pbs, clip_norm=0.65)
    torch.cuda.synchronize()
    _lib.PATH_COUNTS.clear()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(5):
        training.train_batch(model, opt, pbs, clip_norm=0.65, sync_loss=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    r = {k: v // 5 for k, v in _lib.PATH_COUNTS.items() if k in ("pair_trunk", "head_use", "pair_second_evaluation", "logit_rows_gathered", "fused_hidden1", "pair_forward_fused") or k.startswith("fallback")}
    print("%-30s %.2f ms per eager step, peak %.1f GB, routes per step %s" % (name, ms, torch.cuda.max_memory_allocated() / 1e9, r))
    if os.environ.get("RAGGED_GRAPH", "1") == "1":              # the same step replayed as a HIP graph (what bench.py --mode train times)
        from dfol_vqa_amd import parallel
        params = [p for p in model.parameters() if p.requires_grad]
        gopt = torch.optim.Adam(params, lr=1e-4, capturable=True)
        gstep = training.GraphedTrainStep(model, gopt, pbs, 0.65, bucket=parallel.GradBucket(params), warmup=1)
        for _ in range(3):
            gstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            gstep()
        torch.cuda.synchronize()
        print("    replayed as a graph: %.2f ms per step" % ((time.perf_counter() - t0) / 10 * 1e3))
        del gstep
    _lib.enable_kernel_timing(list(_lib.SIGNATURES))
    for _ in range(3):
        training.train_batch(model, opt, pbs, clip_norm=0.65, sync_loss=False)
    torch.cuda.synchronize()
    tm = _lib.disable_kernel_timing()
    top = sorted(((t / 3 * 1e3, n // 3, k) for k, (n, t) in tm.items() if n), reverse=True)[:12]
    print("    library launches per step %d, %.2f ms: %s" % (sum(n for _, n, _ in top), sum(t for t, _, _ in top), ", ".join("%s x%d %.2f" % (k.replace("dfol_", "").replace("_f32", ""), n, t) for t, n, k in top)))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        training.train_batch(model, opt, pbs, clip_norm=0.65, sync_loss=False)
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:10]
    print("    device time by kernel (one step): " + "; ".join("%s x%d %.2f ms" % (e.key[:38], e.count, e.device_time_total / 1e3) for e in rows))
