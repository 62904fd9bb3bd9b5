#!/usr/bin/env python3
"""Forward throughput with the attention calibrator on (config/sample_config.yaml's default `activate_attention_transfer: True`):
the LSTM passes over the program plus apply_modulations around every operator, eager and as a replayed HIP graph.

usage: python tools/bench_calibrated.py [objects]
"""
import os, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import dfol_vqa_amd as D
from dfol_vqa_amd import experiment
from dfol_vqa_amd import synthetic as syn
class A: pass
args = A(); args.objects = int(sys.argv[1]) if len(sys.argv) > 1 else 100; args.batch = 256
MIXED = len(sys.argv) > 2 and sys.argv[2].startswith("mixed")   # fresh batches: 1..3-hop programs, eight terminal operators in turn (default: the bench's
ONLY = sys.argv[2].split(":")[1] if MIXED and ":" in sys.argv[2] else None      # program); mixed:<kind>: that terminal operator only
device = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
paths, names = syn.write_synthetic_ontology(tmp)
for calib in (False, True):
    cfg = syn.reference_config(paths, activate_attention_transfer=calib)
    ontology = experiment.build_ontology(cfg)
    model = experiment.build_model(cfg, ontology); bench.init_weights(model); model = model.to(device).eval()
    voc = list(ontology._vocabulary["idx_to_arg"])
    emb = torch.randn(len(voc), 300) * 0.1
    class Collater(D.ProgramCollaterBase):
        def __init__(self): super(Collater, self).__init__("select", "relate", "filter", 1, ontology=ontology)
        def collate_object_features(self, qs):
            return torch.cat([torch.from_numpy(q["scene"]["X"]) for q in qs], 0), torch.cat([torch.full((q["scene"]["n"],), i, dtype=torch.int64) for i, q in enumerate(qs)])
        def collate_meta_data(self, qs): return {"index": {t: i for i, t in enumerate(voc)}, "embedding": emb}
    nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
    qs = []
    for i in range(args.batch):
        br, last = syn.three_hop_program(i, nouns, attrs, rels)
        qs.append(syn.question(i, br, last, "yes", syn.feature_scene(i, args.objects, 2048)))
    pbs = Collater().collate(qs)
    for pb in pbs: pb.create_sparse_tensors()
    pbs = [pb.to_cuda(device) for pb in pbs]
    with torch.no_grad():
        for _ in range(3): model(pbs, False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): model(pbs, False)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("calibration", calib, "ms/step %.3f" % (dt * 1e3), "q/s %.0f" % (args.batch / dt))
    from dfol_vqa_amd.interpreter import GraphedForward
    with torch.no_grad():
        eager = model(pbs, False)
    g = GraphedForward(model, pbs)
    r = g()
    assert torch.equal(r["log_probability"], eager["log_probability"]) and r["answer"] == eager["answer"]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("calibration", calib, "graph replay ms/step %.3f" % (dt * 1e3), "q/s %.0f" % (args.batch / dt))
    replay_ms = dt * 1e3
    # fresh programs: every step ANOTHER ProgramBatch (plans lowered at collate time, as a DataLoader worker would; features device-resident),
    # two batches in flight (forward_async) - the native executor against the Python operator loop
    from dfol_vqa_amd import native_exec, _lib
    spec = native_exec.model_spec(model, calibrate=calib)
    class FreshCollater(Collater):
        def __init__(self, spec_):
            super(FreshCollater, self).__init__()
            self._native_spec = spec_
    for route in ("1", "0"):
        os.environ["DFOL_NATIVE"] = route
        fresh = []
        for k in range(24):
            qk = []
            if MIXED:                                           # programs of differing lengths and terminal operators (what GQA's files hold)
                import json as _json
                cats_ = _json.load(open(paths["attribute_file"]))
                kinds_ = [ONLY] if ONLY else ["exist", "verify_rel", "choose_attr", "and", "query_attr", "verify_attrs", "or", "choose_rel"]
                qk = syn.full_size_questions(kinds_[k % len(kinds_)], args.batch, args.objects, args.objects, names, cats_, 5000 + k, with_scene=False)
                for i, q in enumerate(qk):
                    q["scene"] = qs[i]["scene"]
            for i in range(0 if MIXED else args.batch):
                br, last = syn.three_hop_program(100000 * (k + 1) + i, nouns, attrs, rels)
                qk.append(syn.question(100000 * (k + 1) + i, br, last, "yes", qs[i]["scene"]))
            pb = FreshCollater(spec if route == "1" else None).collate(qk)[0]
            pb.create_sparse_tensors()
            pb = pb.to_cuda(device)
            pb._object_features = pbs[0]._object_features          # (the same resident features: the leg times programs, not uploads)
            fresh.append([pb])
        _lib.PATH_COUNTS.clear()
        with torch.no_grad():
            for pbk in fresh[:4]:
                model(pbk, False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pend = []
            import contextlib
            lanes_ = [torch.cuda.Stream(device=device) for _ in range(2)] if (route == "1" and os.environ.get("CALIB_STREAMS", "1") == "2") else None
            if lanes_:                                          # CALIB_STREAMS=2: the executor's batches alternate two streams (bench.py's loops)
                for s_ in lanes_:
                    s_.wait_stream(torch.cuda.current_stream(device))
            for j, pbk in enumerate(fresh[4:]):
                with (torch.cuda.stream(lanes_[j % 2]) if lanes_ else contextlib.nullcontext()):
                    pend.append(model.forward_async(pbk, False))
                if len(pend) > 2:
                    pend.pop(0).result()
            for x in pend:
                x.result()
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (len(fresh) - 4)
        print("calibration", calib, "fresh programs, %s: ms/batch %.3f" % ("native executor" if route == "1" else "Python loop", dt * 1e3), "q/s %.0f" % (args.batch / dt),
              "= %.2f x the graph-replay rate" % (replay_ms / (dt * 1e3)), dict(_lib.PATH_COUNTS))
    os.environ["DFOL_NATIVE"] = "1"
