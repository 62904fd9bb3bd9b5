"""Lab: soak of the executor on two alternating streams (the loop shape of bench.py's unseen-batch legs): B distinct batches of mixed programs, each
forwarded once on its own for reference, then R rounds over all of them alternating two streams with two pending - every result compared bit for
bit with its reference.  Optional: calibrated model (second argument "calib")."""
import os, sys, time, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import dfol_vqa_amd as D
from dfol_vqa_amd import experiment, synthetic as syn, _lib
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calib = len(sys.argv) > 2 and sys.argv[2] == "calib"
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
paths, names = syn.write_synthetic_ontology(tmp)
cfg = syn.reference_config(paths, activate_attention_transfer=calib)
ont = experiment.build_ontology(cfg)
model = experiment.build_model(cfg, ont)
syn.load_seeded_weights(model, 23)
if calib:
    syn.load_seeded_calibrator(model, 5)
model = model.to(dev).eval()
cats = json.load(open(paths["attribute_file"]))
voc = list(ont._vocabulary["idx_to_arg"])
emb = torch.randn(len(voc), 300, generator=torch.Generator().manual_seed(3)) * 0.1
index = {t: i for i, t in enumerate(voc)}
class Coll(D.ProgramCollaterBase):
    def __init__(self):
        super(Coll, self).__init__("select", "relate", "filter", 1, ontology=ont)
    def collate_object_features(self, qs):
        return torch.cat([torch.from_numpy(q["scene"]["X"]) for q in qs], 0), torch.cat([torch.full((q["scene"]["n"],), i, dtype=torch.int64) for i, q in enumerate(qs)])
    def collate_meta_data(self, qs):
        return {"index": index, "embedding": emb}
kinds = ["exist", "verify_rel", "choose_attr", "and", "query_attr", "verify_attrs", "or", "choose_rel"]
batches, want = [], []
os.environ["DFOL_NATIVE"] = "1"
with torch.no_grad():
    for b in range(24):
        qs = syn.full_size_questions(kinds[b % 8], 48, 20, 60, names, cats, 8100 + b)
        pbs = [pb.to_cuda(dev) for pb in Coll().collate(qs)]
        r = model(pbs, False)
        batches.append(pbs)
        want.append((r["log_probability"].clone(), r["answer"], r["answer_log_probability"]))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    _lib.PATH_COUNTS.clear()
    bad = n = 0
    t0 = time.perf_counter()
    pend = []
    def take(k, p):
        global bad
        r = p.result()
        w = want[k]
        bad += int(not (torch.equal(r["log_probability"], w[0]) and r["answer"] == w[1] and r["answer_log_probability"] == w[2]))
    for rnd in range(rounds):
        for k, pbs in enumerate(batches):
            with torch.cuda.stream(streams[n % 2]):
                pend.append((k, model.forward_async(pbs, False)))
            n += 1
            if len(pend) > 2:
                take(*pend.pop(0))
    for k, p in pend:
        take(k, p)
    torch.cuda.synchronize()
print("soak: %d forwards of 24 mixed batches (48 questions x 20..60 objects%s) on two alternating streams, two pending, %.1f s; results differing from the "
      "batch's own serial forward: %d; routes %s" % (n, ", calibrator on" if calib else "", time.perf_counter() - t0, bad, {k: v for k, v in _lib.PATH_COUNTS.items() if "program" in k}))
