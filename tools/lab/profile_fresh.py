"""Host profile (cProfile) of the fresh-programs path: collate -> create_sparse_tensors -> to_cuda -> model -> metrics, one thread, per batch.
usage: python tools/lab/profile_fresh.py [batches]"""
import cProfile
import json
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
import dfol_vqa_amd as D  # noqa: E402
from dfol_vqa_amd import synthetic as syn, training  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
args = bench.parse([])
dev = torch.device("cuda", 0)
model, ontology, paths, names = bench.build_model(args, dev, train=False)
N, B = args.objects, args.batch
kinds = ["exist", "verify_rel", "choose_attr", "and", "query_attr", "verify_attrs", "or", "choose_rel"]
cats = json.load(open(paths["attribute_file"]))
feats = torch.rand(B * N, 2054).to(dev)
feats[:, 2048:2050] *= 400.0
feats[:, 2050:2052] = feats[:, 2050:2052] * 100.0 + 5.0
feats[:, 2052], feats[:, 2053] = 640.0, 480.0
bindex = torch.arange(B, dtype=torch.int64).repeat_interleave(N)


class Collater(D.ProgramCollaterBase):
    def __init__(self):
        super(Collater, self).__init__("select", "relate", "filter", 1, ontology=ontology)

    def collate_object_features(self, questions):
        return feats, bindex

    def collate_meta_data(self, questions):
        return {"index": {}, "embedding": torch.zeros(1, 1)}


batches = [syn.full_size_questions(kinds[b % len(kinds)], B, N, N, names, cats, 5000 + b, with_scene=False) for b in range(3 * nb + 2)]
coll = Collater()
phase = {"collate": 0.0, "sparse": 0.0, "to_cuda": 0.0, "model": 0.0, "metrics": 0.0}


def one(qs):
    t = [time.perf_counter()]
    pbs = coll.collate(qs); t.append(time.perf_counter())
    for pb in pbs:
        pb.create_sparse_tensors()
    t.append(time.perf_counter())
    pbs = [pb.to_cuda(dev) for pb in pbs]; t.append(time.perf_counter())
    res = model(pbs, False); t.append(time.perf_counter())
    training.compute_evaluation_metrics(pbs, res); t.append(time.perf_counter())
    for k, a, b in zip(phase, t[:-1], t[1:]):
        phase[k] += b - a


with torch.no_grad():
    for qs in batches[:2]:
        one(qs)
    torch.cuda.synchronize()
    for rep in range(3):                                     # (the first pass meets six of the eight terminal operators for the first time)
        for k in phase:
            phase[k] = 0.0
        t0 = time.perf_counter()
        for qs in batches[2 + rep * nb:2 + (rep + 1) * nb]:   # (every pass its own batches: content never repeats)
            one(qs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("unprofiled pass %d: %.2f ms per batch; phases (ms): %s" % (rep, dt / nb * 1e3, {k: round(v / nb * 1e3, 2) for k, v in phase.items()}))
    pr = cProfile.Profile()
    pr.enable()
    for qs in batches[2:2 + nb]:
        one(qs)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative")
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((ct / nb * 1e3, tt / nb * 1e3, nc / nb, "%s:%d %s" % (os.path.basename(fn), line, name)))
rows.sort(reverse=True, key=(lambda r: r[1]) if os.environ.get("SORT") == "own" else None)
print("cumulative ms / own ms / calls per batch")
for r in rows[:int(os.environ.get("ROWS", "70"))]:
    print("%8.3f %8.3f %8.1f  %s" % r)

# the same stream with collate one batch ahead on a worker thread (bench.py's leg), at several interpreter switch intervals
from concurrent.futures import ThreadPoolExecutor


def prepare(qs):
    pbs = coll.collate(qs)
    for pb in pbs:
        pb.create_sparse_tensors()
    return pbs


with torch.no_grad():
    for si in (5e-3, 1e-3, 2e-4, 5e-5):
        sys.setswitchinterval(si)
        for rep in range(2):
            mine = batches[2 + rep * nb:2 + (rep + 1) * nb]
            with ThreadPoolExecutor(1) as ex:
                t0 = time.perf_counter()
                nxt = ex.submit(prepare, mine[0])
                for i in range(nb):
                    pbs = nxt.result()
                    if i + 1 < nb:
                        nxt = ex.submit(prepare, mine[i + 1])
                    pbs = [pb.to_cuda(dev) for pb in pbs]
                    res = model(pbs, False)
                    training.compute_evaluation_metrics(pbs, res)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            print("threaded, switch interval %.0e s, pass %d: %.2f ms per batch" % (si, rep, dt / nb * 1e3))
