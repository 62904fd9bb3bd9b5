"""GPU-box host timing of the torch-CPU restatement vs thread count (why did bench.py's cpu_baseline leg take > 10 min?)."""
import os, sys, time, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dfol_vqa_amd import synthetic as syn
from oracle import dfol_oracle as orc, dfol_oracle_torch as orct
N, Qn = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 16
paths, names = syn.write_synthetic_ontology(tempfile.mkdtemp())
ont = orc.Ontology(paths["attribute_file"], paths["class_file"], paths["vocabulary_file"], paths["relation_file"])
w = syn.seeded_weights(3)
nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
qs = []
for i in range(Qn):
    br, last = syn.three_hop_program(i, nouns, attrs, rels)
    qs.append(syn.question(i, br, last, "yes", syn.feature_scene(i, N, 2048)))
print("cpu_count", os.cpu_count(), "torch threads default", torch.get_num_threads(), flush=True)
for th in (8, 16, 32, 64, 128):
    if th > (os.cpu_count() or 8):
        continue
    torch.set_num_threads(th)
    for size in (2, 4):
        t0 = time.perf_counter()
        orct.run_questions(ont, qs, [q["scene"] for q in qs], w, split=max(1, Qn // size))
        print("threads", th, "ProgramBatch", size, "%.2f s -> %.2f q/s" % (time.perf_counter() - t0, Qn / (time.perf_counter() - t0)), flush=True)
t0 = time.perf_counter()
orc.run_questions(ont, qs, [q["scene"] for q in qs], np.float32, split=Qn // 4, weights=w)
print("numpy port ProgramBatch 4: %.2f s" % (time.perf_counter() - t0), flush=True)
torch.set_num_threads(32)
t0 = time.perf_counter()
orct.run_questions(ont, qs, [q["scene"] for q in qs], w, split=Qn // 2)
print("restatement again after numpy (32 threads): %.2f s" % (time.perf_counter() - t0), flush=True)
