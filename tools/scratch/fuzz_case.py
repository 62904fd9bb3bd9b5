# Scratch: one fuzz case of tests/test_interpreter_gpu.py::test_all_ops_against_oracle in detail: python tools/scratch/fuzz_case.py <kind> <round>
import os, sys, zlib
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_interpreter_gpu as T
from oracle import dfol_oracle as orc
import dfol_vqa_amd as D
kind, rnd = sys.argv[1], int(sys.argv[2])
seed, split, n_range = zlib.crc32(kind.encode()) % 1000 + 7919 * rnd, 2 + rnd % 3, (2, 40) if rnd % 2 == 0 else (1, 17)
d = os.path.join("tests", "golden", "mini_ontology")
p = {"attribute_file": os.path.join(d, "attribute.json"), "class_file": os.path.join(d, "class.json"), "relation_file": os.path.join(d, "relation.json"),
     "vocabulary_file": os.path.join(d, "vocab.json"), "word_embedding_file": os.path.join(d, "glove.txt")}
ontology = D.GQAOntology(p["attribute_file"], p["class_file"], p["vocabulary_file"], p["word_embedding_file"], relation_json_path=p["relation_file"])
oont = orc.Ontology(p["attribute_file"], p["class_file"], p["vocabulary_file"], p["relation_file"])
C, CR = len(ontology._vocabulary["idx_to_arg"]), len(ontology._relation_index)
qs, scenes = T.random_questions(kind, 24, n_range[0], n_range[1], C, CR, seed=seed)
model = T.table_model(ontology)
res, _ = T.run(model, qs, scenes, ontology, split=split)
lp = res["log_probability"].cpu().numpy().astype(np.float64)
r32 = orc.run_questions(oont, qs, scenes, np.float32, split=split)["log_probability"].astype(np.float64)
r64 = orc.run_questions(oont, qs, scenes, np.float64, split=split)["log_probability"]
for i in range(len(lp)):
    print("%2d n=%2d lp64 %+.6f  ours-64 %+.2e  ref32-64 %+.2e   dp ours %+.2e ref32 %+.2e   %s" % (
        i, scenes[i]["n"], r64[i], lp[i] - r64[i], r32[i] - r64[i], np.exp(lp[i]) - np.exp(r64[i]), np.exp(r32[i]) - np.exp(r64[i]),
        " ".join(str(o) for o in qs[i]["program"])[:150] if abs(np.exp(lp[i]) - np.exp(r64[i])) > 8e-7 else ""))
if len(sys.argv) > 3:
    i = int(sys.argv[3])
    print(qs[i]["program"])
    # the same question alone, and under the relate kernels' general (non-product) code: DFOL_RELATE_FAST=0 if the library honours it
    r1, _ = T.run(model, [qs[i]], [scenes[i]], ontology, split=1)
    print("alone: ours-64 %+.3e" % (float(r1["log_probability"].cpu().numpy()[0]) - r64[i]))
