"""Lab: one g10 calibration case through the executor and the Python loop, next to the golden."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
import dfol_vqa_amd as D
from dfol_vqa_amd import native_plan as NP, native_exec, _lib
from test_interpreter_gpu import CalibrationCollater, neural_model, DEV
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mini_ontology
p = mini_ontology.write(os.path.join(ROOT, "tests", "golden", "mini_ontology"))
ont = D.GQAOntology(p["attribute_file"], p["class_file"], p["vocabulary_file"], p["word_embedding_file"], relation_json_path=p["relation_file"])
a, meta = gu.load("g10_calibration")
weights = {k[2:]: a[k] for k in a.files if k.startswith("w:")}
model = neural_model(ont, meta["config"], weights)
name = sys.argv[1] if len(sys.argv) > 1 else "exist"
run_meta = meta["runs"][name]
qs = [{"program": q["program"], "answer": q["answer"], "question_id": q["question_id"], "image_id": "img000", "tokens": [], "original_dict": None,
       "question": None, "scene": {"n": q["n"], "X": a["%s:X_%d" % (name, i)]}} for i, q in enumerate(run_meta["questions"])]
for q in qs:
    print(q["program"])
names = {v: k for k, v in vars(NP).items() if k.startswith("OP_")}
for native in ("1", "0"):
    os.environ["DFOL_NATIVE"] = native
    pbs = [pb.to_cuda(DEV) for pb in CalibrationCollater(ont).collate([dict(q) for q in qs])]
    with torch.no_grad():
        res = model(pbs, False)
    print("native" if native == "1" else "python", res["log_probability"].cpu().numpy(), dict(_lib.PATH_COUNTS))
    if native == "1":
        plan = pbs[0]._native_plan
        for row in plan.instrs:
            print("  ", names[int(row[0])], [int(x) for x in row[1:12]])
print("golden f32", a[name + ":lp_f32"], "off", a[name + ":lp_off_f32"])
