#!/usr/bin/env python3
"""Times the reference's own Python (imported from /root/reference/src) and the CPU oracle (oracle/dfol_oracle.py, the
`cpu_baseline` leg of bench.py) on IDENTICAL inputs, in the build container.

    python tools/time_reference.py [--questions 100] [--objects 36] [--threads 8]

Workload = BASELINE.json configs[0]: synthetic N-object scenes, select->filter->relate->exist programs, the full-size model
(2048->512, 516/1036->256->300->2335) with seeded random weights and random GloVe, ProgramBatch sizes {5, 10, 20} (the
reference's cost is super-linear in it).  For every size: the reference's forward time, the oracle's forward time, their ratio,
and the agreement of their log-probabilities.  The result goes to profiles/reference_timing.json, whose `summary` string
bench.py attaches to `cpu_baseline` on the GPU box, where the reference cannot run.  The reference's Python never travels.
"""

import argparse
import copy
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_harness  # noqa: E402
from dfol_vqa_amd import synthetic as syn  # noqa: E402
from oracle import dfol_oracle as orc  # noqa: E402
from oracle import dfol_oracle_torch as orct  # noqa: E402


def write_glove(path, names, dim=300, seed=3):
    rng = np.random.RandomState(seed)
    words = sorted({w for n in names for w in n.split()})
    with open(path, "w") as f:
        for w in words:
            f.write(w + " " + " ".join("%.4f" % x for x in rng.normal(0, 0.3, dim)) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--questions", type=int, default=100)
    ap.add_argument("--objects", type=int, default=36)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--sizes", type=str, default="5,10,20")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--out", type=str, default=os.path.join(ROOT, "profiles", "reference_timing.json"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)                      # trainer.py:57-62
    ref = ref_harness.import_reference()
    sys.path.insert(0, ref_harness.REF_SRC)
    import gqa_interpreter_experiments as gie

    tmp = tempfile.mkdtemp(prefix="dfol_reftime_")
    paths, names = syn.write_synthetic_ontology(tmp)
    with open(paths["vocabulary_file"]) as f:
        vocab = json.load(f)
    paths["word_embedding_file"] = os.path.join(tmp, "glove.txt")
    write_glove(paths["word_embedding_file"], vocab["idx_to_arg"])
    cfg = syn.reference_config(paths)
    exp = gie.GQAObjectBoxExperiment()
    exp._local_rank = 0
    ontology = exp.build_ontology(cfg, None)
    torch.manual_seed(0)
    model = exp.build_model(cfg, ontology, None)
    lin = model._oracle._embedding_network._network[1]
    with torch.no_grad():                                    # the magnitudes bench.py uses (sparse concept probabilities)
        lin.weight.normal_(0.0, 0.1)
        lin.bias.fill_(-2.0)
    model.eval()
    weights = {k: v.detach().numpy() for k, v in model.state_dict().items() if k.startswith("_featurizer.") or k.startswith("_oracle.")}
    oont = orc.Ontology(paths["attribute_file"], paths["class_file"], paths["vocabulary_file"], paths["relation_file"])

    nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
    qs = []
    for i in range(args.questions):
        br, last = syn.three_hop_program(i, nouns, attrs, rels)
        qs.append(syn.question(i, br, last, "yes", syn.feature_scene(i, args.objects, 2048)))

    rows = []
    for size in [int(x) for x in args.sizes.split(",")]:
        split = max(1, -(-args.questions // size))
        t_ref, t_orc, t_rst = [], [], []

        def run_ref():
            collater = ref_harness.make_collater(ref, split, "feature")
            t0 = time.perf_counter()
            pbs = collater.collate(copy.deepcopy(qs))
            for pb in pbs:
                pb.create_sparse_tensors()
            with torch.no_grad():
                out = model(pbs, False)
            t_ref.append(time.perf_counter() - t0)
            return out

        def run_rst():
            t0 = time.perf_counter()
            out = orct.run_questions(oont, qs, [q["scene"] for q in qs], weights, split=split)
            t_rst.append(time.perf_counter() - t0)
            return out

        # reference and restatement alternate (whichever runs right after another library's thread pool has been busy is slowed by
        # 20-30 %: the order is swapped every repetition and the minimum taken); the numpy port is timed afterwards, on its own
        for rep in range(args.reps):
            if rep % 2 == 0:
                res, rt = run_ref(), run_rst()
            else:
                rt, res = run_rst(), run_ref()
        for _ in range(max(1, args.reps // 2)):
            t0 = time.perf_counter()
            r = orc.run_questions(oont, qs, [q["scene"] for q in qs], np.float32, split=split, weights=weights)
            t_orc.append(time.perf_counter() - t0)
        lp_ref = res["log_probability"].numpy()
        row = {"program_batch_size": size, "reference_s": min(t_ref), "oracle_s": min(t_orc),
               "reference_qps": args.questions / min(t_ref), "oracle_qps": args.questions / min(t_orc),
               "oracle_over_reference_time": min(t_orc) / min(t_ref),
               "max_abs_dlp": float(np.abs(lp_ref - r["log_probability"]).max()),
               "max_abs_dp": float(np.abs(np.exp(lp_ref) - np.exp(r["log_probability"])).max()),
               "answers_agree": int(sum(a == b for a, b in zip(res["answer"], r["answer"]))),
               "restatement_s": min(t_rst), "restatement_qps": args.questions / min(t_rst), "restatement_over_reference_time": min(t_rst) / min(t_ref),
               "restatement_max_abs_dlp": float(np.abs(lp_ref - rt["log_probability"]).max()),
               "restatement_answers_agree": int(sum(a == b for a, b in zip(res["answer"], rt["answer"])))}
        rows.append(row)
        print(json.dumps(row))
    best_ref = max(rows, key=lambda x: x["reference_qps"])
    best_orc = max(rows, key=lambda x: x["oracle_qps"])
    best_rst = max(rows, key=lambda x: x["restatement_qps"])
    out = {"workload": "BASELINE configs[0]: %d questions, N=%d, select->filter->relate->exist, full-size model, fp32" % (args.questions, args.objects),
           "threads": args.threads, "torch": torch.__version__, "numpy": np.__version__, "rows": rows,
           "summary": "build container, %d threads, N=%d: reference %.1f q/s (ProgramBatch %d) vs this port %.1f q/s (ProgramBatch %d); port time / "
                      "reference time at equal ProgramBatch size: %s; max |dp| between them %.1e"
                      % (args.threads, args.objects, best_ref["reference_qps"], best_ref["program_batch_size"], best_orc["oracle_qps"],
                         best_orc["program_batch_size"], ", ".join("%.2f (size %d)" % (x["oracle_over_reference_time"], x["program_batch_size"]) for x in rows),
                         max(x["max_abs_dp"] for x in rows)),
           "restatement_summary": "build container, %d threads, N=%d: reference %.1f q/s (ProgramBatch %d) vs the torch-CPU restatement "
                                  "(oracle/dfol_oracle_torch.py) %.1f q/s (ProgramBatch %d); restatement time / reference time at equal ProgramBatch size: %s; "
                                  "max |dlp| between them %.1e"
                                  % (args.threads, args.objects, best_ref["reference_qps"], best_ref["program_batch_size"], best_rst["restatement_qps"],
                                     best_rst["program_batch_size"],
                                     ", ".join("%.2f (size %d)" % (x["restatement_over_reference_time"], x["program_batch_size"]) for x in rows),
                                     max(x["restatement_max_abs_dlp"] for x in rows))}
    key = "N%d" % args.objects
    allres = {}
    if os.path.exists(args.out):
        with open(args.out) as f:
            allres = json.load(f)
    allres[key] = out
    allres["summary"] = " | ".join(allres[k]["summary"] for k in sorted(allres) if k.startswith("N"))
    allres["restatement_summary"] = " | ".join(allres[k]["restatement_summary"] for k in sorted(allres) if k.startswith("N") and "restatement_summary" in allres[k])
    with open(args.out, "w") as f:
        json.dump(allres, f, indent=1)
    print(allres["summary"])


if __name__ == "__main__":
    main()
