#!/usr/bin/env python3
"""Capture golden vectors from the imported reference (runs only in the build container).

    python tools/capture_goldens.py            # rewrites tests/golden/*.npz|json

The reference has no tests of its own, so parity is pinned by running the
reference's Python here, on inputs authored in this repository, and committing the inputs
together with the reference's outputs.  Every case is stored twice: the reference run in
fp32 and the same reference run in fp64 (`model.double()`), which gives the tolerance
policy of DESIGN.md something to measure fp32 conditioning against.

Families:
  g1  log-space primitives                     util.py:17-47
  g2  BatchBayesianLogicCell.forward           batch_base_ops.py:153-215, 62-151
  g3  FilterBatch / RelateBatch.forward        batch_base_ops.py:311-405, 483-596
  g4  whole-interpreter runs, all 16 ops+end   batch_base_interpreter.py:72-183
  g5  neural oracle (reduced dims) end-to-end  classifier_oracle.py:145-156 + MLPs
  g6  loss values and gradients                trainer.py:181-262
  g7  collate_programs                         data_pipeline.py:647-746
  g8  gather_results                           data_parallel.py:15-50
"""

import copy
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import mini_ontology  # noqa: E402
import ref_harness  # noqa: E402
from dfol_vqa_amd import synthetic as syn  # noqa: E402

import torch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
ref = ref_harness.import_reference()
paths = mini_ontology.write(os.path.join(OUT, "mini_ontology"))
ontology = ref_harness.build_ontology(ref, paths)
C = len(ontology._vocabulary["idx_to_arg"])
CR = len(ontology._relation_index)
op = syn.op


def both_dtypes():
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        ref_harness.set_fp64(ref, dt == torch.float64)
        yield dt, tag
    ref_harness.set_fp64(ref, False)


def save(name, arrays, meta):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", name, "arrays:", len(arrays), "bytes:", os.path.getsize(os.path.join(OUT, name + ".npz")))


# ---------------------------------------------------------------------------------------- g1
def g1():
    u = ref.util
    arrays = {}
    x = np.concatenate([np.linspace(-60, 0, 241), -np.logspace(-8, 1.7, 120), [-30.0, -46.0517, -1e-7, 0.0]])
    y = np.concatenate([np.linspace(0, -60, 241)[::-1] * 0.37, -np.logspace(-7, 1.5, 120), [-0.5, -30.0, -2e-7, 0.0]])
    p = np.concatenate([np.linspace(0, 1, 200), np.logspace(-30, 0, 100), [0.0, 1e-20, 1e-21, 1.0]])
    for dt, tag in both_dtypes():
        tx, ty, tp = (torch.tensor(v, dtype=dt) for v in (x, y, p))
        arrays["safe_log_" + tag] = u.safe_log(tp).numpy()
        arrays["safe_exp_" + tag] = u.safe_exp(tx).numpy()
        arrays["log_and_" + tag] = u.log_and(tx, ty).numpy()
        arrays["log_or_" + tag] = u.log_or(tx, ty).numpy()
        arrays["log_not_" + tag] = u.log_not(tx).numpy()
        for a in (0.0, 1.0, 0.25):
            arrays["log_pnot_a%g_%s" % (a, tag)] = u.log_parametric_not(tx, a, 1).numpy()
        arrays["log_or_tensor_" + tag] = u.log_or_tensor(tx.view(-1, 5), 1).numpy()
    arrays["x"], arrays["y"], arrays["p"] = x, y, p
    save("g1_primitives", arrays, {"source": "util.py:17-47", "torch": torch.__version__})


# ---------------------------------------------------------------------------------------- g2
def flat_problem(rng, n_list, k_list, arity, family="mix10"):
    """A flat-layout logic-cell problem as the reference sees it.
    n_list: objects per image (= per question); k_list: predicates per question."""
    Q = len(n_list)
    O = int(sum(n_list))
    img = np.repeat(np.arange(Q), n_list)
    pq = np.repeat(np.arange(Q), k_list)
    P = len(pq)
    prior = np.minimum(syn.table_log_likelihood(rng, (Q, arity, O), "unif") * 0.3, 0).astype(np.float32)
    if arity == 1:
        ll = np.full((P, O, 1), -30, np.float32)
        for p_ in range(P):
            m = img == pq[p_]
            ll[p_, m, 0] = syn.table_log_likelihood(rng, (int(m.sum()),), family)
    else:
        ll = np.full((P, O, O, 1), -30, np.float32)
        for p_ in range(P):
            idx = np.nonzero(img == pq[p_])[0]
            n = len(idx)
            t = syn.table_log_likelihood(rng, (n, n), family)
            t[np.arange(n), np.arange(n)] = -30
            ll[np.ix_([p_], idx, idx, [0])] = t[None, :, :, None]
    return {"img": img, "pq": pq, "prior": prior, "ll": ll, "Q": Q, "O": O, "P": P}


def sparse_map(rows, cols, shape, dtype):
    ind = torch.stack([torch.as_tensor(rows, dtype=torch.int64), torch.as_tensor(cols, dtype=torch.int64)])
    return torch.sparse_coo_tensor(ind, torch.ones(len(rows), dtype=dtype), shape)


def g2():
    arrays, meta = {}, {"source": "batch_base_ops.py:153-215,62-151", "cases": []}
    rng = np.random.RandomState(20)
    specs = [
        # name, n_list, k_list, arity, quantifier mode, negation
        ("a1_single", [5], [1], 1, "exists", False),
        ("a1_batch", [1, 2, 5], [1, 1, 1], 1, "mixed", True),
        ("a1_expand", [4, 6, 3], [2, 1, 3], 1, "mixed", True),
        ("a2_single", [5], [1], 2, "exists", False),
        ("a2_n1", [1, 3], [1, 1], 2, "exists", False),
        ("a2_batch", [2, 5, 4], [1, 1, 1], 2, "mixed", False),
        ("a2_batch_neg", [3, 6, 2], [1, 1, 1], 2, "mixed", True),
        ("a2_expand", [4, 5, 3], [2, 1, 2], 2, "mixed", True),
        ("a2_forall", [6, 4], [1, 1], 2, "forall", False),
        ("a2_n36", [36, 36, 36], [1, 1, 1], 2, "mixed", False),
        ("a2_stress_mix05", [12, 9], [1, 2], 2, "exists", False),
        ("a2_stress_weak", [12, 9], [1, 1], 2, "exists", False),
    ]
    for name, n_list, k_list, arity, qmode, neg in specs:
        fam = "mix05" if "mix05" in name else ("weak" if "weak" in name else "mix10")
        pr = flat_problem(rng, n_list, k_list, arity, fam)
        Q, O, P = pr["Q"], pr["O"], pr["P"]
        if qmode == "exists":
            quant = np.ones((P, arity), np.float32)
        elif qmode == "forall":
            quant = np.zeros((P, arity), np.float32)
        else:
            quant = (rng.uniform(size=(Q, arity)) < 0.6).astype(np.float32)[pr["pq"]]
        is_neg = (rng.uniform(size=P) < 0.5).astype(np.float32) if neg else None
        for dt, tag in both_dtypes():
            cell = ref.base_ops.BatchBayesianLogicCell(arity)
            bom = sparse_map(pr["img"], np.arange(O), (Q, O), dt)
            pqm = sparse_map(np.arange(P), pr["pq"], (P, Q), dt) if P != Q else None
            out = cell(torch.tensor(pr["prior"], dtype=dt), torch.tensor(pr["ll"], dtype=dt),
                       torch.tensor(quant, dtype=dt), list(range(arity)), bom, pqm,
                       None if is_neg is None else torch.tensor(is_neg, dtype=dt))
            arrays[name + "_out_" + tag] = out.numpy()
        arrays[name + "_prior"] = pr["prior"]
        arrays[name + "_ll"] = pr["ll"]
        arrays[name + "_quant"] = quant
        arrays[name + "_img"] = pr["img"]
        arrays[name + "_pq"] = pr["pq"]
        if is_neg is not None:
            arrays[name + "_neg"] = is_neg
        meta["cases"].append({"name": name, "n": n_list, "k": k_list, "arity": arity, "family": fam, "neg": neg})
    save("g2_logic_cell", arrays, meta)


# ---------------------------------------------------------------------------------------- interpreter runs
def run_reference(questions, split=1, dtype=torch.float32, training=False, return_trace=True, normalize=True,
                  grad_tables=False, hard=False, threshold=0):
    qs = copy.deepcopy(questions)
    collater = ref_harness.make_collater(ref, split, "table")
    pbs = collater.collate(qs)
    model = ref_harness.build_table_interpreter(ref, ontology, normalize)
    model._hard_mode = hard                                  # batch_gqa_interpreter.py:23,73
    model._likelihood_threshold = threshold                  # :22,73
    leaves = []
    for pb in pbs:
        pb.create_sparse_tensors()
        if dtype == torch.float64:
            pb.to(torch.float64)
            pb._object_batch_index = pb._object_batch_index.long()
        if grad_tables:
            pb._object_features = pb._object_features.clone().requires_grad_(True)
            pb._meta_data["R"] = pb._meta_data["R"].clone().requires_grad_(True)
            leaves.append((pb._object_features, pb._meta_data["R"]))
    if dtype == torch.float64:
        model = model.double()
    if training:
        model.train()
    with torch.set_grad_enabled(grad_tables):
        res = model(pbs, training, return_trace=return_trace)
    return res, pbs, leaves


def scene_for(qid, n, family="mix10"):
    return syn.table_scene(qid, n, C, CR, family)


def pack_result(arrays, meta, tag, res, traces):
    lp = res["log_probability"]
    arrays["lp_" + tag] = lp.detach().numpy()
    if tag == "f32":
        meta["answer"] = res["answer"]
        meta["options"] = res["options"]
        meta["type"] = int(res["type"])
        meta["answer_log_probability"] = res["answer_log_probability"]
    k = 0
    for b, trace in enumerate(traces):
        for i, x in enumerate(trace):
            if isinstance(x, ref.base_types.BatchVariableSet):
                arrays["trace_%s_b%d_op%d_att" % (tag, b, i)] = x._log_attention.detach().numpy()
                arrays["trace_%s_b%d_op%d_quant" % (tag, b, i)] = x._quantifier.detach().numpy()
                if tag == "f32":
                    meta.setdefault("trace_names", {})["b%d_op%d" % (b, i)] = list(x._name)
                k += 1
    return k


def questions_to_meta(questions):
    out = []
    for q in questions:
        out.append({"program": q["program"], "answer": q["answer"], "question_id": q["question_id"], "n": q["scene"]["n"]})
    return out


def capture_run(name, questions, split=1, normalize=True, arrays=None, meta=None, hard=False, threshold=0):
    arrays = {} if arrays is None else arrays
    meta = {} if meta is None else meta
    meta.update({"questions": questions_to_meta(questions), "split": split, "normalize": normalize, "hard_mode": hard,
                 "likelihood_threshold": threshold,
                 "source": "batch_base_interpreter.py:72-183"})
    for i, q in enumerate(questions):
        arrays["A_%d" % i] = q["scene"]["A"]
        arrays["R_%d" % i] = q["scene"]["R"]
    for dt, tag in both_dtypes():
        (res, traces), pbs, _ = run_reference(questions, split, dt, normalize=normalize, hard=hard, threshold=threshold)
        pack_result(arrays, meta, tag, res, traces)
        if tag == "f32":
            meta["op_names"] = [[ob._op_name for ob in pb._op_batch_list] for pb in pbs]
            meta["dependencies"] = [pb._dependencies for pb in pbs]
    save(name, arrays, meta)


def g4():
    Q = syn.question
    n_of = lambda i: [5, 7, 3, 6, 4, 8, 2, 5][i % 8]

    def mk(i, branches, last, answer="yes", fam="mix10"):
        return Q(100 + i, branches, last, answer, scene_for(100 + i, n_of(i), fam))

    # exist: ragged program lengths, negation, '_' select, both relate directions, two relates
    qs = [
        mk(0, [[op("select", "dog")]], op("exist")),
        mk(1, [[op("select", "cat"), op("filter", "red")]], op("exist"), "no"),
        mk(2, [[op("select", "_"), op("filter", "not(blue)"), op("relate", "on", True, "table")]], op("exist")),
        mk(3, [[op("select", "man"), op("relate", "to the left of", False, "car"), op("filter", "small"),
                op("relate", "near", True, "_")]], op("exist")),
        mk(4, [[op("select", "chair"), op("filter", "wood"), op("filter", "large"),
                op("relate", "not(under)", False, "cup")]], op("exist"), "no"),
        mk(5, [[op("select", "scene"), op("relate", "holding", True, "tree"), op("relate", "behind", False, "dog"),
                op("filter", "black")]], op("exist")),
    ]
    capture_run("g4_exist", qs)
    capture_run("g4_exist_split3", qs, split=3)

    # the BASELINE configs[0] shape at N=36 (3 questions)
    nouns, attrs, rels = ["dog", "table", "cup", "car", "tree", "cat", "chair", "man"], \
        ["red", "blue", "small", "large", "white", "black"], ["on", "to the left of", "to the right of", "under", "near"]
    qs36 = []
    for i in range(3):
        br, last = syn.three_hop_program(500 + i, nouns, attrs, rels)
        qs36.append(Q(500 + i, br, last, "yes", scene_for(500 + i, 36)))
    capture_run("g4_threehop_n36", qs36)

    qs = [
        mk(10, [[op("select", "dog"), op("filter", "small")]], op("verify_attrs", ["red", "large"])),
        mk(11, [[op("select", "cup")]], op("verify_attrs", ["glass"]), "no"),
        mk(12, [[op("select", "car"), op("relate", "near", True, "bus")]], op("verify_attrs", ["not(white)", "metal"])),
    ]
    capture_run("g4_verify_attrs", qs)

    qs = [
        mk(20, [[op("select", "dog")]], op("choose_attr", ["red", "blue"]), "red"),
        mk(21, [[op("select", "table"), op("filter", "wood")]], op("choose_attr", ["small", "large"]), "large"),
        mk(22, [[op("select", "man"), op("relate", "holding", True, "cup")]], op("choose_attr", ["sitting", "standing"]), "sitting"),
    ]
    capture_run("g4_choose_attr", qs)
    capture_run("g4_choose_attr_nonorm", qs, normalize=False)

    qs = [
        mk(30, [[op("select", "dog")]], op("query_attr", "color"), "black"),
        mk(31, [[op("select", "animal"), op("filter", "small")]], op("query_attr", "name"), "cat"),
        mk(32, [[op("select", "chair")]], op("query_attr", "material"), "wood"),
        mk(33, [[op("select", "cup")]], op("query_attr", "name"), "cup"),
    ]
    capture_run("g4_query_attr", qs)

    qs = [
        mk(40, [[op("select", "dog")]], op("verify_rel", "on", True, "couch")),
        mk(41, [[op("select", "man"), op("filter", "standing")]], op("verify_rel", "to the right of", False, "bus"), "no"),
        mk(42, [[op("select", "cup")]], op("verify_rel", "not(near)", True, "_")),
    ]
    capture_run("g4_verify_rel", qs)

    qs = [
        mk(50, [[op("select", "dog")]], op("choose_rel", ["to the left of", "to the right of"], True, "cat"), "to the left of"),
        mk(51, [[op("select", "woman"), op("filter", "large")]], op("choose_rel", ["on", "under"], False, "table"), "on"),
        mk(52, [[op("select", "boy")]], op("choose_rel", ["near", "behind"], True, "_"), "near"),
    ]
    capture_run("g4_choose_rel", qs)

    two = lambda i, last, ans="yes": mk(i, [[op("select", "dog"), op("filter", "red")],
                                            [op("select", "cat"), op("relate", "near", bool(i % 2), "table")]], last, ans)
    two_b = lambda i, last, ans="yes": mk(i, [[op("select", "car")], [op("select", "bus"), op("filter", "not(large)")]], last, ans)
    for name in ("and", "or"):
        capture_run("g4_" + name, [two(60, op(name)), two_b(61, op(name), "no"), two(62, op(name))])
    for name in ("two_same", "two_different"):
        capture_run("g4_" + name, [two(70, op(name, "color")), two_b(71, op(name, "material"), "no"), two(73, op(name, "size"))])
    for name in ("all_same", "all_different"):
        capture_run("g4_" + name, [mk(80, [[op("select", "dog")]], op(name, "color")),
                                   mk(81, [[op("select", "furniture"), op("filter", "wood")]], op(name, "name"), "no"),
                                   mk(82, [[op("select", "cup"), op("relate", "on", False, "table")]], op(name, "size"))])
    capture_run("g4_compare", [two(90, op("compare", "large", False), "dog"), two_b(91, op("compare", "red", True), "bus"),
                               two(92, op("compare", "not(small)", True), "cat")])

    # stress families (probability-space comparison only)
    for fam in ("mix05", "weak"):
        qsf = [mk(200 + i, [[op("select", "dog"), op("filter", "red"), op("relate", "on", bool(i % 2), "table")]],
                  op("exist"), "yes", fam) for i in range(4)]
        capture_run("g4_stress_" + fam, qsf)

    # implicit `end`: a hand-built ProgramBatch whose last op is not terminal (batch_gqa_interpreter.py:75-76)
    arrays, meta = {}, {"source": "batch_gqa_interpreter.py:72-78", "implicit_end": True}
    qs = [mk(95, [[op("select", "dog"), op("filter", "red")]], op("exist")), mk(96, [[op("select", "cat"), op("filter", "blue")]], op("exist"))]
    for i, q in enumerate(qs):
        arrays["A_%d" % i], arrays["R_%d" % i] = q["scene"]["A"], q["scene"]["R"]
    for dt, tag in both_dtypes():
        collater = ref_harness.make_collater(ref, 1, "table")
        pb = collater.collate(copy.deepcopy(qs))[0]
        pb2 = ref.data_pipeline.ProgramBatch(pb.device, pb._op_batch_list[:-1], pb._dependencies[:-1], pb._answers,
                                             pb._object_features, pb._object_batch_index, pb._original_dicts, pb._meta_data)
        pb2.create_sparse_tensors()
        model = ref_harness.build_table_interpreter(ref, ontology)
        if dt == torch.float64:
            pb2.to(torch.float64)
            pb2._object_batch_index = pb2._object_batch_index.long()
            model = model.double()
        with torch.no_grad():
            res, traces = model([pb2], False, return_trace=True)
        pack_result(arrays, meta, tag, res, [[t for t in traces[0] if not isinstance(t, dict)]])
    meta["questions"] = questions_to_meta(qs)
    save("g4_end", arrays, meta)


def g11():
    """hard_mode (batch_base_types.py:104-112): the test-time min/max aggregation, through whole-interpreter runs."""
    Q = syn.question
    n_of = lambda i: [5, 7, 3, 6, 4, 8, 2, 5][i % 8]

    def mk(i, branches, last, answer="yes"):
        return Q(600 + i, branches, last, answer, scene_for(600 + i, n_of(i)))

    two = lambda i, last, ans="yes": mk(i, [[op("select", "dog"), op("filter", "red")],
                                            [op("select", "cat"), op("relate", "near", bool(i % 2), "table")]], last, ans)
    capture_run("g11_hard_exist", [mk(0, [[op("select", "dog")]], op("exist")),
                                   mk(1, [[op("select", "_"), op("filter", "not(blue)"), op("relate", "on", True, "table")]], op("exist")),
                                   mk(2, [[op("select", "chair"), op("filter", "wood")]], op("exist"), "no")], hard=True)
    capture_run("g11_hard_single", [mk(3, [[op("select", "cat"), op("filter", "small")]], op("exist"))], hard=True)
    capture_run("g11_hard_verify_attrs", [mk(10, [[op("select", "dog"), op("filter", "small")]], op("verify_attrs", ["red", "large"])),
                                          mk(11, [[op("select", "cup")]], op("verify_attrs", ["glass"]), "no")], hard=True)
    capture_run("g11_hard_query_attr", [mk(30, [[op("select", "dog")]], op("query_attr", "color"), "black"),
                                        mk(31, [[op("select", "animal"), op("filter", "small")]], op("query_attr", "name"), "cat")], hard=True)
    capture_run("g11_hard_choose_rel", [mk(50, [[op("select", "dog")]], op("choose_rel", ["to the left of", "to the right of"], True, "cat"),
                                           "to the left of"),
                                        mk(51, [[op("select", "woman")]], op("choose_rel", ["on", "under"], False, "table"), "on")], hard=True)
    capture_run("g11_hard_and", [two(60, op("and")), two(62, op("and"))], hard=True)
    capture_run("g11_hard_two_same", [two(70, op("two_same", "color")), two(73, op("two_same", "size"))], hard=True)
    capture_run("g11_hard_all_same", [mk(80, [[op("select", "dog")]], op("all_same", "color")),
                                      mk(81, [[op("select", "furniture"), op("filter", "wood")]], op("all_same", "name"), "no")], hard=True)
    # all_different / two_different / query_attr do not forward hard_mode to the operator they wrap (batch_gqa_ops.py:306,628,703)
    capture_run("g11_hard_all_different", [mk(82, [[op("select", "dog")]], op("all_different", "color")),
                                           mk(83, [[op("select", "furniture"), op("filter", "wood")]], op("all_different", "name"), "no")], hard=True)
    capture_run("g11_hard_two_different", [two(74, op("two_different", "color")), two(75, op("two_different", "size"))], hard=True)
    capture_run("g11_hard_compare", [two(90, op("compare", "large", False), "dog"), two(92, op("compare", "not(small)", True), "cat")], hard=True)


# ---------------------------------------------------------------------------------------- g3
def g3():
    """FilterBatch / RelateBatch called directly, incl. None/'_'/'not(x)' tokens and list predicate maps."""
    arrays, meta = {}, {"source": "batch_base_ops.py:311-405,483-596", "cases": []}
    n_list = [4, 6, 3]
    scenes = [scene_for(300 + i, n) for i, n in enumerate(n_list)]
    for i, s in enumerate(scenes):
        arrays["A_%d" % i], arrays["R_%d" % i] = s["A"], s["R"]
    Qn, O = len(n_list), sum(n_list)
    img = np.repeat(np.arange(Qn), n_list)
    rng = np.random.RandomState(33)
    att0 = np.minimum(syn.table_log_likelihood(rng, (Qn, O), "unif") * 0.3, 0)
    att1 = np.minimum(syn.table_log_likelihood(rng, (Qn, O), "unif") * 0.3, 0)
    arrays["att0"], arrays["att1"], arrays["img"] = att0, att1, img
    cases = [
        ("filter_plain", "filter", ["red", "small", "wood"], None, True),
        ("filter_none", "filter", ["red", None, "_"], None, True),
        ("filter_neg", "filter", ["not(red)", "large", " blue "], None, True),
        ("filter_neg_none", "filter", [None, "not(metal)", "glass"], None, True),
        ("filter_expand", "filter", ["red", "blue", "small", "wood", "metal", "glass"], [0, 0, 1, 2, 2, 2], True),
        ("filter_expand_nonorm", "filter", ["red", "blue", "small", "wood", "metal", "glass"], [0, 0, 1, 2, 2, 2], False),
        ("relate_plain", "relate", ["on", "near", "under"], None, True),
        ("relate_none", "relate", ["on", None, "_"], None, True),
        ("relate_neg", "relate", ["not(on)", "to the left of", "not(behind)"], None, True),
        ("relate_expand", "relate", ["on", "under", "near", "holding", "behind"], [0, 0, 1, 2, 2], True),
        ("relate_forall", "relate", ["on", "near", "under"], None, True),
    ]
    for name, kind, tokens, pqm, normalized in cases:
        for dt, tag in both_dtypes():
            oracle = ref.ClassifierOracle(ontology, None, None, None, normalize=True, cached=True)
            A = torch.tensor(np.concatenate([s["A"] for s in scenes]), dtype=dt)
            R = torch.tensor(np.concatenate([s["R"] for s in scenes]), dtype=dt)
            bi = torch.tensor(img)
            ind = ref.util.find_sparse_pair_indices(bi, bi, torch.device("cpu"), True)
            world = ref.base_types.BatchWorld(torch.device("cpu"), O, A, {"features": R, "index": list(ind)}, bi,
                                              {"index": {}, "embedding": torch.zeros(1, 1)}).to(dt)
            q0 = torch.tensor([1.0, 0.0, 1.0] if name == "relate_forall" else [1.0] * 3, dtype=dt)
            q1 = torch.tensor([0.0, 1.0, 1.0] if name == "relate_forall" else [1.0] * 3, dtype=dt)
            vs0 = world.variable_set(["a", "b", "c"], quantifier=q0, log_attention=torch.tensor(att0, dtype=dt))
            vs1 = world.variable_set(["d", "e", "f"], quantifier=q1, log_attention=torch.tensor(att1, dtype=dt))
            # A python-list map works in fp32 only (the reference builds it with the legacy FloatTensor
            # constructor, batch_base_ops.py:324-335); the fp64 pass hands over a ready sparse map instead.
            if pqm is None:
                pq_arg = None
            elif dt == torch.float32:
                pq_arg = list(pqm)
            else:
                pq_arg = sparse_map(np.arange(len(pqm)), pqm, (len(pqm), Qn), dt)
            if kind == "filter":
                f = ref.base_ops.FilterBatch(oracle)
                out = f("id", world, vs0, list(tokens), pq_arg, normalized_probability=normalized)
                arrays[name + "_att_" + tag] = out._log_attention.numpy()
                arrays[name + "_quant_" + tag] = out._quantifier.numpy()
            else:
                r = ref.base_ops.RelateBatch(oracle)
                s_, o_ = r("id", world, vs0, vs1, list(tokens), pq_arg, normalized_probability=normalized)
                arrays[name + "_satt_" + tag] = s_._log_attention.numpy()
                arrays[name + "_oatt_" + tag] = o_._log_attention.numpy()
                arrays[name + "_quant_" + tag] = s_._quantifier.numpy()
        meta["cases"].append({"name": name, "kind": kind, "tokens": tokens, "pqm": pqm, "normalized": normalized,
                              "quant0": [1.0, 0.0, 1.0] if name == "relate_forall" else [1.0] * 3,
                              "quant1": [0.0, 1.0, 1.0] if name == "relate_forall" else [1.0] * 3})
    meta["n"] = n_list
    save("g3_filter_relate", arrays, meta)


# ---------------------------------------------------------------------------------------- g5
def g5():
    """Neural oracle at reduced dims with deterministic weights: raw object features -> tables -> answers."""
    sys.path.insert(0, ref_harness.REF_SRC)
    import gqa_interpreter_experiments as gie
    cfg = dict(box_features_dim=32, oracle_input_dim=16, oracle_output_dim=1, word_embedding_dim=mini_ontology.EMBEDDING_DIM,
               classifier_oracle=True, featurizer_layers_config=[], attribute_network_layers_config=[8],
               relation_network_layers_config=[8], operator_layers_config=[], normalize_oracle=True, dropout=0.0,
               freeze_featurizer=False, freeze_attribute_network=False, freeze_relation_network=False,
               freeze_embedding_network=False, activate_attention_transfer=False, attention_transfer_state_dim=0,
               freeze_attention_network=False, trainable_gate=False, likelihood_threshold=0, hard_mode=False,
               verbose=False, model_name="g5", gpu_num=1)
    exp = gie.GQAObjectBoxExperiment()
    exp._local_rank = 0
    torch.manual_seed(0)
    model = exp.build_model(cfg, ontology, None)
    model.eval()
    arrays, meta = {}, {"source": "classifier_oracle.py:145-156; gqa_interpreter_experiments.py:18-77,107-240", "config": cfg}
    for k, v in model.state_dict().items():
        if k.startswith("_featurizer.") or k.startswith("_oracle."):   # every op module aliases the same oracle
            arrays["w:" + k] = v.numpy()
    n_list = [5, 7, 4]
    nouns, attrs, rels = ["dog", "table", "cup", "car"], ["red", "blue", "small"], ["on", "near", "under"]
    qs = []
    for i, n in enumerate(n_list):
        br, last = syn.three_hop_program(700 + i, nouns, attrs, rels)
        q = syn.question(700 + i, br, last, "yes", syn.feature_scene(700 + i, n, cfg["box_features_dim"]))
        qs.append(q)
        arrays["X_%d" % i] = q["scene"]["X"]
    for dt, tag in both_dtypes():
        m = copy.deepcopy(model).double() if dt == torch.float64 else model
        collater = ref_harness.make_collater(ref, 1, "feature")
        pbs = collater.collate(copy.deepcopy(qs))
        for pb in pbs:
            pb.create_sparse_tensors()
            if dt == torch.float64:
                pb.to(torch.float64)
                pb._object_batch_index = pb._object_batch_index.long()
        with torch.no_grad():
            world = m.build_scene(pbs[0].device, pbs[0]._object_features, pbs[0]._object_batch_index, pbs[0]._meta_data)
            arrays["A_" + tag] = world._attribute_features.numpy()
            arrays["R_" + tag] = world._relation_features["features"].numpy()
            res, traces = m(pbs, False, return_trace=True)
        pack_result(arrays, meta, tag, res, traces)
    meta["questions"] = questions_to_meta(qs)
    meta["relation_index"] = list(ontology._relation_index)
    save("g5_neural_oracle", arrays, meta)


# ---------------------------------------------------------------------------------------- g12
def g14():
    """likelihood_threshold > 0 (batch_gqa_interpreter.py:22; util.py:64-66): QUERY answers below the threshold are dropped."""
    Q = syn.question
    n_of = lambda i: [5, 7, 3, 6, 4, 8, 2, 5][i % 8]
    mk = lambda i, branches, last, answer: Q(1000 + i, branches, last, answer, scene_for(1000 + i, n_of(i)))
    qa = [mk(i, [[op("select", n)] + ([op("filter", "small")] if i % 2 else [])], op("query_attr", c), "red")
          for i, (n, c) in enumerate([("dog", "color"), ("animal", "name"), ("chair", "material"), ("cup", "size"), ("man", "pose"), ("car", "color")])]
    capture_run("g14_threshold_query_attr", qa, threshold=0.03)
    ca = [mk(10 + i, [[op("select", n)]], op("choose_attr", o), o[0])
          for i, (n, o) in enumerate([("dog", ["red", "blue"]), ("table", ["small", "large"]), ("cat", ["black", "white"]), ("bus", ["metal", "glass"])])]
    capture_run("g14_threshold_choose_attr", ca, threshold=0.3)
    cr = [mk(20 + i, [[op("select", n)]], op("choose_rel", o, bool(i % 2), "table"), o[0])
          for i, (n, o) in enumerate([("dog", ["on", "under"]), ("man", ["near", "behind"]), ("cup", ["to the left of", "to the right of"])])]
    capture_run("g14_threshold_choose_rel", cr, threshold=0.01)


def g12():
    """Gradients of the train-step loss w.r.t. every weight of the neural model (featurizer, attribute / relation MLPs, embedding
    layer) at reduced dims, from the reference's own autograd (trainer.py:181-262, 429-442), for a BINARY and a QUERY batch."""
    sys.path.insert(0, ref_harness.REF_SRC)
    import gqa_interpreter_experiments as gie
    from nsvqa.train.trainer import VQATrainer
    fake = types.SimpleNamespace(_device=torch.device("cpu"), _config={})
    cfg = dict(box_features_dim=32, oracle_input_dim=16, oracle_output_dim=1, word_embedding_dim=mini_ontology.EMBEDDING_DIM,
               classifier_oracle=True, featurizer_layers_config=[], attribute_network_layers_config=[8],
               relation_network_layers_config=[8], operator_layers_config=[], normalize_oracle=True, dropout=0.0,
               freeze_featurizer=False, freeze_attribute_network=False, freeze_relation_network=False,
               freeze_embedding_network=False, activate_attention_transfer=False, attention_transfer_state_dim=0,
               freeze_attention_network=False, trainable_gate=False, likelihood_threshold=0, hard_mode=False,
               verbose=False, model_name="g12", gpu_num=1)
    exp = gie.GQAObjectBoxExperiment()
    exp._local_rank = 0
    torch.manual_seed(1)
    model = exp.build_model(cfg, ontology, None)
    model.train()
    Q = syn.question
    n_list = [5, 7, 4, 6]
    sets = {
        "binary": [Q(900 + i, [[op("select", "dog"), op("filter", "red"), op("relate", "on", bool(i % 2), "table")]], op("exist"),
                     "yes" if i % 2 == 0 else "no", syn.feature_scene(900 + i, n, cfg["box_features_dim"])) for i, n in enumerate(n_list)],
        "query_rel": [Q(920 + i, [[op("select", "man"), op("filter", "not(small)")]], op("choose_rel", ["on", "under"], bool(i % 2), "table"),
                        "on" if i % 2 == 0 else "under", syn.feature_scene(920 + i, n, cfg["box_features_dim"])) for i, n in enumerate(n_list)],
    }
    arrays, meta = {}, {"source": "trainer.py:181-262,429-442; gqa_interpreter_experiments.py:18-77,107-240", "config": cfg, "sets": {}}
    for k, v in model.state_dict().items():
        if k.startswith("_featurizer.") or k.startswith("_oracle."):
            arrays["w:" + k] = v.numpy().copy()
    for name, qs in sets.items():
        for i, q in enumerate(qs):
            arrays["%s:X_%d" % (name, i)] = q["scene"]["X"]
        for dt, tag in both_dtypes():
            m = copy.deepcopy(model).double() if dt == torch.float64 else copy.deepcopy(model)
            m.train()
            collater = ref_harness.make_collater(ref, 1, "feature")
            pbs = collater.collate(copy.deepcopy(qs))
            for pb in pbs:
                pb.create_sparse_tensors()
                if dt == torch.float64:
                    pb.to(torch.float64)
                    pb._object_batch_index = pb._object_batch_index.long()
            res = m(pbs, True, return_trace=False)
            lp = res["log_probability"]
            if dt == torch.float64:           # trainer.py:185-194,207-230 restated in fp64 (its targets are built in fp32)
                if res["type"] == ref.base_types.QuestionType.BINARY:
                    target = torch.tensor([a == "yes" for pb in pbs for a in pb._answers], dtype=dt)
                    loss = torch.nn.functional.binary_cross_entropy(lp.exp(), target, reduction="sum")
                else:
                    answers = [a for pb in pbs for a in pb._answers]
                    target = [[a == o for o in opt] for a, opt in zip(answers, res["options"])]
                    seg = torch.tensor([i for i, t in enumerate(target) for _ in t])
                    tflat = torch.tensor([x for t in target for x in t], dtype=dt)
                    denom = torch.zeros(len(target), dtype=dt).index_add(0, seg, lp.exp())
                    loss = ref.util.safe_log(denom).sum() - (tflat * lp).sum()
            else:
                loss = VQATrainer._compute_loss(fake, pbs, res)
            loss = loss / sum(pb.batch_size() for pb in pbs)
            loss.backward()
            arrays["%s:loss_%s" % (name, tag)] = loss.detach().numpy()
            arrays["%s:lp_%s" % (name, tag)] = lp.detach().numpy()
            seen = set()
            for k, prm in m.named_parameters():
                if (k.startswith("_featurizer.") or k.startswith("_oracle.")) and id(prm) not in seen:
                    seen.add(id(prm))
                    arrays["%s:g:%s:%s" % (name, k, tag)] = (torch.zeros_like(prm) if prm.grad is None else prm.grad).detach().numpy()
        meta["sets"][name] = {"questions": questions_to_meta(qs)}
    save("g12_weight_gradients", arrays, meta)


# ---------------------------------------------------------------------------------------- g6
def g6():
    """Loss values and gradients w.r.t. the tables (trainer.py:181-262) for BINARY and QUERY batches."""
    from nsvqa.train.trainer import VQATrainer
    fake = types.SimpleNamespace(_device=torch.device("cpu"), _config={})
    Q = syn.question
    sets = {
        "binary": [Q(800 + i, [[op("select", "dog"), op("filter", "red"), op("relate", "on", bool(i % 2), "table")]], op("exist"),
                     "yes" if i % 2 == 0 else "no", scene_for(800 + i, 4 + i)) for i in range(3)],
        "query": [Q(810 + i, [[op("select", "cat"), op("filter", "small")]], op("choose_attr", ["red", "blue"]),
                    "red" if i % 2 == 0 else "blue", scene_for(810 + i, 5 + i)) for i in range(3)],
        "query_rel": [Q(820 + i, [[op("select", "man")]], op("choose_rel", ["on", "under"], bool(i % 2), "table"),
                        "on" if i % 2 == 0 else "under", scene_for(820 + i, 4 + i)) for i in range(3)],
    }
    for name, qs in sets.items():
        arrays, meta = {}, {"source": "trainer.py:181-262,429-442", "questions": questions_to_meta(qs)}
        for i, q in enumerate(qs):
            arrays["A_%d" % i], arrays["R_%d" % i] = q["scene"]["A"], q["scene"]["R"]
        for dt, tag in both_dtypes():
            res, pbs, leaves = run_reference(qs, 1, dt, training=True, return_trace=False, grad_tables=True)
            if dt == torch.float64:
                # _compute_loss builds fp32 targets; run it on an fp32 view of fp64 log-probs would lose the point,
                # so restate the two formulas here in fp64 exactly as trainer.py:185-194,207-230 does.
                lp = res["log_probability"]
                if res["type"] == ref.base_types.QuestionType.BINARY:
                    target = torch.tensor([a == "yes" for pb in pbs for a in pb._answers], dtype=dt)
                    loss = torch.nn.functional.binary_cross_entropy(lp.exp(), target, reduction="sum")
                else:
                    answers = [a for pb in pbs for a in pb._answers]
                    target = [[a == o for o in opt] for a, opt in zip(answers, res["options"])]
                    seg = torch.tensor([i for i, t in enumerate(target) for _ in t])
                    tflat = torch.tensor([x for t in target for x in t], dtype=dt)
                    denom = torch.zeros(len(target), dtype=dt).index_add(0, seg, lp.exp())
                    loss = ref.util.safe_log(denom).sum() - (tflat * lp).sum()
            else:
                loss = VQATrainer._compute_loss(fake, pbs, res)
            loss = loss / sum(pb.batch_size() for pb in pbs)
            loss.backward()
            arrays["loss_" + tag] = loss.detach().numpy()
            arrays["lp_" + tag] = res["log_probability"].detach().numpy()
            for key, leaf in (("gA_", leaves[0][0]), ("gR_", leaves[0][1])):
                arrays[key + tag] = (torch.zeros_like(leaf) if leaf.grad is None else leaf.grad).detach().numpy()
            if tag == "f32":
                meta["options"] = res["options"]
                meta["type"] = int(res["type"])
        save("g6_loss_" + name, arrays, meta)


# ---------------------------------------------------------------------------------------- g7 / g8
def g7():
    Q = syn.question
    qs = [
        Q(0, [[op("select", "dog")]], op("exist")),
        Q(1, [[op("select", "cat"), op("filter", "red"), op("filter", "small")]], op("exist")),
        Q(2, [[op("select", "_"), op("relate", "on", True, "table"), op("filter", "blue"), op("relate", "near", False, "cup")]], op("exist")),
        Q(3, [[op("select", "man"), op("filter", "large"), op("relate", "under", True, "_"), op("filter", "wood"), op("filter", "black")]], op("exist")),
    ]
    qs2 = [
        Q(4, [[op("select", "dog"), op("filter", "red")], [op("select", "cat")]], op("two_same", "color")),
        Q(5, [[op("select", "car")], [op("select", "bus"), op("relate", "near", True, "tree"), op("filter", "small")]], op("two_same", "size")),
    ]
    qs3 = [Q(6, [[op("select", "dog")]], op("choose_attr", ["red", "blue"])), Q(7, [[op("select", "cat"), op("filter", "small")]], op("choose_attr", ["white", "black"]))]
    meta = {"source": "data_pipeline.py:647-746,31-143", "cases": []}
    for name, questions in (("ragged_exist", qs), ("two_branch", qs2), ("choose", qs3)):
        collater = ref.data_pipeline.ProgramCollaterBase("select", "relate", "filter", 1)
        obl, deps = collater.collate_programs(copy.deepcopy(questions))
        meta["cases"].append({
            "name": name, "questions": [{"program": q["program"]} for q in questions], "dependencies": deps,
            "ops": [{"op_name": ob._op_name, "is_terminal": ob._is_terminal, "arguments": ob._arguments,
                     "mask": ob._mask.tolist(), "predicate_num": ob._predicate_num,
                     "question_index": None if ob._question_index is None else ob._question_index.tolist()} for ob in obl]})
    collater = ref.data_pipeline.ProgramCollaterBase("select", "relate", "filter", 3)
    pbs = collater.collate(copy.deepcopy(qs))
    meta["split3_sizes"] = [pb.batch_size() for pb in pbs]
    meta["split3_ops"] = [[ob._op_name for ob in pb._op_batch_list] for pb in pbs]
    save("g7_collate", {"dummy": np.zeros(1)}, meta)


def g8():
    QT = ref.base_types.QuestionType
    from nsvqa.nn.interpreter.data_parallel import gather_results
    outs = [{"answer": [["yes"], ["no"]], "log_probability": torch.tensor([-0.1, -2.0]), "options": ["no", "yes"], "variable_set": None,
             "type": QT.BINARY, "cumulative_loss": 0, "variable_sets_num": 3, "answer_log_probability": [[-0.1], [-0.14]]},
            {"answer": [["no"], ["yes"]], "log_probability": torch.tensor([-3.0, -0.2]), "options": ["no", "yes"], "variable_set": None,
             "type": QT.BINARY, "cumulative_loss": 0, "variable_sets_num": 4, "answer_log_probability": [[-0.05], [-0.2]]}]
    res = gather_results(outs, torch.device("cpu"), False)
    outs_q = [{"answer": [["red"]], "log_probability": torch.tensor([-0.1, -2.0]), "options": [["red", "blue"]], "variable_set": None,
               "type": QT.QUERY, "cumulative_loss": 0, "variable_sets_num": 1, "answer_log_probability": [[-0.1]]},
              {"answer": [["on"]], "log_probability": torch.tensor([-0.3, -1.0]), "options": [["on", "under"]], "variable_set": None,
               "type": QT.QUERY, "cumulative_loss": 0, "variable_sets_num": 2, "answer_log_probability": [[-0.3]]}]
    res_q = gather_results(outs_q, torch.device("cpu"), False)
    meta = {"source": "data_parallel.py:15-50",
            "binary": {k: (v if not isinstance(v, torch.Tensor) else v.tolist()) for k, v in res.items() if k != "type"},
            "query": {k: (v if not isinstance(v, torch.Tensor) else v.tolist()) for k, v in res_q.items() if k != "type"}}
    meta["binary"]["type"], meta["query"]["type"] = int(res["type"]), int(res_q["type"])
    save("g8_gather", {"dummy": np.zeros(1)}, meta)


# ---------------------------------------------------------------------------------------- g9
def g9():
    """Program bytecode (the HDF5 question format): the reference's GQAH5Encoder.encode (gqa_preprocess.py:51-94) and
    ProgramDataset.__getitem__/_decode_*/_transform_line (data_pipeline.py:337-453, 593-622), run for real with an
    in-memory stand-in for the h5py.File *container* (h5py is not installed; only dataset storage is stubbed)."""
    import random
    import tempfile
    store = {}

    class FakeFile(object):
        def __init__(self, path, mode="r"):
            self._path = path
            if mode == "w":
                store[path] = {}
            self._d = store[path]

        def create_dataset(self, name, data=None):
            self._d[name] = np.array(data)

        def __getitem__(self, k):
            return self._d[k]

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def close(self):
            pass

    sys.modules["h5py"].File = FakeFile
    for name in ("pattern", "pattern.text", "pattern.text.en"):       # `pattern` (singularize) is not installed; the encoder never calls it
        m = types.ModuleType(name)
        m.__path__ = []
        m.singularize = lambda w: w
        sys.modules.setdefault(name, m)
    sys.path.insert(0, ref_harness.REF_SRC)
    import gqa_preprocess
    Q = lambda i, br, last, ans: {"imageId": "img%03d" % i, "answer": ans, "question": "q%d" % i, "question_id": str(i),
                                  "program": {"branches": br, "last_op": last}}
    files = {
        "exist": [Q(0, [[op("select", "dog"), op("filter", "not(red)"), op("relate", "on", True, "table")]], op("exist"), "yes"),
                  Q(1, [[op("select", "_"), op("relate", "to the left of", False, "_"), op("filter", "small")]], op("exist"), "no")],
        "choose_rel": [Q(2, [[op("select", "man")]], op("choose_rel", ["on", "under"], True, "table"), "on"),
                       Q(3, [[op("select", "cat"), op("filter", "black")]], op("choose_rel", ["near", "behind"], False, "_"), "behind")],
        "verify_attrs": [Q(4, [[op("select", "cup")]], op("verify_attrs", ["red", "glass"]), "yes"),
                         Q(5, [[op("select", "car")]], op("verify_attrs", ["blue"]), "no")],
        "two_same": [Q(6, [[op("select", "dog")], [op("select", "cat"), op("filter", "white")]], op("two_same", "color"), "yes")],
        "compare": [Q(7, [[op("select", "dog")], [op("select", "cat")]], op("compare", "large", True), "dog")],
        "query_attr": [Q(8, [[op("select", "animal"), op("filter", "small")]], op("query_attr", "name"), "cat"),
                       Q(9, [[op("select", "chair")]], op("query_attr", "color"), "red")],
    }
    tmp_in, tmp_out = tempfile.mkdtemp(), tempfile.mkdtemp()
    for name, qs in files.items():
        with open(os.path.join(tmp_in, name + ".json"), "w") as f:
            for q in qs:
                f.write(json.dumps(q) + "\n")
    gqa_preprocess.GQAH5Encoder(ontology).encode(tmp_in, tmp_out)
    arrays, meta = {}, {"source": "gqa_preprocess.py:51-94; data_pipeline.py:337-453,593-622", "files": {}}
    orig_shuffle = ref.data_pipeline.shuffle
    ref.data_pipeline.shuffle = lambda x: None          # the reference shuffles choose-options at load time; goldens keep file order
    for name, qs in files.items():
        path = os.path.join(tmp_out, name + ".h5")
        for k, v in store[path].items():
            arrays[name + ":" + k] = v
        ds = ref.data_pipeline.ProgramDataset(path, ontology, in_memory=True)
        decoded = []
        for i in range(len(ds)):
            r = ds[i]
            decoded.append({"program": r["program"], "image_id": r["image_id"], "answer": r["answer"], "tokens": sorted(map(str, r["tokens"]))})
        jl = ref.data_pipeline.ProgramDataset(copy.deepcopy(qs), ontology, in_memory=True)
        from_json = []
        for i in range(len(jl)):
            r = jl[i]
            from_json.append({"program": r["program"], "image_id": r["image_id"], "answer": r["answer"], "tokens": sorted(map(str, r["tokens"])),
                              "question": r["question"], "question_id": r["question_id"]})
        meta["files"][name] = {"questions": qs, "decoded": decoded, "from_json": from_json}
    ref.data_pipeline.shuffle = orig_shuffle
    save("g9_program_bytecode", arrays, meta)


# ---------------------------------------------------------------------------------------- g16
def g16():
    """The HDF5 containers themselves: the reference's GQAH5Encoder writes REAL .h5 files (h5py is not installed here; its slot in
    sys.modules is taken by dfol_vqa_amd/h5lite.py, a ctypes binding of the same HDF5 C library h5py wraps) and the reference's
    ProgramDataset and BatchGQABoxFeaturesCollator read them back (gqa_preprocess.py:51-94, data_pipeline.py:328-389,
    batch_gqa_boxfeatures_pipeline.py:15-81).  The files are committed under tests/golden/h5/ as data fixtures."""
    import shutil
    import tempfile
    from dfol_vqa_amd import h5lite
    saved = sys.modules.get("h5py")
    sys.modules["h5py"] = h5lite
    ref.data_pipeline.h5py = h5lite
    sys.path.insert(0, ref_harness.REF_SRC)
    for name in ("pattern", "pattern.text", "pattern.text.en"):
        m = types.ModuleType(name)
        m.__path__ = []
        m.singularize = lambda w: w
        sys.modules.setdefault(name, m)
    import gqa_preprocess
    gqa_preprocess.h5py = h5lite
    from nsvqa.data import batch_gqa_boxfeatures_pipeline as bfp
    bfp.h5py = h5lite
    with open(os.path.join(OUT, "g9_program_bytecode.json")) as fh:
        meta9 = json.load(fh)
    files = {k: v["questions"] for k, v in meta9["files"].items()}
    tmp_in, tmp_out = tempfile.mkdtemp(), tempfile.mkdtemp()
    for name, qs in files.items():
        with open(os.path.join(tmp_in, name + ".json"), "w") as f:
            for q in qs:
                f.write(json.dumps(q) + "\n")
    gqa_preprocess.GQAH5Encoder(ontology).encode(tmp_in, tmp_out)
    h5dir = os.path.join(OUT, "h5")
    os.makedirs(h5dir, exist_ok=True)
    meta = {"source": "gqa_preprocess.py:51-94; data_pipeline.py:328-389; batch_gqa_boxfeatures_pipeline.py:15-81", "files": {}}
    orig_shuffle = ref.data_pipeline.shuffle
    ref.data_pipeline.shuffle = lambda x: None
    for name in files:
        shutil.copy(os.path.join(tmp_out, name + ".h5"), os.path.join(h5dir, "ref_" + name + ".h5"))
        ds = ref.data_pipeline.ProgramDataset(os.path.join(h5dir, "ref_" + name + ".h5"), ontology, in_memory=False)
        decoded = []
        for i in range(len(ds)):
            r = ds[i]
            decoded.append({"program": r["program"], "image_id": r["image_id"], "answer": r["answer"], "tokens": sorted(map(str, r["tokens"]))})
        meta["files"][name] = {"decoded": decoded}
    ref.data_pipeline.shuffle = orig_shuffle
    # object-feature chunks: two chunk files of three images, read through the reference's collator
    rng = np.random.RandomState(16)
    F, max_obj = 16, 6
    info = {}
    for c in range(2):
        feats = rng.uniform(0, 1, (3, max_obj, F)).astype(np.float32)
        boxes = np.zeros((3, max_obj, 4), np.float32)
        boxes[..., :2] = rng.uniform(0, 300, (3, max_obj, 2))
        boxes[..., 2:] = boxes[..., :2] + rng.uniform(5, 100, (3, max_obj, 2))
        with h5lite.File(os.path.join(h5dir, "gqa_objects_%d.h5" % c), "w") as f:
            f.create_dataset("features", data=feats)
            f.create_dataset("bboxes", data=boxes)
        for i in range(3):
            info["img%03d" % (3 * c + i)] = {"idx": i, "file": c, "objectsNum": int(rng.randint(1, max_obj + 1)), "width": 640, "height": 480}
    with open(os.path.join(h5dir, "gqa_objects_info.json"), "w") as f:
        json.dump(info, f)
    coll = bfp.BatchGQABoxFeaturesCollator(h5dir, "gqa_objects", 2, os.path.join(h5dir, "gqa_objects_info.json"), ontology, 1)
    order = ["img004", "img000", "img005", "img002"]
    feats, bi = coll.collate_object_features([{"image_id": im} for im in order])
    arrays = {"features": feats.numpy(), "batch_index": bi.numpy()}
    meta["chunks"] = {"order": order, "feature_dim": F, "max_objects": max_obj}
    save("g16_hdf5_containers", arrays, meta)
    if saved is not None:
        sys.modules["h5py"] = saved


# ---------------------------------------------------------------------------------------- g10
def g10():
    """Attention calibration on (activate_attention_transfer): LSTM forward/backward passes + apply_modulations
    (batch_base_interpreter.py:87-140, batch_base_ops.py:407-467,598-684, batch_base_types.py:170-187), reduced dims."""
    sys.path.insert(0, ref_harness.REF_SRC)
    import gqa_interpreter_experiments as gie
    cfg = dict(box_features_dim=32, oracle_input_dim=16, oracle_output_dim=1, word_embedding_dim=mini_ontology.EMBEDDING_DIM,
               classifier_oracle=True, featurizer_layers_config=[], attribute_network_layers_config=[8],
               relation_network_layers_config=[8], operator_layers_config=[], normalize_oracle=True, dropout=0.0,
               freeze_featurizer=True, freeze_attribute_network=True, freeze_relation_network=True,
               freeze_embedding_network=True, activate_attention_transfer=True, attention_transfer_state_dim=6,
               freeze_attention_network=False, trainable_gate=False, likelihood_threshold=0, hard_mode=False,
               verbose=False, model_name="g10", gpu_num=1)
    exp = gie.GQAObjectBoxExperiment()
    exp._local_rank = 0
    torch.manual_seed(3)
    model = exp.build_model(cfg, ontology, None)
    out_net = model._ops["select"]._filter._attention_output_network
    with torch.no_grad():
        out_net[0].weight.normal_(0.0, 0.8)               # the reference starts it at zero; make the modulations state-dependent
    model.eval()
    arrays, meta = {}, {"source": "batch_base_interpreter.py:87-140; batch_base_ops.py:407-467,598-684; batch_base_types.py:170-187",
                        "config": cfg, "runs": {}}
    pre = "_ops.select._filter."
    for k, v in model.state_dict().items():
        if k.startswith("_featurizer.") or k.startswith("_oracle.") or (k.startswith(pre) and "attention" in k):
            arrays["w:" + k] = v.numpy()
    Q = syn.question
    n_of = lambda i: [5, 7, 3, 6][i % 4]

    def mk(i, branches, last, answer="yes"):
        return Q(900 + i, branches, last, answer, syn.feature_scene(900 + i, n_of(i), cfg["box_features_dim"]))

    two = lambda i, last: mk(i, [[op("select", "dog"), op("filter", "red")], [op("select", "cat"), op("relate", "near", bool(i % 2), "table")]], last)
    runs = {
        "exist": [mk(0, [[op("select", "dog")]], op("exist")),
                  mk(1, [[op("select", "cat"), op("filter", "not(red)"), op("relate", "on", True, "table")]], op("exist")),
                  mk(2, [[op("select", "_"), op("relate", "to the left of", False, "car"), op("filter", "small"), op("relate", "near", True, "_")]], op("exist"))],
        "verify_attrs": [mk(3, [[op("select", "dog"), op("filter", "small")]], op("verify_attrs", ["red", "large"])),
                         mk(4, [[op("select", "cup")]], op("verify_attrs", ["glass"]))],
        "choose_attr": [mk(5, [[op("select", "dog")]], op("choose_attr", ["red", "blue"]), "red"),
                        mk(6, [[op("select", "table"), op("filter", "wood")]], op("choose_attr", ["small", "large"]), "large")],
        "query_attr": [mk(7, [[op("select", "dog")]], op("query_attr", "color"), "black"),
                       mk(8, [[op("select", "animal"), op("filter", "small")]], op("query_attr", "name"), "cat")],
        "verify_rel": [mk(9, [[op("select", "dog")]], op("verify_rel", "on", True, "couch")),
                       mk(10, [[op("select", "man"), op("filter", "standing")]], op("verify_rel", "to the right of", False, "bus"))],
        "choose_rel": [mk(11, [[op("select", "dog")]], op("choose_rel", ["to the left of", "to the right of"], True, "cat"), "to the left of"),
                       mk(12, [[op("select", "woman"), op("filter", "large")]], op("choose_rel", ["on", "under"], False, "table"), "on")],
        "and": [two(13, op("and")), two(14, op("and"))],
        "two_same": [two(15, op("two_same", "color")), two(16, op("two_same", "size"))],
        "all_same": [mk(17, [[op("select", "dog")]], op("all_same", "color")), mk(18, [[op("select", "furniture"), op("filter", "wood")]], op("all_same", "name"))],
        "compare": [two(19, op("compare", "large", False)), two(20, op("compare", "red", True))],
    }
    for name, qs in runs.items():
        for i, q in enumerate(qs):
            arrays["%s:X_%d" % (name, i)] = q["scene"]["X"]
        entry = {"questions": questions_to_meta(qs)}
        for dt, tag in both_dtypes():
            m = copy.deepcopy(model).double() if dt == torch.float64 else model
            collater = ref_harness.make_collater(ref, 1, "feature", ontology)
            pbs = collater.collate(copy.deepcopy(qs))
            for pb in pbs:
                pb.create_sparse_tensors()
                if dt == torch.float64:
                    pb.to(torch.float64)
                    pb._object_batch_index = pb._object_batch_index.long()
            with torch.no_grad():
                res = m(pbs, False, return_trace=False, modulator_switch=True)
                res_off = m(pbs, False, return_trace=False, modulator_switch=False)
            arrays["%s:lp_%s" % (name, tag)] = res["log_probability"].numpy()
            arrays["%s:lp_off_%s" % (name, tag)] = res_off["log_probability"].numpy()
            if tag == "f32":
                entry["answer"], entry["type"] = res["answer"], int(res["type"])
        meta["runs"][name] = entry
    save("g10_calibration", arrays, meta)


def g13():
    """Verdicts of the reference's GQAProgramVerifier (nn/parser/parse_utils.py:24-240) on well-formed and malformed programs."""
    for name in ("pattern", "pattern.text", "pattern.text.en"):       # only `normalize` uses singularize; the verifier does not
        m = types.ModuleType(name)
        m.__path__ = []
        m.singularize = lambda w: w
        sys.modules.setdefault(name, m)
    sys.path.insert(0, ref_harness.REF_SRC)
    from nsvqa.nn.parser import parse_utils
    p = paths
    ver = parse_utils.GQAProgramVerifier(p["attribute_file"], p["class_file"], p["vocabulary_file"], p["relation_file"])
    sel = lambda n="dog": op("select", n)
    P = lambda br, last: {"branches": br, "last_op": last}
    two = [[sel("dog"), op("filter", "red")], [sel("cat")]]
    cases = [
        P([[sel(), op("filter", "not(red)"), op("relate", "on", True, "table")]], op("exist")),
        P([[sel("_"), op("relate", "to the left of", False, "_")]], op("exist")),
        P([[sel("scene")]], op("exist")),
        P([[sel("unicorn")]], op("exist")),                                   # noun not in the vocabulary
        P([[sel(), op("filter", "shiny")]], op("exist")),                     # unknown attribute
        P([[sel(), op("relate", "red", True, "table")]], op("exist")),        # not a relation
        P([[sel(), op("relate", "riding", True, "table")]], op("exist")),     # a relation the vocabulary lacks
        P([[sel(), op("relate", "on", 1, "table")]], op("exist")),            # flag must be a bool
        P([[sel(), op("relate", "on", True, "unicorn")]], op("exist")),
        P([[sel(), op("relate", "on", True)]], op("exist")),                  # wrong argument count
        P([[op("filter", "red")]], op("exist")),                              # branch must start with select
        P([[sel(), sel("cat")]], op("exist")),                                # select inside a branch
        P([[sel(), op("exist")]], op("exist")),                               # terminal operator inside a branch
        P([[sel()]], op("filter", "red")),                                    # non-terminal last_op
        P([[sel()]], op("jump")),                                             # unknown operator
        P([[sel()], [sel("cat")]], op("exist")),                              # branch count
        P(two, op("and")), P(two, op("or")), P([[sel()]], op("and")),
        P([[sel()]], op("query_attr", "color")), P([[sel()]], op("query_attr", "name")), P([[sel()]], op("query_attr", "animal")),
        P([[sel()]], op("query_attr", "flavour")),
        P([[sel()]], op("choose_attr", ["red", "blue"])), P([[sel()]], op("choose_attr", ["red"])), P([[sel()]], op("choose_attr", ["red", "shiny"])),
        P([[sel()]], op("verify_attrs", ["red", "not(large)"])), P([[sel()]], op("verify_attrs", [])), P([[sel()]], op("verify_attrs", ["shiny"])),
        P([[sel()]], op("verify_rel", "on", True, "table")), P([[sel()]], op("verify_rel", "over", True, "table")),
        P([[sel()]], op("choose_rel", ["on", "under"], False, "_")), P([[sel()]], op("choose_rel", [], False, "_")),
        P([[sel()]], op("choose_rel", ["on", "red"], False, "_")),
        P([[sel()]], op("all_same", "color")), P([[sel()]], op("all_different", "type")), P([[sel()]], op("all_same", "flavour")),
        P(two, op("two_same", "material")), P(two, op("two_different", "flavour")), P([[sel()]], op("two_same", "color")),
        P(two, op("compare", "large", False)), P(two, op("compare", "not(small)", True)), P(two, op("compare", "shiny", True)),
        P(two, op("compare", "large", 0)),
        {"branches": [[sel()]]}, {"last_op": op("exist")}, P([[{"arguments": []}]], op("exist")), P([[{"operator": "select"}]], op("exist")),
    ]
    verdicts = []
    for prog in cases:
        try:
            verdicts.append(bool(ver.verify(copy.deepcopy(prog))))
        except parse_utils.ParserError:
            verdicts.append(False)
    save("g13_program_verifier", {}, {"source": "nn/parser/parse_utils.py:24-240", "programs": cases, "valid": verdicts})


def g15():
    """The reference's GQA-JSON -> program preprocessor (src/gqa_preprocess.py:98-361) on questions authored here, with an operator map
    authored here: batch format (branches / last_op) and flat format (operators / arguments / dependencies), verify-and merging,
    logical-branch rewriting, argument parsing for every operator family, dropped questions, and the segregated per-line output files.
    `pattern.singularize` is not installed: it is replaced by the identity, and the authored vocabulary only uses words the identity is
    right for or that `normalize` handles itself (its plurale-tantum / irregular lists) - word singularisation itself stays unpinned."""
    import tempfile
    for name in ("pattern", "pattern.text", "pattern.text.en"):
        m = types.ModuleType(name)
        m.__path__ = []
        m.singularize = lambda w: w
        sys.modules.setdefault(name, m)
    if "h5py" not in sys.modules:
        sys.modules["h5py"] = types.ModuleType("h5py")
    sys.path.insert(0, ref_harness.REF_SRC)
    import gqa_preprocess as ref_pre
    op_map = {"select": "select", "filter color": "filter", "filter": "filter", "filter size": "filter", "relate": "relate", "query": "query_attr",
              "verify color": "verify_attrs", "verify size": "verify_attrs", "verify": "verify_attrs", "choose color": "choose_attr",
              "choose rel": "choose_rel", "verify rel": "verify_rel", "exist": "exist", "and": "and", "or": "or", "same color": "two_same",
              "same material shape": "two_same", "different color": "two_different", "same": "all_same", "different": "all_different",
              "choose older": "compare", "choose healthier": "compare", "choose less healthy": "compare", "choose more healthy": "compare",
              "choose larger": "compare", "common": None}
    S = lambda operation, argument, deps: {"operation": operation, "argument": argument, "dependencies": deps}
    Qn = lambda semantic, answer: {"semantic": semantic, "answer": answer, "question": "?", "imageId": "7"}
    questions = {
        "q01": Qn([S("select", "dog (1234)", []), S("filter color", "Brown ", [0]), S("relate", "table,on,s (55)", [1]), S("exist", "?", [2])], "Yes"),
        "q02": Qn([S("select", "shelves (1,2)", []), S("verify color", "red", [0]), S("verify size", "large", [0]), S("and", "", [1, 2])], "no"),
        "q03": Qn([S("select", "cat (3)", []), S("exist", "?", [0]), S("select", "glasses (-)", []), S("exist", "?", [2]), S("and", "", [1, 3])], "yes"),
        "q04": Qn([S("select", "man (9)", []), S("verify rel", "horse,riding,o (10)", [0]), S("select", "woman (11)", []),
                   S("verify color", "blue", [2]), S("or", "", [1, 3])], "no"),
        "q05": Qn([S("select", "cup (4)", []), S("choose color", "red|green", [0])], "Red"),
        "q06": Qn([S("select", "boy (5)", []), S("choose rel", "girl,to the left of|to the right of,s (6)", [0])], "to the left of"),
        "q07": Qn([S("select", "bus (7)", []), S("query", "color", [0])], "yellow"),
        "q08": Qn([S("select", "fork (1)", []), S("select", "spoon (2)", []), S("same color", "", [0, 1])], "no"),
        "q09": Qn([S("select", "fork (1)", []), S("select", "spoon (2)", []), S("different color", "", [0, 1])], "yes"),
        "q10": Qn([S("select", "plate (1)", []), S("same", "color", [0])], "yes"),
        "q11": Qn([S("select", "plate (1)", []), S("different", "shape", [0])], "no"),
        "q12": Qn([S("select", "man (1)", []), S("select", "woman (2)", []), S("choose older", "", [0, 1])], "man"),
        "q13": Qn([S("select", "apple (1)", []), S("select", "cake (2)", []), S("choose healthier", "", [0, 1])], "apple"),
        "q14": Qn([S("select", "apple (1)", []), S("select", "cake (2)", []), S("choose less healthy", "", [0, 1])], "cake"),
        "q15": Qn([S("select", "apple (1)", []), S("select", "cake (2)", []), S("choose more healthy", "", [0, 1])], "apple"),
        "q16": Qn([S("select", "scene", []), S("query", "weather", [0])], "sunny"),
        "q17": Qn([S("select", "dog (1)", []), S("teleport", "x", [0])], "yes"),                  # operator the map does not know
        "q18": Qn([S("select", "dog (1)", []), S("common", "", [0])], "color"),                    # operator mapped to null
        "q19": Qn([S("select", "dress (8)", []), S("filter size", "small", [0]), S("filter color", "not(white)", [1]),
                   S("relate", "_,wearing,o (9)", [2]), S("query", "name", [3])], "girl"),
        "q20": Qn([S("select", "car (1)", []), S("verify color", "red", [0]), S("select", "truck (2)", []), S("verify color", "red", [2]),
                   S("and", "", [1, 3])], "yes"),                                                  # verify-and on two different traces: not merged
        "q21": Qn([S("select", "apple (1)", []), S("select", "cake (2)", []), S("same material shape", "", [0, 1])], "no"),
        "q22": Qn([S("select", "box (1)", []), S("select", "bag (2)", []), S("choose larger", "", [0, 1])], "box"),
        "q23": Qn([S("select", "sky (-) ", []), S("verify", "cloudy", [0])], "yes"),
    }
    tmp = tempfile.mkdtemp()
    map_path = os.path.join(tmp, "op_map.json")
    json.dump(op_map, open(map_path, "w"))
    out = {"op_map": op_map, "questions": questions, "parsed": {}}
    for tag, batch_format in (("batch", True), ("flat", False)):
        pre = ref_pre.GQAPreprocessor(map_path, batch_format)
        for discard in (False, True):
            res = {}
            for qid, q in questions.items():
                res[qid] = pre.parse_question(json.loads(json.dumps(q)), discard)
            out["parsed"]["%s_discard%d" % (tag, int(discard))] = res
    # the file-level driver: segregated per-line outputs
    in_file = os.path.join(tmp, "questions.json")
    json.dump(questions, open(in_file, "w"))
    files = {}
    for seg, by_len in ((True, False), (True, True), (False, False)):
        od = tempfile.mkdtemp()
        ref_pre.GQAPreprocessor(map_path, True).preprocess(in_file, os.path.join(od, "p.json"), seg, by_len, discard_global=True)
        files["seg%d_len%d" % (int(seg), int(by_len))] = {f: [json.loads(l) for l in open(os.path.join(od, f))] for f in sorted(os.listdir(od))}
    out["files"] = files
    with open(os.path.join(OUT, "g15_preprocess.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote g15_preprocess", len(questions), "questions")


# ---------------------------------------------------------------------------------------- g17
G17_KINDS = ["exist", "and", "or", "verify_attrs", "verify_rel", "choose_attr", "query_attr", "choose_rel", "two_same", "two_different",
             "all_same", "all_different", "compare"]
G17_WEIGHT_SEED = 17


def g17():
    """The reference at FULL model size (2048 -> 512, 516 / 1036 -> 256 -> 300 -> 2335 concepts, 333 relation columns)
    on the synthetic full-size ontology: BASELINE configs[1] verbatim (64 questions, 36 objects, select -> filter -> relate -> exist) and
    every terminal operator on ragged scenes of 60..100 objects (configs[2]'s shape).  Stored: seeds, programs, object counts and the
    reference's fp32 and fp64 outputs; weights (synthetic.seeded_weights), ontology (synthetic.write_synthetic_ontology) and scenes
    (synthetic.feature_scene) are regenerated from their seeds.  classifier_oracle.py:145-156, gqa_interpreter_experiments.py:107-198."""
    import tempfile
    import zlib
    sys.path.insert(0, ref_harness.REF_SRC)
    import gqa_interpreter_experiments as gie
    tmp = tempfile.mkdtemp(prefix="dfol_g17_")
    fpaths, names = syn.write_synthetic_ontology(tmp)
    with open(fpaths["vocabulary_file"]) as f:
        vocab = json.load(f)
    with open(fpaths["attribute_file"]) as f:
        categories = json.load(f)
    fpaths["word_embedding_file"] = os.path.join(tmp, "glove.txt")
    rng = np.random.RandomState(3)
    with open(fpaths["word_embedding_file"], "w") as f:          # build_model wants a GloVe file; every weight is overwritten below
        for wd in sorted({x for nme in vocab["idx_to_arg"] for x in nme.split()}):
            f.write(wd + " " + " ".join("%.4f" % x for x in rng.normal(0, 0.3, 300)) + "\n")
    cfg = syn.reference_config(fpaths)
    exp = gie.GQAObjectBoxExperiment()
    exp._local_rank = 0
    full_ontology = exp.build_ontology(cfg, None)
    torch.manual_seed(0)
    model = exp.build_model(cfg, full_ontology, None)
    syn.load_seeded_weights(model, G17_WEIGHT_SEED)
    model.eval()
    model64 = copy.deepcopy(model).double()

    def run(qs, split, dt):
        m = model64 if dt == torch.float64 else model
        collater = ref_harness.make_collater(ref, split, "feature")
        pbs = collater.collate(copy.deepcopy(qs))
        for pb in pbs:
            pb.create_sparse_tensors()
            if dt == torch.float64:
                pb.to(torch.float64)
                pb._object_batch_index = pb._object_batch_index.long()
        with torch.no_grad():
            return m(pbs, False)

    cases = {}
    nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
    c1 = []
    for i in range(64):
        br, last = syn.three_hop_program(i, nouns, attrs, rels)
        c1.append(syn.question(i, br, last, "yes", syn.feature_scene(i, 36, 2048)))
    cases["c1_n36"] = (c1, 8)
    for kind in G17_KINDS:
        seed = zlib.crc32(kind.encode()) % 1000 + 31
        cases["%s_n60_100" % kind] = (syn.full_size_questions(kind, 6, 60, 100, names, categories, seed), 2)
    arrays, meta = {}, {"source": "classifier_oracle.py:145-156; gqa_interpreter_experiments.py:107-198; batch_base_interpreter.py:72-183",
                        "weight_seed": G17_WEIGHT_SEED, "ontology": "synthetic.write_synthetic_ontology (defaults)", "feature_dim": 2048,
                        "torch": torch.__version__, "cases": {}}
    for name, (qs, split) in cases.items():
        cm = {"questions": questions_to_meta(qs), "split": split}
        for dt, tag in both_dtypes():
            res = run(qs, split, dt)
            arrays["%s:lp_%s" % (name, tag)] = res["log_probability"].detach().numpy()
            if tag == "f32":
                cm["answer"], cm["options"], cm["type"] = res["answer"], res["options"], int(res["type"])
        e = np.abs(arrays[name + ":lp_f32"] - arrays[name + ":lp_f64"])
        print(name, "lp range %.3f .. %.3f" % (arrays[name + ":lp_f64"].min(), arrays[name + ":lp_f64"].max()), "ref32 vs ref64 max %.2e" % e.max())
        meta["cases"][name] = cm
    save("g17_full_size", arrays, meta)



# ---------------------------------------------------------------------------------------- g23
G23_KINDS = ["exist", "verify_attrs", "verify_rel", "choose_rel", "query_attr", "and", "two_same", "compare"]
G23_CALIBRATOR_SEED = 29


def g23():
    """The CALIBRATED forward of the reference at FULL model size (activate_attention_transfer: True, config/sample_config.yaml's default: the
    LSTMCell(318 -> 50) walks + Linear(100 -> 4) modulations of batch_base_interpreter.py:87-140 around the full-size oracle) - g10 pins the
    calibration at reduced dims only.  Eight terminal operators x 6 questions on ragged 10..40-object scenes, fp32 + fp64; everything but the
    reference's outputs is regenerated from seeds (synthetic.seeded_weights, load_seeded_calibrator, write_synthetic_glove, feature_scene)."""
    import zlib
    model, names, categories = _full_size_reference("g23", activate_attention_transfer=True)
    ont_full = model._dfol_ontology
    syn.load_seeded_weights(model, G17_WEIGHT_SEED)
    syn.load_seeded_calibrator(model, G23_CALIBRATOR_SEED)
    model.eval()
    model64 = copy.deepcopy(model).double()
    arrays, meta = {}, {"source": "batch_base_interpreter.py:87-140; batch_base_ops.py:407-467,598-684; batch_base_types.py:170-187; "
                                  "gqa_interpreter_experiments.py:115-138", "weight_seed": G17_WEIGHT_SEED, "calibrator_seed": G23_CALIBRATOR_SEED,
                        "glove": "synthetic.write_synthetic_glove (seed 3)", "feature_dim": 2048, "torch": torch.__version__, "cases": {}}
    for kind in G23_KINDS:
        seed = zlib.crc32(kind.encode()) % 1000 + 2300
        qs = syn.full_size_questions(kind, 6, 10, 40, names, categories, seed)
        cm = {"questions": questions_to_meta(qs), "split": 1}
        for dt, tag in both_dtypes():
            m = model64 if dt == torch.float64 else model
            collater = ref_harness.make_collater(ref, 1, "feature", ont_full)
            pbs = collater.collate(copy.deepcopy(qs))
            for pb in pbs:
                pb.create_sparse_tensors()
                if dt == torch.float64:
                    pb.to(torch.float64)
                    pb._object_batch_index = pb._object_batch_index.long()
                    pb._meta_data["embedding"] = pb._meta_data["embedding"].double()
            with torch.no_grad():
                res = m(pbs, False, return_trace=False, modulator_switch=True)
                off = m(pbs, False, return_trace=False, modulator_switch=False)
            arrays["%s:lp_%s" % (kind, tag)] = res["log_probability"].detach().numpy()
            arrays["%s:lp_off_%s" % (kind, tag)] = off["log_probability"].detach().numpy()
            if tag == "f32":
                cm["answer"], cm["options"], cm["type"] = res["answer"], res["options"], int(res["type"])
        e = np.abs(arrays[kind + ":lp_f32"] - arrays[kind + ":lp_f64"])
        d = np.abs(arrays[kind + ":lp_f64"] - arrays[kind + ":lp_off_f64"])
        print(kind, "lp range %.3f .. %.3f" % (arrays[kind + ":lp_f64"].min(), arrays[kind + ":lp_f64"].max()), "ref32 vs ref64 max %.2e" % e.max(),
              "calibrated vs not: max %.3f" % d.max())
        meta["cases"][kind] = cm
    save("g23_calibrated_full_size", arrays, meta)


# ---------------------------------------------------------------------------------------- g24
G24_KINDS = ["exist", "verify_rel", "choose_rel", "query_attr", "and"]


def g24():
    """The reference's `_train_batch` in the CALIBRATOR phases of its curriculum (cur6-7: everything frozen but the two LSTM cells and the
    attention-output layer; trainer.py:181-262, 429-442 over batch_base_interpreter.py:87-140) at FULL model size: loss, log-probabilities and the
    gradient of every calibrator tensor (ten of them; norm + the g19 sample of 4096 entries), fp32 + fp64, for BINARY and QUERY batches on ragged
    10..40-object scenes.  Weights / GloVe / scenes regenerate from seeds as for g23."""
    import zlib
    from nsvqa.train.trainer import VQATrainer
    model, names, categories = _full_size_reference("g24", activate_attention_transfer=True, dropout=0.0)
    ont_full = model._dfol_ontology
    syn.load_seeded_weights(model, G17_WEIGHT_SEED)
    syn.load_seeded_calibrator(model, G23_CALIBRATOR_SEED)
    trainable = sorted(k for k, p in model.named_parameters() if p.requires_grad)
    assert trainable and all("attention" in k for k in trainable), trainable
    fake = types.SimpleNamespace(_device=torch.device("cpu"), _config={})
    arrays, meta = {}, {"source": "trainer.py:181-262,429-442; batch_base_interpreter.py:87-140; batch_base_ops.py:407-467,598-684", "weight_seed": G17_WEIGHT_SEED,
                        "calibrator_seed": G23_CALIBRATOR_SEED, "feature_dim": 2048, "torch": torch.__version__, "cases": {}}
    flt_prefix = "_ops.select._filter."
    for kind in G24_KINDS:
        seed = zlib.crc32(kind.encode()) % 1000 + 2400
        qs = syn.full_size_questions(kind, 6, 10, 40, names, categories, seed)
        rng = np.random.RandomState(seed)
        for q in qs:                                            # answers the loss can score: yes / no, or one of the question's options
            last = q["program"]["last_op"]
            if kind == "choose_rel":
                q["answer"] = last["arguments"][0][rng.randint(2)]
            elif kind == "query_attr":
                q["answer"] = categories[last["arguments"][0]][rng.randint(len(categories[last["arguments"][0]]))]
            else:
                q["answer"] = "yes" if rng.uniform() < 0.5 else "no"
        for dt, tag in both_dtypes():
            m = copy.deepcopy(model).double() if dt == torch.float64 else copy.deepcopy(model)
            m.train()
            collater = ref_harness.make_collater(ref, 1, "feature", ont_full)
            pbs = collater.collate(copy.deepcopy(qs))
            for pb in pbs:
                pb.create_sparse_tensors()
                if dt == torch.float64:
                    pb.to(torch.float64)
                    pb._object_batch_index = pb._object_batch_index.long()
                    pb._meta_data["embedding"] = pb._meta_data["embedding"].double()
            res = m(pbs, True, return_trace=False)
            lp = res["log_probability"]
            if dt == torch.float64:
                if res["type"] == ref.base_types.QuestionType.BINARY:
                    target = torch.tensor([a_ == "yes" for pb in pbs for a_ in pb._answers], dtype=dt)
                    loss = torch.nn.functional.binary_cross_entropy(lp.exp(), target, reduction="sum")
                else:
                    answers = [a_ for pb in pbs for a_ in pb._answers]
                    target = [[a_ == o for o in opt] for a_, opt in zip(answers, res["options"])]
                    seg = torch.tensor([i for i, t in enumerate(target) for _ in t])
                    tflat = torch.tensor([x for t in target for x in t], dtype=dt)
                    denom = torch.zeros(len(target), dtype=dt).index_add(0, seg, lp.exp())
                    loss = ref.util.safe_log(denom).sum() - (tflat * lp).sum()
            else:
                loss = VQATrainer._compute_loss(fake, pbs, res)
            loss = loss / sum(pb.batch_size() for pb in pbs)
            loss.backward()
            arrays["%s:loss_%s" % (kind, tag)] = loss.detach().numpy()
            arrays["%s:lp_%s" % (kind, tag)] = lp.detach().numpy()
            for k, prm in m.named_parameters():
                if k.startswith(flt_prefix) and "attention" in k:
                    g = (torch.zeros_like(prm) if prm.grad is None else prm.grad).detach().numpy()
                    pname = k[len(flt_prefix):]
                    arrays["%s:gn:%s:%s" % (kind, pname, tag)] = np.sqrt((g.astype(np.float64) ** 2).sum())
                    arrays["%s:gs:%s:%s" % (kind, pname, tag)] = g.reshape(-1)[syn.gradient_sample_index(pname, g.size)]
            print(kind, tag, "loss", float(loss), "|g out.weight|", float(arrays["%s:gn:_attention_output_network.0.weight:%s" % (kind, tag)]))
        meta["cases"][kind] = {"questions": questions_to_meta(qs), "type": int(res["type"])}
    save("g24_calibrator_train_step", arrays, meta)


# ---------------------------------------------------------------------------------------- g22
def g22():
    """The argument vocabulary the reference ships (data/metadata/gqa_vocab.json: idx_to_arg, 2335 names - the table columns of SURVEY 8(a) a3):
    the names its authors' preprocessing produced WITH the `pattern` library, which neither container has.  Stored as a plain list: the input
    vectors of the singulariser's consistency test (tests/test_data_path.py: the restated `pattern_singularize` must leave the vocabulary fixed,
    and must still 'need' every entry of the reference's own exception tables, parse_utils.py:9-20)."""
    with open(os.path.join(ref_harness.REF_SRC, "nsvqa", "data", "metadata", "gqa_vocab.json")) as f:
        v = json.load(f)
    with open(os.path.join(OUT, "g22_vocabulary_args.json"), "w") as f:
        json.dump({"source": "src/nsvqa/data/metadata/gqa_vocab.json: idx_to_arg", "args": list(v["idx_to_arg"])}, f)
    print("wrote g22_vocabulary_args", len(v["idx_to_arg"]))


# ---------------------------------------------------------------------------------------- g21
def g21():
    """The reference's batch samplers (data_pipeline.py:787-871): batches never mix files; MultiSetSequencialSampler walks the files in order,
    MultiSetSampler draws the next file with probability proportional to what it has left (torch.multinomial) and a random permutation per
    file - reproducible under torch.manual_seed (CPU generator).  Stored: file lengths, batch size, drop_last, replacement, seed and the
    batches (indices into the concatenation) the reference hands out."""
    dp = ref.data_pipeline
    cases = []
    for lengths, bs, drop in (([5, 3, 7], 3, False), ([5, 3, 7], 3, True), ([1, 40, 2, 9], 4, False), ([64], 10, False)):
        dss = [list(range(n)) for n in lengths]
        def run(make):                # (with drop_last and a remainder the reference's generators run off their file's end: RuntimeError)
            try:
                return [list(map(int, b)) for b in make()]
            except RuntimeError:
                return None
        seq = run(lambda: dp.MultiSetSequencialSampler(dss, bs, drop))
        case = {"lengths": lengths, "batch_size": bs, "drop_last": drop, "sequential": seq, "random": []}
        for seed, repl in ((0, False), (7, False), (11, True)):
            torch.manual_seed(seed)
            smp = dp.MultiSetSampler(dss, bs, drop, replacement=repl)
            case["random"].append({"seed": seed, "replacement": repl, "batches": run(lambda: smp), "len": len(smp)})
        cases.append(case)
    with open(os.path.join(OUT, "g21_samplers.json"), "w") as f:
        json.dump({"source": "data_pipeline.py:787-871", "torch": torch.__version__, "cases": cases}, f, indent=1, sort_keys=True)
    print("wrote g21_samplers", sum(len(c["sequential"] or []) for c in cases), "sequential batches")


# ---------------------------------------------------------------------------------------- g20
G20_QUESTIONS = 3


def _full_size_reference(tag, **cfg_over):
    """The reference's own experiment object at FULL model size on the synthetic ontology, seeded weights to be loaded by the caller
    (gqa_interpreter_experiments.py:83-240)."""
    import tempfile
    sys.path.insert(0, ref_harness.REF_SRC)
    import gqa_interpreter_experiments as gie
    tmp = tempfile.mkdtemp(prefix="dfol_%s_" % tag)
    fpaths, names = syn.write_synthetic_ontology(tmp)
    with open(fpaths["vocabulary_file"]) as f:
        vocab = json.load(f)
    with open(fpaths["attribute_file"]) as f:
        categories = json.load(f)
    # build_model wants a GloVe file (the oracle's weights are overwritten by the caller; the calibrator's token embeddings are read from it)
    fpaths["word_embedding_file"] = syn.write_synthetic_glove(os.path.join(tmp, "glove.txt"), vocab["idx_to_arg"])
    cfg = syn.reference_config(fpaths, **cfg_over)
    exp = gie.GQAObjectBoxExperiment()
    exp._local_rank = 0
    full_ontology = exp.build_ontology(cfg, None)
    torch.manual_seed(0)
    model = exp.build_model(cfg, full_ontology, None)
    model._dfol_ontology = full_ontology
    return model, names, categories


def g20():
    """BASELINE configs[4] verbatim through the reference at FULL model size: 256-object scenes, 8-hop open programs
    select -> (filter -> relate) x 4 -> query_attr(category) with 26 options per question (synthetic.open_program, the generator bench.py's
    c4 workload uses), 3 questions in one ProgramBatch (O = 768 objects, 195 840 ordered pairs: the reference's [pairs, 2335] relation
    intermediate is 1.8 GB in fp32, 3.7 GB in fp64 - what a 64 GB host takes).  Stored: seeds, programs and the reference's fp32 and fp64
    log-probabilities [78], answers, options.  batch_gqa_ops.py:296-310, classifier_oracle.py:84-137, batch_base_ops.py:62-151."""
    model, names, categories = _full_size_reference("g20")
    syn.load_seeded_weights(model, G17_WEIGHT_SEED)
    model.eval()
    nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
    qs = []
    for i in range(G20_QUESTIONS):
        qid = 20000 + i
        br, last = syn.open_program(qid, nouns, attrs, rels, names["categories"][:3], hops=4)
        qs.append(syn.question(qid, br, last, attrs[i % len(attrs)], syn.feature_scene(qid, 256, 2048)))
    arrays, meta = {}, {"source": "batch_gqa_ops.py:296-310,354-390; classifier_oracle.py:84-137,145-156; batch_base_ops.py:62-151; "
                                  "batch_base_interpreter.py:72-183", "weight_seed": G17_WEIGHT_SEED, "feature_dim": 2048,
                        "ontology": "synthetic.write_synthetic_ontology (defaults)", "torch": torch.__version__, "cases": {}}
    cm = {"questions": questions_to_meta(qs), "split": 1}
    for dt, tag in both_dtypes():
        m = copy.deepcopy(model).double() if dt == torch.float64 else model
        collater = ref_harness.make_collater(ref, 1, "feature")
        pbs = collater.collate(copy.deepcopy(qs))
        for pb in pbs:
            pb.create_sparse_tensors()
            if dt == torch.float64:
                pb.to(torch.float64)
                pb._object_batch_index = pb._object_batch_index.long()
        with torch.no_grad():
            res = m(pbs, False)
        arrays["c4_n256:lp_%s" % tag] = res["log_probability"].detach().numpy()
        if tag == "f32":
            cm["answer"], cm["options"], cm["type"] = res["answer"], res["options"], int(res["type"])
            cm["answer_log_probability"] = res["answer_log_probability"]
        del m, pbs, res
    e = np.abs(arrays["c4_n256:lp_f32"] - arrays["c4_n256:lp_f64"])
    print("c4_n256 lp range %.3f .. %.3f" % (arrays["c4_n256:lp_f64"].min(), arrays["c4_n256:lp_f64"].max()), "ref32 vs ref64 max %.2e" % e.max(),
          "in p: %.2e" % np.abs(np.exp(arrays["c4_n256:lp_f32"]) - np.exp(arrays["c4_n256:lp_f64"])).max(), "answers", cm["answer"])
    meta["cases"]["c4_n256"] = cm
    save("g20_c4_open_programs", arrays, meta)


# ---------------------------------------------------------------------------------------- g19
def g19():
    """The reference's own `_train_batch` gradients (trainer.py:181-262, 429-442) at FULL model size (2048 -> 512, 516 / 1036 -> 256 -> 300 ->
    2335 concepts; gqa_interpreter_experiments.py:147-167), dropout 0, every weight trainable, for ragged BINARY and QUERY (choose_rel)
    batches with 0..3 relate hops: loss, log-probabilities, and per weight tensor the gradient's norm plus its values at a fixed sample of
    4096 flat indices (synthetic.gradient_sample_index) - fp32 and fp64 runs of the reference.  Weights, ontology and scenes are
    regenerated from their seeds (synthetic.seeded_weights / write_synthetic_ontology / feature_scene)."""
    import tempfile
    sys.path.insert(0, ref_harness.REF_SRC)
    import gqa_interpreter_experiments as gie
    from nsvqa.train.trainer import VQATrainer
    tmp = tempfile.mkdtemp(prefix="dfol_g19_")
    fpaths, names = syn.write_synthetic_ontology(tmp)
    with open(fpaths["vocabulary_file"]) as f:
        vocab = json.load(f)
    with open(fpaths["attribute_file"]) as f:
        categories = json.load(f)
    fpaths["word_embedding_file"] = os.path.join(tmp, "glove.txt")
    rng = np.random.RandomState(3)
    with open(fpaths["word_embedding_file"], "w") as f:          # build_model wants a GloVe file; every weight is overwritten below
        for wd in sorted({x for nme in vocab["idx_to_arg"] for x in nme.split()}):
            f.write(wd + " " + " ".join("%.4f" % x for x in rng.normal(0, 0.3, 300)) + "\n")
    cfg = syn.reference_config(fpaths, freeze_featurizer=False, freeze_attribute_network=False, freeze_relation_network=False,
                               freeze_embedding_network=False, dropout=0.0)
    exp = gie.GQAObjectBoxExperiment()
    exp._local_rank = 0
    full_ontology = exp.build_ontology(cfg, None)
    torch.manual_seed(0)
    model = exp.build_model(cfg, full_ontology, None)
    syn.load_seeded_weights(model, syn.TRAIN_PARITY_WEIGHT_SEED)
    fake = types.SimpleNamespace(_device=torch.device("cpu"), _config={})
    arrays, meta = {}, {"source": "trainer.py:181-262,429-442; gqa_interpreter_experiments.py:18-77,107-240", "weight_seed": syn.TRAIN_PARITY_WEIGHT_SEED,
                        "ontology": "synthetic.write_synthetic_ontology (defaults)", "torch": torch.__version__, "cases": {}}
    have = os.path.join(OUT, "g19_full_size_train_step.npz")
    if os.path.exists(have) and os.environ.get("G19_RECAPTURE_ALL", "0") != "1":
        # cases captured in an earlier round keep their stored values (a re-run differs in the last bits with the host BLAS's threading):
        # only the cases the fixture does not hold yet are run
        old = np.load(have)
        arrays.update({k: old[k] for k in old.files if k.split(":")[0] in syn.TRAIN_PARITY_CASES})
        with open(have[:-4] + ".json") as f:
            meta["cases"].update({k: v for k, v in json.load(f)["cases"].items() if k in syn.TRAIN_PARITY_CASES})
    for case in sorted(syn.TRAIN_PARITY_CASES):
        if case in meta["cases"]:
            continue
        qs = syn.train_parity_questions(case, names, categories)
        for dt, tag in both_dtypes():
            m = copy.deepcopy(model).double() if dt == torch.float64 else copy.deepcopy(model)
            m.train()
            collater = ref_harness.make_collater(ref, 1, "feature")
            pbs = collater.collate(copy.deepcopy(qs))
            for pb in pbs:
                pb.create_sparse_tensors()
                if dt == torch.float64:
                    pb.to(torch.float64)
                    pb._object_batch_index = pb._object_batch_index.long()
            res = m(pbs, True, return_trace=False)
            lp = res["log_probability"]
            if dt == torch.float64:           # trainer.py:185-194,207-230 restated in fp64 (its targets are built in fp32)
                if res["type"] == ref.base_types.QuestionType.BINARY:
                    target = torch.tensor([a == "yes" for pb in pbs for a in pb._answers], dtype=dt)
                    loss = torch.nn.functional.binary_cross_entropy(lp.exp(), target, reduction="sum")
                else:
                    answers = [a for pb in pbs for a in pb._answers]
                    target = [[a == o for o in opt] for a, opt in zip(answers, res["options"])]
                    seg = torch.tensor([i for i, t in enumerate(target) for _ in t])
                    tflat = torch.tensor([x for t in target for x in t], dtype=dt)
                    denom = torch.zeros(len(target), dtype=dt).index_add(0, seg, lp.exp())
                    loss = ref.util.safe_log(denom).sum() - (tflat * lp).sum()
            else:
                loss = VQATrainer._compute_loss(fake, pbs, res)
            loss = loss / sum(pb.batch_size() for pb in pbs)
            loss.backward()
            arrays["%s:loss_%s" % (case, tag)] = loss.detach().numpy()
            arrays["%s:lp_%s" % (case, tag)] = lp.detach().numpy()
            seen = set()
            for k, prm in m.named_parameters():
                if (k.startswith("_featurizer.") or k.startswith("_oracle.")) and id(prm) not in seen:
                    seen.add(id(prm))
                    g = (torch.zeros_like(prm) if prm.grad is None else prm.grad).detach().numpy().reshape(-1)
                    arrays["%s:gn:%s:%s" % (case, k, tag)] = np.asarray(np.sqrt((g.astype(np.float64) ** 2).sum()))
                    arrays["%s:gs:%s:%s" % (case, k, tag)] = g[syn.gradient_sample_index(k, g.size)]
            print(case, tag, "loss", float(loss), "pairs", sum(q["scene"]["n"] * (q["scene"]["n"] - 1) for q in qs))
        meta["cases"][case] = {"questions": questions_to_meta(qs), "type": int(res["type"])}
    save("g19_full_size_train_step", arrays, meta)


# ---------------------------------------------------------------------------------------- g18
def g18():
    """End to end FROM THE REFERENCE'S FILE FORMATS: questions authored here are written as program-bytecode .h5 files by the reference's
    GQAH5Encoder (gqa_preprocess.py:51-94) and object features as chunk .h5 files + info JSON in the layout its collator reads
    (batch_gqa_boxfeatures_pipeline.py:29-55) - committed under tests/golden/h5/g18_* as data fixtures - and then run through the reference's
    own ProgramDataset -> BatchGQABoxFeaturesCollator -> BatchGQAInterpreter (data_pipeline.py:328-367, 391-453;
    batch_gqa_boxfeatures_pipeline.py:29-92; batch_base_interpreter.py:72-183) with the neural model at reduced dims (16 features per
    object).  Stored: the model's weights, and per program file the reference's log-probabilities (fp32 and fp64), answers, options and
    question type.  (h5py is not installed: its slot is taken by dfol_vqa_amd/h5lite.py, a ctypes binding of the same HDF5 C library.)"""
    import shutil
    import tempfile
    from dfol_vqa_amd import h5lite
    saved = sys.modules.get("h5py")
    sys.modules["h5py"] = h5lite
    ref.data_pipeline.h5py = h5lite
    sys.path.insert(0, ref_harness.REF_SRC)
    for name in ("pattern", "pattern.text", "pattern.text.en"):
        m = types.ModuleType(name)
        m.__path__ = []
        m.singularize = lambda w: w
        sys.modules.setdefault(name, m)
    import gqa_preprocess
    gqa_preprocess.h5py = h5lite
    from nsvqa.data import batch_gqa_boxfeatures_pipeline as bfp
    bfp.h5py = h5lite
    import gqa_interpreter_experiments as gie
    h5dir = os.path.join(OUT, "h5")
    F, max_obj, per_chunk, n_img = 16, 8, 8, 24
    # ---- questions: eight terminal operators x six questions, 1..3 hops, negations, '_' names, second branches
    rng = np.random.RandomState(18)
    nouns = [n for v in mini_ontology.CLASSES.values() for n in v]
    attrs = [a_ for v in mini_ontology.ATTRIBUTES.values() for a_ in v]
    rels = [r for r in mini_ontology.RELATIONS if r != "riding"]
    pick = lambda xs: xs[rng.randint(len(xs))]
    Q = lambda i, im, br, last, ans: {"imageId": "img%03d" % im, "answer": ans, "question": "q%d" % i, "question_id": str(i),
                                      "program": {"branches": br, "last_op": last}}

    def branch():
        b_ = [op("select", pick(nouns + ["_"]))]
        for _ in range(rng.randint(1, 4)):
            if rng.uniform() < 0.5:
                a_ = pick(attrs)
                b_.append(op("filter", "not(%s)" % a_ if rng.uniform() < 0.25 else a_))
            else:
                b_.append(op("relate", pick(rels), bool(rng.uniform() < 0.5), pick(nouns + ["_"])))
        return b_

    files, qid = {}, 100
    for kind in ("exist", "verify_rel", "verify_attrs", "choose_attr", "choose_rel", "query_attr", "and", "two_same"):
        qs = []
        for j in range(6):
            im = 10 + int(rng.randint(n_img))
            cat = pick(sorted(mini_ontology.ATTRIBUTES))
            br = [branch()]
            if kind in ("and", "two_same"):
                br.append([op("select", pick(nouns)), op("filter", pick(attrs))])
            last, ans = {
                "exist": (op("exist"), pick(["yes", "no"])),
                "and": (op("and"), pick(["yes", "no"])),
                "verify_rel": (op("verify_rel", pick(rels), bool(rng.uniform() < 0.5), pick(nouns)), pick(["yes", "no"])),
                "verify_attrs": (op("verify_attrs", [pick(attrs)] + ([pick(attrs)] if j % 2 else [])), pick(["yes", "no"])),
                "choose_attr": (op("choose_attr", list(mini_ontology.ATTRIBUTES[cat][:2])), mini_ontology.ATTRIBUTES[cat][0]),
                "choose_rel": (op("choose_rel", [rels[j % 3], rels[3 + j % 3]], bool(rng.uniform() < 0.5), pick(nouns + ["_"])), rels[j % 3]),
                "query_attr": (op("query_attr", pick(["color", "material", "name"])), pick(attrs)),
                "two_same": (op("two_same", pick(["color", "material"])), pick(["yes", "no"])),
            }[kind]
            qs.append(Q(qid, im, br, last, ans))
            qid += 1
        files["g18_" + kind] = qs
    tmp_in, tmp_out = tempfile.mkdtemp(), tempfile.mkdtemp()
    for name, qs in files.items():
        with open(os.path.join(tmp_in, name + ".json"), "w") as f:
            for q in qs:
                f.write(json.dumps(q) + "\n")
    gqa_preprocess.GQAH5Encoder(ontology).encode(tmp_in, tmp_out)
    for name in files:
        shutil.copy(os.path.join(tmp_out, name + ".h5"), os.path.join(h5dir, name + ".h5"))
    # ---- object features: three chunk files of eight images (batch_gqa_boxfeatures_pipeline.py:29-33, 52-55)
    info = {}
    for c in range(n_img // per_chunk):
        feats = rng.uniform(0, 1, (per_chunk, max_obj, F)).astype(np.float32)
        boxes = np.zeros((per_chunk, max_obj, 4), np.float32)
        boxes[..., :2] = rng.uniform(0, 300, (per_chunk, max_obj, 2))
        boxes[..., 2:] = boxes[..., :2] + rng.uniform(5, 100, (per_chunk, max_obj, 2))
        with h5lite.File(os.path.join(h5dir, "g18_objects_%d.h5" % c), "w") as f:
            f.create_dataset("features", data=feats)
            f.create_dataset("bboxes", data=boxes)
        for i in range(per_chunk):
            info["img%03d" % (10 + per_chunk * c + i)] = {"idx": i, "file": c, "objectsNum": int(rng.randint(3, max_obj + 1)), "width": 640, "height": 480}
    with open(os.path.join(h5dir, "g18_objects_info.json"), "w") as f:
        json.dump(info, f)
    # ---- the reference's model at reduced dims
    cfg = dict(box_features_dim=F, oracle_input_dim=16, oracle_output_dim=1, word_embedding_dim=mini_ontology.EMBEDDING_DIM,
               classifier_oracle=True, featurizer_layers_config=[], attribute_network_layers_config=[8],
               relation_network_layers_config=[8], operator_layers_config=[], normalize_oracle=True, dropout=0.0,
               freeze_featurizer=False, freeze_attribute_network=False, freeze_relation_network=False,
               freeze_embedding_network=False, activate_attention_transfer=False, attention_transfer_state_dim=0,
               freeze_attention_network=False, trainable_gate=False, likelihood_threshold=0, hard_mode=False,
               verbose=False, model_name="g18", gpu_num=1)
    exp = gie.GQAObjectBoxExperiment()
    exp._local_rank = 0
    torch.manual_seed(18)
    model = exp.build_model(cfg, ontology, None)
    with torch.no_grad():                                     # informative likelihoods: a fresh embedding layer saturates every concept
        lin = model._oracle._embedding_network._network[1]
        lin.weight.normal_(0.0, 0.6)
        lin.bias.fill_(-1.0)
    model.eval()
    model64 = copy.deepcopy(model).double()
    arrays, meta = {}, {"source": "gqa_preprocess.py:51-94; data_pipeline.py:328-367,391-453; batch_gqa_boxfeatures_pipeline.py:29-92; "
                                  "batch_base_interpreter.py:72-183",
                        "config": cfg, "h5_dir": "tests/golden/h5", "feature_prefix": "g18_objects", "chunk_num": n_img // per_chunk,
                        "info": "g18_objects_info.json", "torch": torch.__version__, "files": {}}
    for k, v in model.state_dict().items():
        if k.startswith("_featurizer.") or k.startswith("_oracle."):
            arrays["w:" + k] = v.numpy().copy()
    orig_shuffle = ref.data_pipeline.shuffle
    ref.data_pipeline.shuffle = lambda x: None                # choose-options stay in file order (data_pipeline.py:596-597 shuffles them)
    coll = bfp.BatchGQABoxFeaturesCollator(h5dir, "g18_objects", n_img // per_chunk, os.path.join(h5dir, "g18_objects_info.json"), ontology, 1)
    for name in sorted(files):
        # The dataset gets its OWN copy of the ontology, as it has in the reference's DataLoader worker processes: its token extraction
        # appends the category name to the list GQAOntology.query returns - the ontology's own list (data_pipeline.py:487-488, 523-539) - so
        # with one shared object the interpreter's query_attr would see option lists that grow by one bogus entry per decoded question.
        ds = ref.data_pipeline.ProgramDataset(os.path.join(h5dir, name + ".h5"), copy.deepcopy(ontology), in_memory=False)
        items = [ds[i] for i in range(len(ds))]
        fm = {"image_ids": [it["image_id"] for it in items], "programs": [it["program"] for it in items], "gold": [it["answer"] for it in items]}
        for dt, tag in both_dtypes():
            pbs = coll.collate(copy.deepcopy(items))
            for pb in pbs:
                pb.create_sparse_tensors()
                if dt == torch.float64:
                    pb.to(torch.float64)
                    pb._object_batch_index = pb._object_batch_index.long()
            with torch.no_grad():
                res = (model64 if dt == torch.float64 else model)(pbs, False)
            arrays["%s:lp_%s" % (name, tag)] = res["log_probability"].detach().numpy()
            if tag == "f32":
                fm["answer"], fm["options"], fm["type"] = res["answer"], res["options"], int(res["type"])
                fm["objects"] = [int(x) for x in torch.bincount(pbs[0]._object_batch_index).tolist()]
        e = np.abs(arrays[name + ":lp_f32"] - arrays[name + ":lp_f64"])
        print(name, "lp", np.round(arrays[name + ":lp_f64"], 2), "ref32 vs ref64 max %.2e" % e.max())
        meta["files"][name] = fm
    ref.data_pipeline.shuffle = orig_shuffle
    save("g18_h5_end_to_end", arrays, meta)
    if saved is not None:
        sys.modules["h5py"] = saved


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15"]
    for w in which:
        globals()[w]()

