"""The native executor (include/dfol_vqa.h: dfol_run_program; dfol_vqa_amd/native_plan.py + native_exec.py) against the Python operator loop
it replaces (batch_base_interpreter.py:145-172 restated in interpreter.py): the same kernels with the same arguments, so every output is
compared BIT FOR BIT - log-probabilities, answers, answer log-probabilities, options, type, variable_sets_num - and, through the golden
tests of test_interpreter_gpu.py (which now run on the executor by default), against the reference itself."""

import json
import os
import sys
import zlib

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu  # noqa: E402
import dfol_vqa_amd as D  # noqa: E402
from dfol_vqa_amd import _lib, native_exec, native_plan  # noqa: E402
from dfol_vqa_amd import synthetic as syn  # noqa: E402
from oracle import dfol_oracle as orc  # noqa: E402
from test_interpreter_gpu import DEV, TableCollater, neural_model  # noqa: E402

pytestmark = pytest.mark.gpu
ALL_KINDS = ["exist", "and", "or", "verify_attrs", "verify_rel", "choose_attr", "query_attr", "choose_rel", "two_same", "two_different", "all_same",
             "all_different", "compare"]


@pytest.fixture(scope="module")
def full(tmp_path_factory):
    from dfol_vqa_amd import experiment
    d = str(tmp_path_factory.mktemp("native"))
    paths, names = syn.write_synthetic_ontology(d)
    cfg = syn.reference_config(paths)
    ont = experiment.build_ontology(cfg)
    model = experiment.build_model(cfg, ont)
    syn.load_seeded_weights(model, 23)
    with open(paths["attribute_file"]) as f:
        categories = json.load(f)
    oont = orc.Ontology(paths["attribute_file"], paths["class_file"], paths["vocabulary_file"], paths["relation_file"])
    return model.to(DEV).eval(), ont, oont, names, categories


def both_routes(model, ont, qs, split=1, monkeypatch=None, expect_native=True, collater=None):
    """The same questions through the native executor and through the Python loop (DFOL_NATIVE=0)."""
    out = []
    for native in ("1", "0"):
        monkeypatch.setenv("DFOL_NATIVE", native)
        pbs = (collater or TableCollater(split, ont, "X")).collate([dict(q) for q in qs])
        for pb in pbs:
            pb.create_sparse_tensors()
        pbs = [pb.to_cuda(DEV) for pb in pbs]
        _lib.PATH_COUNTS.clear()
        with torch.no_grad():
            res = model(pbs, False)
        taken = _lib.PATH_COUNTS.get("native_program", 0)
        assert taken == (len(pbs) if (native == "1" and expect_native) else 0), (native, taken, len(pbs))
        out.append(res)
    return out


def same_results(a, b, what=""):
    assert torch.equal(a["log_probability"], b["log_probability"]), (what, (a["log_probability"] - b["log_probability"]).abs().max().item())
    assert a["answer"] == b["answer"], what
    assert a["answer_log_probability"] == b["answer_log_probability"], what
    assert [list(o) if isinstance(o, (list, tuple)) else o for o in a["options"]] == [list(o) if isinstance(o, (list, tuple)) else o for o in b["options"]], what
    assert int(a["type"]) == int(b["type"]) and a["variable_sets_num"] == b["variable_sets_num"] and a["cumulative_loss"] == b["cumulative_loss"], what


@pytest.mark.parametrize("kind", ALL_KINDS)
def test_native_equals_python_loop_full_size(full, kind, monkeypatch):
    """Every terminal operator, ragged 20..64-object scenes, 1..3 filter / relate hops of differing lengths (masks, no-op tokens), `_`
    names and negated tokens, two ProgramBatches: executor == Python loop, bit for bit."""
    model, ont, oont, names, categories = full
    seed = zlib.crc32(kind.encode()) % 1000 + 77
    qs = syn.full_size_questions(kind, 10, 20, 64, names, categories, seed)
    nat, py = both_routes(model, ont, qs, split=2, monkeypatch=monkeypatch)
    same_results(nat, py, kind)


def test_native_bench_workload(full, monkeypatch):
    """BASELINE configs[1]'s program (select -> filter -> relate -> exist) at 36 objects and a uniform 100-object batch."""
    model, ont, oont, names, categories = full
    nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
    for n, count in ((36, 32), (100, 16)):
        qs = []
        for i in range(count):
            br, last = syn.three_hop_program(i, nouns, attrs, rels, negate_prob=0.2)
            qs.append(syn.question(i, br, last, "yes", syn.feature_scene(i, n, 2048)))
        nat, py = both_routes(model, ont, qs, monkeypatch=monkeypatch)
        same_results(nat, py, "three-hop n=%d" % n)


def test_native_open_programs_and_implicit_end(full, monkeypatch):
    """8-hop open programs (BASELINE configs[4]'s shape at small N) and a program without a terminal operator (the appended `end`)."""
    model, ont, oont, names, categories = full
    nouns, attrs, rels, cats = names["nouns"][:8], names["attributes"][:6], names["relations"][:5], sorted(categories)[:3]
    qs = []
    for i in range(6):
        br, last = syn.open_program(i, nouns, attrs, rels, cats, hops=4)
        qs.append(syn.question(500 + i, br, last, "x", syn.feature_scene(500 + i, 12 + i, 2048)))
    nat, py = both_routes(model, ont, qs, monkeypatch=monkeypatch)
    same_results(nat, py, "open")
    # implicit `end`: a ProgramBatch whose last operator batch is a (ragged, masked) relate - the interpreter appends `end` to the operator's own
    # un-gated result (batch_gqa_interpreter.py:75-76)
    qs = [syn.question(600 + i, [[syn.op("select", nouns[i]), syn.op("filter", attrs[i])] + ([syn.op("relate", rels[0], True, nouns[2])] if i % 2 else [])],
                       syn.op("exist"), "yes", syn.feature_scene(600 + i, 7 + i, 2048)) for i in range(4)]
    out = []
    for native in ("1", "0"):
        monkeypatch.setenv("DFOL_NATIVE", native)
        pb = TableCollater(1, ont, "X").collate([dict(q) for q in qs])[0]
        pb2 = D.ProgramBatch(pb.device, pb._op_batch_list[:-1], pb._dependencies[:-1], pb._answers, pb._object_features, pb._object_batch_index,
                             pb._original_dicts, pb._meta_data)
        pb2.create_sparse_tensors()
        _lib.PATH_COUNTS.clear()
        with torch.no_grad():
            out.append(model([pb2.to_cuda(DEV)], False))
        assert _lib.PATH_COUNTS.get("native_program", 0) == (1 if native == "1" else 0)
    same_results(out[0], out[1], "implicit end")
    assert int(out[0]["type"]) == int(D.QuestionType.STATEMENT)


def test_same_relation_with_both_orientations_in_one_batch(full, monkeypatch):
    """Two relate operators naming the SAME relations with different subject flags (their lowered token lists are one memoised object):
    each must read tiles of its own orientation.  Both routes against the oracle (round 5 fixed the Python route's tile lookup)."""
    model, ont, oont, names, categories = full
    nouns, rels = names["nouns"][:4], names["relations"][:3]
    qs = [syn.question(700 + i, [[syn.op("select", nouns[i % 4]), syn.op("relate", rels[i % 3], True, nouns[(i + 1) % 4]),
                                  syn.op("relate", rels[i % 3], False, nouns[(i + 2) % 4])]], syn.op("exist"), "yes", syn.feature_scene(700 + i, 9 + i, 2048))
          for i in range(3)]
    nat, py = both_routes(model, ont, qs, monkeypatch=monkeypatch)
    same_results(nat, py, "orientations")
    weights = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if k.startswith("_featurizer.") or k.startswith("_oracle.")}
    plain = [{k: v for k, v in q.items() if k != "scene"} for q in qs]
    scenes = [q["scene"] for q in qs]
    r32 = orc.run_questions(oont, plain, scenes, np.float32, weights=weights)
    r64 = orc.run_questions(oont, plain, scenes, np.float64, weights=weights)
    for res in (nat, py):
        gu.check_logprob(res["log_probability"].cpu().numpy(), r32["log_probability"], r64["log_probability"], "orientations")


def test_native_reduced_dims_models(monkeypatch, mini_ontology_paths):
    """The g5 model (8-wide hidden layers: the fp32 dense kernel and the un-packed pair kernel) - the executor's other dispatch arms."""
    p = mini_ontology_paths
    ont = D.GQAOntology(p["attribute_file"], p["class_file"], p["vocabulary_file"], p["word_embedding_file"], relation_json_path=p["relation_file"])
    a, meta = gu.load("g5_neural_oracle")
    weights = {k[2:]: a[k] for k in a.files if k.startswith("w:")}
    model = neural_model(ont, meta["config"], weights)
    qs, scenes = gu.questions_and_scenes(a, meta, "X")
    qq = [dict(q, scene=s) for q, s in zip(qs, scenes)]
    nat, py = both_routes(model, ont, qq, monkeypatch=monkeypatch)
    same_results(nat, py, "g5")
    gu.check_logprob(nat["log_probability"].cpu().numpy(), a["lp_f32"], a["lp_f64"], "g5 native")


def test_plan_built_at_collate_time_travels_through_pickle(full, monkeypatch):
    """A collater that was given the model's spec lowers every ProgramBatch in the worker; the plan survives pickling (DataLoader workers)
    and to_cuda, and the interpreter uses it as it is."""
    import pickle
    model, ont, oont, names, categories = full
    spec = native_exec.model_spec(model)
    assert spec is not None
    qs = syn.full_size_questions("choose_attr", 8, 10, 30, names, categories, 5)
    coll = D.ProgramCollaterBase.__new__(TableCollater)
    TableCollater.__init__(coll, 2, ont, "X")
    coll._native_spec = spec
    pbs = pickle.loads(pickle.dumps(coll.collate([dict(q) for q in qs])))
    assert all(isinstance(pb._native_plan, native_plan.NativePlan) for pb in pbs)
    plans = [pb._native_plan for pb in pbs]
    pbs = [pb.to_cuda(DEV) for pb in pbs]
    assert [pb._native_plan for pb in pbs] == plans
    monkeypatch.setenv("DFOL_NATIVE", "1")
    _lib.PATH_COUNTS.clear()
    with torch.no_grad():
        res = model(pbs, False)
        pend = model.forward_async(pbs, False)
        res2 = pend.result()
    assert _lib.PATH_COUNTS.get("native_program", 0) == 4
    monkeypatch.setenv("DFOL_NATIVE", "0")
    with torch.no_grad():
        ref = model(pbs, False)
    same_results(res, ref, "collate-time plan")
    same_results(res2, ref, "collate-time plan, forward_async")


def test_native_sees_new_weights(full, monkeypatch):
    """The C view of the model is keyed on the parameters' versions: an in-place update is seen by the next forward."""
    model, ont, oont, names, categories = full
    qs = syn.full_size_questions("exist", 4, 10, 20, names, categories, 9)
    pbs = [pb.to_cuda(DEV) for pb in TableCollater(1, ont, "X").collate([dict(q) for q in qs])]
    monkeypatch.setenv("DFOL_NATIVE", "1")
    with torch.no_grad():
        before = model(pbs, False)["log_probability"].clone()
        lin = model._oracle._attribute_network._network[1]
        saved = lin.weight.detach().clone()
        lin.weight.mul_(0.5)
        after = model(pbs, False)["log_probability"].clone()
        monkeypatch.setenv("DFOL_NATIVE", "0")
        ref = model(pbs, False)["log_probability"].clone()
        lin.weight.copy_(saved)
    assert not torch.equal(before, after) and torch.equal(after, ref)


@pytest.mark.parametrize("native", ["1", "0"])
def test_fp16_range_overflow_raises_instead_of_nan(full, native, monkeypatch):
    """An object feature of 1e5 does not fit the two UNSCALED fp16 pieces of the default dense arithmetic: the reference accepts any fp32
    feature (batch_gqa_boxfeatures_pipeline.py:199-213), so the forward raises - on the executor and on the Python loop, from forward() and
    from forward_async().result() - instead of answering with NaN; the next clean batch runs normally (the flag is cleared), and
    `mlp_math: bf16x3` (fp32's exponent range) takes the same features without complaint."""
    model, ont, oont, names, categories = full
    monkeypatch.setenv("DFOL_NATIVE", native)
    qs = syn.full_size_questions("exist", 4, 10, 20, names, categories, 21)
    good = [pb.to_cuda(DEV) for pb in TableCollater(1, ont, "X").collate([dict(q) for q in qs])]
    bad_qs = [dict(q, scene=dict(q["scene"], X=q["scene"]["X"].copy())) for q in qs]
    bad_qs[1]["scene"]["X"][3, 100] = 1.0e5
    bad = [pb.to_cuda(DEV) for pb in TableCollater(1, ont, "X").collate(bad_qs)]
    with torch.no_grad():
        ref = model(good, False)["log_probability"].clone()
        with pytest.raises(_lib.DfolError, match="bf16x3"):
            model(bad, False)
        with pytest.raises(_lib.DfolError, match="fp16 range"):
            model.forward_async(bad, False).result()
        again = model(good, False)["log_probability"]
        assert torch.equal(ref, again)                           # the flag was cleared; a clean batch is a clean batch
        model._mlp_math = "bf16x3"
        try:
            lp = model(bad, False)["log_probability"]
        finally:
            model._mlp_math = None
        assert bool(torch.isfinite(lp).all())


@pytest.mark.parametrize("native", ["1", "0"])
def test_pair_kernel_saturation_raises(full, native, monkeypatch):
    """the fused pair kernel clamps its ELU outputs at 4.16e4 (fp16 pieces).  First-layer sums scaled to ~3e5 used to come back
    as a silently wrong relation tile; now the bound kernel in front of it flags DFOL_RANGE_PAIR_SATURATED and the forward raises.  A model
    whose sums stay in range is not flagged (the same batch, unscaled), on either route."""
    from torch import nn
    model, ont, oont, names, categories = full
    monkeypatch.setenv("DFOL_NATIVE", native)
    qs = syn.full_size_questions("exist", 4, 10, 20, names, categories, 33)
    pbs = [pb.to_cuda(DEV) for pb in TableCollater(1, ont, "X").collate([dict(q) for q in qs])]
    lin1 = [m for m in model._oracle._relation_network._network if isinstance(m, nn.Linear)][0]
    with torch.no_grad():
        ref = model(pbs, False)["log_probability"].clone()
        saved_w, saved_b = lin1.weight.clone(), lin1.bias.clone()
        try:
            lin1.weight.mul_(1.0e6)                              # sums of ~3e5 (the features are Sigmoid outputs: within fp16's range themselves)
            lin1.bias.mul_(1.0e6)
            with pytest.raises(_lib.DfolError, match="saturation"):
                model(pbs, False)
        finally:
            lin1.weight.copy_(saved_w)
            lin1.bias.copy_(saved_b)
        assert torch.equal(model(pbs, False)["log_probability"], ref)


def test_range_flag_belongs_to_its_own_forward(full, monkeypatch):
    """with two forwards in flight the status word is cleared on the stream behind each forward's copy, so an overflow flagged by
    batch A does not also condemn batch B queued behind it."""
    model, ont, oont, names, categories = full
    qs = syn.full_size_questions("exist", 4, 10, 20, names, categories, 21)
    good = [pb.to_cuda(DEV) for pb in TableCollater(1, ont, "X").collate([dict(q) for q in qs])]
    bad_qs = [dict(q, scene=dict(q["scene"], X=q["scene"]["X"].copy())) for q in qs]
    bad_qs[1]["scene"]["X"][3, 100] = 1.0e5
    bad = [pb.to_cuda(DEV) for pb in TableCollater(1, ont, "X").collate(bad_qs)]
    with torch.no_grad():
        ref = model(good, False)["log_probability"].clone()
        a = model.forward_async(bad, False)
        b = model.forward_async(good, False)
        with pytest.raises(_lib.DfolError, match="fp16 range"):
            a.result()
        assert torch.equal(b.result()["log_probability"], ref)


# ---- round 6: what real runs look like stays on the executor -----------------------------------------------------------------
@pytest.mark.parametrize("kind", ["exist", "verify_rel", "choose_rel", "choose_attr", "query_attr", "two_same"])
def test_native_shared_scenes(full, kind, monkeypatch):
    """24 questions on 4 images collated with share_scenes=True (one copy of every image; image-level pair-kernel requests, per-operator tile
    gathers): executor == Python loop bit for bit, also split over two ProgramBatches, and == the per-question layout."""
    model, ont, oont, names, categories = full
    qs = syn.full_size_questions(kind, 24, 10, 20, names, categories, 300 + ALL_KINDS.index(kind))
    images = [syn.feature_scene(9100 + i, n, 2048) for i, n in enumerate((17, 40, 9, 28))]
    pick = np.random.RandomState(7).randint(0, 4, size=len(qs))
    for q, i in zip(qs, pick):
        q["image_id"], q["scene"] = "img%d" % i, images[i]
    for split in (1, 2):
        nat, py = both_routes(model, ont, qs, monkeypatch=monkeypatch, collater=TableCollater(split, ont, "X", share_scenes=True))
        same_results(nat, py, "shared %s split %d" % (kind, split))
        monkeypatch.setenv("DFOL_NATIVE", "1")
        per_q = [pb.to_cuda(DEV) for pb in TableCollater(split, ont, "X").collate([dict(q) for q in qs])]
        with torch.no_grad():
            ref = model(per_q, False)
        same_results(nat, ref, "shared vs per-question %s" % kind)


@pytest.mark.parametrize("kind", ["exist", "verify_rel", "choose_rel", "and"])
@pytest.mark.parametrize("n_list", [[1, 9, 14, 1, 6], [1, 1, 1]])
def test_native_one_object_images(full, kind, n_list, monkeypatch):
    """Real GQA scenes have 1..100 objects: an image of ONE object has no pairs (its relate posteriors come from absent tiles); a batch of
    such images requests no pair kernel at all.  Executor == Python loop, and against the oracle."""
    model, ont, oont, names, categories = full
    qs = syn.full_size_questions(kind, len(n_list), 10, 20, names, categories, 410 + ALL_KINDS.index(kind))
    for q, n in zip(qs, n_list):
        q["scene"] = syn.feature_scene(q["question_id"], n, 2048)
    nat, py = both_routes(model, ont, qs, monkeypatch=monkeypatch)
    same_results(nat, py, "one-object %s %s" % (kind, n_list))
    weights = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if k.startswith("_featurizer.") or k.startswith("_oracle.")}
    plain = [{k: v for k, v in q.items() if k != "scene"} for q in qs]
    scenes = [q["scene"] for q in qs]
    r32 = orc.run_questions(oont, plain, scenes, np.float32, weights=weights)
    r64 = orc.run_questions(oont, plain, scenes, np.float64, weights=weights)
    gu.check_logprob(nat["log_probability"].cpu().numpy(), r32["log_probability"], r64["log_probability"], "one-object %s" % kind)


@pytest.mark.parametrize("kind", ["choose_rel", "choose_attr", "verify_attrs"])
def test_native_noop_token_inside_an_option_list(full, kind, monkeypatch):
    """A no-op token (`_`) among a question's options.  (The reference itself cannot run this shape: with more predicates than questions its
    row restore indexes the [Q, O] attention with a [P] mask, batch_base_ops.py:385 / :563-564, an IndexError - so there is no oracle to
    compare with.)  The Python operators normalise the COMPRESSED list and put default blocks back (classifier_oracle.py:56-60, 72-75 read
    that way); the executor does the same with two row gathers around the normalisation: == Python loop bit for bit, and finite."""
    model, ont, oont, names, categories = full
    qs = syn.full_size_questions(kind, 6, 10, 20, names, categories, 520 + ALL_KINDS.index(kind))
    qs[0]["program"]["last_op"]["arguments"][0][1] = "_"
    qs[3]["program"]["last_op"]["arguments"][0][0] = "_"
    nat, py = both_routes(model, ont, qs, monkeypatch=monkeypatch)
    same_results(nat, py, "no-op option %s" % kind)
    assert bool(torch.isfinite(nat["log_probability"]).all())


@pytest.mark.parametrize("shared", [False, True])
def test_native_bf16_relation_tiles(full, shared, monkeypatch):
    """relation_tile_dtype: bf16 (BASELINE configs[4]) on the executor: the pair kernel writes bf16 tiles, dfol_relate_one_fwd_bf16 reads them -
    NS % 8 == 0 batches; others, and batches with a choose_rel, keep fp32 tiles - bit for bit the Python loop, per question and shared."""
    model, ont, oont, names, categories = full
    saved = model._oracle._tile_dtype
    model._oracle._tile_dtype = torch.bfloat16
    model.__dict__.pop("_native_model", None)
    try:
        for kind, n_lo, n_hi, expect_bf16 in (("exist", 40, 40, True), ("verify_rel", 33, 40, True), ("exist", 30, 36, False), ("choose_rel", 40, 40, False)):
            qs = syn.full_size_questions(kind, 8, n_lo, n_hi, names, categories, 630 + ALL_KINDS.index(kind))
            qs[0]["scene"] = syn.feature_scene(qs[0]["question_id"], n_hi, 2048)       # (the largest image fixes NS)
            if shared:
                for i, q in enumerate(qs):
                    q["image_id"], q["scene"] = "img%d" % (i % 3), qs[i % 3]["scene"]
            coll = TableCollater(1, ont, "X", share_scenes=True) if shared else None
            nat, py = both_routes(model, ont, qs, monkeypatch=monkeypatch, collater=coll)
            same_results(nat, py, "bf16 tiles %s" % kind)
            model._oracle._tile_dtype = torch.float32
            monkeypatch.setenv("DFOL_NATIVE", "1")
            pbs = [pb.to_cuda(DEV) for pb in (coll or TableCollater(1, ont, "X")).collate([dict(q) for q in qs])]
            with torch.no_grad():
                f32 = model(pbs, False)
            model._oracle._tile_dtype = torch.bfloat16
            assert torch.equal(f32["log_probability"], nat["log_probability"]) != expect_bf16, (kind, n_lo, n_hi)
            assert (f32["log_probability"] - nat["log_probability"]).abs().max().item() <= 5e-2
    finally:
        model._oracle._tile_dtype = saved


# ---- the calibrated forward (activate_attention_transfer: the reference's default, sample_config.yaml) on the executor ---------------------------
from test_interpreter_gpu import CalibrationCollater  # noqa: E402


def _no_state_left(model):
    for mod in model.modules():
        for attr in ("_modulations", "_subject_modulations", "_object_modulations", "_forward_state", "_forward_subject_state", "_forward_object_state"):
            assert not getattr(mod, attr, None), (type(mod).__name__, attr)


@pytest.mark.parametrize("name", ["exist", "verify_attrs", "choose_attr", "query_attr", "verify_rel", "choose_rel", "and", "two_same", "all_same", "compare"])
def test_native_calibrated_forward_g10(mini_ontology_paths, name, monkeypatch):
    """The attention-calibration passes (forward / backward LSTM walks over the aligned program + apply_modulations around every operator:
    batch_base_interpreter.py:87-140, batch_base_types.py:170-187) lowered into the executor's plan: golden g10's questions (the reference's
    own calibrated runs) through dfol_run_program == the Python operator loop BIT FOR BIT, == the reference (golden), also on shared scenes."""
    p = mini_ontology_paths
    ont = D.GQAOntology(p["attribute_file"], p["class_file"], p["vocabulary_file"], p["word_embedding_file"], relation_json_path=p["relation_file"])
    a, meta = gu.load("g10_calibration")
    weights = {k[2:]: a[k] for k in a.files if k.startswith("w:")}
    model = neural_model(ont, meta["config"], weights)
    assert model._has_modulator and native_exec.calibrator(model) is not None
    run_meta = meta["runs"][name]
    qs = [{"program": q["program"], "answer": q["answer"], "question_id": q["question_id"], "image_id": "img000", "tokens": [], "original_dict": None,
           "question": None, "scene": {"n": q["n"], "X": a["%s:X_%d" % (name, i)]}} for i, q in enumerate(run_meta["questions"])]
    nat, py = both_routes(model, ont, qs, monkeypatch=monkeypatch, collater=CalibrationCollater(ont))
    same_results(nat, py, "calibrated " + name)
    gu.check_logprob(nat["log_probability"].cpu().numpy(), a[name + ":lp_f32"], a[name + ":lp_f64"], name + " (calibrated, executor)")
    assert nat["answer"] == run_meta["answer"]
    _no_state_left(model)
    # shared scenes: every question three times on two or three images
    scenes = [q["scene"] for q in qs]
    rep = []
    for r in range(3):
        for i, q in enumerate(qs):
            img = (i + r) % min(3, len(scenes))
            rep.append(dict(q, image_id="img%d" % img, scene=scenes[img]))
    coll = CalibrationCollater(ont)
    coll._share_scenes = True
    nat2, py2 = both_routes(model, ont, rep, monkeypatch=monkeypatch, collater=coll)
    same_results(nat2, py2, "calibrated shared " + name)
    # tokens missing from the batch's embedding index: both routes fall back to the ontology's word embeddings (base_oracle.py:45-55)
    nat3, py3 = both_routes(model, ont, qs, monkeypatch=monkeypatch, collater=TableCollater(1, ont, "X"))
    same_results(nat3, py3, "calibrated, ontology embeddings " + name)
    # the switch off (the reference's test loop switches it off for QUERY batches, trainer.py:97): another plan, the uncalibrated numbers
    monkeypatch.setenv("DFOL_NATIVE", "1")
    pbs = [pb.to_cuda(DEV) for pb in CalibrationCollater(ont).collate([dict(q) for q in qs])]
    _lib.PATH_COUNTS.clear()
    with torch.no_grad():
        off = model(pbs, False, modulator_switch=False)
        on = model(pbs, False, modulator_switch=True)
    assert _lib.PATH_COUNTS.get("native_program", 0) == 2 * len(pbs)
    gu.check_logprob(off["log_probability"].cpu().numpy(), a[name + ":lp_off_f32"], a[name + ":lp_off_f64"], name + " (switch off, executor)")
    assert torch.equal(on["log_probability"], nat["log_probability"])


@pytest.fixture(scope="module")
def calibrated_full(tmp_path_factory):
    from dfol_vqa_amd import experiment
    d = str(tmp_path_factory.mktemp("native_calib"))
    paths, names = syn.write_synthetic_ontology(d)
    cfg = syn.reference_config(paths, activate_attention_transfer=True)
    ont = experiment.build_ontology(cfg)
    torch.manual_seed(5)
    model = experiment.build_model(cfg, ont)
    syn.load_seeded_weights(model, 23)
    with torch.no_grad():                                        # (the reference initialises the output layer's weight to zero: every modulation would be Sigmoid(bias))
        out = model._ops['filter']._filter._attention_output_network[0]
        out.weight.normal_(0.0, 0.5)
        out.bias.normal_(0.0, 0.5)
    with open(paths["attribute_file"]) as f:
        categories = json.load(f)
    return model.to(DEV).eval(), ont, names, categories


class _FullCalibrationCollater(TableCollater):
    """The GQA collator's meta data (batch_gqa_boxfeatures_pipeline.py:88-92): token -> row of an embedding table of the batch's tokens."""

    def __init__(self, split, ont, share=False, spec=None):
        super(_FullCalibrationCollater, self).__init__(split, ont, "X", share_scenes=share)
        self._ont = ont
        self._native_spec = spec

    def collate_meta_data(self, questions):
        toks = set()
        for q in questions:
            for o in [o for br in q["program"]["branches"] for o in br] + [q["program"]["last_op"]]:
                for arg in o["arguments"]:
                    for t in (arg if isinstance(arg, list) else [arg]):
                        if isinstance(t, str):
                            toks.add(t[4:-1] if t.startswith("not(") else t)
                            toks.update(self._ont.query(t) if t in getattr(self._ont, "_attribute_dict", {}) else [])
        names = sorted(t for t in toks if t not in ("_", ""))
        rng = np.random.RandomState(len(names))
        return {"index": {t: i for i, t in enumerate(names)}, "embedding": torch.from_numpy(rng.normal(0, 0.3, (len(names), 300)).astype(np.float32))}


@pytest.mark.parametrize("kind", ALL_KINDS)
def test_native_calibrated_full_size_ragged_1_to_100(calibrated_full, kind, monkeypatch):
    """VERDICT r5 #2's bar: a ragged 1..100-object stream with activate_attention_transfer: True at FULL model size (LSTMCell(318 -> 50) x 2,
    Linear(100 -> 4)) - every ProgramBatch on the executor (`native_program` == the batch count), bit for bit the Python loop, per question and
    with shared scenes, for every terminal operator."""
    model, ont, names, categories = calibrated_full
    qs = syn.full_size_questions(kind, 12, 1, 100, names, categories, 900 + ALL_KINDS.index(kind))
    qs[0]["scene"] = syn.feature_scene(qs[0]["question_id"], 1, 2048)            # one image of ONE object for sure
    qs[1]["scene"] = syn.feature_scene(qs[1]["question_id"], 100, 2048)
    nat, py = both_routes(model, ont, qs, split=2, monkeypatch=monkeypatch, collater=_FullCalibrationCollater(2, ont))
    same_results(nat, py, "calibrated full-size " + kind)
    _no_state_left(model)
    for i, q in enumerate(qs):
        q["image_id"], q["scene"] = "img%d" % (i % 4), qs[i % 4]["scene"]
    nat, py = both_routes(model, ont, qs, split=1, monkeypatch=monkeypatch, collater=_FullCalibrationCollater(1, ont, share=True))
    same_results(nat, py, "calibrated full-size shared " + kind)
    # the modulations do something (the calibrator is not the identity)
    monkeypatch.setenv("DFOL_NATIVE", "1")
    pbs = [pb.to_cuda(DEV) for pb in _FullCalibrationCollater(1, ont, share=True).collate([dict(q) for q in qs])]
    with torch.no_grad():
        off = model(pbs, False, modulator_switch=False)
    assert not torch.equal(off["log_probability"], nat["log_probability"])


def test_native_calibrated_plan_from_the_collate_worker(calibrated_full, monkeypatch):
    """The calibrated plan is built where the reference builds its ProgramBatches (the collate worker: CPU meta data), travels through pickle
    and runs as it is."""
    import pickle
    model, ont, names, categories = calibrated_full
    spec = native_exec.model_spec(model, calibrate=True)
    assert spec is not None and spec.calib["state_dim"] == 50 and spec.calib["lstm_in"] == 318
    qs = syn.full_size_questions("query_attr", 8, 10, 30, names, categories, 77)
    pbs = pickle.loads(pickle.dumps(_FullCalibrationCollater(2, ont, spec=spec).collate([dict(q) for q in qs])))
    assert all(isinstance(pb._native_plan, native_plan.NativePlan) for pb in pbs)
    ops = [int(x) for x in pbs[0]._native_plan.instrs[:, 0]]
    assert ops.count(native_plan.OP_LSTM_CELL) >= 4 and native_plan.OP_MODULATE in ops and native_plan.OP_ATT_MODULATIONS in ops
    pbs = [pb.to_cuda(DEV) for pb in pbs]
    monkeypatch.setenv("DFOL_NATIVE", "1")
    _lib.PATH_COUNTS.clear()
    with torch.no_grad():
        res = model(pbs, False)
    assert _lib.PATH_COUNTS.get("native_program", 0) == 2
    monkeypatch.setenv("DFOL_NATIVE", "0")
    with torch.no_grad():
        ref = model(pbs, False)
    same_results(res, ref, "collate-time calibrated plan")


def test_calibrated_full_batch_properties(calibrated_full, monkeypatch):
    """BASELINE-size batch (256 questions x 100 objects, full-size model, calibrator ON) on the executor through size-independent properties:
    (1) permuting the objects of every scene leaves every log-probability unchanged to rounding (the calibrator reads tokens and LSTM states,
    never the scene geometry); (2) reversing the question order reverses the outputs bit for bit; (3) the batch run in four ProgramBatches
    equals the batch run at once bit for bit; (4) the Python operator loop gives the same bits; (5) the calibrator is not the identity."""
    model, ont, names, categories = calibrated_full
    nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
    Q, N = 256, 100
    qs = []
    for i in range(Q):
        br, last = syn.three_hop_program(8000 + i, nouns, attrs, rels, negate_prob=0.1)
        qs.append(syn.question(8000 + i, br, last, "yes", syn.feature_scene(8000 + i, N, 2048)))

    def go(questions, split=1, native="1", switch=True):
        monkeypatch.setenv("DFOL_NATIVE", native)
        pbs = [pb.to_cuda(DEV) for pb in _FullCalibrationCollater(split, ont).collate([dict(q) for q in questions])]
        _lib.PATH_COUNTS.clear()
        with torch.no_grad():
            res = model(pbs, False, modulator_switch=switch)
        assert _lib.PATH_COUNTS.get("native_program", 0) == (len(pbs) if native == "1" else 0)
        return res["log_probability"].cpu().numpy()

    # the reference's initial calibrator (alpha = beta = c = 1, d = 0.5: gqa_interpreter_experiments.py:119-132) with small weights on top, so that
    # the modulations differ from question to question without driving every attention to the floor (the fixture's N(0, 0.5) weights do, at
    # 100 objects and three hops)
    import math
    out = model._ops['filter']._filter._attention_output_network[0]
    saved_w, saved_b = out.weight.detach().clone(), out.bias.detach().clone()
    try:
        with torch.no_grad():
            out.weight.normal_(0.0, 0.05, generator=torch.Generator(device=out.weight.device).manual_seed(4))
            out.bias.copy_(torch.tensor([-math.log(9.0)] * 3 + [0.0], device=out.bias.device))
        lp = go(qs)
        assert lp.shape == (Q,) and np.all(np.isfinite(lp)) and np.all(lp <= 1e-6)
        assert 0.02 < np.mean(np.exp(lp) > 0.5) < 0.98, "degenerate batch: every answer the same"
        rng = np.random.RandomState(0)
        permuted = [dict(q, scene=dict(q["scene"], X=q["scene"]["X"][rng.permutation(N)])) for q in qs]
        lp_perm = go(permuted)
        assert np.abs(np.exp(lp_perm) - np.exp(lp)).max() <= 2e-5 and np.abs(lp_perm - lp).max() <= 1e-3 * max(1.0, np.abs(lp).max())
        assert np.array_equal(go(qs[::-1])[::-1], lp)
        assert np.array_equal(go(qs, split=4), lp)
        assert np.array_equal(go(qs, native="0"), lp)
        assert not np.array_equal(go(qs, switch=False), lp)
    finally:
        with torch.no_grad():
            out.weight.copy_(saved_w)
            out.bias.copy_(saved_b)


@pytest.fixture(scope="module")
def g23_model(tmp_path_factory):
    """This repository's interpreter as golden g23's reference model: full size, calibrator on, numpy-seeded oracle and calibrator weights, the
    synthetic GloVe file the capture tool wrote (regenerated from its seed)."""
    from dfol_vqa_amd import experiment
    d = str(tmp_path_factory.mktemp("g23"))
    paths, names = syn.write_synthetic_ontology(d)
    with open(paths["vocabulary_file"]) as f:
        vocab = json.load(f)
    paths["word_embedding_file"] = syn.write_synthetic_glove(os.path.join(d, "glove.txt"), vocab["idx_to_arg"])
    cfg = syn.reference_config(paths, activate_attention_transfer=True)
    ont = experiment.build_ontology(cfg)
    model = experiment.build_model(cfg, ont)
    a, meta = gu.load("g23_calibrated_full_size")
    syn.load_seeded_weights(model, meta["weight_seed"])
    syn.load_seeded_calibrator(model, meta["calibrator_seed"])
    return model.to(DEV).eval(), ont, a, meta


@pytest.mark.parametrize("kind", gu.G23_KINDS)
def test_g23_calibrated_full_size_against_the_reference(g23_model, kind, monkeypatch):
    """The calibrated forward at FULL model size against the REFERENCE ITSELF (golden g23: the imported reference with
    activate_attention_transfer: True - LSTMCell(318 -> 50) x 2 + Linear(100 -> 4) around the full-size oracle - on ragged 10..40-object
    scenes, eight terminal operators; g10 pins the calibration at reduced dims).  Executor and Python loop, calibrated and with the switch
    off, at the policy's defaults; the two routes bit for bit."""
    model, ont, a, meta = g23_model
    assert model._has_modulator and native_exec.calibrator(model) is not None
    qs, cm = gu.g23_case(kind, a, meta)
    nat, py = both_routes(model, ont, qs, monkeypatch=monkeypatch, collater=CalibrationCollater(ont))
    same_results(nat, py, "g23 " + kind)
    lp32, lp64 = a[kind + ":lp_f32"], a[kind + ":lp_f64"]
    lp = nat["log_probability"].cpu().numpy()
    if kind == "compare":                                          # (renormalises two aggregations: the named exception of the policy)
        assert np.abs(np.exp(lp.astype(np.float64)) - np.exp(lp64)).max() <= 4 * np.abs(np.exp(lp32.astype(np.float64)) - np.exp(lp64)).max() + 4e-6
    else:
        gu.check_logprob(lp, lp32, lp64, "g23 " + kind)
    assert int(nat["type"]) == cm["type"]
    decided = gu.decided_answers(cm, lp32, lp64)
    assert [x for x, dd in zip(nat["answer"], decided) if dd] == [x for x, dd in zip(cm["answer"], decided) if dd], kind
    monkeypatch.setenv("DFOL_NATIVE", "1")
    pbs = [pb.to_cuda(DEV) for pb in CalibrationCollater(ont).collate([dict(q) for q in qs])]
    with torch.no_grad():
        off = model(pbs, False, modulator_switch=False)["log_probability"].cpu().numpy()
    if kind != "compare":
        gu.check_logprob(off, a[kind + ":lp_off_f32"], a[kind + ":lp_off_f64"], "g23 " + kind + " (switch off)")
    assert np.abs(off - lp).max() > 0.1                            # the calibrator moves the answers


@pytest.mark.parametrize("kind", ["exist", "choose_rel", "query_attr", "and"])
def test_calibration_walk_equals_separate_launches(g23_model, kind, monkeypatch):
    """DFOL_OP_CALIB_WALK (opt-in, DFOL_CALIB_WALK=1: a run of the calibration passes' row-wise steps - cells, state gates, sums, attention-output products -
    in ONE launch, the workgroup that owns 16 rows walking the run) against the same plan lowered to one launch per step (the default): the same bits, on golden
    g23's full-size calibrated model (binary questions: one run; option lists: runs around the gathers of the terminal operator)."""
    model, ont, a, meta = g23_model
    qs, cm = gu.g23_case(kind, a, meta)
    monkeypatch.setenv("DFOL_NATIVE", "1")
    out = {}
    for walk in ("1", "0"):
        monkeypatch.setenv("DFOL_CALIB_WALK", walk)
        pbs = [pb.to_cuda(DEV) for pb in CalibrationCollater(ont).collate([dict(q) for q in qs])]
        _lib.PATH_COUNTS.clear()
        with torch.no_grad():
            out[walk] = model(pbs, False)
        assert _lib.PATH_COUNTS.get("native_program", 0) == len(pbs)
        ops = [int(x) for pb in pbs for x in pb._native_plan.instrs[:, 0]]          # (lowered at the first forward: the collater had no model spec)
        assert (native_plan.OP_CALIB_WALK in ops) == (walk == "1"), ops
        out[walk + "n"] = len(ops)
    same_results(out["1"], out["0"], "walk " + kind)
    assert out["1n"] < out["0n"]


def test_executor_batches_on_two_streams_keep_their_results_and_flags(full, monkeypatch):
    """bench.py's unseen-batch loops alternate two HIP streams (the executor's batch is self-contained: its blob, arena, read-back buffer and
    event, and - from a ring per device - its own fp16-range word).  Six batches launched alternately on two streams with at most two pending:
    everyone returns the bits of its own serial forward, and the one batch with a feature beyond fp16's range raises at ITS result() only."""
    model, ont, oont, names, categories = full
    monkeypatch.setenv("DFOL_NATIVE", "1")
    kinds = ["exist", "verify_rel", "choose_attr", "query_attr", "and", "exist"]
    batches, want = [], []
    for k, kind in enumerate(kinds):
        qs = syn.full_size_questions(kind, 6, 12, 30, names, categories, 900 + k)
        pbs = TableCollater(1, ont, "X").collate([dict(q) for q in qs])
        for pb in pbs:
            pb.create_sparse_tensors()
        pbs = [pb.to_cuda(DEV) for pb in pbs]
        if k == 3:
            pbs[0]._object_features[0, 0] = 1e6
            want.append(None)
        else:
            with torch.no_grad():
                r = model(pbs, False)
            want.append((r["log_probability"].clone(), r["answer"], r["answer_log_probability"]))
        batches.append(pbs)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)]
    _lib.PATH_COUNTS.clear()
    pend, got = [], []

    def take(k, p):
        if want[k] is None:
            with pytest.raises(_lib.DfolError):
                p.result()
            got.append(None)
        else:
            got.append(p.result())
    with torch.no_grad():
        for k, pbs in enumerate(batches):
            with torch.cuda.stream(streams[k % 2]):
                pend.append((k, model.forward_async(pbs, False)))
            if len(pend) > 2:
                take(*pend.pop(0))
        for k, p in pend:
            take(k, p)
    torch.cuda.synchronize()
    assert _lib.PATH_COUNTS.get("native_program", 0) == len(batches)
    for k, (r, w) in enumerate(zip(got, want)):
        if w is None:
            assert r is None
            continue
        assert torch.equal(r["log_probability"], w[0]) and r["answer"] == w[1] and r["answer_log_probability"] == w[2], kinds[k]
