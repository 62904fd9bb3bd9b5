"""Host logic of the native executor's lowering (dfol_vqa_amd/native_plan.py) - no GPU: every operator kind lowers, operands stay inside
the blob / the workspace, plans pickle (collate workers), the shapes the executor does not take step aside, answers decode as the Python
operators decode them.  The executor itself is compared with the Python loop on the GPU (tests/test_native_gpu.py)."""

import json
import os
import pickle
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dfol_vqa_amd as D  # noqa: E402
from dfol_vqa_amd import native_plan as NP  # noqa: E402
from dfol_vqa_amd import synthetic as syn  # noqa: E402

KINDS = ["exist", "and", "or", "verify_attrs", "verify_rel", "choose_attr", "query_attr", "choose_rel", "two_same", "two_different", "all_same",
         "all_different", "compare"]
W_OPERANDS = {NP.OP_DENSE: [6], NP.OP_BOX_POSITIONS: [1], NP.OP_FILL: [1], NP.OP_PAIR_LL: [1, 3, 9], NP.OP_ATTR_LL: [1, 6], NP.OP_OPTION_NORMALIZE: [1],
              NP.OP_GATHER_TILES: [1, 4], NP.OP_CALIB_FEATURES: [7], NP.OP_LSTM_CELL: [3, 4, 6, 7], NP.OP_SELECT_ROWS: [1, 2, 6],
              NP.OP_ATT_MODULATIONS: [1, 2, 4], NP.OP_MODULATE: [1, 2, 5],
              NP.OP_FILTER: [1, 2, 7], NP.OP_RELATE_ONE: [1, 2, 3, 10], NP.OP_RELATE: [1, 2, 3, 13, 14], NP.OP_QUANTIFY: [1, 5], NP.OP_GATE: [1, 2, 7, 8],
              NP.OP_LOGIC: [2, 5], NP.OP_SEGMENT_SUM_ROWS: [1, 5], NP.OP_SEGMENT_OR: [1, 4], NP.OP_IMPLICATION: [1, 2, 5], NP.OP_COMPARE: [1, 2, 5],
              NP.OP_FIND_MAX_IND: [1, 5]}
B_OPERANDS = {NP.OP_CALIB_WALK: [1], NP.OP_GATHER_TILES: [2], NP.OP_CALIB_FEATURES: [1, 3, 5], NP.OP_LSTM_CELL: [8, 10, 12], NP.OP_SELECT_ROWS: [3], NP.OP_MODULATE: [3], NP.OP_PAIR_LL: [5, 6, 7], NP.OP_ATTR_LL: [3, 4], NP.OP_OPTION_NORMALIZE: [2, 4], NP.OP_FILTER: [3, 4, 5], NP.OP_RELATE_ONE: [4, 5, 6, 7],
              NP.OP_RELATE: [4, 5, 6, 7, 8, 9], NP.OP_QUANTIFY: [2, 3], NP.OP_GATE: [3, 4, 5], NP.OP_SEGMENT_SUM_ROWS: [2], NP.OP_SEGMENT_OR: [2],
              NP.OP_IMPLICATION: [3], NP.OP_COMPARE: [3], NP.OP_FIND_MAX_IND: [2]}


class FeatureCollater(D.ProgramCollaterBase):
    def __init__(self, split, ontology, spec=None):
        super(FeatureCollater, self).__init__("select", "relate", "filter", split, ontology=ontology, native_spec=spec)

    def collate_object_features(self, questions):
        feats = torch.cat([torch.as_tensor(q["scene"]["X"]) for q in questions], 0)
        bi = torch.cat([torch.full((q["scene"]["n"],), i, dtype=torch.int64) for i, q in enumerate(questions)])
        return feats, bi

    def collate_meta_data(self, questions):
        return {"index": {}, "embedding": torch.zeros(1, 1)}


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    from dfol_vqa_amd import experiment
    paths, names = syn.write_synthetic_ontology(str(tmp_path_factory.mktemp("plan")))
    ont = experiment.build_ontology(syn.reference_config(paths))
    with open(paths["attribute_file"]) as f:
        categories = json.load(f)
    spec = NP.ModelSpec([512], [256, 300], 256, 516, True, 0.0, ont._relation_index)
    return ont, names, categories, spec


@pytest.mark.parametrize("kind", KINDS)
def test_every_operator_lowers_and_operands_stay_in_bounds(setup, kind):
    ont, names, categories, spec = setup
    qs = syn.full_size_questions(kind, 7, 5, 12, names, categories, 3 + KINDS.index(kind))
    for q in qs:                                                  # small feature width: this test never touches the features
        q["scene"]["X"] = q["scene"]["X"][:, -22:]
    pbs = FeatureCollater(2, ont, spec).collate(qs)
    assert len(pbs) == 2
    for pb in pickle.loads(pickle.dumps(pbs)):
        plan = pb._native_plan
        assert isinstance(plan, NP.NativePlan), kind
        assert plan.instrs.dtype == np.int64 and plan.instrs.shape[1] == NP.INSTR_WIDTH and plan.launches == plan.instrs.shape[0]
        assert 0 < plan.out_bytes <= plan.ws_bytes and plan.key == spec.key()
        for row in plan.instrs:
            for k in W_OPERANDS.get(int(row[0]), []):
                assert -1 <= row[k] < plan.ws_bytes, (kind, row)
            for k in B_OPERANDS.get(int(row[0]), []):
                assert -1 <= row[k] < plan.blob.nbytes and (row[k] < 0 or row[k] % 16 == 0), (kind, row)
        r = plan.result
        assert r["lp"] + 4 * r["count"] <= plan.out_bytes
        Q = len(pb._answers)
        assert r["count"] == {"binary": Q, "end": Q, "compare": 2 * Q}.get(r["kind"], r["count"])
        if kind != "compare":                                     # (compare answers as a QUERY although the batch type list does not name it)
            assert int(r["type"]) == int(pb._question_type)
        ops = [int(x) for x in plan.instrs[:, 0]]
        assert ops[:2] == [NP.OP_DENSE, NP.OP_BOX_POSITIONS] and ops.count(NP.OP_ATTR_LL) <= 1 and ops.count(NP.OP_PAIR_LL) <= 1
        # the scene header's arrays are the batch's geometry
        n = np.asarray(pb._object_nums, np.int32)
        assert np.array_equal(plan.blob[plan.scene["n_obj"]:plan.scene["n_obj"] + 4 * Q].view(np.int32), n)
        assert np.array_equal(plan.blob[plan.scene["obj_off"]:plan.scene["obj_off"] + 4 * (Q + 1)].view(np.int32), np.concatenate([[0], np.cumsum(n)]))


WALK_W = {NP.WALK_FILL: [1], NP.WALK_SELECT: [1, 2, 6], NP.WALK_ADD: [1, 2, 6], NP.WALK_LSTM: [2, 3, 4, 5], NP.WALK_ATT_MODULATIONS: [1, 2, 3]}
WALK_B = {NP.WALK_SELECT: [3], NP.WALK_LSTM: [6, 8, 10]}


def _walk_steps(plan):
    """The steps of every OP_CALIB_WALK table of the plan (rows of INSTR_WIDTH int64 in the blob)."""
    out = []
    for row in plan.instrs:
        if int(row[0]) == NP.OP_CALIB_WALK:
            off, n = int(row[1]), int(row[2])
            assert off % 16 == 0 and n >= 2 and int(row[3]) > 0
            out.extend(plan.blob[off:off + n * NP.INSTR_WIDTH * 8].view(np.int64).reshape(n, NP.INSTR_WIDTH))
    return out


def _in_bounds(plan):
    for row in plan.instrs:
        for k in W_OPERANDS.get(int(row[0]), []):
            assert -1 <= row[k] < plan.ws_bytes, row
        for k in B_OPERANDS.get(int(row[0]), []):
            assert -1 <= row[k] < plan.blob.nbytes and (row[k] < 0 or row[k] % 16 == 0), row
    for step in _walk_steps(plan):
        for k in WALK_W[int(step[0])]:
            assert 0 <= step[k] < plan.ws_bytes, step
        for k in WALK_B.get(int(step[0]), []):
            assert 0 <= step[k] < plan.blob.nbytes and step[k] % 16 == 0, step


def test_round6_shapes_lower(setup):
    """What used to step aside to the Python loop and lowers since round 6: a no-op token inside an option list (the
    compressed list is normalised through two row gathers), shared scenes (image-level requests + per-operator tile gathers; the scene
    header keeps the two geometries apart), an image of ONE object (no pair-kernel request), bf16 relation tiles."""
    ont, names, categories, spec = setup
    qs = syn.full_size_questions("choose_rel", 4, 5, 9, names, categories, 11)
    for q in qs:
        q["scene"]["X"] = q["scene"]["X"][:, -22:]
    qs[0]["program"]["last_op"]["arguments"][0][1] = "_"          # a no-op token inside an option list
    plan = FeatureCollater(1, ont, spec).collate(qs)[0]._native_plan
    assert isinstance(plan, NP.NativePlan)
    _in_bounds(plan)
    ops = [int(x) for x in plan.instrs[:, 0]]
    assert ops.count(NP.OP_GATHER_TILES) == 2 and ops.count(NP.OP_OPTION_NORMALIZE) == 1
    # shared scenes: 6 questions on 2 images
    shared = D.ProgramCollaterBase("select", "relate", "filter", 1, ontology=ont, share_scenes=True, native_spec=spec)
    shared.collate_object_features = FeatureCollater(1, ont).collate_object_features
    shared.collate_meta_data = FeatureCollater(1, ont).collate_meta_data
    qs2 = syn.full_size_questions("exist", 6, 5, 9, names, categories, 12)
    for i, q in enumerate(qs2):
        q["scene"] = qs2[i % 2]["scene"] if i >= 2 else dict(q["scene"], X=q["scene"]["X"][:, -22:])
        q["image_id"] = "shared%d" % (i % 2)
    pb = shared.collate(qs2)[0]
    assert pb._question_image is not None and len(pb._object_nums) == 2
    plan = pb._native_plan
    assert isinstance(plan, NP.NativePlan)
    _in_bounds(plan)
    sc = plan.scene
    n_q = plan.blob[sc["n_obj"]:sc["n_obj"] + 4 * 6].view(np.int32)
    n_i = plan.blob[sc["img_n_obj"]:sc["img_n_obj"] + 4 * 2].view(np.int32)
    assert list(n_i) == [int(n) for n in pb._object_nums] and list(n_q) == [int(pb._object_nums[i]) for i in pb._question_image]
    assert sc["O"] == int(sum(pb._object_nums)) and sc["Q"] == 6
    pair = plan.instrs[[int(x) == NP.OP_PAIR_LL for x in plan.instrs[:, 0]]]
    assert len(pair) == 1 and int(pair[0][10]) == 2               # requests over the two IMAGES
    assert NP.OP_GATHER_TILES in [int(x) for x in plan.instrs[:, 0]]
    # an image of one object next to larger ones, and a batch of one-object images only (no pair kernel)
    for n_list in ([1, 5, 7], [1, 1]):
        qs3 = syn.full_size_questions("verify_rel", len(n_list), 5, 9, names, categories, 13)
        for q, n in zip(qs3, n_list):
            q["scene"] = {"n": n, "X": q["scene"]["X"][:n, -22:]}
        plan = FeatureCollater(1, ont, spec).collate(qs3)[0]._native_plan
        assert isinstance(plan, NP.NativePlan), n_list
        assert [int(x) for x in plan.instrs[:, 0]].count(NP.OP_PAIR_LL) == (1 if max(n_list) > 1 else 0)
    # bf16 relation tiles: NS % 8 == 0 and no choose_rel in the batch
    bspec = NP.ModelSpec([512], [256, 300], 256, 516, True, 0.0, ont._relation_index, tile_bf16=True)
    assert bspec.key() != spec.key()
    for n_list, want in (([8, 5, 7], NP.TILE_BF16), ([9, 5, 12], NP.TILE_F32)):
        qs4 = syn.full_size_questions("verify_rel", len(n_list), 5, 9, names, categories, 14)
        for q, n in zip(qs4, n_list):
            q["scene"] = {"n": n, "X": syn.feature_scene(q["question_id"], n, 16)["X"]}
        plan = FeatureCollater(1, ont, bspec).collate(qs4)[0]._native_plan
        rel = plan.instrs[[int(x) == NP.OP_RELATE_ONE for x in plan.instrs[:, 0]]]
        assert len(rel) >= 1 and all(int(r[11]) == want for r in rel), n_list
        _in_bounds(plan)


@pytest.mark.parametrize("kind", KINDS)
def test_calibrated_programs_lower(setup, kind, monkeypatch):
    """The attention-calibration passes in the plan (round 6): with a calibration spec every operator kind lowers, the LSTM walks are there
    (as many backward cells as forward cells, a modulation per calibrated Filter / Relate output), operands stay in bounds, token embeddings
    travel in the blob (index entries where the batch has them, the ontology's embeddings otherwise - none here: the plan steps aside), and
    the uncalibrated plan of the same batch is another plan (another key)."""
    ont, names, categories, spec = setup
    cspec = NP.ModelSpec([512], [256, 300], 256, 516, True, 0.0, ont._relation_index,
                         calib=dict(state_dim=50, lstm_in=18 + 300, ops_index=D.BatchGQAInterpreter._OPS_INDEX))
    assert cspec.key() != spec.key()
    qs = syn.full_size_questions(kind, 6, 5, 12, names, categories, 40 + KINDS.index(kind))
    for q in qs:
        q["scene"]["X"] = q["scene"]["X"][:, -22:]
    voc = list(ont._vocabulary["idx_to_arg"])

    class Coll(FeatureCollater):
        def collate_meta_data(self, questions):
            return {"index": {t: i for i, t in enumerate(voc)}, "embedding": torch.arange(len(voc) * 300, dtype=torch.float32).view(len(voc), 300)}

    monkeypatch.setenv("DFOL_CALIB_WALK", "1")                    # (opt-in: runs of row-wise launches as one; the default plan is checked below)
    pb = Coll(1, ont, cspec).collate(qs)[0]
    plan = pickle.loads(pickle.dumps(pb._native_plan))
    assert isinstance(plan, NP.NativePlan), kind
    _in_bounds(plan)
    ops = [int(x) for x in plan.instrs[:, 0]]
    steps = _walk_steps(plan)
    assert NP.OP_CALIB_WALK in ops and NP.OP_CALIB_FEATURES not in ops      # runs of row-wise launches travel as one; token rows are built inside the cells
    # every cell, stand-alone (token form: x == -1) or as a step of a walk, as (which, head width, table, embedding width)
    cells = [(int(r[1]), int(r[9]), int(r[10]), int(r[11])) for r in plan.instrs if int(r[0]) == NP.OP_LSTM_CELL and int(r[2]) == -1]
    cells += [(int(r[1]), int(r[7]), int(r[8]), int(r[9])) for r in steps if int(r[0]) == NP.WALK_LSTM]
    assert not [r for r in plan.instrs if int(r[0]) == NP.OP_LSTM_CELL and int(r[2]) != -1]
    assert len(cells) >= 2 and sum(1 for c in cells if c[0] == 0) == sum(1 for c in cells if c[0] == 1)
    n_mods = ops.count(NP.OP_ATT_MODULATIONS) + sum(1 for r in steps if int(r[0]) == NP.WALK_ATT_MODULATIONS)
    assert n_mods >= ops.count(NP.OP_MODULATE) >= 1
    assert all(c[1] == 18 and c[3] == 300 for c in cells)
    table_off = cells[0][2]
    row0 = plan.blob[table_off:table_off + 1200].view(np.float32)
    assert row0[1] - row0[0] == 1.0 and int(row0[0]) % 300 == 0                       # a row of the batch's embedding table
    # the same batch without an embedding for its tokens (the synthetic ontology has no embedding file): the Python loop's business
    # the same batch as separate launches (the default): the same steps, one instruction each
    monkeypatch.delenv("DFOL_CALIB_WALK")
    flat = Coll(1, ont, cspec).collate(qs)[0]._native_plan
    _in_bounds(flat)
    fops = [int(x) for x in flat.instrs[:, 0]]
    assert NP.OP_CALIB_WALK not in fops and len(fops) == len(ops) - ops.count(NP.OP_CALIB_WALK) + len(steps)
    assert FeatureCollater(1, ont, cspec).collate(qs)[0]._native_plan is None
    assert isinstance(FeatureCollater(1, ont, spec).collate(qs)[0]._native_plan, NP.NativePlan)


def test_programs_of_differing_lengths_lower(setup):
    """synthetic.ragged_hop_program (bench.py --hops ragged: select -> 1..3 filter / relate hops -> exist, a length per question): seeded by the
    question id, lengths 2..4, and a collated batch - shorter programs padded with no-op tokens - lowers to ONE plan (with and without the
    calibration passes) whose operands stay in bounds."""
    ont, names, categories, spec = setup
    nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
    progs = [syn.ragged_hop_program(900 + i, nouns, attrs, rels) for i in range(40)]
    assert progs == [syn.ragged_hop_program(900 + i, nouns, attrs, rels) for i in range(40)]
    lengths = sorted(set(len(br[0]) for br, last in progs))
    assert lengths == [2, 3, 4] and all(last["operator"] == "exist" for br, last in progs)
    qs = [syn.question(900 + i, br, last, "yes", syn.feature_scene(900 + i, 5 + i % 7, 22)) for i, (br, last) in enumerate(progs)]
    pb = FeatureCollater(1, ont, spec).collate(qs)[0]
    assert isinstance(pb._native_plan, NP.NativePlan)
    _in_bounds(pb._native_plan)
    voc = list(ont._vocabulary["idx_to_arg"])
    cspec = NP.ModelSpec([512], [256, 300], 256, 516, True, 0.0, ont._relation_index,
                         calib=dict(state_dim=50, lstm_in=18 + 300, ops_index=D.BatchGQAInterpreter._OPS_INDEX))

    class Coll(FeatureCollater):
        def collate_meta_data(self, questions):
            return {"index": {t: i for i, t in enumerate(voc)}, "embedding": torch.zeros(len(voc), 300)}
    cal = Coll(1, ont, cspec).collate(qs)[0]._native_plan
    assert isinstance(cal, NP.NativePlan) and cal.launches > pb._native_plan.launches
    _in_bounds(cal)


def test_shapes_the_executor_does_not_take_step_aside(setup):
    ont, names, categories, spec = setup
    qs = syn.full_size_questions("exist", 3, 5, 9, names, categories, 15)
    for q in qs:
        q["scene"]["X"] = q["scene"]["X"][:, -22:]
    pb = FeatureCollater(1, ont).collate(qs)[0]
    pb._object_nums = None                                       # (used to raise TypeError out of build_plan instead of stepping aside)
    assert NP.build_plan(pb, ont, spec) is None


def test_decode_matches_the_python_operators(setup):
    plan = NP.NativePlan()
    out = np.zeros(64, np.uint8)
    lp = np.log(np.asarray([0.9, 0.2, 0.5000001, 0.4], np.float32))
    out[:16] = lp.view(np.uint8)
    plan.result = dict(kind="binary", lp=0, count=4)
    ans, alp = NP.decode(plan, out)
    assert ans == [["yes"], ["no"], ["yes"], ["no"]]
    p = np.exp(lp).tolist()
    assert alp == [[np.log(p[0])], [np.log(1 - p[1])], [np.log(p[2])], [np.log(1 - p[3])]]
    out[16:20] = np.asarray([1, 0, 0, 1], np.uint8)
    plan.result = dict(kind="choose", lp=0, count=4, flags=16, flat=["a", "b", "c", "d"], batch_index=[0, 0, 1, 1], options=None)
    ans, alp = NP.decode(plan, out)
    assert ans == [["a"], ["d"]] and alp == [[float(lp[0])], [float(lp[3])]]
    plan.result = dict(kind="compare", lp=0, count=4, options=[("x", "y"), ("u", "v")])
    ans, alp = NP.decode(plan, out)
    assert ans == [["x"], ["u"]] and alp == [[float(lp[0])], [float(lp[2])]]
    plan.result = dict(kind="end", lp=0, count=2, names=["dog", "entity"])
    assert NP.decode(plan, out) == ([["dog"], ["entity"]], [])
    assert NP.decode(plan, out, give_answer=False) == ([], [])


def test_lazy_operator_upload_keeps_the_host_view(setup):
    """A ProgramBatch that carries a plan moves its operator batches to the device only when somebody reads them."""
    from dfol_vqa_amd.program import _LazyOps
    ont, names, categories, spec = setup
    qs = syn.full_size_questions("exist", 3, 5, 9, names, categories, 13)
    for q in qs:
        q["scene"]["X"] = q["scene"]["X"][:, -22:]
    pb = FeatureCollater(1, ont, spec).collate(qs)[0]
    lazy = _LazyOps(pb._op_batch_list, "cpu", True)
    assert len(lazy) == len(pb._op_batch_list) and list.__len__(lazy) == 0 and not lazy._moved
    assert pickle.loads(pickle.dumps(lazy)) is not None and isinstance(pickle.loads(pickle.dumps(lazy)), list)
    assert pb.terminal_op_name() == "exist"


def test_operator_batches_travel_as_bytes_beside_a_plan(setup):
    """A ProgramBatch pickled WITH a plan (collate worker -> launching process) carries its operator batches as bytes: length and the
    terminal operator's name are there without opening them, the first read opens them, and a batch without a plan pickles as before."""
    from dfol_vqa_amd.program import _PickledOps
    ont, names, categories, spec = setup
    qs = syn.full_size_questions("verify_rel", 5, 5, 9, names, categories, 21)
    for q in qs:
        q["scene"]["X"] = q["scene"]["X"][:, -22:]
    pb = FeatureCollater(1, ont, spec).collate(qs)[0]
    want = [ob._op_name for ob in pb._op_batch_list]
    got = pickle.loads(pickle.dumps(pb))
    ops = got._op_batch_list
    assert isinstance(ops, _PickledOps) and len(ops) == len(want) and got.terminal_op_name() == want[-1] and not ops._loaded
    assert isinstance(got._native_plan, NP.NativePlan) and got._answers == pb._answers and got.batch_size() == pb.batch_size()
    again = pickle.loads(pickle.dumps(got))                                  # unopened: the bytes travel on as they are
    assert isinstance(again._op_batch_list, _PickledOps) and not again._op_batch_list._loaded
    assert [ob._op_name for ob in ops] == want and ops._loaded and ops[-1]._op_name == want[-1]
    assert [ob._op_name for ob in pickle.loads(pickle.dumps(got))._op_batch_list] == want
    plain = pickle.loads(pickle.dumps(FeatureCollater(1, ont).collate(qs)[0]))
    assert type(plain._op_batch_list) is list and [ob._op_name for ob in plain._op_batch_list] == want


def test_fused_clip_adam_steps_aside_without_a_gpu_bucket():
    """training.FusedClipAdam.make -> None for CPU parameters (and train_batch then runs torch's clip_grad_norm_ + Adam.step())."""
    from dfol_vqa_amd import parallel, training
    params = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(4))]
    opt = torch.optim.Adam(params, lr=1e-2)
    bucket = parallel.GradBucket(params)
    assert training.FusedClipAdam.make(opt, bucket) is None and training.FusedClipAdam.make(opt, None) is None
    before = [p.detach().clone() for p in params]
    for p in params:
        p.grad.fill_(1.0)
    training.clip_and_step(torch.nn.ParameterList(params), opt, 0.65, bucket, training._fused_for(opt, bucket))
    assert all(not torch.equal(a, b.detach()) for a, b in zip(before, params))
