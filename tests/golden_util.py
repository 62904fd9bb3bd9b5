"""Helpers shared by the golden tests: loading fixtures, rebuilding questions, the tolerance policy."""

import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    arrays = np.load(os.path.join(GOLDEN, name + ".npz"))
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    return arrays, meta


def questions_and_scenes(arrays, meta, feature_key=None):
    qs, scenes = [], []
    for i, q in enumerate(meta["questions"]):
        qs.append({"program": q["program"], "answer": q["answer"], "question_id": q["question_id"],
                   "image_id": "img%03d" % (q["question_id"] % 64), "tokens": [], "original_dict": None, "question": None})
        if feature_key is None:
            scenes.append({"n": q["n"], "A": arrays["A_%d" % i], "R": arrays["R_%d" % i]})
        else:
            scenes.append({"n": q["n"], "X": arrays["%s_%d" % (feature_key, i)]})
    return qs, scenes


G4_CASES = ["g4_exist", "g4_exist_split3", "g4_threehop_n36", "g4_verify_attrs", "g4_choose_attr", "g4_choose_attr_nonorm",
            "g4_query_attr", "g4_verify_rel", "g4_choose_rel", "g4_and", "g4_or", "g4_two_same", "g4_two_different",
            "g4_all_same", "g4_all_different", "g4_compare"]
G4_STRESS = ["g4_stress_mix05", "g4_stress_weak"]
G14_CASES = ["g14_threshold_query_attr", "g14_threshold_choose_attr", "g14_threshold_choose_rel"]      # likelihood_threshold > 0
G11_CASES = ["g11_hard_exist", "g11_hard_single", "g11_hard_verify_attrs", "g11_hard_query_attr", "g11_hard_choose_rel", "g11_hard_and",
             "g11_hard_two_same", "g11_hard_all_same", "g11_hard_all_different", "g11_hard_two_different", "g11_hard_compare"]       # hard_mode = True (batch_base_types.py:104-112)


def check_logprob(got, ref32, ref64, what="", lp_tol=1e-4, p_tol=1e-6, K=2.0, floor=-5.0):
    """Tolerance policy for fp32 log-probabilities (SURVEY.md §7 hard part 1).

    The reference evaluates log(1 - e^x) naively in fp32, so some outputs are ill-conditioned: the
    reference's own fp32 run then differs from its fp64 run by far more than 1e-4 (and `compare`
    renormalises two such values).  Every golden therefore carries both runs, and the reference's own
    fp32-vs-fp64 deviation is the yardstick for how much rounding noise an output carries:

    (i)   where the fp64 golden says lp >= floor AND the reference's own fp32 run is within lp_tol/4 of it
          (the output is demonstrably well-conditioned):   |got - ref32| <= lp_tol           (the 1e-4 bar)
    (ii)  everywhere, in probability space:  |e^got - e^ref64| <= K * max|e^ref32 - e^ref64| + p_tol
    (iii) where lp >= floor:                 |got - ref64|     <= K * max|ref32 - ref64|     + lp_tol
          (max over the case: one fp32 sample of the noise is not a bound for another)
    """
    got, ref32, ref64 = (np.asarray(a, np.float64).reshape(-1) for a in (got, ref32, ref64))
    assert got.shape == ref32.shape == ref64.shape, (what, got.shape, ref32.shape)
    assert np.all(np.isfinite(got)), what
    err_ref = np.abs(ref32 - ref64)
    region = ref64 >= floor
    good = region & (err_ref <= lp_tol / 4)
    d = np.abs(got - ref32)
    if good.any():
        assert np.all(d[good] <= lp_tol), "%s: |dlp| %.3g > %g on well-conditioned outputs" % (what, d[good].max(), lp_tol)
    ep_ref = np.abs(np.exp(ref32) - np.exp(ref64)).max()
    ep_got = np.abs(np.exp(got) - np.exp(ref64))
    assert ep_got.max() <= K * ep_ref + p_tol, "%s: |dp| %.3g vs reference's own %.3g" % (what, ep_got.max(), ep_ref)
    if region.any():
        e_got = np.abs(got - ref64)[region]
        assert e_got.max() <= K * err_ref[region].max() + lp_tol, \
            "%s: |dlp vs fp64| %.3g vs reference's own %.3g" % (what, e_got.max(), err_ref[region].max())
    return int(good.sum()), int(len(got))


G17_CASES = ["c1_n36"] + [k + "_n60_100" for k in ("exist", "and", "or", "verify_attrs", "verify_rel", "choose_attr", "query_attr", "choose_rel",
                                                    "two_same", "two_different", "all_same", "all_different", "compare")]


def g17_case(name, arrays, meta):
    """Questions and scenes of one case of golden family g17 (the reference at full model size): everything but the reference's outputs
    is regenerated from seeds by dfol_vqa_amd.synthetic."""
    from dfol_vqa_amd import synthetic as syn
    cm = meta["cases"][name]
    qs = [syn.question(q["question_id"], q["program"]["branches"], q["program"]["last_op"], q["answer"]) for q in cm["questions"]]
    scenes = [syn.feature_scene(q["question_id"], q["n"], meta["feature_dim"]) for q in cm["questions"]]
    return qs, scenes, cm, arrays[name + ":lp_f32"], arrays[name + ":lp_f64"]


def decided_answers(cm, lp32, lp64):
    """Which questions' answers the reference's own fp32 and fp64 runs agree on with a margin (an arg-max over near-ties, or a yes / no
    at p ~ 0.5, is decided by rounding noise)."""
    lp32, lp64 = np.asarray(lp32, np.float64), np.asarray(lp64, np.float64)
    if cm["type"] == 1 and cm["options"] and isinstance(cm["options"][0], list):            # QUERY: per-question option lists
        sizes = [len(o) for o in cm["options"]]
        off = np.concatenate([[0], np.cumsum(sizes)])
        out = []
        for i in range(len(sizes)):
            a64, a32 = lp64[off[i]:off[i + 1]], lp32[off[i]:off[i + 1]]
            top = np.sort(a64)[::-1]
            out.append(len(top) < 2 or top[0] - top[1] > 4 * np.abs(a32 - a64).max() + 1e-4)
        return out
    return list(np.abs(np.exp(lp64) - 0.5) > 4 * np.abs(np.exp(lp32) - np.exp(lp64)) + 1e-5)


G19_CASES = ["binary_small", "query_rel_small", "binary_tall", "query_rel_tall"]
# round 6: the attribute-side and two-branch terminals (QUERY loss over a category's 26 options, choose_attr, verify_attrs, and / or,
# compare, two_same, all_different) - VERDICT r5 #1
G19_ATTR_CASES = ["query_attr_small", "choose_attr_small", "verify_attrs_small", "and_small", "and_tall", "or_small", "compare_small", "two_same_small",
                  "all_different_small"]


def g19_case(name, arrays, meta):
    """One case of golden family g19 (the reference's `_train_batch` gradients at full model size): questions with their scenes (regenerated
    from seeds), the reference's loss / log-probabilities and, per weight tensor, gradient norm + sampled values, fp32 and fp64."""
    from dfol_vqa_amd import synthetic as syn
    cm = meta["cases"][name]
    qs = [syn.question(q["question_id"], q["program"]["branches"], q["program"]["last_op"], q["answer"], syn.feature_scene(q["question_id"], q["n"], 2048))
          for q in cm["questions"]]
    grads = {}
    for k in arrays.files:
        if k.startswith(name + ":gs:") and k.endswith(":f64"):
            pname = k[len(name) + 4:-4]
            grads[pname] = {"sample64": arrays[k], "sample32": arrays[k[:-3] + "f32"],
                            "norm64": float(arrays["%s:gn:%s:f64" % (name, pname)]), "norm32": float(arrays["%s:gn:%s:f32" % (name, pname)])}
    return qs, cm, {t: (float(arrays["%s:loss_%s" % (name, t)]), arrays["%s:lp_%s" % (name, t)]) for t in ("f32", "f64")}, grads


def check_g19_gradients(got, grads, what, rtol=2e-3, K=8.0):
    """got: {parameter name: full gradient array}.  Sampled values and norm against the reference's fp64 run, with the reference's own
    fp32-vs-fp64 deviation as the yardstick (the policy of test_backward_gpu.grad_close)."""
    from dfol_vqa_amd import synthetic as syn
    assert len(grads) == 12, sorted(grads)
    for pname, g in grads.items():
        full = np.asarray(got[pname], np.float64).reshape(-1)
        smp = full[syn.gradient_sample_index(pname, full.size)]
        ref64, ref32 = g["sample64"].astype(np.float64), g["sample32"].astype(np.float64)
        scale = max(np.abs(ref64).max(), g["norm64"] / np.sqrt(full.size)) + 1e-30
        own = np.abs(ref32 - ref64).max()
        err = np.abs(smp - ref64).max()
        assert err <= K * own + rtol * scale, "%s d%s: |dgrad| %.3g vs the reference's own %.3g (scale %.3g)" % (what, pname, err, own, scale)
        norm = np.sqrt((full ** 2).sum())
        assert abs(norm - g["norm64"]) <= K * abs(g["norm32"] - g["norm64"]) + rtol * g["norm64"] + 1e-30, \
            "%s |d%s| %.6g vs %.6g (the reference's own fp32 run %.6g)" % (what, pname, norm, g["norm64"], g["norm32"])


G23_KINDS = ["exist", "verify_attrs", "verify_rel", "choose_rel", "query_attr", "and", "two_same", "compare"]


def g23_case(kind, arrays, meta):
    """One case of golden family g23 (the reference's CALIBRATED forward at full model size): questions with their scenes regenerated from seeds."""
    from dfol_vqa_amd import synthetic as syn
    cm = meta["cases"][kind]
    qs = [syn.question(q["question_id"], q["program"]["branches"], q["program"]["last_op"], q["answer"], syn.feature_scene(q["question_id"], q["n"], meta["feature_dim"]))
          for q in cm["questions"]]
    return qs, cm
