"""Seeded synthetic scenes and GQA-style programs (SURVEY.md §8(d) "Common synthetic scene").

Every scene is keyed by its question id, never by draw order, so that sharding a
question list across ranks does not change anyone's inputs.
Pure numpy: used by the golden capture tool, the tests and bench.py.
"""

import numpy as np

_IMG_W, _IMG_H = 640.0, 480.0


def _rng(qid, salt=0):
    return np.random.RandomState((1000003 * int(qid) + 7919 * int(salt) + 12345) % (2 ** 31 - 1))


def table_log_likelihood(rng, shape, family="mix10"):
    """Log-probability tables with the value mixtures of SURVEY.md §8(c).

    mix10 : 10 % strong p~U(.5,1), 90 % weak p~U(0,.05)  (default; final lp spread -5..-0.4)
    mix05 : 5 % U(.9,1) / 95 % U(0,.01)                  (stress, probability-space comparison)
    weak  : all p~U(0,.02)                               (stress)
    unif  : p~U(.02,.98)
    """
    u = rng.uniform(size=shape)
    pick = rng.uniform(size=shape)
    if family == "mix10":
        p = np.where(pick < 0.10, 0.5 + 0.5 * u, 0.05 * u)
    elif family == "mix05":
        p = np.where(pick < 0.05, 0.9 + 0.1 * u, 0.01 * u)
    elif family == "weak":
        p = 0.02 * u
    elif family == "unif":
        p = 0.02 + 0.96 * u
    else:
        raise ValueError(family)
    return np.log(np.maximum(p, 1e-5)).astype(np.float32)


def table_scene(qid, n, concept_num, relation_num, family="mix10"):
    """One image's cached oracle tables: A [n, concept_num], R [n(n-1), relation_num] (pairs row-major in subject)."""
    rng = _rng(qid, 1)
    A = table_log_likelihood(rng, (n, concept_num), family)
    R = table_log_likelihood(rng, (n * (n - 1), relation_num), family)
    return {"n": int(n), "A": A, "R": R}


def feature_scene(qid, n, feature_dim):
    """One image's raw object features [n, feature_dim + 6]; the tail is (W, H, x, y, w, h)
    as the reference's collator lays it out (batch_gqa_boxfeatures_pipeline.py:57-71)."""
    rng = _rng(qid, 2)
    feats = rng.uniform(0.0, 1.0, (n, feature_dim))
    x = rng.uniform(0, 500, n)
    y = rng.uniform(0, 400, n)
    w = rng.uniform(5, 105, n)
    h = rng.uniform(5, 105, n)
    tail = np.stack([np.full(n, _IMG_W), np.full(n, _IMG_H), x, y, w, h], 1)
    return {"n": int(n), "X": np.concatenate([feats, tail], 1).astype(np.float32)}


def op(operator, *arguments):
    return {"operator": operator, "arguments": list(arguments)}


def question(qid, branches, last_op, answer="yes", scene=None):
    q = {"program": {"branches": branches, "last_op": last_op}, "image_id": "img%03d" % (int(qid) % 64),
         "answer": answer, "tokens": [], "original_dict": None, "question": None, "question_id": int(qid)}
    if scene is not None:
        q["scene"] = scene
    return q


def three_hop_program(qid, nouns, attributes, relations, negate_prob=0.0):
    """select(n) -> filter(a) -> relate(r, is_subject, n') -> exist   (BASELINE.json configs[0]/[1])."""
    rng = _rng(qid, 3)
    n1 = nouns[rng.randint(len(nouns))]
    a = attributes[rng.randint(len(attributes))]
    r = relations[rng.randint(len(relations))]
    n2 = nouns[rng.randint(len(nouns))]
    subj = bool(rng.uniform() < 0.5)
    if rng.uniform() < negate_prob:
        a = "not(" + a + ")"
    return [[op("select", n1), op("filter", a), op("relate", r, subj, n2)]], op("exist")
