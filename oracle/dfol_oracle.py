"""CPU oracle: a numpy restatement of the reference's ∇-FOL interpreter hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `dfol_vqa_amd/` imports this file; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and there only as the
checker / the timed CPU baseline — never as a fallback for the HIP path.

It follows the reference's algorithm and its FLAT layout ([P, O] attention over all objects of a
ProgramBatch, [P, O, O] relation likelihoods with cross-image entries at the default -30), so that
its arithmetic can be laid beside the reference's line by line.  Every function cites the reference
file:line it restates.  dtype is a parameter (np.float32 / np.float64).

Pinning: checked against the golden vectors under tests/golden/, which were produced by the
reference's own Python (tools/capture_goldens.py) in fp32 and fp64 — see tests/test_oracle_golden.py.
In fp64 the oracle reproduces the reference's fp64 outputs to ~1e-12 (same algorithm); in fp32 it
agrees within the tolerance policy of DESIGN.md §Numerics (numpy's exp/log are not torch's).
"""

import json
import math
import re

import numpy as np

DEFAULT_LL = -30.0

# Element-wise transcendentals over the flat tables ([pairs, 2335] LogSigmoid, [P, O, O] exp / log) are what the reference spends its
# time in, and torch evaluates them on every core (trainer.py:57-62: torch.set_num_threads(cpu_cores_num)); numpy's ufuncs are single-
# threaded but release the GIL, so large arrays are cut into row blocks over a thread pool: the same values, a baseline that is not
# handicapped by one core (bench.py's cpu_baseline).  DFOL_ORACLE_THREADS=1 turns it off.
import os as _os
from concurrent.futures import ThreadPoolExecutor as _Pool

_THREADS = max(1, int(_os.environ.get("DFOL_ORACLE_THREADS", _os.cpu_count() or 1)))
_pool = None


def _rowwise(fn, x):
    """fn(x) for an element-wise fn, in row blocks on the thread pool when x is large."""
    global _pool
    if _THREADS == 1 or x.ndim == 0 or x.size < (1 << 20) or x.shape[0] < 2:
        return fn(x)
    if _pool is None:
        _pool = _Pool(_THREADS)
    parts = min(_THREADS, x.shape[0])
    bounds = np.linspace(0, x.shape[0], parts + 1).astype(np.int64)
    out = np.empty_like(x)

    def work(i):
        out[bounds[i]:bounds[i + 1]] = fn(x[bounds[i]:bounds[i + 1]])
    list(_pool.map(work, range(parts)))
    return out


# ------------------------------------------------------------------------------------------------
# a1  log-space primitives                                                       util.py:17-47
# ------------------------------------------------------------------------------------------------
def safe_exp(x):                                   # util.py:17-19
    return _rowwise(np.exp, x) if isinstance(x, np.ndarray) else np.exp(x)


def safe_log(x):                                   # util.py:22-25 (fp32/fp64 floor 1e-20)
    f = lambda v: np.log(np.maximum(v, v.dtype.type(1e-20)))
    return _rowwise(f, x) if isinstance(x, np.ndarray) else f(x)


def log_and(a, b):                                 # util.py:29-30
    return a + b


def log_or(a, b):                                  # util.py:32-33
    one = a.dtype.type(1.0)
    return safe_log(one - (one - safe_exp(a)) * (one - safe_exp(b)))


def log_not(x):                                    # util.py:35-36
    return safe_log(x.dtype.type(1.0) - safe_exp(x))


def log_parametric_not(x, alpha, beta=1.0):        # util.py:46-47
    alpha = np.asarray(alpha, dtype=x.dtype)
    one, two = x.dtype.type(1.0), x.dtype.type(2.0)
    return safe_log(alpha + x.dtype.type(beta) * (one - two * alpha) * safe_exp(x))


def log_or_tensor(x, axis):                        # util.py:41-44
    return log_not(log_not(x).sum(axis))


def detect_negations(tokens):                      # util.py:68-85
    neg = [re.match(r"not\((\w|\s)+\)", a.strip()) is not None for a in tokens]
    if any(neg):
        out = [a.strip()[4:-1] if n else a.strip() for a, n in zip(tokens, neg)]
    else:
        out = list(tokens)
    return any(neg), neg, out


def flatten_list(ll):                              # util.py:52-57
    a = [x if x is not None else [None] for x in ll]
    batch_index = [i for i, sub in enumerate(a) for _ in sub]
    return [item for sub in a for item in sub], batch_index


def unflatten_list(a_list, batch_index, flags):    # util.py:59-62
    d = {i: [] for i in set(batch_index)}
    for x, y, z in zip(batch_index, a_list, flags):
        if z > 0:
            d.setdefault(x, []).append(y)
    return list(d.values())


def find_max_ind(lp, pq, question_num, threshold=0):   # util.py:64-66
    temp = np.zeros((len(lp), question_num), lp.dtype)
    temp[np.arange(len(lp)), pq] = np.exp(lp)
    mx = temp.max(0)[None, :]
    return ((np.abs(temp - mx) <= 0) & (temp > threshold)).sum(1)


def pair_indices(img):                             # util.py:87-103 (same image, s != o, row-major in s)
    img = np.asarray(img)
    flags = img[:, None] == img[None, :]
    np.fill_diagonal(flags, False)
    ind1, ind2 = np.nonzero(flags)
    return img[ind2], ind1, ind2


# ------------------------------------------------------------------------------------------------
# ontology                                                               batch_gqa_ops.py:25-148
# ------------------------------------------------------------------------------------------------
class Ontology(object):
    def __init__(self, attribute_file, class_file, vocabulary_file, relation_file):
        self.attribute_dict = json.load(open(attribute_file))
        self.class_dict = json.load(open(class_file))
        self.nouns = set(sum(self.class_dict.values(), []))
        vocab = json.load(open(vocabulary_file))
        self.arg_to_idx = vocab["arg_to_idx"]
        self.idx_to_arg = vocab["idx_to_arg"]
        rel = set(json.load(open(relation_file)))
        self.relation_index = sorted(self.arg_to_idx[r] - 1 for r in rel if r in self.arg_to_idx)   # :59
        self.relation_reversed = {i: j for j, i in enumerate(self.relation_index)}                    # :62
        self.concept_num = len(self.idx_to_arg)

    def query(self, name):                         # batch_gqa_ops.py:114-124
        if name in self.attribute_dict:
            return list(self.attribute_dict[name])
        if name in self.class_dict:
            return list(self.class_dict[name])
        if name is None:
            return [None]
        if name == "entity":
            return list(self.nouns)
        return [name]


# ------------------------------------------------------------------------------------------------
# a2  featurizer                                       batch_gqa_boxfeatures_pipeline.py:199-281
# a3  cached tables            classifier_oracle.py:145-156; gqa_interpreter_experiments.py:18-77
# ------------------------------------------------------------------------------------------------
def _sigmoid(x):
    return _rowwise(lambda v: 1.0 / (1.0 + np.exp(-v)), x)


def _elu(x):
    return _rowwise(lambda v: np.where(v > 0, v, np.expm1(np.minimum(v, 0))), x)


def _log_sigmoid(x):
    # torch's LogSigmoid: min(x,0) - log1p(exp(-|x|))
    return _rowwise(lambda v: np.minimum(v, 0) - np.log1p(np.exp(-np.abs(v))), x)


def _linear(x, w, b):
    return x @ w.T + b


def regular_mlp(x, layers):
    """RegularMLP (gqa_interpreter_experiments.py:18-36): [Linear, ELU]* then Linear, Sigmoid. Dropout is eval-mode."""
    for w, b in layers[:-1]:
        x = _elu(_linear(x, w, b))
    w, b = layers[-1]
    return _sigmoid(_linear(x, w, b))


def featurize_scene(X, img, featurizer_layers):
    """-> object features [O, D+4], pair features [pairs, 2(D+4)+4], (ind0, ind1, ind2)."""
    dt = X.dtype
    f = regular_mlp(X[:, :-6], featurizer_layers) if featurizer_layers else X[:, :-6]      # :203-206
    wh = np.maximum(np.stack([X[:, -6], X[:, -5], X[:, -6], X[:, -5]], 1), dt.type(1.0))   # :208-209 clamp(1)
    pos = X[:, -4:] / wh
    obj = np.concatenate([f, pos], 1).astype(dt)
    ind0, ind1, ind2 = pair_indices(img)                                                    # :252
    if len(ind1) == 0:
        return obj, None, (ind0, ind1, ind2)
    x1, y1, w1, h1 = (pos[ind1, k] for k in range(4))
    x2, y2, w2, h2 = (pos[ind2, k] for k in range(4))
    half = dt.type(2.0)
    dx = x1 + w1 / half - x2 - w2 / half
    dy = y1 + h1 / half - y2 - h2 / half
    dist = np.sqrt(dx ** 2 + dy ** 2)                                                       # :271-272
    angle = np.arcsin(dy / np.maximum(dist, dt.type(1e-10)))                                # :275
    pair = np.concatenate([obj[ind1], obj[ind2], dist[:, None], angle[:, None],
                           np.sign(x2 - x1)[:, None], np.sign(y2 - y1)[:, None]], 1).astype(dt)   # :276-279
    return obj, pair, (ind0, ind1, ind2)


def compute_all_log_likelihood_2(obj, pair, attr_layers, rel_layers, emb_w, emb_b, relation_index):
    """classifier_oracle.py:145-156: the full cached tables A [O, C] and R [pairs, |relation_index|]."""
    A = _log_sigmoid(_linear(regular_mlp(obj, attr_layers), emb_w, emb_b))
    R = None
    if pair is not None:
        R = _log_sigmoid(_linear(regular_mlp(pair, rel_layers), emb_w, emb_b))[:, relation_index]
    return A, R


# ------------------------------------------------------------------------------------------------
# world / variable set                                              batch_base_types.py:34-252
# ------------------------------------------------------------------------------------------------
class World(object):
    """Flat world: A [O, C], R [pairs, CR], img [O] (object -> image/question index)."""

    def __init__(self, ontology, A, R, img, dtype, normalize=True):
        self.ontology = ontology
        self.dtype = np.dtype(dtype)
        self.A = None if A is None else np.asarray(A, self.dtype)
        self.R = None if R is None else np.asarray(R, self.dtype)
        self.img = np.asarray(img, np.int64)
        self.O = len(self.img)
        self.Q = int(self.img.max()) + 1
        self.pair = pair_indices(self.img)
        self.normalize = normalize
        self.bom = np.zeros((self.Q, self.O), self.dtype)          # batch_object_map, dense
        self.bom[self.img, np.arange(self.O)] = 1

    def variable_set(self, names, quantifier=1.0, att=None):
        q = np.full(self.Q, quantifier, self.dtype) if np.isscalar(quantifier) else np.asarray(quantifier, self.dtype)
        return VarSet(list(names), np.zeros((self.Q, self.O), self.dtype) if att is None else att, q, None, self)


class VarSet(object):
    def __init__(self, names, att, quant, pq, world):
        self.names, self.att, self.quant, self.pq, self.world = names, att, quant, pq, world

    def gate(self, other, flag):                   # batch_base_types.py:149-168
        g = np.asarray([0 if f is None else f for f in flag], self.att.dtype)
        one = g.dtype.type(1.0)
        quant = self.quant * g + other.quant * (one - g)
        att = self.att * g[:, None] + other.att * (one - g[:, None])
        names = [x if f > 0 else y for x, y, f in zip(self.names, other.names, g)]
        return VarSet(names, att, quant, self.pq, self.world)

    def log_probability(self, hard=False):         # batch_base_types.py:103-125
        w = self.world
        if hard:                                   # :104-112 (test-time option: min/max instead of the soft aggregation)
            t = log_parametric_not(self.att, self.quant[:, None], 1)             # [P, O]
            mask = w.bom if self.pq is None else w.bom[self.pq, :]
            return log_parametric_not((mask * t).min(1), self.quant, 1)
        t = log_parametric_not(self.att.T.copy(), self.quant[None, :], 1)        # [O, P]
        s = w.bom @ t                                                            # [Q, P]
        if self.pq is not None:
            s = s[self.pq, :]                                                    # pqm @ . -> [P, P]
        return log_parametric_not(np.diag(s).copy(), self.quant, 1)


# ------------------------------------------------------------------------------------------------
# a4 / a5  oracle gathers                                        classifier_oracle.py:22-137
# ------------------------------------------------------------------------------------------------
def _cluster_index(image_map):                     # torch.unique_consecutive(..., return_inverse=True)
    image_map = np.asarray(image_map)
    if len(image_map) == 0:
        return image_map, 0
    change = np.concatenate([[0], (image_map[1:] != image_map[:-1]).astype(np.int64)])
    cl = np.cumsum(change)
    return cl, int(cl.max()) + 1


def _normalize_clusters(result, image_map):        # classifier_oracle.py:72-75, 124-127 with _build_map :22-42
    cl, num = _cluster_index(image_map)
    if len(cl) == num:
        return result                              # every cluster a singleton -> cluster_map is None
    cm = np.zeros((num, len(cl)), result.dtype)
    cm[cl, np.arange(len(cl))] = 1
    denom = cm.T @ safe_log(cm @ np.exp(result))
    return result - denom


def attribute_log_likelihood(world, tokens, image_map, default=DEFAULT_LL, normalized_probability=True):
    """classifier_oracle.py:44-82 (cached=True): -> [P', O, 1]; column = arg_to_idx-1 of the FULL table."""
    ont = world.ontology
    col = np.asarray([ont.arg_to_idx[t.strip()] - 1 for t in tokens], np.int64)
    image_map = np.asarray(image_map, np.int64)
    match = image_map[:, None] == world.img[None, :]                       # find_sparse_pair_indices(..., False)
    ind1, ind2 = np.nonzero(match)
    result = np.full((len(tokens), world.O), default, world.dtype)
    result[ind1, ind2] = world.A[ind2, col[ind1]]
    if world.normalize and normalized_probability:
        result = _normalize_clusters(result, image_map)
    return result[:, :, None]


def relation_log_likelihood(world, tokens, image_map, default=DEFAULT_LL, normalized_probability=True):
    """classifier_oracle.py:84-137 (cached=True, no relation_pairobject_map): -> [P', O, O, 1]."""
    ont = world.ontology
    col = np.asarray([ont.relation_reversed[ont.arg_to_idx[t.strip()] - 1] for t in tokens], np.int64)
    image_map = np.asarray(image_map, np.int64)
    pimg, ps, po = world.pair
    match = image_map[:, None] == pimg[None, :]
    ind1, ind2 = np.nonzero(match)
    temp = np.full((len(tokens), len(pimg)), default, world.dtype)
    if len(ind1):
        temp[ind1, ind2] = world.R[ind2, col[ind1]]
    if world.normalize and normalized_probability:
        temp = _normalize_clusters(temp, image_map)
    result = np.full((len(tokens), world.O, world.O, 1), default, world.dtype)
    result[:, ps, po, 0] = temp                                             # :134-135
    return result


# ------------------------------------------------------------------------------------------------
# a6  BatchBayesianLogicCell                                  batch_base_ops.py:153-215, 62-151
# ------------------------------------------------------------------------------------------------
def logic_cell(prior, ll, quant, bom, pq=None, is_negated=None):
    """prior [Q, a, O]; ll [P, O, (O,) 1]; quant [P, a]; bom [Q, O] dense 0/1; pq [P] or None.
    Literal flat restatement, dim_order = [0, 1].  -> [P, a, O]."""
    dt = prior.dtype
    arity = prior.shape[1]
    Q, O, P = prior.shape[0], prior.shape[2], ll.shape[0]
    ll = -np.maximum(-ll.mean(-1), dt.type(0))                              # :194  -relu(-mean)
    if is_negated is not None:                                              # :212-213
        ll = log_parametric_not(ll, np.asarray(is_negated, dt).reshape([-1] + [1] * arity), 1)
    log_p = prior[pq] if (pq is not None and P != Q) else prior            # :74-77
    result = np.zeros((P, arity, O), dt)
    if arity == 1:
        result[:, 0, :] = ll + log_p[:, 0, :]                               # :138 (no other variable to sum out)
        return result
    img_of = bom.argmax(0)                                                  # object -> image
    for a in range(2):
        i, j = a + 1, 2 - a                                                 # this variable, the other variable
        shape_j = [P, 1, O] if j == 2 else [P, O, 1]                        # _reshape_dim :54-55
        shape_i = [P, 1, O] if i == 2 else [P, O, 1]
        lp = ll + log_p[:, j - 1, :].reshape(shape_j)                       # :102
        qj = quant[:, j - 1].reshape(P, 1, 1)
        # :104-108  a single predicate takes a literal branch: EXISTS -> log_not, FOR_ALL -> untouched
        lone_forall = P == 1 and quant[0, j - 1] != 1
        if not lone_forall:
            lp = log_parametric_not(lp, qj, 1)
        if O > 1:
            d = np.arange(O)
            lp[:, d, d] = 0                                                 # :112
        # sum over axis j restricted to each image (mm with batch_object_map), :114-127
        moved = np.moveaxis(lp, j, 0).reshape(O, -1)
        summed = (bom @ moved).reshape([Q] + [P, O])                        # [Q, P, O(i)]
        summed = np.moveaxis(summed, 0, j)                                  # back: axis j now has size Q
        if not lone_forall:
            summed = log_parametric_not(summed, qj, 1)                      # :129-133
        summed = summed + log_p[:, i - 1, :].reshape(shape_i)               # :138
        # :140-147  pick, for every object o, the column q = image(o)
        x = np.moveaxis(summed, i, 1).reshape(P, O, -1)                     # [P, O, Q]
        if Q > 1:
            x = (x * bom.T[None, :, :]).sum(2)
        else:
            x = x[:, :, 0]
        result[:, i - 1, :] = x
    return result


def relate_block(a, b, l, qs, qo, neg, any_neg):
    """Per-predicate block form of the arity-2 cell: a,b [n]; l [n,n] raw tile."""
    dt = a.dtype
    l = np.minimum(l, dt.type(0))
    if any_neg:
        l = log_parametric_not(l, dt.type(neg), 1)
    eye = np.eye(len(a), dtype=bool)
    t = log_parametric_not(l + b[None, :], dt.type(qo), 1)
    t[eye] = 0
    post_s = a + log_parametric_not(t.sum(1), dt.type(qo), 1)
    w = log_parametric_not(l + a[:, None], dt.type(qs), 1)
    w[eye] = 0
    post_o = b + log_parametric_not(w.sum(0), dt.type(qs), 1)
    return post_s, post_o


# ------------------------------------------------------------------------------------------------
# a7 / a8  FilterBatch / RelateBatch                       batch_base_ops.py:311-405, 483-596
# ------------------------------------------------------------------------------------------------
def _valid(tokens):
    return [t is not None and t.strip() not in ("", "_") for t in tokens]


def filter_batch(world, vs, tokens, pq=None, default=DEFAULT_LL, normalized_probability=True):
    if not isinstance(tokens, list):
        tokens = [tokens]
    ind = _valid(tokens)
    if not any(ind):
        return vs
    dt = world.dtype
    P = len(tokens)
    pq_arr = None if pq is None else np.asarray(pq, np.int64)
    quant = vs.quant[pq_arr] if pq_arr is not None else vs.quant           # :341-343
    kept = [t for t, k in zip(tokens, ind) if k]
    any_neg, is_neg, names = detect_negations(kept)                        # :348
    image_map = (pq_arr if pq_arr is not None else np.arange(P))[np.asarray(ind)]   # :351-354
    llk = attribute_log_likelihood(world, names, image_map, default, normalized_probability)
    indb = np.asarray(ind)
    if not all(ind):
        ll = np.full((P, world.O, 1), default, dt)                         # :364
        ll[indb] = llk
        negv = None
        if any_neg:
            negv = np.zeros(P, dt)
            negv[indb] = np.asarray(is_neg, dt)
        out = logic_cell(vs.att[:, None, :], ll, quant[:, None], world.bom, pq_arr, negv)
        src = vs.att[pq_arr] if (pq_arr is not None and P != vs.att.shape[0]) else vs.att
        out[~indb, 0, :] = src[~indb]                                      # :385
    else:
        negv = np.asarray(is_neg, dt) if any_neg else None
        out = logic_cell(vs.att[:, None, :], llk, quant[:, None], world.bom, pq_arr, negv)
    return VarSet(vs.names, out[:, 0, :], quant, pq_arr, world)            # :392-399


def relate_batch(world, svs, ovs, tokens, pq=None, default=DEFAULT_LL, normalized_probability=True):
    if not isinstance(tokens, list):
        tokens = [tokens]
    ind = _valid(tokens)
    if not any(ind):
        return svs, ovs
    dt = world.dtype
    P = len(tokens)
    pq_arr = None if pq is None else np.asarray(pq, np.int64)
    prior = np.stack([svs.att, ovs.att], 1)                                # :516
    quant = np.stack([svs.quant, ovs.quant], 1)                            # :518
    if pq_arr is not None:
        quant = quant[pq_arr]
    kept = [t for t, k in zip(tokens, ind) if k]
    any_neg, is_neg, names = detect_negations(kept)
    image_map = (pq_arr if pq_arr is not None else np.arange(P))[np.asarray(ind)]
    llk = relation_log_likelihood(world, names, image_map, default, normalized_probability)
    indb = np.asarray(ind)
    if not all(ind):
        ll = np.full((P, world.O, world.O, 1), default, dt)
        ll[indb] = llk
        negv = None
        if any_neg:
            negv = np.zeros(P, dt)
            negv[indb] = np.asarray(is_neg, dt)
        out = logic_cell(prior, ll, quant, world.bom, pq_arr, negv)
        out[~indb, 0, :] = svs.att[~indb]                                  # :563-564
        out[~indb, 1, :] = ovs.att[~indb]
    else:
        negv = np.asarray(is_neg, dt) if any_neg else None
        out = logic_cell(prior, llk, quant, world.bom, pq_arr, negv)
    q_out = svs.quant[pq_arr] if pq_arr is not None else svs.quant         # :571-574 (subject's, for both)
    return (VarSet(svs.names, out[:, 0, :], q_out, pq_arr, world),
            VarSet(ovs.names, out[:, 1, :], q_out, pq_arr, world))


# ------------------------------------------------------------------------------------------------
# a9 / a11  GQA operators                                              batch_gqa_ops.py:160-783
# ------------------------------------------------------------------------------------------------
BINARY, QUERY, STATEMENT = 0, 1, 2


def _yes_no(lp):                                   # e.g. batch_gqa_ops.py:404-407
    p = np.exp(lp).tolist()
    ans = [["yes"] if x > 0.5 else ["no"] for x in p]
    alp = [[math.log(x)] if x > 0.5 else [math.log(1 - x)] for x in p]
    return ans, alp


def _result(answer, lp, options, vs, qtype, alp):
    return {"answer": answer, "log_probability": lp, "options": options, "variable_set": vs, "type": qtype,
            "answer_log_probability": alp}


def gqa_select(world, attribute_list=None, **kw):  # :168-183
    Q = world.Q
    if attribute_list is None:
        return world.variable_set(["entity"] * Q)
    name = ["entity" if a is None or a.lower() in ("_", "scene") else a for a in attribute_list][:Q]
    att = [None if a is None or a.lower() in ("_", "scene") else a for a in attribute_list][:Q]
    x = world.variable_set(name)
    return x if all(a is None for a in att) else filter_batch(world, x, att)


def gqa_filter(world, vs, attribute_list, **kw):   # :322-335
    return filter_batch(world, vs, attribute_list)


def gqa_relate(world, vs, relation_list, is_subject, attribute_list=None, **kw):   # :364-371
    x = gqa_select(world, attribute_list)
    subj = x.gate(vs, is_subject)
    obj = vs.gate(x, is_subject)
    subj, obj = relate_batch(world, subj, obj, relation_list)
    return subj.gate(obj, is_subject)


def gqa_exist(world, vs, give_answer=True, hard=False, **kw):  # :399-410
    lp = vs.log_probability(give_answer and hard)
    ans, alp = _yes_no(lp) if give_answer else ([], [])
    return _result(ans, lp, ["no", "yes"], vs, BINARY, alp)


def gqa_end(world, vs, give_answer=True, hard=False, **kw):    # :773-780
    return _result([[n] for n in vs.names] if give_answer else [], vs.log_probability(give_answer and hard), [], vs, STATEMENT, [])


def gqa_verify_attrs(world, vs, attribute_list_list, give_answer=True, pq=None, hard=False, **kw):   # :452-473
    tokens, bi = flatten_list(attribute_list_list)
    x = filter_batch(world, vs, tokens, bi if pq is None else pq, normalized_probability=False)
    xpq = x.pq if x.pq is not None else np.arange(len(tokens))
    att = np.zeros_like(vs.att)
    np.add.at(att, xpq, x.att)                                            # pqm^T @ att  :457
    y = VarSet(vs.names, att, vs.quant, None, world)
    lp = y.log_probability(give_answer and hard)
    ans, alp = _yes_no(lp) if give_answer else ([], [])
    return _result(ans, lp, ["no", "yes"], y, BINARY, alp)


def gqa_verify_rel(world, vs, relation_list, is_subject, attribute_list=None, give_answer=True, hard=False, **kw):   # :489-501
    x = gqa_relate(world, vs, relation_list, is_subject, attribute_list)
    lp = x.log_probability(give_answer and hard)
    ans, alp = _yes_no(lp) if give_answer else ([], [])
    return _result(ans, lp, ["no", "yes"], x, BINARY, alp)


def _choose_answer(world, lp, x, tokens, bi, threshold, give_answer):
    if not give_answer:
        return [], []
    pqv = x.pq if x.pq is not None else np.arange(len(lp))
    flags = find_max_ind(lp, pqv, world.Q, threshold).tolist()
    return unflatten_list(tokens, bi, flags), unflatten_list(lp.tolist(), bi, flags)


def gqa_choose_attr(world, vs, attribute_list_list, give_answer=True, pq=None, threshold=0, hard=False, **kw):   # :215-228
    tokens, bi = flatten_list(attribute_list_list)
    x = filter_batch(world, vs, tokens, bi if pq is None else pq)
    lp = x.log_probability(give_answer and hard)
    ans, alp = _choose_answer(world, lp, x, tokens, bi, threshold, give_answer)
    return _result(ans, lp, attribute_list_list, x, QUERY, alp)


def gqa_query_attr(world, vs, category_list, give_answer=True, pq=None, threshold=0, **kw):   # :304-306
    ont = world.ontology
    lists = [ont.query(c if c not in ("name", "type") else n) for c, n in zip(category_list, vs.names)]
    return gqa_choose_attr(world, vs, lists, give_answer, pq, threshold)      # hard_mode is NOT forwarded (:306)


def gqa_choose_rel(world, vs, relation_list_list, is_subject, attribute_list=None, give_answer=True, pq=None,
                   threshold=0, hard=False, **kw):             # :246-267
    tokens, bi = flatten_list(relation_list_list)
    x = gqa_select(world, attribute_list)
    subj = x.gate(vs, is_subject)
    obj = vs.gate(x, is_subject)
    subj, obj = relate_batch(world, subj, obj, tokens, bi if pq is None else pq)
    flag = np.asarray(is_subject, world.dtype)[subj.pq]                   # pqm @ is_subject  :254-255
    x = subj.gate(obj, flag.tolist())
    lp = x.log_probability(give_answer and hard)
    ans, alp = _choose_answer(world, lp, x, tokens, bi, threshold, give_answer)
    return _result(ans, lp, relation_list_list, x, QUERY, alp)


def _lp_of(v, hard=False):
    return v.log_probability(hard) if isinstance(v, VarSet) else v["log_probability"]


def gqa_and(world, v1, v2, give_answer=True, hard=False, **kw):   # :513-534
    lp = log_and(_lp_of(v1, give_answer and hard), _lp_of(v2, give_answer and hard))
    ans, alp = _yes_no(lp) if give_answer else ([], [])
    return _result(ans, lp, ["no", "yes"], None, BINARY, alp)


def gqa_or(world, v1, v2, give_answer=True, hard=False, **kw):    # :546-567
    lp = log_or(_lp_of(v1, give_answer and hard), _lp_of(v2, give_answer and hard))
    ans, alp = _yes_no(lp) if give_answer else ([], [])
    return _result(ans, lp, ["no", "yes"], None, BINARY, alp)


def _segment_or(world, lp, pq):                    # pqm^T @ log_not(lp), then log_not  e.g. :597-598
    s = np.zeros(world.Q, lp.dtype)
    np.add.at(s, pq, log_not(lp))
    return log_not(s)


def gqa_all_same(world, vs, category_list, give_answer=True, pq=None, hard=False, **kw):   # :582-608
    ont = world.ontology
    lists = [ont.query(c if c not in ("name", "type") else n) for c, n in zip(category_list, vs.names)]
    tokens, bi = flatten_list(lists)
    x = filter_batch(world, vs, tokens, bi if pq is None else pq)
    post = log_not(log_and(vs.att[x.pq], log_not(x.att)))                 # :588-589
    temp = VarSet(x.names, post, np.zeros(len(tokens), world.dtype), x.pq, world)   # FOR_ALL
    lp = _segment_or(world, temp.log_probability(give_answer and hard), x.pq)
    ans, alp = _yes_no(lp) if give_answer else ([], [])
    return _result(ans, lp, ["no", "yes"], None, BINARY, alp)


def gqa_all_different(world, vs, category_list, give_answer=True, pq=None, **kw):   # :627-639
    r = gqa_all_same(world, vs, category_list, give_answer, pq)           # hard_mode is NOT forwarded (:628)
    lp = log_not(r["log_probability"])
    ans, alp = _yes_no(lp) if give_answer else ([], [])
    return _result(ans, lp, ["no", "yes"], None, BINARY, alp)


def gqa_two_same(world, v1, v2, category_list, give_answer=True, pq=None, hard=False, **kw):   # :654-681
    ont = world.ontology
    lists = [ont.query(c if c not in ("name", "type") else n) for c, n in zip(category_list, v1.names)]
    tokens, bi = flatten_list(lists)
    x1 = filter_batch(world, v1, tokens, bi if pq is None else pq)
    x2 = filter_batch(world, v2, tokens, bi if pq is None else pq)
    h = give_answer and hard
    lp = _segment_or(world, log_and(x1.log_probability(h), x2.log_probability(h)), x1.pq)
    ans, alp = _yes_no(lp) if give_answer else ([], [])
    return _result(ans, lp, ["no", "yes"], None, BINARY, alp)


def gqa_two_different(world, v1, v2, category_list, give_answer=True, pq=None, **kw):   # :702-714
    r = gqa_two_same(world, v1, v2, category_list, give_answer, pq)       # hard_mode is NOT forwarded (:703)
    lp = log_not(r["log_probability"])
    ans, alp = _yes_no(lp) if give_answer else ([], [])
    return _result(ans, lp, ["no", "yes"], None, BINARY, alp)


def gqa_compare(world, v1, v2, attribute_list, is_less, give_answer=True, hard=False, **kw):   # :730-758
    x1 = filter_batch(world, v1, attribute_list)
    x2 = filter_batch(world, v2, attribute_list)
    h = give_answer and hard
    lp = np.stack([x1.log_probability(h), x2.log_probability(h)], 1)
    m = lp.max(1, keepdims=True)
    lp = lp - (m + np.log(np.exp(lp - m).sum(1, keepdims=True)))          # LogSoftmax(dim=1)
    lp = log_parametric_not(lp, np.asarray(is_less, lp.dtype)[:, None], 1)
    options = list(zip(v1.names, v2.names))
    ans, alp = [], []
    if give_answer:
        k = lp.argmax(1)
        ans = [[options[i][k[i]]] for i in range(len(k))]
        alp = [[float(lp[i, k[i]])] for i in range(len(k))]
    return _result(ans, lp.reshape(-1), [list(o) for o in options], None, QUERY, alp)


OPS = {"select": (gqa_select, False), "filter": (gqa_filter, False), "relate": (gqa_relate, False),
       "exist": (gqa_exist, True), "end": (gqa_end, True), "verify_attrs": (gqa_verify_attrs, True),
       "verify_rel": (gqa_verify_rel, True), "choose_attr": (gqa_choose_attr, True),
       "query_attr": (gqa_query_attr, True), "choose_rel": (gqa_choose_rel, True), "and": (gqa_and, True),
       "or": (gqa_or, True), "all_same": (gqa_all_same, True), "all_different": (gqa_all_different, True),
       "two_same": (gqa_two_same, True), "two_different": (gqa_two_different, True), "compare": (gqa_compare, True)}


# ------------------------------------------------------------------------------------------------
# a13  program batching                                     data_pipeline.py:31-143, 647-746
# ------------------------------------------------------------------------------------------------
def _transpose(a):
    return [list(x) for x in zip(*a)]


def operator_batch(op_name, arguments, question_num, is_terminal, mask):
    """OperatorBatch.__init__ (data_pipeline.py:33-74): pads/truncates, replaces None rows, transposes."""
    arguments = list(arguments)
    if 0 < len(arguments) < question_num:
        arguments = arguments + [None] * (question_num - len(arguments))
    elif len(arguments) >= question_num:
        arguments = arguments[:question_num]
    else:
        arguments = []
    width = 0
    for x in arguments:
        if isinstance(x, list):
            width = len(x)
            break
    arguments = [[None] * width if x is None else x for x in arguments]
    args_t = _transpose(arguments)
    predicate_num, question_index = question_num, None
    if args_t and any(len(el) > 1 if isinstance(el, list) else False for el in args_t[0]):
        flat, bi = flatten_list(args_t[0])
        predicate_num = len(flat)
        if predicate_num != question_num:
            question_index = bi
    return {"op_name": op_name, "arguments": args_t, "is_terminal": is_terminal,
            "mask": None if mask is None else [float(m) for m in mask],
            "predicate_num": predicate_num, "question_index": question_index, "question_num": question_num}


def collate_programs(questions, starter="select", sep="relate", filler="filter"):
    """ProgramCollaterBase.collate_programs (data_pipeline.py:647-746) -> (op batches, dependencies)."""
    B = len(questions)
    ops, deps, offset, last_dep = [], [], -1, []
    for i in range(max(len(q["program"]["branches"]) for q in questions)):
        args = [q["program"]["branches"][i][0]["arguments"] if q["program"]["branches"][i][0]["operator"] == starter
                else ["_"] for q in questions]
        ops.append(operator_batch(starter, args, B, False, [1.0] * B))
        deps.append([])
        offset += 1
        fillers, seps = [], []
        for k, q in enumerate(questions):
            f_i, s_i = 0, 0
            for o in q["program"]["branches"][i][1:]:
                if o["operator"] == filler:
                    if s_i >= len(fillers):
                        fillers.extend([] for _ in range(s_i - len(fillers) + 1))
                        f_i = 0
                    if f_i >= len(fillers[s_i]):
                        fillers[s_i].append({"arguments": [None] * B, "mask": [0.0] * B})
                    fillers[s_i][f_i]["mask"][k] = 1.0
                    fillers[s_i][f_i]["arguments"][k] = o["arguments"]
                    f_i += 1
                elif o["operator"] == sep:
                    if s_i >= len(seps):
                        seps.append({"arguments": [None] * B, "mask": [0.0] * B})
                    seps[s_i]["mask"][k] = 1.0
                    seps[s_i]["arguments"][k] = o["arguments"]
                    s_i += 1
                    f_i = 0
        for n in range(max(len(seps), len(fillers))):
            if len(fillers) > n:
                for d in fillers[n]:
                    ops.append(operator_batch(filler, d["arguments"], B, False, d["mask"]))
                    deps.append([offset])
                    offset += 1
            if len(seps) > n:
                ops.append(operator_batch(sep, seps[n]["arguments"], B, False, seps[n]["mask"]))
                deps.append([offset])
                offset += 1
        last_dep.append(offset)
    terminal = {}
    for k, q in enumerate(questions):
        o = q["program"]["last_op"]
        if o["operator"] not in terminal:
            terminal[o["operator"]] = {"arguments": [None] * B, "mask": [0.0] * B}
        terminal[o["operator"]]["arguments"][k] = o["arguments"]
        terminal[o["operator"]]["mask"][k] = 1.0
    for name, val in terminal.items():
        ops.append(operator_batch(name, val["arguments"], B, True, val["mask"]))
        deps.append(last_dep)
    return ops, deps


def split_questions(questions, split_num):         # ProgramCollaterBase.collate :754-783
    n = len(questions)
    split_num = min(split_num, n)
    size = math.ceil(n / split_num)
    out, start, end = [], 0, size
    for _ in range(split_num):
        if start >= end:
            break
        out.append(questions[start:end])
        start += size
        end = min(end + size, n)
    return out


# ------------------------------------------------------------------------------------------------
# a12  interpreter                     batch_base_interpreter.py:72-183; batch_gqa_interpreter.py:72-78
# ------------------------------------------------------------------------------------------------
def execute_program_batch(world, ops, deps, give_answer=True, threshold=0, return_trace=False):
    trace = []
    for i, ob in enumerate(ops):
        inputs = tuple(trace[d] for d in deps[i])
        fn, _ = OPS[ob["op_name"]]
        x = fn(world, *(inputs + tuple(ob["arguments"])), give_answer=give_answer, pq=ob["question_index"],
               threshold=threshold, hard=getattr(world, "hard_mode", False))
        last = i == len(ops) - 1
        if last and not ob["is_terminal"]:                                 # batch_gqa_interpreter.py:75-76
            x = gqa_end(world, x, give_answer, hard=getattr(world, "hard_mode", False))
        if isinstance(x, VarSet) and len(inputs) > 0 and ob["mask"] is not None:
            x = x.gate(inputs[0], ob["mask"])                              # batch_base_interpreter.py:166-167
        trace.append(x)
    return (trace[-1], trace) if return_trace else trace[-1]


def gather_results(outputs):                       # data_parallel.py:15-50
    lp = np.concatenate([np.asarray(o["log_probability"]).reshape(-1) for o in outputs])
    res = {"answer": sum([o["answer"] for o in outputs], []), "log_probability": lp,
           "options": sum([o["options"] for o in outputs], []) if outputs[0]["type"] == QUERY else outputs[0]["options"],
           "type": outputs[0]["type"],
           "answer_log_probability": sum([o["answer_log_probability"] for o in outputs], [])}
    return res


def run_questions(ontology, questions, scenes, dtype=np.float32, split=1, normalize=True, give_answer=True,
                  weights=None, return_trace=False, hard_mode=False, threshold=0):
    """The reference's forward over a list of questions: collate (split) -> build_scene -> execute -> gather.
    scenes[i] is {'n', 'A', 'R'} (cached tables) or {'n', 'X'} with `weights` (neural oracle)."""
    dtype = np.dtype(dtype)
    outs, traces = [], []
    start = 0
    for chunk in split_questions(questions, split):
        sc = scenes[start:start + len(chunk)]
        start += len(chunk)
        img = np.repeat(np.arange(len(sc)), [s["n"] for s in sc])
        if "A" in sc[0]:
            A = np.concatenate([s["A"] for s in sc]).astype(dtype)
            R = np.concatenate([s["R"] for s in sc]).astype(dtype)
        else:
            A, R = tables_from_features(np.concatenate([s["X"] for s in sc]).astype(dtype), img, weights, ontology, dtype)
        world = World(ontology, A, R, img, dtype, normalize)
        world.hard_mode = bool(hard_mode)                                  # BatchGQAInterpreter._hard_mode (:23, :73)
        ops, deps = collate_programs(chunk)
        r = execute_program_batch(world, ops, deps, give_answer, threshold=threshold, return_trace=return_trace)
        if return_trace:
            traces.append(r[1])
            r = r[0]
        outs.append(r)
    res = gather_results(outs)
    return (res, traces) if return_trace else res


def tables_from_features(X, img, weights, ontology, dtype):
    """build_scene (batch_base_interpreter.py:45-70) with the classifier oracle's cached tables."""
    w = {k: np.asarray(v, dtype) for k, v in weights.items()}

    def layers(prefix):
        idx = sorted({int(k.split(".")[-2]) for k in w if k.startswith(prefix) and k.endswith(".weight")})
        return [(w["%s%d.weight" % (prefix, i)], w["%s%d.bias" % (prefix, i)]) for i in idx]

    obj, pair, _ = featurize_scene(X, img, layers("_featurizer._featurizer_network._network."))
    return compute_all_log_likelihood_2(obj, pair, layers("_oracle._attribute_network._network."),
                                        layers("_oracle._relation_network._network."),
                                        w["_oracle._embedding_network._network.1.weight"],
                                        w["_oracle._embedding_network._network.1.bias"], ontology.relation_index)


# ------------------------------------------------------------------------------------------------
# a14  loss                                                               trainer.py:181-262
# ------------------------------------------------------------------------------------------------
def compute_loss(result, answers):
    lp = result["log_probability"]
    dt = lp.dtype
    if result["type"] == STATEMENT:
        return -lp.sum()
    if result["type"] == BINARY:                                           # :185-194  BCE(exp(lp), target, 'sum')
        target = np.asarray([a in ("yes", "yeah", "yep", "yup", "aye", "yea") for a in answers], dt)
        p = np.exp(lp)
        # torch's binary_cross_entropy clamps each log term at -100
        return -(target * np.maximum(np.log(p), -100) + (1 - target) * np.maximum(np.log(1 - p), -100)).sum()
    target = [[a == o for o in opt] for a, opt in zip(answers, result["options"])]   # :207-230
    seg = np.asarray([i for i, t in enumerate(target) for _ in t])
    tflat = np.asarray([x for t in target for x in t], dt)
    denom = np.zeros(len(target), dt)
    np.add.at(denom, seg, np.exp(lp))
    return safe_log(denom).sum() - (tflat * lp).sum()
