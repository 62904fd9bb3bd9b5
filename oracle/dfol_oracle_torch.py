"""TEST INFRASTRUCTURE - the reference's forward path restated in torch-CPU ops, for the `cpu_baseline` leg of bench.py.

This module exists for ONE purpose: a CPU baseline that is the reference's algorithm AT THE REFERENCE'S SPEED on the GPU box, where the
reference's own Python cannot travel.  oracle/dfol_oracle.py (numpy) is the parity checker; it is 2-7x slower than the reference on equal
cores (numpy's elementwise transcendentals are single-threaded, its batch_object_map products dense).  Here every step is the
reference's own torch operator sequence in the reference's flat layout - the same nn.functional.linear / ELU / Sigmoid / LogSigmoid
calls on the full [O, 2335] and [pairs, 2335] tables, the same COO sparse maps and torch.sparse.mm products, the same [P, O, O] relation
tensors - so its wall time tracks the reference's (validated in the build container by tools/time_reference.py: within +-15 % at
ProgramBatch sizes 5 / 10 / 20, outputs equal to 1e-6; profiles/reference_timing.json).  Only tests/, bench.py's cpu_baseline leg and
tools/ import it; the product never does.

Second purpose (round 5): the reference's TRAIN STEP arithmetic under torch autograd (`train_loss`): the same flat-layout operator sequence
in any dtype with the weights as differentiable leaves, so the fused full-size training kernels are checked end to end against
d(loss)/d(weight) of the reference's own formulation (tests/test_backward_gpu.py), pinned on golden g19 - the reference's own
`_train_batch` gradients at full model size (tests/test_oracle_golden.py).

Scope of `run_questions` (the timed inference leg): select, filter, relate, exist.  Scope of `train_loss` / `execute_collated` since round 6:
all 16 program operators + the implicit `end` (select, filter, relate, exist, verify_rel, choose_rel, verify_attrs, choose_attr,
query_attr, and, or, all_same, all_different, two_same, two_different, compare), predicate lists longer than the question list
(predicate_question_map) included.

Each function cites the reference lines it restates (paths under /root/reference/src/nsvqa).
"""

import math
import re

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_LL = -30.0
_NEG = re.compile(r"not\((\w|\s)+\)")


def safe_log(x):                                     # nn/interpreter/util.py:22-25
    return x.clamp(min=1e-20).log()


def log_parametric_not(op, alpha, beta):             # util.py:46-47
    return safe_log(alpha + beta * (1 - 2 * alpha) * op.exp())


def find_sparse_pair_indices(m1, m2, exclude_self_relations=True):      # util.py:87-103
    size1, size2 = m1.size(0), m2.size(0)
    flags = m1.unsqueeze(1) == m2.unsqueeze(0)
    if exclude_self_relations and size1 == size2 and bool((m1 == m2).prod() == 1):
        flags = (flags.float() - torch.eye(size1, dtype=torch.float32)).bool()
    ind = torch.nonzero(flags)
    return m2.repeat(size1, 1)[flags], ind[:, 0], ind[:, 1]


def sparse_map(rows, cols, shape, dtype=torch.float32):                # the legacy torch.sparse.FloatTensor(ind, ones, size) constructor
    return torch.sparse_coo_tensor(torch.stack([rows, cols]), torch.ones(rows.numel(), dtype=dtype), shape)


def predicate_question_map(batch_index, question_num, dtype):          # batch_base_ops.py:324-335, 497-508: [P, Q], one 1 per row
    bi = torch.as_tensor(batch_index, dtype=torch.int64)
    return sparse_map(torch.arange(bi.numel()), bi, (bi.numel(), question_num), dtype)


def detect_negations(a_list):                        # util.py:68-85
    neg = [_NEG.match(a.strip()) is not None for a in a_list]
    return any(neg), neg, ([a.strip()[4:-1] if n else a.strip() for a, n in zip(a_list, neg)] if any(neg) else a_list)


# ---- featurizer and cached tables -----------------------------------------------------------------------------------------------------
def _mlp(x, layers):                                 # gqa_interpreter_experiments.py:18-36 (Dropout is the identity in eval mode)
    for w, b in layers[:-1]:
        x = F.elu(F.linear(x, w, b))
    w, b = layers[-1]
    return torch.sigmoid(F.linear(x, w, b))


def build_scene(X, img, weights, relation_index):
    """BatchGQABoxFeaturizer.featurize_scene (data/batch_gqa_boxfeatures_pipeline.py:199-281) + ClassifierOracle.compute_all_log_likelihood_2
    (nn/vision/classifier_oracle.py:145-156): -> A [O, C], R [pairs, |relation_index|], (pair image, pair subject, pair object)."""
    def layers(prefix):
        idx = sorted({int(k.split(".")[-2]) for k in weights if k.startswith(prefix) and k.endswith(".weight")})
        return [(weights["%s%d.weight" % (prefix, i)], weights["%s%d.bias" % (prefix, i)]) for i in idx]

    f = _mlp(X[:, :-6], layers("_featurizer._featurizer_network._network."))                     # :203-206
    wh = torch.stack([X[:, -6], X[:, -5], X[:, -6], X[:, -5]], 1).clamp(min=1.0)                 # :208-209
    obj = torch.cat([f, X[:, -4:] / wh], 1)                                                      # :210-211
    ind0, ind1, ind2 = find_sparse_pair_indices(img, img)                                        # :252
    pos = obj[:, -4:]
    x1, y1, w1, h1 = (pos[ind1, k] for k in range(4))
    x2, y2, w2, h2 = (pos[ind2, k] for k in range(4))
    dx = x1 + w1 / 2.0 - x2 - w2 / 2.0
    dy = y1 + h1 / 2.0 - y2 - h2 / 2.0
    dist = torch.sqrt(dx ** 2 + dy ** 2)                                                         # :271-272
    angle = torch.asin(dy / dist.clamp(min=1e-10))                                               # :275
    pair = torch.cat([obj[ind1], obj[ind2], dist.unsqueeze(1), angle.unsqueeze(1), torch.sign(x2 - x1).unsqueeze(1),
                      torch.sign(y2 - y1).unsqueeze(1)], 1)                                      # :276-279
    ew, eb = weights["_oracle._embedding_network._network.1.weight"], weights["_oracle._embedding_network._network.1.bias"]
    emb = lambda h: F.logsigmoid(F.linear(h, ew, eb))                                            # gqa_interpreter_experiments.py:60-77
    A = emb(_mlp(obj, layers("_oracle._attribute_network._network.")))
    R = emb(_mlp(pair, layers("_oracle._relation_network._network.")))[:, relation_index]        # classifier_oracle.py:154
    return A, R, (ind0, ind1, ind2)


class World(object):                                 # nn/interpreter/batch_base_types.py:191-252
    def __init__(self, ontology, A, R, pair, img):
        self.ontology, self.A, self.R, self.pair, self.img = ontology, A, R, pair, img
        self.dtype = A.dtype
        self.O = img.numel()
        self.Q = int(img.max().item()) + 1
        self.bom = sparse_map(img, torch.arange(self.O), (self.Q, self.O), self.dtype)           # _batch_object_map :218-222


class VarSet(object):                                # batch_base_types.py:34-187
    def __init__(self, names, att, quant, world, pqm=None):
        self.names, self.att, self.quant, self.world, self.pqm = names, att, quant, world, pqm

    def gate(self, other, flag):                     # :149-168 (the result keeps SELF's predicate_question_map)
        if isinstance(flag, torch.Tensor):
            g = flag
        else:
            g = torch.tensor([0.0 if f is None else float(f) for f in flag], dtype=self.att.dtype)
        return VarSet([x if f > 0 else y for x, y, f in zip(self.names, other.names, g.tolist())],
                      self.att * g.unsqueeze(1) + other.att * (1 - g.unsqueeze(1)), self.quant * g + other.quant * (1 - g), self.world, self.pqm)

    def log_probability(self):                       # :103-125 (soft mode): a full [Q, P] product, then its diagonal
        t = log_parametric_not(self.att.transpose(0, 1).contiguous(), self.quant.unsqueeze(0), 1)
        s = torch.sparse.mm(self.world.bom, t)
        if self.pqm is not None:
            s = torch.sparse.mm(self.pqm, s)
        return log_parametric_not(s.diag(), self.quant, 1)


# ---- oracle gathers (classifier_oracle.py:44-137, cached tables) -------------------------------------------------------------------------
def _normalize(result, image_map):                   # :22-42, :72-75
    _, cl = torch.unique_consecutive(image_map, return_inverse=True)
    size, num = cl.numel(), int(cl.max().item()) + 1
    if size == num:
        return result
    cm = sparse_map(cl, torch.arange(size), (num, size), result.dtype)
    return result - torch.sparse.mm(cm.transpose(0, 1), safe_log(torch.sparse.mm(cm, result.exp())))


def attribute_ll(world, tokens, image_map, normalize=True):
    ind = torch.tensor([world.ontology.arg_to_idx[t.strip()] - 1 for t in tokens], dtype=torch.int64)
    _, ind1, ind2 = find_sparse_pair_indices(image_map, world.img, exclude_self_relations=False)
    result = DEFAULT_LL * torch.ones(len(tokens), world.O, dtype=world.dtype)
    result[ind1, ind2] = world.A[ind2, ind[ind1]]
    if normalize:
        result = _normalize(result, image_map)
    return result.unsqueeze(2)


def relation_ll(world, tokens, image_map, normalize=True):
    ont = world.ontology
    ind = torch.tensor([ont.relation_reversed[ont.arg_to_idx[t.strip()] - 1] for t in tokens], dtype=torch.int64)
    _, ind1, ind2 = find_sparse_pair_indices(image_map, world.pair[0], exclude_self_relations=False)
    temp = DEFAULT_LL * torch.ones(len(tokens), world.pair[0].numel(), dtype=world.dtype)
    temp[ind1, ind2] = world.R[ind2, ind[ind1]]
    if normalize:
        temp = _normalize(temp, image_map)
    result = DEFAULT_LL * torch.ones(len(tokens), world.O, world.O, 1, dtype=world.dtype)
    result[:, world.pair[1], world.pair[2], :] = temp.unsqueeze(2)                               # :134-135
    return result


# ---- BatchBayesianLogicCell (nn/interpreter/batch_base_ops.py:62-215), dim_order [0, 1] --------------------------------------------------
def logic_cell(prior, ll, quant, bom, is_negated=None, pqm=None):
    """prior [Q, arity, O]; ll [P, O(, O), 1]; quant [P, arity]; pqm: sparse [P, Q] when the predicates outnumber the questions."""
    arity, P, O = prior.size(1), ll.size(0), prior.size(2)
    Q = prior.size(0)
    ll = -F.relu(-ll.mean(dim=ll.dim() - 1))                                                     # :194
    if is_negated is not None:
        ll = log_parametric_not(ll, is_negated.view([-1] + arity * [1]), 1)                      # :212-213
    if pqm is not None and P != Q:                                                               # :74-77
        log_p = torch.sparse.mm(pqm, prior.reshape(Q, -1)).view(P, arity, -1)
    else:
        log_p = prior
    result = torch.zeros(P, arity, O, dtype=prior.dtype)
    if arity == 1:
        result[:, 0, :] = ll + log_p[:, 0, :]
        return result
    reshape = [[P, O, 1], [P, 1, O]]                                                             # _reshape_dim :54-55
    coeff = (Q ** (arity - 1) - Q) / (Q - 1) if Q > 1 else 0
    ind = torch.arange(Q, dtype=torch.int64)
    ind = ind + (ind * coeff).long()                                                             # :82-86
    diag = list(range(O))
    for a in range(arity):
        i = a + 1
        lp = ll
        for b in range(arity):
            j = b + 1
            if i == j:
                continue
            lp = lp + log_p[:, j - 1, :].view(reshape[j - 1])                                    # :102
            lone = quant[:, j - 1].numel() == 1
            if lone:                                                                              # :104-108
                if quant[0, j - 1] == 1:
                    lp = safe_log(1.0 - lp.exp())
            else:
                lp = log_parametric_not(lp, quant[:, j - 1].view([-1] + arity * [1]), 1)
            lp = lp.clone()
            lp[:, diag, diag] = 0                                                                # :112
            s1 = lp.size()
            lp = lp.transpose(0, j)
            s2 = list(lp.size())
            lp = torch.sparse.mm(bom, lp.contiguous().view(s1[j], -1))                           # :124-125
            s2[0] = Q
            lp = lp.view(s2).transpose(0, j)
            if lone:                                                                              # :129-133
                if quant[0, j - 1] == 1:
                    lp = safe_log(1.0 - lp.exp())
            else:
                lp = log_parametric_not(lp, quant[:, j - 1].view([-1] + arity * [1]), 1)
        lp = lp + log_p[:, i - 1, :].view(reshape[i - 1])                                        # :138
        lp = lp.transpose(1, i).contiguous().view(P, O, -1)
        if Q > 1:                                                                                 # :142-147
            lp = lp[:, :, ind]
            lp = (lp * bom.transpose(0, 1).to_dense().unsqueeze(0)).sum(dim=2)
        else:
            lp = lp.squeeze(2)
        result[:, i - 1, :] = lp
    return result


def _valid(tokens):                                  # batch_base_ops.py:315
    return [t is not None and t.strip() not in ("", "_") for t in tokens]


def _image_map(pq, P, indb):                         # :352-355, :530-533
    return (torch.arange(P, dtype=torch.int64) if pq is None else torch.as_tensor(pq, dtype=torch.int64))[indb]


def filter_batch(world, vs, tokens, pq=None, normalize=True):           # batch_base_ops.py:311-405; pq: question of every predicate (list) or None
    ind = _valid(tokens)
    if not any(ind):
        return vs
    P, dt = len(tokens), world.dtype
    pqm = None if pq is None else predicate_question_map(pq, vs.att.size(0), dt)
    quant = vs.quant.unsqueeze(1)
    if pqm is not None:
        quant = torch.sparse.mm(pqm, quant)                                                      # :341-343
    kept = [t for t, k in zip(tokens, ind) if k]
    any_neg, is_neg, names = detect_negations(kept)
    indb = torch.tensor(ind)
    llk = attribute_ll(world, names, _image_map(pq, P, indb), normalize)
    if not all(ind):
        ll = DEFAULT_LL * torch.ones(P, world.O, 1, dtype=dt)                                    # :364
        ll[indb] = llk
        negv = None
        if any_neg:
            negv = torch.zeros(P, dtype=dt)
            negv[indb] = torch.tensor(is_neg, dtype=dt)
        out = logic_cell(vs.att.unsqueeze(1), ll, quant, world.bom, negv, pqm)
        out[~indb, 0, :] = vs.att[~indb]                                                         # :385
    else:
        out = logic_cell(vs.att.unsqueeze(1), llk, quant, world.bom, torch.tensor(is_neg, dtype=dt) if any_neg else None, pqm)
    q_out = vs.quant if pqm is None else torch.sparse.mm(pqm, vs.quant.unsqueeze(1)).squeeze(1)  # :391-394
    return VarSet(vs.names, out[:, 0, :], q_out, world, pqm)


def relate_batch(world, svs, ovs, tokens, pq=None, normalize=True):     # batch_base_ops.py:483-596
    ind = _valid(tokens)
    if not any(ind):
        return svs, ovs
    P, dt = len(tokens), world.dtype
    pqm = None if pq is None else predicate_question_map(pq, world.Q, dt)
    prior = torch.stack([svs.att, ovs.att], 1)
    quant = torch.stack([svs.quant, ovs.quant], 1)
    if pqm is not None:
        quant = torch.sparse.mm(pqm, quant)                                                      # :518-521
    kept = [t for t, k in zip(tokens, ind) if k]
    any_neg, is_neg, names = detect_negations(kept)
    indb = torch.tensor(ind)
    llk = relation_ll(world, names, _image_map(pq, P, indb), normalize)
    if not all(ind):
        ll = DEFAULT_LL * torch.ones(P, world.O, world.O, 1, dtype=dt)
        ll[indb] = llk
        negv = None
        if any_neg:
            negv = torch.zeros(P, dtype=dt)
            negv[indb] = torch.tensor(is_neg, dtype=dt)
        out = logic_cell(prior, ll, quant, world.bom, negv, pqm)
        out[~indb, 0, :] = svs.att[~indb]                                                        # :563-564
        out[~indb, 1, :] = ovs.att[~indb]
    else:
        out = logic_cell(prior, llk, quant, world.bom, torch.tensor(is_neg, dtype=dt) if any_neg else None, pqm)
    q_out = svs.quant if pqm is None else torch.sparse.mm(pqm, svs.quant.unsqueeze(1)).squeeze(1)  # :568-571 (both take the subject's)
    return VarSet(svs.names, out[:, 0, :], q_out, world, pqm), VarSet(ovs.names, out[:, 1, :], q_out, world, pqm)


# ---- GQA operators (nn/interpreter/batch_gqa_ops.py) ------------------------------------------------------------------------------------
def gqa_select(world, attribute_list):               # :168-183
    Q = world.Q
    name = ["entity" if a is None or a.lower() in ("_", "scene") else a for a in attribute_list][:Q]
    att = [None if a is None or a.lower() in ("_", "scene") else a for a in attribute_list][:Q]
    x = VarSet(name, torch.zeros(Q, world.O, dtype=world.dtype), torch.ones(Q, dtype=world.dtype), world)
    return x if all(a is None for a in att) else filter_batch(world, x, att)


def gqa_relate(world, vs, relation_list, is_subject, attribute_list):      # :364-371
    x = gqa_select(world, attribute_list)
    subj, obj = relate_batch(world, x.gate(vs, is_subject), vs.gate(x, is_subject), relation_list)
    return subj.gate(obj, is_subject)


def gqa_choose_rel(world, vs, relation_list_list, is_subject, attribute_list):   # :246-267 (log-probabilities only)
    relation_list = [r for rl in relation_list_list for r in rl]                 # util.flatten_list :52-57
    batch_index = [i for i, rl in enumerate(relation_list_list) for _ in rl]
    x = gqa_select(world, attribute_list)
    subj, obj = relate_batch(world, x.gate(vs, is_subject), vs.gate(x, is_subject), relation_list, batch_index)
    flags = torch.tensor([0.0 if f is None else float(f) for f in is_subject], dtype=world.dtype).unsqueeze(1)
    flags = torch.sparse.mm(subj.pqm, flags).squeeze(1)                          # :254-255
    return subj.gate(obj, flags).log_probability(), relation_list_list


def _weights_and_index(ontology, weights):
    w = {k: torch.as_tensor(np.asarray(v, np.float32)) for k, v in weights.items()}
    return w, torch.as_tensor(np.asarray(ontology.relation_index, np.int64))


def run_questions(ontology, questions, scenes, weights, split=1):
    """The reference's inference forward over a list of questions (collate into `split` ProgramBatches -> build_scene -> execute ->
    gather, nn/interpreter/batch_base_interpreter.py:45-183) for select -> (filter | relate)* -> exist programs with ONE branch whose
    operator sequence is the same for every question of the list.  -> {"log_probability": float32 numpy [Q], "answer": [["yes"] | ["no"]]}"""
    w, rel_index = _weights_and_index(ontology, weights)
    n, lps = len(questions), []
    split = min(split, n)
    size = math.ceil(n / split)
    with torch.no_grad():
        for i in range(split):
            chunk, sc = questions[i * size:(i + 1) * size], scenes[i * size:(i + 1) * size]
            if not chunk:
                break
            img = torch.as_tensor(np.repeat(np.arange(len(sc)), [s["n"] for s in sc]).astype(np.int64))
            X = torch.as_tensor(np.concatenate([s["X"] for s in sc]).astype(np.float32))
            world = World(ontology, *build_scene(X, img, w, rel_index), img)
            branches = [q["program"]["branches"] for q in chunk]
            if any(len(b) != 1 for b in branches) or any(q["program"]["last_op"]["operator"] != "exist" for q in chunk):
                raise NotImplementedError("the torch restatement times select -> (filter | relate)* -> exist programs only")
            ops = [[o["operator"] for o in b[0]] for b in branches]
            if any(o != ops[0] for o in ops):
                raise NotImplementedError("every question of a ProgramBatch must have the same operator sequence here")
            vs = None
            for k, name in enumerate(ops[0]):
                args = [b[0][k]["arguments"] for b in branches]
                if name == "select":
                    vs = gqa_select(world, [a[0] for a in args])
                elif name == "filter":
                    vs = filter_batch(world, vs, [a[0] for a in args])
                elif name == "relate":
                    vs = gqa_relate(world, vs, [a[0] for a in args], [a[1] for a in args], [a[2] for a in args])
                else:
                    raise NotImplementedError(name)
            lps.append(vs.log_probability())
    lp = torch.cat(lps).numpy()
    return {"log_probability": lp, "answer": [["yes"] if x > 0.5 else ["no"] for x in np.exp(lp).tolist()]}


# ---- the train step's arithmetic under autograd (train/trainer.py:181-262, 429-442) --------------------------------------------------------
_YES = ("yes", "yeah", "yep", "yup", "aye", "yea")


def log_not(op):                                     # util.py:35-36
    return safe_log(1.0 - op.exp())


def log_or(a, b):                                    # util.py:32-33
    return safe_log(1.0 - (1.0 - a.exp()) * (1.0 - b.exp()))


def _flatten(list_of_lists):                         # util.flatten_list :52-57
    return [x for l in list_of_lists for x in l], [i for i, l in enumerate(list_of_lists) for _ in l]


def _category_options(world, category_list, names):  # batch_gqa_ops.py:305 (GQAOntology.query :114-124)
    return [list(world.ontology.query(c if c not in ("name", "type") else n)) for c, n in zip(category_list, names)]


def execute_collated(world, ops, deps):
    """The execution loop (batch_base_interpreter.py:145-172) over operator batches collated by oracle.dfol_oracle.collate_programs
    (data_pipeline.py:647-746): dependency-ordered dispatch, per-question mask gating (:166-167), implicit `end` (:75-76 of
    batch_gqa_interpreter.py).  -> (log_probability [P], type, options)"""
    trace = []
    for i, ob in enumerate(ops):
        name, args, last = ob["op_name"], ob["arguments"], i == len(ops) - 1
        inputs = [trace[d] for d in deps[i]]
        if name == "select":
            x = gqa_select(world, args[0] if args else ["_"] * world.Q)
        elif name == "filter":
            x = filter_batch(world, inputs[0], args[0])
        elif name in ("relate", "verify_rel"):
            x = gqa_relate(world, inputs[0], args[0], args[1], args[2])
            if name == "verify_rel":                                                         # batch_gqa_ops.py:489-501
                return x.log_probability(), "binary", None
        elif name == "exist":                                                                 # :399-410
            return inputs[0].log_probability(), "binary", None
        elif name == "choose_rel":
            lp, options = gqa_choose_rel(world, inputs[0], args[0], args[1], args[2])
            return lp, "query", options
        elif name in ("choose_attr", "query_attr"):                                           # :215-228, :304-306
            lists = args[0] if name == "choose_attr" else _category_options(world, args[0], inputs[0].names)
            flat, bi = _flatten(lists)
            return filter_batch(world, inputs[0], flat, bi).log_probability(), "query", lists
        elif name == "verify_attrs":                                                          # :452-473
            flat, bi = _flatten(args[0])
            x = filter_batch(world, inputs[0], flat, bi, normalize=False)
            att = torch.sparse.mm(x.pqm.transpose(0, 1), x.att)                               # the prior is counted once per attribute (:457)
            return VarSet(inputs[0].names, att, inputs[0].quant, world).log_probability(), "binary", None
        elif name in ("and", "or"):                                                           # :513-567
            v1, v2 = inputs[0].log_probability(), inputs[1].log_probability()
            return (v1 + v2 if name == "and" else log_or(v1, v2)), "binary", None
        elif name in ("all_same", "all_different"):                                           # :582-608, :627-639
            flat, bi = _flatten(_category_options(world, args[0], inputs[0].names))
            x = filter_batch(world, inputs[0], flat, bi)
            post = log_not(torch.sparse.mm(x.pqm, inputs[0].att) + log_not(x.att))           # (pre-condition ==> the same), before aggregation
            lp = VarSet(x.names, post, torch.zeros(len(flat), dtype=world.dtype), world, x.pqm).log_probability()      # FOR_ALL
            lp = log_not(torch.sparse.mm(x.pqm.transpose(0, 1), log_not(lp.unsqueeze(1))).squeeze(1))
            return (lp if name == "all_same" else log_not(lp)), "binary", None
        elif name in ("two_same", "two_different"):                                           # :654-681, :702-714
            flat, bi = _flatten(_category_options(world, args[0], inputs[0].names))
            x1, x2 = filter_batch(world, inputs[0], flat, bi), filter_batch(world, inputs[1], flat, bi)
            lp = x1.log_probability() + x2.log_probability()
            lp = log_not(torch.sparse.mm(x1.pqm.transpose(0, 1), log_not(lp.unsqueeze(1))).squeeze(1))
            return (lp if name == "two_same" else log_not(lp)), "binary", None
        elif name == "compare":                                                               # :730-758
            x1, x2 = filter_batch(world, inputs[0], args[0]), filter_batch(world, inputs[1], args[0])
            lp = F.log_softmax(torch.stack([x1.log_probability(), x2.log_probability()]).transpose(0, 1).contiguous(), dim=1)
            alpha = torch.tensor([float(bool(v)) for v in args[1]], dtype=world.dtype).unsqueeze(1)
            return log_parametric_not(lp, alpha, 1).view(-1), "query", list(zip(inputs[0].names, inputs[1].names))
        else:
            raise NotImplementedError(name)
        if inputs and ob["mask"] is not None:
            x = x.gate(inputs[0], ob["mask"])
        trace.append(x)
    return trace[-1].log_probability(), "statement", None                                    # the appended `end` (:768-783)


def compute_loss(lp, kind, answers, options):        # trainer.py:181-262 (the fp64 run builds its targets in the run's dtype)
    if kind == "statement":
        return -lp.sum()
    if kind == "binary":                             # :185-194
        target = torch.tensor([float(a in _YES) for a in answers], dtype=lp.dtype)
        return F.binary_cross_entropy(lp.exp(), target, reduction="sum")
    target = [[a == o for o in op] for a, op in zip(answers, options)]                        # :207-230
    seg = torch.tensor([i for i, t in enumerate(target) for _ in t], dtype=torch.int64)
    tflat = torch.tensor([float(x) for t in target for x in t], dtype=lp.dtype)
    qpm = sparse_map(seg, torch.arange(seg.numel()), (len(target), seg.numel()), lp.dtype)
    return safe_log(torch.sparse.mm(qpm, lp.unsqueeze(1).exp())).sum() - (tflat * lp).sum()


def train_loss(ontology, questions, scenes, weights, dtype=torch.float64, collate=None):
    """_train_batch up to the backward (trainer.py:429-437) for ONE ProgramBatch: forward in training mode (dropout 0), `_compute_loss`,
    `/ batch size`, `.backward()`.  `weights`: {reference parameter name: array}; -> (loss, log_probability, {name: gradient}) as numpy, the
    whole computation in `dtype`.  `collate`: oracle.dfol_oracle.collate_programs (passed in so this module stays importable alone)."""
    if collate is None:
        from oracle.dfol_oracle import collate_programs as collate
    w = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in weights.items()}
    rel_index = torch.as_tensor(np.asarray(ontology.relation_index, np.int64))
    img = torch.as_tensor(np.repeat(np.arange(len(scenes)), [s["n"] for s in scenes]).astype(np.int64))
    X = torch.tensor(np.concatenate([s["X"] for s in scenes]), dtype=dtype)
    world = World(ontology, *build_scene(X, img, w, rel_index), img)
    ops, deps = collate(questions)
    lp, kind, options = execute_collated(world, ops, deps)
    loss = compute_loss(lp, kind, [q["answer"] for q in questions], options) / len(questions)
    loss.backward()
    grads = {k: (np.zeros(tuple(v.shape), np.float64) if v.grad is None else v.grad.numpy().astype(np.float64)) for k, v in w.items()}
    return float(loss.detach()), lp.detach().numpy(), grads
