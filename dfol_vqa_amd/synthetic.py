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


def ragged_hop_program(qid, nouns, attributes, relations):
    """select(n) -> 1..3 hops, each filter(a) or relate(r, is_subject, n') -> exist: programs of DIFFERING lengths (collate pads the shorter ones with
    no-op tokens) - what a file of GQA programs holds; bench.py --mode train --hops ragged."""
    rng = _rng(qid, 5)
    branch = [op("select", nouns[rng.randint(len(nouns))])]
    for _ in range(rng.randint(1, 4)):
        if rng.uniform() < 0.5:
            branch.append(op("filter", attributes[rng.randint(len(attributes))]))
        else:
            branch.append(op("relate", relations[rng.randint(len(relations))], bool(rng.uniform() < 0.5), nouns[rng.randint(len(nouns))]))
    return [branch], op("exist")


def open_program(qid, nouns, attributes, relations, categories, hops=4):
    """select(n) -> (filter(a) -> relate(r, is_subject, n')) x hops -> query_attr(category): the 8-hop open (QUERY) programs of
    BASELINE.json configs[4]."""
    rng = _rng(qid, 4)
    branch = [op("select", nouns[rng.randint(len(nouns))])]
    for _ in range(hops):
        branch.append(op("filter", attributes[rng.randint(len(attributes))]))
        branch.append(op("relate", relations[rng.randint(len(relations))], bool(rng.uniform() < 0.5), nouns[rng.randint(len(nouns))]))
    return [branch], op("query_attr", categories[rng.randint(len(categories))])


def write_synthetic_ontology(directory, concept_num=2335, relation_num=333, seed=11):
    """Metadata files with the reference's schema and the real vocabulary's dimensions (2335 concepts of which
    333 are relations), but made-up names: the GQA metadata itself belongs to the reference."""
    import json
    import os
    os.makedirs(directory, exist_ok=True)
    rng = np.random.RandomState(seed)
    n_rel = relation_num
    n_noun = 1200
    n_attr = concept_num - n_rel - n_noun - 40          # 40 category / class names
    nouns = ["noun%04d" % i for i in range(n_noun)]
    attrs = ["attr%04d" % i for i in range(n_attr)]
    rels = ["rel %03d of" % i for i in range(n_rel)]
    cats = ["category%02d" % i for i in range(20)]
    classes = ["class%02d" % i for i in range(20)]
    attribute = {c: attrs[i * 26:(i + 1) * 26] for i, c in enumerate(cats)}
    klass = {c: nouns[i * 40:(i + 1) * 40] for i, c in enumerate(classes)}
    args = nouns + attrs + rels + cats + classes
    assert len(args) == concept_num
    perm = rng.permutation(concept_num)
    idx_to_arg = [args[i] for i in perm]
    vocab = {"op_to_idx": {}, "idx_to_op": [], "arg_to_idx": {a: i + 1 for i, a in enumerate(idx_to_arg)}, "idx_to_arg": idx_to_arg,
             "img_to_idx": {}, "idx_to_img": []}
    paths = {k: os.path.join(directory, v) for k, v in (("attribute_file", "attribute.json"), ("class_file", "class.json"),
                                                         ("relation_file", "relation.json"), ("vocabulary_file", "vocab.json"))}
    for key, obj in (("attribute_file", attribute), ("class_file", klass), ("relation_file", rels + ["rel missing"]), ("vocabulary_file", vocab)):
        with open(paths[key], "w") as f:
            json.dump(obj, f)
    paths["word_embedding_file"] = None
    return paths, {"nouns": nouns, "attributes": attrs, "relations": rels, "categories": cats}


def write_synthetic_glove(path, idx_to_arg, seed=3, dim=300):
    """A GloVe-format text file (`word v1 ... v300` per line) with one seeded N(0, 0.3) row per WORD of the vocabulary's names, sorted - what
    `GQAOntology.get_embeddings` (batch_gqa_ops.py:135-148) reads: a multi-word name's embedding is the sum of its words' rows.  Deterministic
    (numpy RandomState, values printed with four decimals): the capture tool and the tests regenerate the same file."""
    rng = np.random.RandomState(seed)
    with open(path, "w") as f:
        for wd in sorted({x for name in idx_to_arg for x in name.split()}):
            f.write(wd + " " + " ".join("%.4f" % x for x in rng.normal(0, 0.3, dim)) + "\n")
    return path


def load_seeded_calibrator(model, seed):
    """Numpy-seeded weights for the attention calibrator of an interpreter - the two LSTM cells and the attention-output layer every operator
    shares (batch_base_ops.py:251-254; built at gqa_interpreter_experiments.py:115-138) - written into the model in place; works on the
    reference's model and on this repository's alike (same attribute names).  LSTM weights U(-1/sqrt(H), 1/sqrt(H)) like torch's default; the
    output layer keeps the reference's bias (alpha = beta = c = 1, d = 0.5) and gets N(0, 0.05) weights instead of zeros, so that the
    modulations depend on the LSTM states."""
    import math
    import torch
    flt = model._ops['filter']._filter
    rng = np.random.RandomState(seed)
    with torch.no_grad():
        for net in (flt._forward_attention_network, flt._backward_attention_network):
            k = 1.0 / math.sqrt(net.hidden_size)
            for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                p = getattr(net, name)
                p.copy_(torch.as_tensor(rng.uniform(-k, k, tuple(p.shape))).to(p.dtype))
        lin = flt._attention_output_network[0]
        lin.weight.copy_(torch.as_tensor(rng.normal(0.0, 0.05, tuple(lin.weight.shape))).to(lin.weight.dtype))
        lin.bias.copy_(torch.as_tensor(np.asarray([-math.log(9.0)] * 3 + [0.0])).to(lin.bias.dtype))
    return model


def reference_config(paths, **over):
    """config/sample_config.yaml's model section (reference config/sample_config.yaml:37-60), with the calibrator off."""
    cfg = dict(model_name="bench", version="v0", box_features_dim=2048, oracle_input_dim=512, oracle_output_dim=1, word_embedding_dim=300,
               classifier_oracle=True, featurizer_layers_config=[], attribute_network_layers_config=[256],
               relation_network_layers_config=[256], operator_layers_config=[], normalize_oracle=True, dropout=0.0,
               freeze_featurizer=True, freeze_attribute_network=True, freeze_relation_network=True, freeze_embedding_network=True,
               activate_attention_transfer=False, attention_transfer_state_dim=50, freeze_attention_network=False, trainable_gate=False,
               likelihood_threshold=0, hard_mode=False, verbose=False, gpu_num=1)
    cfg.update(paths)
    cfg.update(over)
    return cfg


def seeded_weights(seed, box_features_dim=2048, oracle_input_dim=512, hidden=256, word_embedding_dim=300, concept_num=2335):
    """A numpy-seeded state dict of the classifier oracle at the reference's full size (2048 -> 512, 516 / 1036 -> 256 -> 300 -> 2335),
    under the reference's parameter names.  Golden family g17 stores only the seed: the capture tool loads these arrays into the
    reference's model, the tests load the same arrays into this build's (numpy's RandomState stream is stable across versions and
    machines, torch.manual_seed initialisers are not a contract).  Linear layers: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch's
    default; embedding rows N(0, 0.1) with bias -2 (GloVe-like magnitudes: sparse concept probabilities instead of saturated ones)."""
    rng = np.random.RandomState(int(seed))
    D = oracle_input_dim + 4

    def linear(n_out, n_in):
        b = 1.0 / np.sqrt(n_in)
        return rng.uniform(-b, b, (n_out, n_in)).astype(np.float32), rng.uniform(-b, b, n_out).astype(np.float32)

    w = {}
    w["_featurizer._featurizer_network._network.1.weight"], w["_featurizer._featurizer_network._network.1.bias"] = linear(oracle_input_dim, box_features_dim)
    for name, n_in in (("_oracle._attribute_network", D), ("_oracle._relation_network", 2 * D + 4)):
        w[name + "._network.1.weight"], w[name + "._network.1.bias"] = linear(hidden, n_in)
        w[name + "._network.4.weight"], w[name + "._network.4.bias"] = linear(word_embedding_dim, hidden)
    w["_oracle._embedding_network._network.1.weight"] = rng.normal(0.0, 0.1, (concept_num, word_embedding_dim)).astype(np.float32)
    w["_oracle._embedding_network._network.1.bias"] = np.full(concept_num, -2.0, np.float32)
    return w


def load_seeded_weights(model, seed):
    """Copy seeded_weights(seed) into a model built from reference_config (this build's or the reference's: same parameter names)."""
    import torch
    w = seeded_weights(seed)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in w.items():
            assert tuple(sd[k].shape) == v.shape, (k, tuple(sd[k].shape), v.shape)
            sd[k].copy_(torch.from_numpy(v).to(sd[k].dtype))
    return w


def full_size_questions(kind, count, n_lo, n_hi, names, categories, seed, with_scene=True):
    """Seeded ragged questions for one terminal operator at BASELINE configs[2]'s shape (golden family g17 and
    tests/test_interpreter_gpu.py): select -> 1..3 filter / relate hops -> <kind>; second branch for the binary-branch operators."""
    rng = np.random.RandomState(seed)
    nouns, rels = names["nouns"][:8], names["relations"][:5]
    cats = sorted(categories)[:3]
    attrs = [a for c in cats for a in categories[c][:4]]
    qs = []
    for i in range(count):
        qid = seed * 1000 + i
        pick = lambda xs: xs[rng.randint(len(xs))]
        branch = [op("select", pick(nouns + ["_"]))]
        for _ in range(rng.randint(1, 4)):
            if rng.uniform() < 0.5:
                a_ = pick(attrs)
                branch.append(op("filter", "not(%s)" % a_ if rng.uniform() < 0.2 else a_))
            else:
                branch.append(op("relate", pick(rels), bool(rng.uniform() < 0.5), pick(nouns + ["_"])))
        branches = [branch]
        if kind in ("and", "or", "two_same", "two_different", "compare"):
            branches.append([op("select", pick(nouns)), op("filter", pick(attrs))])
        cat = pick(cats)
        last = {"exist": op("exist"), "and": op("and"), "or": op("or"), "verify_attrs": op("verify_attrs", [pick(attrs), pick(attrs)]),
                "verify_rel": op("verify_rel", pick(rels), bool(rng.uniform() < 0.5), pick(nouns)),
                "choose_attr": op("choose_attr", [categories[cat][0], categories[cat][1]]), "query_attr": op("query_attr", cat),
                "choose_rel": op("choose_rel", [rels[0], rels[1]], bool(rng.uniform() < 0.5), pick(nouns)),
                "two_same": op("two_same", cat), "two_different": op("two_different", cat), "all_same": op("all_same", cat),
                "all_different": op("all_different", cat), "compare": op("compare", pick(attrs), bool(rng.uniform() < 0.5))}[kind]
        n = int(rng.randint(n_lo, n_hi + 1))
        # (with_scene=False: programs only - the caller serves the object features, e.g. from a device-resident pool: bench.py's fresh-programs leg)
        qs.append(question(qid, branches, last, "yes", feature_scene(qid, n, 2048) if with_scene else {"n": n}))
    return qs


# case -> (kind, questions, fewest objects, most objects, seed).  (The four round-5 cases keep the seeds they were captured with.)
TRAIN_PARITY_CASES = {"binary_small": ("binary", 8, 20, 40, 1900), "binary_tall": ("binary", 20, 27, 34, 1901),
                      "query_rel_small": ("query_rel", 8, 20, 40, 1902), "query_rel_tall": ("query_rel", 20, 27, 34, 1903),
                      # round 6: the attribute-side terminals and the two-branch ones
                      "query_attr_small": ("query_attr", 8, 20, 40, 1910), "choose_attr_small": ("choose_attr", 8, 20, 40, 1911),
                      "verify_attrs_small": ("verify_attrs", 8, 20, 40, 1912), "and_small": ("and", 8, 20, 40, 1913),
                      "compare_small": ("compare", 8, 20, 40, 1914), "two_same_small": ("two_same", 6, 20, 32, 1915),
                      "all_different_small": ("all_different", 6, 20, 32, 1916), "and_tall": ("and", 20, 27, 34, 1917),
                      "or_small": ("or", 8, 20, 40, 1918)}
TRAIN_PARITY_WEIGHT_SEED = 19


def train_parity_questions(case, names, categories):
    """Seeded questions for the full-size TRAIN STEP parity cases (golden family g19, tests/test_backward_gpu.py, tests/test_oracle_golden.py):
    select -> optional filters -> 1..3 relate hops (ragged: the aligned relate batches carry no-op tokens for the shorter programs; an
    occasional `_` name and a negated token) -> exist (BINARY), choose_rel (QUERY, two options per question) or - round 6 - one of the
    attribute-side / two-branch terminals: query_attr (a 26-option category per question: the QUERY loss of trainer.py:207-230 over
    208 predicates), choose_attr, verify_attrs, and / or, compare, two_same, all_different.  The `_small` cases keep the pair rows below the
    persistent kernels' threshold (16384 rows), the `_tall` cases above it."""
    kind, count, n_lo, n_hi, seed = TRAIN_PARITY_CASES[case]
    rng = np.random.RandomState(seed)
    nouns, rels = names["nouns"][:8], names["relations"][:5]
    cats = sorted(categories)[:3]
    attrs = [a for c in cats for a in categories[c][:4]]
    pick = lambda xs: xs[rng.randint(len(xs))]
    qs = []
    for i in range(count):
        qid = seed * 1000 + i
        branch = [op("select", pick(nouns + ["_"]))]
        hops = i % 3 if kind == "query_rel" else 1 + i % 3        # the choose_rel programs end with a relation operator of their own
        for _ in range(hops):
            if rng.uniform() < 0.4:
                a_ = pick(attrs)
                branch.append(op("filter", "not(%s)" % a_ if rng.uniform() < 0.25 else a_))
            r_ = pick(rels)
            branch.append(op("relate", "not(%s)" % r_ if rng.uniform() < 0.15 else r_, bool(rng.uniform() < 0.5), pick(nouns + ["_"])))
        branches = [branch]
        if kind == "binary":
            last, answer = op("exist"), ("yes" if rng.uniform() < 0.5 else "no")
        elif kind == "query_rel":
            ra, rb = rng.choice(len(rels), 2, replace=False)
            last = op("choose_rel", [rels[ra], rels[rb]], bool(rng.uniform() < 0.5), pick(nouns))
            answer = rels[ra] if rng.uniform() < 0.5 else rels[rb]
        else:
            if kind in ("and", "or", "compare", "two_same"):
                # the second branch: select (a real name: compare's options are the two names) -> filter, every other one with a relate
                b2 = [op("select", nouns[(nouns.index(branch[0]["arguments"][0]) + 1 + i % 3) % len(nouns)] if branch[0]["arguments"][0] != "_" else pick(nouns)),
                      op("filter", pick(attrs))]
                if i % 2:
                    b2.append(op("relate", pick(rels), bool(rng.uniform() < 0.5), pick(nouns + ["_"])))
                branches.append(b2)
            cat = pick(cats)
            yes_no = "yes" if rng.uniform() < 0.5 else "no"
            if kind == "query_attr":
                last, answer = op("query_attr", cat), categories[cat][rng.randint(len(categories[cat]))]
            elif kind == "choose_attr":
                ia, ib = rng.choice(len(categories[cat]), 2, replace=False)
                last, answer = op("choose_attr", [categories[cat][ia], categories[cat][ib]]), categories[cat][ia if rng.uniform() < 0.5 else ib]
            elif kind == "verify_attrs":
                k = 1 + i % 2                                       # one or two attributes per question (the h5 encoder stores <= 2)
                last, answer = op("verify_attrs", [pick(attrs) for _ in range(k)]), yes_no
            elif kind in ("and", "or"):
                last, answer = op(kind), yes_no
            elif kind == "compare":
                last = op("compare", pick(attrs), bool(rng.uniform() < 0.5))
                n1 = branch[0]["arguments"][0]
                answer = (n1 if n1 != "_" else "entity") if rng.uniform() < 0.5 else b2[0]["arguments"][0]
            elif kind in ("two_same", "all_different"):
                last, answer = op(kind, cat), yes_no
            else:
                raise ValueError(kind)
        n = int(rng.randint(n_lo, n_hi + 1))
        qs.append(question(qid, branches, last, answer, feature_scene(qid, n, 2048)))
    return qs


def gradient_sample_index(name, numel, count=4096):
    """The fixed flat indices at which golden g19 stores a weight gradient (all of it when the tensor is small)."""
    import zlib
    if numel <= count:
        return np.arange(numel)
    return np.sort(np.random.RandomState(zlib.crc32(name.encode()) % (2 ** 31)).choice(numel, count, replace=False))
