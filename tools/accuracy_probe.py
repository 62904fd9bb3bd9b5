#!/usr/bin/env python3
"""Where does the HIP path's rounding noise come from?  (VERDICT r2 "weak" #1.)  Runs on the GPU box.

    python tools/accuracy_probe.py [--objects 36] [--questions 64] [--lib path/to/alternative/libdfolvqa.so]

One seeded batch of BASELINE configs[1] (select -> filter -> relate -> exist, full-size model) goes through the product's forward;
every intermediate the needed-columns dataflow keeps (object features, attribute hidden layer, the per-object halves of the pair
MLP's first layer, the requested attribute blocks, the requested relation tiles, the final log-probabilities) is compared with a
float64 evaluation of the same network.  Beside it: the same stages evaluated in fp32 by numpy (the oracle's arithmetic) and by
torch on the CPU (the reference's arithmetic: nn.Linear / ELU / Sigmoid / LogSigmoid), i.e. the reference's OWN fp32-vs-fp64 noise,
which is the yardstick of the tolerance policy.  Prints one JSON object.
"""

import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stats(got, ref, mask=None):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    if mask is not None:
        d = d[mask]
    return {"max": float(d.max()), "mean": float(d.mean()), "rms": float(np.sqrt((d * d).mean()))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=36)
    ap.add_argument("--questions", type=int, default=64)
    ap.add_argument("--lib", type=str, default=None)
    ap.add_argument("--tag", type=str, default="default")
    ap.add_argument("--dump", type=str, default=None, help="write the exact likelihoods and the logic kernels' outputs to this .npz")
    args = ap.parse_args()
    if args.lib:
        os.environ["DFOL_LIB"] = os.path.abspath(args.lib)
    import torch
    import dfol_vqa_amd as D
    from dfol_vqa_amd import _lib as L
    from dfol_vqa_amd import experiment
    from dfol_vqa_amd import synthetic as syn
    from oracle import dfol_oracle as orc

    dev = torch.device("cuda:0")
    tmp = tempfile.mkdtemp(prefix="dfol_probe_")
    paths, names = syn.write_synthetic_ontology(tmp)
    cfg = syn.reference_config(paths)
    ont = experiment.build_ontology(cfg)
    torch.manual_seed(0)
    model = experiment.build_model(cfg, ont)
    with torch.no_grad():
        model._oracle._embedding_network.linear.weight.normal_(0.0, 0.1)
        model._oracle._embedding_network.linear.bias.fill_(-2.0)
    model = model.to(dev).eval()
    weights = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if k.startswith("_featurizer.") or k.startswith("_oracle.")}

    class Collater(D.ProgramCollaterBase):
        def __init__(self):
            super(Collater, self).__init__("select", "relate", "filter", 1, ontology=ont)

        def collate_object_features(self, questions):
            feats = torch.cat([torch.from_numpy(q["scene"]["X"]) for q in questions], 0)
            bi = torch.cat([torch.full((q["scene"]["n"],), i, dtype=torch.int64) for i, q in enumerate(questions)])
            return feats, bi

        def collate_meta_data(self, questions):
            return {"index": {}, "embedding": torch.zeros(1, 1)}

    nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
    qs = []
    for i in range(args.questions):
        br, last = syn.three_hop_program(i, nouns, attrs, rels)
        qs.append(syn.question(i, br, last, "yes", syn.feature_scene(i, args.objects, 2048)))
    pbs = Collater().collate(qs)
    for pb in pbs:
        pb.create_sparse_tensors()
    pbs = [pb.to_cuda(dev) for pb in pbs]

    worlds = []
    orig_build = model.build_scene

    def build_scene(*a, **k):
        w = orig_build(*a, **k)
        worlds.append(w)
        return w

    model.build_scene = build_scene
    with torch.no_grad():
        res, traces = model(pbs, False, return_trace=True)
    world = worlds[0]
    lp_hip = res["log_probability"].cpu().numpy()

    # ---- float64 / float32 host evaluations of the same stages ---------------------------------------------------------------
    X = np.concatenate([q["scene"]["X"] for q in qs])
    img = np.repeat(np.arange(len(qs)), [q["scene"]["n"] for q in qs])
    oont = orc.Ontology(paths["attribute_file"], paths["class_file"], paths["vocabulary_file"], paths["relation_file"])

    def host_stages(dtype):
        w = {k: np.asarray(v, dtype) for k, v in weights.items()}

        def layers(prefix):
            idx = sorted({int(k.split(".")[-2]) for k in w if k.startswith(prefix) and k.endswith(".weight")})
            return [(w["%s%d.weight" % (prefix, i)], w["%s%d.bias" % (prefix, i)]) for i in idx]

        obj, pair, _ = orc.featurize_scene(X.astype(dtype), img, layers("_featurizer._featurizer_network._network."))
        hid_attr = orc.regular_mlp(obj, layers("_oracle._attribute_network._network."))
        rl = layers("_oracle._relation_network._network.")
        w1, b1 = rl[0]
        Dd = obj.shape[1]
        U = obj @ w1[:, :Dd].T + b1
        V = obj @ w1[:, Dd:2 * Dd].T
        hid_rel = orc.regular_mlp(pair, rl)
        ew, eb = w["_oracle._embedding_network._network.1.weight"], w["_oracle._embedding_network._network.1.bias"]
        return {"obj": obj, "hidden_attr": hid_attr, "uv": np.concatenate([U, V], 1), "hidden_rel": hid_rel, "emb": (ew, eb)}

    def torch_stages():
        """The reference's arithmetic: torch CPU fp32 modules."""
        import torch.nn.functional as F
        t = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in weights.items()}

        def layers(prefix):
            idx = sorted({int(k.split(".")[-2]) for k in t if k.startswith(prefix) and k.endswith(".weight")})
            return [(t["%s%d.weight" % (prefix, i)], t["%s%d.bias" % (prefix, i)]) for i in idx]

        def mlp(x, ls):
            for wgt, b in ls[:-1]:
                x = F.elu(F.linear(x, wgt, b))
            return torch.sigmoid(F.linear(x, *ls[-1]))

        Xt = torch.from_numpy(X.astype(np.float32))
        f = mlp(Xt[:, :-6], layers("_featurizer._featurizer_network._network."))
        wh = torch.stack([Xt[:, -6], Xt[:, -5], Xt[:, -6], Xt[:, -5]], 1).clamp(min=1.0)
        obj = torch.cat([f, Xt[:, -4:] / wh], 1)
        hid_attr = mlp(obj, layers("_oracle._attribute_network._network."))
        obj64, pair64, _ = orc.featurize_scene(X.astype(np.float32), img, [])          # pair geometry on the host in fp32
        ind0, ind1, ind2 = orc.pair_indices(img)
        pos = obj[:, -4:]
        x1, y1, w1_, h1 = (pos[ind1, k] for k in range(4))
        x2, y2, w2_, h2 = (pos[ind2, k] for k in range(4))
        dx = x1 + w1_ / 2.0 - x2 - w2_ / 2.0
        dy = y1 + h1 / 2.0 - y2 - h2 / 2.0
        dist = torch.sqrt(dx ** 2 + dy ** 2)
        ang = torch.asin(dy / dist.clamp(min=1e-10))
        pair = torch.cat([obj[ind1], obj[ind2], dist[:, None], ang[:, None], torch.sign(x2 - x1)[:, None], torch.sign(y2 - y1)[:, None]], 1)
        hid_rel = mlp(pair, layers("_oracle._relation_network._network."))
        return {"obj": obj.numpy(), "hidden_attr": hid_attr.numpy(), "hidden_rel": hid_rel.numpy()}

    s64, s32 = host_stages(np.float64), host_stages(np.float32)
    st = torch_stages()
    out = {"tag": args.tag, "objects": args.objects, "questions": args.questions, "lib": args.lib,
           "env": {k: os.environ.get(k) for k in ("DFOL_PAIR_MATH", "DFOL_DENSE_MATH") if os.environ.get(k)}, "stages": {}}
    S = out["stages"]
    S["obj"] = {"hip": stats(world._obj.cpu().numpy(), s64["obj"]), "numpy32": stats(s32["obj"], s64["obj"]), "torch32": stats(st["obj"], s64["obj"])}
    S["hidden_attr"] = {"hip": stats(world._hidden_attr.cpu().numpy(), s64["hidden_attr"]), "numpy32": stats(s32["hidden_attr"], s64["hidden_attr"]),
                        "torch32": stats(st["hidden_attr"], s64["hidden_attr"])}
    # (the fp16x2 pair kernel takes U | V in units of ln 2: the oracle scales the stacked first-layer weight by log2 e)
    uv_hip = world._uv.cpu().numpy() / (1.4426950408889634 if model._oracle._pair_kind() == "f16x2" else 1.0)
    S["uv"] = {"hip": stats(uv_hip, s64["uv"]), "numpy32": stats(s32["uv"], s64["uv"])}
    S["hidden_rel"] = {"numpy32": stats(s32["hidden_rel"], s64["hidden_rel"]), "torch32": stats(st["hidden_rel"], s64["hidden_rel"])}

    # requested relation tiles: LogSigmoid(hidden_rel . E[col] + b[col]) in the tile layout
    n = args.objects
    NS = world._NS
    ridx = np.asarray(oont.relation_index)
    entries = {id(e[0]): e for e in world._rel_tiles.values()}          # (an entry is filed under its operator's token list AND its lowered tokens)
    (tiles, orient, fused), = [e[:3] for e in entries.values()]
    tiles = tiles.float().cpu().numpy()
    relate_op = [ob for ob in pbs[0]._op_batch_list if ob._op_name == "relate"][0]
    from dfol_vqa_amd.fol_types import TokenType
    from dfol_vqa_amd.host_util import get_lowered
    low = get_lowered(relate_op._arguments[0], ont, TokenType.RELATION)
    full_cols = model._oracle._relation_full_columns(low.cols)
    ii, jj = np.nonzero(~np.eye(n, dtype=bool))

    def host_tiles(hid_rel, emb):
        ew, eb = emb
        ref = np.full((len(qs), n, n), -30.0, hid_rel.dtype)
        for q in range(len(qs)):
            h = hid_rel[q * n * (n - 1):(q + 1) * n * (n - 1)]
            x = h @ ew[full_cols[q]] + eb[full_cols[q]]
            ll = np.minimum(x, 0) - np.log1p(np.exp(-np.abs(x)))
            t = ref[q]
            t[ii, jj] = ll
            if orient[q]:
                ref[q] = t.T
        return ref

    t64 = host_tiles(s64["hidden_rel"], s64["emb"])
    t32 = host_tiles(s32["hidden_rel"], s32["emb"])
    tt = host_tiles(st["hidden_rel"], s32["emb"])
    off = np.broadcast_to(~np.eye(n, dtype=bool), t64.shape)
    S["rel_tiles"] = {"hip": stats(tiles[:, :n, :n], t64, off), "numpy32": stats(t32, t64, off), "torch32": stats(tt, t64, off)}
    S["rel_tiles_prob"] = {"hip": stats(np.exp(tiles[:, :n, :n]), np.exp(t64), off), "numpy32": stats(np.exp(t32), np.exp(t64), off)}

    # the logic alone: the product's logic kernels on the fp64-exact tiles / blocks (rounded to fp32) against the fp64 logic
    r64 = orc.run_questions(oont, qs, [q["scene"] for q in qs], np.float64, split=max(1, len(qs) // 8), weights=weights)
    r32 = orc.run_questions(oont, qs, [q["scene"] for q in qs], np.float32, split=max(1, len(qs) // 8), weights=weights)
    lp64, lp32 = np.asarray(r64["log_probability"], np.float64), np.asarray(r32["log_probability"], np.float64)
    well = lp64 >= -5
    S["final_lp"] = {"hip": stats(lp_hip, lp64, well), "numpy32": stats(lp32, lp64, well), "well_conditioned": int(well.sum())}
    S["final_p"] = {"hip": stats(np.exp(lp_hip), np.exp(lp64)), "numpy32": stats(np.exp(lp32), np.exp(lp64))}

    # ---- isolate the two halves: (A) the product's logic kernels on float64-exact likelihoods (rounded once to fp32);
    #      (B) a float64 evaluation of the logic on the product's likelihoods ------------------------------------------------------
    Q = len(qs)
    ops_by = {ob._op_name: ob for ob in pbs[0]._op_batch_list}
    low_n1 = get_lowered(ops_by["select"]._arguments[0], ont, TokenType.ATTRIBUTE)
    low_a = get_lowered(ops_by["filter"]._arguments[0], ont, TokenType.ATTRIBUTE)
    low_n2 = get_lowered(relate_op._arguments[2], ont, TokenType.ATTRIBUTE)
    hip_blocks = {k: v.cpu().numpy() for k, v in getattr(world, "_attr_blocks", {}).items()}

    def host_attr(hid, emb, low):
        ew, eb = emb
        out = np.full((Q, NS), -30.0, hid.dtype)
        for q in range(Q):
            x = hid[q * n:(q + 1) * n] @ ew[low.cols[q]] + eb[low.cols[q]]
            out[q, :n] = np.minimum(x, 0) - np.log1p(np.exp(-np.abs(x)))
        return out

    def logic64(b1, ba, b2, tl):
        """select -> filter -> relate -> exist in float64 on [Q, NS] attribute blocks and [Q, n, n] tiles (rows = summed-out variable)."""
        lp = np.zeros(Q)
        for q in range(Q):
            prev = np.minimum(b1[q, :n], 0) + np.minimum(ba[q, :n], 0)
            x = np.minimum(b2[q, :n], 0)
            l = np.minimum(tl[q], 0)                                 # l[r, c]: r = object of `prev`, c = object of x
            t = np.log(np.maximum(1 - np.exp(l + prev[:, None]), 1e-20))
            t[np.eye(n, dtype=bool)] = 0
            post = x + np.log(np.maximum(1 - np.exp(t.sum(0)), 1e-20))
            lp[q] = np.log(np.maximum(1 - np.exp(np.log(np.maximum(1 - np.exp(post), 1e-20)).sum()), 1e-20))
        return lp

    a64 = [host_attr(s64["hidden_attr"], s64["emb"], lo).astype(np.float64) for lo in (low_n1, low_a, low_n2)]
    chk = logic64(a64[0], a64[1], a64[2], t64)
    S["check_logic64_vs_oracle64"] = stats(chk, lp64, well)
    if all(id(lo) in hip_blocks for lo in (low_n1, low_a, low_n2)):
        hb = [hip_blocks[id(lo)].astype(np.float64) for lo in (low_n1, low_a, low_n2)]
        S["attr_blocks"] = {"hip": stats(hb[0][:, :n], a64[0][:, :n])}
        lpB = logic64(hb[0], hb[1], hb[2], tiles[:, :n, :n].astype(np.float64))
        S["B_logic64_on_hip_likelihoods"] = {"lp": stats(lpB, lp64, well), "p": stats(np.exp(lpB), np.exp(lp64))}
    a32 = [host_attr(s32["hidden_attr"], s32["emb"], lo).astype(np.float64) for lo in (low_n1, low_a, low_n2)]
    lpB32 = logic64(a32[0], a32[1], a32[2], t32.astype(np.float64))
    S["B_logic64_on_numpy32_likelihoods"] = {"lp": stats(lpB32, lp64, well), "p": stats(np.exp(lpB32), np.exp(lp64))}
    # (A) product logic kernels on exact likelihoods
    tdev = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)
    zeros, ones = torch.zeros(Q, NS, device=dev), torch.ones(Q, device=dev)
    e = [tdev(a) for a in a64]
    tile_pad = np.full((Q, NS, NS), -30.0, np.float32)
    tile_pad[:, :n, :n] = t64
    ident, n_obj = world._ident, world._n_obj
    att = L.filter_fwd(L.filter_fwd(zeros, e[0], ident, n_obj), e[1], ident, n_obj)
    xs = L.filter_fwd(zeros, e[2], ident, n_obj)
    post = L.relate_one_fwd(xs, att, tdev(tile_pad), ident, n_obj, ones)
    lpA = L.quantify_fwd(post, ones, ident, n_obj).cpu().numpy()
    lpA64 = logic64(*(np.asarray(x.cpu().numpy(), np.float64) for x in e), tile_pad[:, :n, :n].astype(np.float64))
    S["A_hip_logic_on_exact_likelihoods"] = {"lp": stats(lpA, lpA64, well), "p": stats(np.exp(lpA), np.exp(lpA64))}
    # the same with the stages split: exact posterior into the product's exist kernel, and the product's posterior into an exact exist
    post64 = np.zeros((Q, NS))
    for q in range(Q):
        prev = np.minimum(a64[0][q, :n], 0) + np.minimum(a64[1][q, :n], 0)
        l = np.minimum(t64[q], 0)
        t = np.log(np.maximum(1 - np.exp(l + prev[:, None]), 1e-20))
        t[np.eye(n, dtype=bool)] = 0
        post64[q, :n] = np.minimum(a64[2][q, :n], 0) + np.log(np.maximum(1 - np.exp(t.sum(0)), 1e-20))
    lp_q = L.quantify_fwd(tdev(post64), ones, ident, n_obj).cpu().numpy()
    ex = lambda pm: np.log(np.maximum(1 - np.exp(np.log(np.maximum(1 - np.exp(pm[:, :n]), 1e-20)).sum(1)), 1e-20))
    S["A1_hip_exist_on_exact_posterior"] = {"lp": stats(lp_q, ex(np.asarray(tdev(post64).cpu().numpy(), np.float64)), well)}
    S["A2_exact_exist_on_hip_posterior"] = {"lp": stats(ex(post.cpu().numpy().astype(np.float64)), lpA64, well)}
    # numpy fp32 logic (the reference's formulation, flat layout collapsed to blocks) on the same exact likelihoods
    f32 = np.float32

    def logic32(b1, ba, b2, tl):
        lp = np.zeros(Q, f32)
        for q in range(Q):
            prev = np.minimum(b1[q, :n], f32(0)) + np.minimum(ba[q, :n], f32(0))
            x = np.minimum(b2[q, :n], f32(0))
            l = np.minimum(tl[q], f32(0))
            t = np.log(np.maximum(f32(1) - np.exp(l + prev[:, None]), f32(1e-20)))
            t[np.eye(n, dtype=bool)] = 0
            post = x + np.log(np.maximum(f32(1) - np.exp(t.sum(0, dtype=f32)), f32(1e-20)))
            lp[q] = np.log(np.maximum(f32(1) - np.exp(np.log(np.maximum(f32(1) - np.exp(post), f32(1e-20))).sum(dtype=f32)), f32(1e-20)))
        return lp

    lpA32 = logic32(*(x.cpu().numpy() for x in e), tile_pad[:, :n, :n])
    if args.dump:
        np.savez_compressed(args.dump, b1=e[0].cpu().numpy(), ba=e[1].cpu().numpy(), b2=e[2].cpu().numpy(), tile=tile_pad, att=att.cpu().numpy(),
                            xs=xs.cpu().numpy(), post=post.cpu().numpy(), lpA=lpA, lpA64=lpA64, n=n)
    S["A_numpy32_logic_on_exact_likelihoods"] = {"lp": stats(lpA32, lpA64, well), "p": stats(np.exp(lpA32.astype(np.float64)), np.exp(lpA64))}

    # attention traces of the product against the fp64 oracle's (per operator), probability space
    try:
        _, tr64 = orc.run_questions(oont, qs, [q["scene"] for q in qs], np.float64, split=1, weights=weights, return_trace=True)
        names_ops = [ob._op_name for ob in pbs[0]._op_batch_list]
        for i, (name, t) in enumerate(zip(names_ops, traces[0])):
            if hasattr(t, "flat_log_attention") and hasattr(tr64[0][i], "att"):
                a = t._log_attention.cpu().numpy()[:, :n].reshape(len(qs), n)
                b = np.asarray(tr64[0][i].att, np.float64)
                b = np.stack([b[q, q * n:(q + 1) * n] for q in range(len(qs))])
                S["trace_%d_%s" % (i, name)] = {"hip_p": stats(np.exp(a), np.exp(b))}
    except Exception as e:          # the trace layout is a convenience, not the point
        S["trace_error"] = repr(e)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
