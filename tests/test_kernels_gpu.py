"""GPU parity of every C-ABI kernel against the CPU oracle (oracle/dfol_oracle.py) on seeded inputs.

The candidate is always called through the C-ABI library (dfol_vqa_amd._lib -> libdfolvqa.so).
Yardstick: the oracle in fp64 is the truth, the oracle in fp32 measures how much rounding noise an
fp32 evaluation of the reference's formulas carries (tests/golden_util.check_logprob).
"""

import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu  # noqa: E402
from oracle import dfol_oracle as orc  # noqa: E402
from dfol_vqa_amd import synthetic as syn  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from dfol_vqa_amd import _lib
    _lib.load()
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    return _lib


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def ns_of(n_max):
    return max(4, (int(n_max) + 3) // 4 * 4)


def scene_arrays(n_list, C=11, CR=5, seed=0, family="mix10"):
    rng = np.random.RandomState(seed)
    A = np.concatenate([syn.table_log_likelihood(rng, (n, C), family) for n in n_list])
    R = np.concatenate([syn.table_log_likelihood(rng, (n * (n - 1), CR), family) for n in n_list])
    img = np.repeat(np.arange(len(n_list)), n_list)
    obj_off = np.concatenate([[0], np.cumsum(n_list)]).astype(np.int32)
    pair_off = np.concatenate([[0], np.cumsum([n * (n - 1) for n in n_list])]).astype(np.int64)
    return A, R, img, obj_off, pair_off


def block_rows(flat, img, pq, n_list, NS, fill=0.0):
    """[P, O] flat rows -> [P, NS] blocks (own image only)."""
    out = np.full((len(pq), NS), fill, flat.dtype)
    for p, q in enumerate(pq):
        idx = np.nonzero(img == q)[0]
        out[p, :len(idx)] = flat[p, idx]
    return out


def block_tiles(flat, img, pq, NS, fill=-30.0, transpose=False):
    out = np.full((len(pq), NS, NS), fill, flat.dtype)
    for p, q in enumerate(pq):
        idx = np.nonzero(img == q)[0]
        t = flat[p][np.ix_(idx, idx)]
        out[p, :len(idx), :len(idx)] = t.T if transpose else t
    return out


class FakeOntology(object):
    def __init__(self, C, CR):
        self.arg_to_idx = {"c%d" % i: i + 1 for i in range(C)}
        self.relation_reversed = {i: i % CR for i in range(C)}


# ---------------------------------------------------------------------------------------------------
def test_gathers_and_normalize(L):
    n_list = [5, 1, 8, 3]
    C, CR = 11, 5
    A, R, img, obj_off, pair_off = scene_arrays(n_list, C, CR, 1)
    NS = ns_of(max(n_list))
    pq = np.array([0, 0, 1, 2, 2, 2, 3], np.int32)
    cols = np.array([3, 7, 0, 10, 2, 5, 9], np.int32)
    world = orc.World(FakeOntology(C, CR), A, R, img, np.float32, normalize=False)
    toks = ["c%d" % c for c in cols]
    ref = orc.attribute_log_likelihood(world, toks, pq)[:, :, 0]
    got = L.attr_gather(dev(A), dev(obj_off), dev(pq), dev(cols), NS).cpu().numpy()
    assert np.array_equal(got, block_rows(ref, img, pq, n_list, NS, -30.0))
    # a no-op column gives an all-default block
    cols2 = cols.copy()
    cols2[2] = -1
    got2 = L.attr_gather(dev(A), dev(obj_off), dev(pq), dev(cols2), NS).cpu().numpy()
    assert np.all(got2[2] == -30.0) and np.array_equal(got2[3], got[3])

    rcols = (cols % CR).astype(np.int32)
    refR = orc.relation_log_likelihood(world, toks, pq)[:, :, :, 0]
    n_obj = np.array(n_list, np.int32)
    for orient in (0, 1):
        gotR = L.rel_gather(dev(R), dev(pair_off), dev(n_obj), dev(pq), dev(rcols), NS, orient).cpu().numpy()
        assert np.array_equal(gotR, block_tiles(refR, img, pq, NS, -30.0, transpose=bool(orient)))

    # option normalisation over segments {0,1} {2} {3,4,5} {6}
    seg = np.array([0, 2, 3, 6, 7], np.int32)
    world_n = orc.World(FakeOntology(C, CR), A, R, img, np.float32, normalize=True)
    world_64 = orc.World(FakeOntology(C, CR), A, R, img, np.float64, normalize=True)
    for rank in (1, 2):
        if rank == 1:
            r32 = orc.attribute_log_likelihood(world_n, toks, pq)[:, :, 0]
            r64 = orc.attribute_log_likelihood(world_64, toks, pq)[:, :, 0]
            t = L.attr_gather(dev(A), dev(obj_off), dev(pq), dev(cols), NS)
            conv = lambda x: block_rows(x, img, pq, n_list, NS, -30.0)
        else:
            r32 = orc.relation_log_likelihood(world_n, toks, pq)[:, :, :, 0]
            r64 = orc.relation_log_likelihood(world_64, toks, pq)[:, :, :, 0]
            t = L.rel_gather(dev(R), dev(pair_off), dev(n_obj), dev(pq), dev(rcols), NS, 0)
            conv = lambda x: block_tiles(x, img, pq, NS, -30.0)
        L.option_normalize_(t, dev(seg), dev(pq), dev(n_obj), NS)
        got = t.cpu().numpy()
        b32, b64 = conv(r32), conv(r64)
        # padding / diagonal untouched (stay at the default)
        assert np.array_equal(got == -30.0, b32 == -30.0)
        assert np.allclose(got, b64, rtol=0, atol=2e-5)


def _logic_inputs(rng, n_list, k_list, family="mix10"):
    Q = len(n_list)
    pq = np.repeat(np.arange(Q), k_list).astype(np.int32)
    NS = ns_of(max(n_list))
    P = len(pq)
    prior_s = np.zeros((Q, NS), np.float32)
    prior_o = np.zeros((Q, NS), np.float32)
    tile = np.full((P, NS, NS), -30, np.float32)
    ll1 = np.full((P, NS), -30, np.float32)
    for q, n in enumerate(n_list):
        prior_s[q, :n] = np.minimum(syn.table_log_likelihood(rng, (n,), "unif") * 0.3, 0)
        prior_o[q, :n] = np.minimum(syn.table_log_likelihood(rng, (n,), "unif") * 0.3, 0)
    for p in range(P):
        n = n_list[pq[p]]
        t = syn.table_log_likelihood(rng, (n, n), family)
        t[np.arange(n), np.arange(n)] = -30
        tile[p, :n, :n] = t
        ll1[p, :n] = syn.table_log_likelihood(rng, (n,), family)
    return pq, NS, prior_s, prior_o, tile, ll1


@pytest.mark.parametrize("n_list,k_list", [([5, 1, 8, 3], [1, 1, 1, 1]), ([36, 36], [1, 2]), ([100, 37, 64, 2], [2, 1, 1, 3]),
                                           ([7], [1]), ([130, 256], [1, 1])])
def test_filter_relate_quantify(L, n_list, k_list):
    rng = np.random.RandomState(sum(n_list))
    pq, NS, prior_s, prior_o, tile, ll1 = _logic_inputs(rng, n_list, k_list)
    P, Q = len(pq), len(n_list)
    n_obj = np.array(n_list, np.int32)
    quant = (rng.uniform(size=(Q, 2)) < 0.6).astype(np.float32)[pq]
    for any_neg in (False, True):
        neg = (rng.uniform(size=P) < 0.5).astype(np.uint8) if any_neg else None
        active = np.ones(P, np.uint8)
        if P > 2:
            active[1] = 0
        # ---- filter
        got = L.filter_fwd(dev(prior_s), dev(ll1), dev(pq), dev(n_obj), None if neg is None else dev(neg), dev(active)).cpu().numpy()
        for p in range(P):
            n = n_list[pq[p]]
            if not active[p]:
                assert np.array_equal(got[p, :n], prior_s[pq[p], :n])
                continue
            outs = []
            for dt in (np.float32, np.float64):
                l = np.minimum(ll1[p, :n].astype(dt), 0)
                if any_neg:
                    l = orc.log_parametric_not(l, dt(neg[p]), 1)
                outs.append(prior_s[pq[p], :n].astype(dt) + l)
            gu.check_logprob(got[p, :n], outs[0], outs[1], "filter")
            assert np.all(got[p, n:] == 0)
        # ---- relate, both orientations, with want masks
        want = np.array([[3, 1, 2][i % 3] for i in range(P)], np.uint8)
        for orient in (0, 1):
            t_in = tile if orient == 0 else np.ascontiguousarray(tile.transpose(0, 2, 1))
            ps, po = L.relate_fwd(dev(prior_s), dev(prior_o), dev(t_in), dev(pq), dev(n_obj), dev(quant[:, 0]), dev(quant[:, 1]),
                                  None if neg is None else dev(neg), dev(active), dev(want), orient)
            # poison check: unwanted rows are left untouched, so pre-fill and compare wanted rows only
            ps, po = ps.cpu().numpy(), po.cpu().numpy()
            for p in range(P):
                q, n = pq[p], n_list[pq[p]]
                if not active[p]:
                    if want[p] & 1:
                        assert np.array_equal(ps[p, :n], prior_s[q, :n])
                    if want[p] & 2:
                        assert np.array_equal(po[p, :n], prior_o[q, :n])
                    continue
                if n < 2:
                    continue
                r = [orc.relate_block(prior_s[q, :n].astype(dt), prior_o[q, :n].astype(dt), tile[p, :n, :n].astype(dt),
                                      quant[p, 0], quant[p, 1], 0.0 if neg is None else float(neg[p]), any_neg)
                     for dt in (np.float32, np.float64)]
                if want[p] & 1:
                    gu.check_logprob(ps[p, :n], r[0][0], r[1][0], "relate post_s orient %d" % orient)
                    assert np.all(ps[p, n:] == 0)
                if want[p] & 2:
                    gu.check_logprob(po[p, :n], r[0][1], r[1][1], "relate post_o orient %d" % orient)
                    assert np.all(po[p, n:] == 0)
        # ---- quantify
        att = np.zeros((P, NS), np.float32)
        for p in range(P):
            att[p, :n_list[pq[p]]] = prior_o[pq[p], :n_list[pq[p]]] * 3
        lp = L.quantify_fwd(dev(att), dev(quant[:, 0]), dev(pq), dev(n_obj)).cpu().numpy()
        refs = []
        for dt in (np.float32, np.float64):
            r = []
            for p in range(P):
                a = att[p, :n_list[pq[p]]].astype(dt)
                r.append(orc.log_parametric_not(orc.log_parametric_not(a, dt(quant[p, 0]), 1).sum(keepdims=True), dt(quant[p, 0]), 1)[0])
            refs.append(np.array(r))
        gu.check_logprob(lp, refs[0], refs[1], "quantify")


def test_relate_lone_forall(L):
    rng = np.random.RandomState(5)
    pq, NS, prior_s, prior_o, tile, _ = _logic_inputs(rng, [6], [1])
    n_obj = np.array([6], np.int32)
    quant = np.array([[0.0, 1.0]], np.float32)
    ps, po = L.relate_fwd(dev(prior_s), dev(prior_o), dev(tile), dev(pq), dev(n_obj), dev(quant[:, 0]), dev(quant[:, 1]),
                          lone_forall_identity=True)
    bom = np.ones((1, 6), np.float64)
    prior = np.stack([prior_s[:, :6], prior_o[:, :6]], 1).astype(np.float64)
    ref = orc.logic_cell(prior, tile[:, :6, :6, None].astype(np.float64), quant.astype(np.float64), bom)
    assert np.allclose(ps.cpu().numpy()[0, :6], ref[0, 0], atol=2e-5)
    assert np.allclose(po.cpu().numpy()[0, :6], ref[0, 1], atol=2e-5)


def test_g2_goldens_through_kernels(L):
    """The reference's own BatchBayesianLogicCell outputs (flat layout) reproduced by the block kernels."""
    a, meta = gu.load("g2_logic_cell")
    for case in meta["cases"]:
        n = case["name"]
        img, pq = a[n + "_img"], a[n + "_pq"].astype(np.int32)
        n_list = case["n"]
        NS = ns_of(max(n_list))
        neg = a[n + "_neg"].astype(np.uint8) if (n + "_neg") in a.files else None
        prior, ll, quant = a[n + "_prior"], a[n + "_ll"], a[n + "_quant"]
        Q = len(n_list)
        ident = np.arange(Q)
        n_obj = np.array(n_list, np.int32)
        own = img[None, :] == pq[:, None]
        if case["arity"] == 1:
            got = L.filter_fwd(dev(block_rows(prior[:, 0, :], img, ident, n_list, NS)), dev(block_rows(ll[:, :, 0], img, pq, n_list, NS, -30.0)),
                               dev(pq), dev(n_obj), None if neg is None else dev(neg)).cpu().numpy()
            gu.check_logprob(np.concatenate([got[p, :n_list[pq[p]]] for p in range(len(pq))]), a[n + "_out_f32"][:, 0, :][own],
                             a[n + "_out_f64"][:, 0, :][own], n)
            continue
        lone = len(pq) == 1
        ps, po = L.relate_fwd(dev(block_rows(prior[:, 0, :], img, ident, n_list, NS)), dev(block_rows(prior[:, 1, :], img, ident, n_list, NS)),
                              dev(block_tiles(ll[:, :, :, 0], img, pq, NS)), dev(pq), dev(n_obj), dev(quant[:, 0]), dev(quant[:, 1]),
                              None if neg is None else dev(neg), lone_forall_identity=lone)
        for k, got in ((0, ps.cpu().numpy()), (1, po.cpu().numpy())):
            flat = np.concatenate([got[p, :n_list[pq[p]]] for p in range(len(pq))])
            if "stress" in n:
                assert np.abs(np.exp(flat) - np.exp(a[n + "_out_f32"][:, k, :][own])).max() <= 2e-6
            else:
                gu.check_logprob(flat, a[n + "_out_f32"][:, k, :][own], a[n + "_out_f64"][:, k, :][own], n)


def test_small_vector_ops(L):
    rng = np.random.RandomState(3)
    P, NS, Q = 37, 12, 9
    x = -rng.gamma(1.0, 1.5, (P, NS)).astype(np.float32)
    y = -rng.gamma(1.0, 1.5, (P, NS)).astype(np.float32)
    xq, yq = (rng.uniform(size=P) < 0.5).astype(np.float32), (rng.uniform(size=P) < 0.5).astype(np.float32)
    g = (rng.uniform(size=P) < 0.5).astype(np.float32)
    o, oq = L.gate(dev(x), dev(y), dev(xq), dev(yq), dev(g))
    assert np.array_equal(o.cpu().numpy(), np.where(g[:, None] > 0, x, y))
    assert np.array_equal(oq.cpu().numpy(), np.where(g > 0, xq, yq))
    idx = rng.randint(0, P, 50).astype(np.int32)
    assert np.array_equal(L.gather_rows(dev(x), dev(idx)).cpu().numpy(), x[idx])
    seg = np.sort(rng.choice(np.arange(1, P), Q - 1, replace=False))
    seg_off = np.concatenate([[0], seg, [P]]).astype(np.int32)
    ref = np.stack([x[seg_off[i]:seg_off[i + 1]].astype(np.float64).sum(0) for i in range(Q)])
    assert np.allclose(L.segment_sum_rows(dev(x), dev(seg_off)).cpu().numpy(), ref, atol=1e-5)
    a, b = x[:, 0].copy(), y[:, 0].copy()
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    assert np.array_equal(L.logic(L.LOGIC_AND, dev(a), dev(b)).cpu().numpy(), a + b)
    gu.check_logprob(L.logic(L.LOGIC_OR, dev(a), dev(b)).cpu().numpy(), orc.log_or(a, b), orc.log_or(a64, b64), "or")
    gu.check_logprob(L.logic(L.LOGIC_NOT, dev(a)).cpu().numpy(), orc.log_not(a), orc.log_not(a64), "not")
    gu.check_logprob(L.parametric_not(dev(x), dev(g)).cpu().numpy(), orc.log_parametric_not(x, g[:, None], 1),
                     orc.log_parametric_not(x.astype(np.float64), g[:, None].astype(np.float64), 1), "pnot")
    so = L.segment_or(dev(a), dev(seg_off)).cpu().numpy()
    r32 = np.array([orc.log_not(orc.log_not(a[seg_off[i]:seg_off[i + 1]]).sum(keepdims=True))[0] for i in range(Q)])
    r64 = np.array([orc.log_not(orc.log_not(a64[seg_off[i]:seg_off[i + 1]]).sum(keepdims=True))[0] for i in range(Q)])
    gu.check_logprob(so, r32, r64, "segment_or")
    # the same aggregate as the reference writes it (dfol_segment_or_ref_f32: the callers that negate it next) - the reference's fp32 values to
    # rounding, and its SATURATION: one option at log 1 - 1e-9 makes the aggregate exactly 0 and its negation exactly log(1e-20)
    sr = L.segment_or(dev(a), dev(seg_off), as_written=True).cpu().numpy()
    assert np.abs(np.exp(sr.astype(np.float64)) - np.exp(r32.astype(np.float64))).max() <= 2e-7
    sat = np.array([-1e-9, -3.0, -0.5, -2.0, -1e-9], np.float32)
    sat_off = np.array([0, 2, 5], np.int32)
    sw = L.segment_or(dev(sat), dev(sat_off), as_written=True)
    assert np.array_equal(sw.cpu().numpy(), np.zeros(2, np.float32))
    assert np.allclose(L.logic(L.LOGIC_NOT, sw).cpu().numpy(), np.log(1e-20), atol=1e-4)
    g = torch.tensor(sat, device="cuda", requires_grad=True)          # and the zero gradient of the reference's clamp
    from dfol_vqa_amd import ops as OPS
    OPS.logic(L.LOGIC_NOT, OPS.segment_or(g, dev(sat_off), as_written=True)).sum().backward()
    assert float(g.grad.abs().max()) == 0.0
    pq = rng.randint(0, Q, P).astype(np.int32)
    n_obj = rng.randint(1, NS + 1, Q).astype(np.int32)
    prior = -rng.gamma(1.0, 1.0, (Q, NS)).astype(np.float32)
    imp = L.implication(dev(prior), dev(x), dev(pq), dev(n_obj)).cpu().numpy()
    for p in range(P):
        n = n_obj[pq[p]]
        r32 = orc.log_not(prior[pq[p], :n] + orc.log_not(x[p, :n]))
        r64 = orc.log_not(prior[pq[p], :n].astype(np.float64) + orc.log_not(x[p, :n].astype(np.float64)))
        gu.check_logprob(imp[p, :n], r32, r64, "implication")
        assert np.all(imp[p, n:] == 0)
    less = (rng.uniform(size=P) < 0.5).astype(np.float32)
    cmp_ = L.compare(dev(a), dev(b), dev(less)).cpu().numpy()
    st = np.stack([a64, b64], 1)
    ls = st - np.log(np.exp(st).sum(1, keepdims=True))
    r64 = orc.log_parametric_not(ls, less[:, None].astype(np.float64), 1)
    assert np.allclose(np.exp(cmp_), np.exp(r64), atol=1e-6)


@pytest.mark.parametrize("M,N,K,ldx_extra,act", [(300, 70, 50, 0, 0), (257, 300, 256, 0, 1), (513, 256, 516, 0, 2), (200, 333, 300, 0, 3),
                                                 (130, 512, 2048, 6, 1), (64, 49, 12, 1, 3), (1, 5, 3, 0, 0)])
def test_linear_act(L, M, N, K, ldx_extra, act):
    rng = np.random.RandomState(M + N + K)
    Xfull = rng.uniform(-1, 1, (M, K + ldx_extra)).astype(np.float32)
    W = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    xt = dev(Xfull)[:, :K]
    got = L.linear_act(xt, dev(W), dev(b), act).cpu().numpy()
    z = Xfull[:, :K].astype(np.float64) @ W.astype(np.float64).T + b
    ref = [z, orc._sigmoid(z), orc._elu(z), orc._log_sigmoid(z)][act]
    assert np.allclose(got, ref, rtol=2e-5, atol=2e-5)
    # asymmetric operands: a transposed write would not survive this
    assert not np.allclose(got[: min(M, N), : min(M, N)], got[: min(M, N), : min(M, N)].T) or min(M, N) < 2


@pytest.mark.parametrize("math", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("M,N,K,ldx_extra,act", [(300, 70, 52, 0, 0), (257, 300, 256, 0, 1), (513, 256, 516, 0, 2), (200, 333, 300, 0, 3),
                                                 (130, 512, 2048, 8, 2), (64, 49, 12, 4, 3), (1, 5, 4, 0, 0), (1000, 768, 516, 0, 0), (700, 512, 2048, 6, 2)])
def test_linear_act_split(L, M, N, K, ldx_extra, act, math):
    """The split GEMMs - bf16x3: three exact bf16 pieces per operand, six piece products on the bf16 matrix pipe; f16x2 (the default of the
    forward products): two fp16 pieces, three products on the fp16 pipe, rows of W scaled by powers of two - are as close to the exact
    product as the fp32 matrix pipe: both against float64, ragged M / N / K edges, strided X.  bf16x3 holds that for rows of X of ANY
    scale (its pieces have fp32's exponent range); f16x2 splits X unscaled, so its operand error is max(2^-22 |x|, 2^-25): the same
    bound for rows of order one, an ABSOLUTE one for smaller rows (which is why the backward products stay on bf16x3)."""
    from dfol_vqa_amd import _lib
    rng = np.random.RandomState(M + N + K)
    Xfull = (rng.uniform(-1, 1, (M, K + ldx_extra)) * np.exp(rng.uniform(-6, 2, (M, 1)))).astype(np.float32)    # rows of very different scale
    W = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    W[N // 2] *= 1e-6                                            # rows of W of any scale (f16x2 scales every row by its own power of two)
    W[N // 3] *= 1e4
    b = rng.normal(size=N).astype(np.float32)
    xt = dev(Xfull)[:, :K]
    z64 = Xfull[:, :K].astype(np.float64) @ W.astype(np.float64).T
    z = z64 + b
    ref = [z, orc._sigmoid(z), orc._elu(z), orc._log_sigmoid(z)][act]
    with _lib.dense_math(math):
        got = _lib.linear_act_split(xt, dev(W), dev(b), act).cpu().numpy()
        # pre-activation accuracy against the fp32 kernel's, relative to the magnitude of the terms
        pre = _lib.linear_act_split(xt, dev(W), None, 0).cpu().numpy()
    # (f16x2: an input element below 2^-3 carries an ABSOLUTE error of 2^-25, which a weight row scaled by 1e4 turns into a visible one:
    # the documented operand model, DESIGN 3.4 - the tolerance of the activations follows it)
    model = (np.maximum(np.abs(Xfull[:, :K]).astype(np.float64), 2.0 ** -3) @ np.abs(W).astype(np.float64).T) * 2.0 ** -21 if math == "f16x2" else 0.0
    assert np.all(np.abs(got - ref) <= 2e-5 * np.abs(ref) + 2e-5 * max(1.0, np.abs(z).max() if act in (0, 2) else 1.0) + model)
    with _lib.dense_math("f32"):
        pre32 = L.linear_act(xt, dev(W), None, 0).cpu().numpy()
    absx = np.abs(Xfull[:, :K]).astype(np.float64)
    if math == "f16x2":
        absx = np.maximum(absx, 2.0 ** -3)                       # an element below 2^-3 carries an absolute error of 2^-25
    scale = absx @ np.abs(W).astype(np.float64).T + 1e-30
    e_split, e_f32 = np.abs(pre - z64) / scale, np.abs(pre32 - z64) / scale
    assert e_split.max() <= max(2.0 ** -21, 1.5 * e_f32.max()) and e_split.mean() <= 1.5 * e_f32.mean() + (2.0 ** -23 if math == "f16x2" else 2.0 ** -25), \
        (e_split.max(), e_f32.max(), e_split.mean(), e_f32.mean())
    if math == "f16x2":                                          # rows of order one: as close as the fp32 pipe, relative to the terms themselves
        big = np.abs(Xfull[:, :K]).max(1) >= 1.0
        if big.any():
            rel = np.abs(Xfull[:, :K]).astype(np.float64) @ np.abs(W).astype(np.float64).T + 1e-30
            e1, e2 = (np.abs(pre - z64) / rel)[big], (np.abs(pre32 - z64) / rel)[big]
            assert e1.mean() <= 1.5 * e2.mean() + 2.0 ** -25, (e1.mean(), e2.mean())
    # the weight image is cached per weight version
    w = dev(W)
    with _lib.dense_math(math):
        a = _lib.linear_act_split(xt, w, None, 0)
        w.mul_(2.0)
        assert torch.allclose(_lib.linear_act_split(xt, w, None, 0), 2 * a, rtol=1e-6, atol=0)
        # the input-gradient product (transpose_w) is the bf16x3 kernel in every mode: exact operand range
        if N % 4 == 0 and M > 1:
            dz = dev((rng.normal(size=(M, N)) * 1e-7).astype(np.float32))                    # gradients of tiny magnitude
            gx = _lib.linear_act_split(dz, w, None, 0, transpose_w=True).cpu().numpy()
            gx64 = dz.cpu().numpy().astype(np.float64) @ w.cpu().numpy().astype(np.float64)
            sc = np.abs(dz.cpu().numpy()).astype(np.float64) @ np.abs(w.cpu().numpy()).astype(np.float64) + 1e-300
            # (bf16x3 pieces are cut by truncation: the three dropped piece products are each below 2^-21 of the product)
            assert (np.abs(gx - gx64) / sc).max() <= 2.0 ** -18


def test_bf16x3_kernels_bitwise_repeatable(L):
    """The two LDS-pipelined bf16x3 kernels at the bench shape, 25 launches each: every result bitwise equal to the first.  (The pair
    kernel's chunk buffers are filled by LDS-DMA, whose completion no barrier waits for by itself: a missing drain shows up here.)"""
    from dfol_vqa_amd import _lib
    torch.manual_seed(0)
    Q, N, HID1, HID2, C, K = 256, 100, 256, 300, 333, 2
    O = Q * N
    uv = torch.randn(O, 2 * HID1, device="cuda") * 0.5
    pos = torch.rand(O, 4, device="cuda") * 0.5 + 0.05
    wg = torch.randn(HID1, 4, device="cuda") * 0.3
    w2 = torch.zeros(320, HID1, device="cuda")
    w2[:HID2] = torch.randn(HID2, HID1, device="cuda") / 16
    b2, E, be = torch.randn(HID2, device="cuda"), torch.randn(C, HID2, device="cuda") / 17, torch.randn(C, device="cuda")
    n_o = torch.full((Q,), N, dtype=torch.int32, device="cuda")
    off = (torch.arange(Q + 1, device="cuda") * N).to(torch.int32)
    req_col = torch.randint(0, C, (K, Q), dtype=torch.int32, device="cuda")
    req_tile = torch.arange(K * Q, dtype=torch.int32, device="cuda").view(K, Q)
    for pack, run in ((_lib.pair_pack_w2_split, _lib.pair_ll_split), (_lib.pair_pack_w2_h2, _lib.pair_ll_h2)):
        w2s = pack(w2, HID2)
        first = None
        for _ in range(25):
            tiles = torch.full((K * Q, 104, 104), -30.0, device="cuda")
            run(uv, HID1, pos, wg, w2s, b2, HID2, E, be, n_o, off, N, req_col, req_tile, None, tiles)
            first = tiles.clone() if first is None else first
            assert torch.equal(first, tiles)
            assert torch.isfinite(tiles).all()
    x = torch.rand(25600, 2054, device="cuda")[:, :2048]
    w, b = torch.randn(512, 2048, device="cuda") / 45, torch.randn(512, device="cuda")
    for math in ("f16x2", "bf16x3"):
        first = None
        with _lib.dense_math(math):
            for _ in range(25):
                y = _lib.linear_act_split(x, w, b, 1)
                first = y.clone() if first is None else first
                assert torch.equal(first, y)


def test_box_and_pair_features(L):
    n_list = [5, 1, 7]
    F = 10
    X = np.concatenate([syn.feature_scene(50 + i, n, F)["X"] for i, n in enumerate(n_list)])
    img = np.repeat(np.arange(len(n_list)), n_list)
    obj_ref, pair_ref, _ = orc.featurize_scene(X.astype(np.float64), img, [])
    O, D = X.shape[0], F + 4
    obj = torch.zeros(O, D, device="cuda")
    obj[:, :F] = dev(X)[:, :F]
    L.box_positions(dev(X), obj, F)
    assert np.allclose(obj.cpu().numpy(), obj_ref, rtol=1e-6, atol=1e-7)
    obj_off = np.concatenate([[0], np.cumsum(n_list)]).astype(np.int32)
    pair_off = np.concatenate([[0], np.cumsum([n * (n - 1) for n in n_list])]).astype(np.int64)
    pair = L.pair_features(obj, D, dev(obj_off), dev(pair_off), len(n_list), max(n_list), int(pair_off[-1])).cpu().numpy()
    assert pair.shape == pair_ref.shape
    ang = 2 * D + 1                                   # asin is ill-conditioned at +-1: compare its sine there
    rest = np.ones(pair.shape[1], bool)
    rest[ang] = False
    assert np.allclose(pair[:, rest], pair_ref[:, rest], rtol=1e-5, atol=2e-6)
    assert np.allclose(np.sin(pair[:, ang]), np.sin(pair_ref[:, ang]), atol=2e-6)
    assert np.allclose(pair[:, ang], pair_ref[:, ang], atol=2e-3)


def test_errors_are_loud(L):
    with pytest.raises(L.DfolError):
        L.filter_fwd(torch.zeros(2, 8), torch.zeros(2, 8), torch.zeros(2, dtype=torch.int32), torch.ones(2, dtype=torch.int32))
    with pytest.raises(L.DfolError):   # NS not a multiple of 4
        z = torch.zeros(2, 6, device="cuda")
        L.filter_fwd(z, z, torch.zeros(2, dtype=torch.int32, device="cuda"), torch.ones(2, dtype=torch.int32, device="cuda"))


@pytest.mark.parametrize("n_list", [[5, 1, 8, 3], [36, 20, 33], [100, 37, 64, 2], [130, 256]])
def test_relate_one_equals_generic(L, n_list):
    """The fused single-posterior kernel == three gates + the generic arity-2 cell, for both directions and ragged images."""
    rng = np.random.RandomState(sum(n_list) + 7)
    pq, NS, prior_s, prior_o, tile, _ = _logic_inputs(rng, n_list, [1] * len(n_list))
    P = len(pq)
    n_obj = np.array(n_list, np.int32)
    is_subject = (rng.uniform(size=P) < 0.5)
    x_att, prev_att = prior_s, prior_o                      # x = fresh select, prev = incoming attention
    quant_prev = (rng.uniform(size=P) < 0.7).astype(np.float32)
    for any_neg in (False, True):
        neg = (rng.uniform(size=P) < 0.5).astype(np.uint8) if any_neg else None
        active = np.ones(P, np.uint8)
        if P > 2:
            active[2] = 0
        # generic route: subject = x if is_subject else prev, object = the other; keep the posterior of x's variable
        flag = is_subject.astype(np.float32)
        subj = np.where(flag[:, None] > 0, x_att, prev_att)
        obj = np.where(flag[:, None] > 0, prev_att, x_att)
        ones = np.ones(P, np.float32)
        qs = np.where(flag > 0, ones, quant_prev)           # x is a fresh EXISTS set
        qo = np.where(flag > 0, quant_prev, ones)
        want = np.where(is_subject, 1, 2).astype(np.uint8)
        ps, po = L.relate_fwd(dev(subj), dev(obj), dev(tile), dev(pq), dev(n_obj), dev(qs), dev(qo), None if neg is None else dev(neg),
                              dev(active), dev(want), 0)
        ref = np.where(flag[:, None] > 0, ps.cpu().numpy(), po.cpu().numpy())
        # fused route: each tile stored with the summed-out (prev's) variable along rows
        oriented = np.stack([tile[p].T if is_subject[p] else tile[p] for p in range(P)]).copy()
        got = L.relate_one_fwd(dev(x_att), dev(prev_att), dev(oriented), dev(pq), dev(n_obj), dev(quant_prev),
                               None if neg is None else dev(neg), dev(active)).cpu().numpy()
        for p in range(P):
            n = n_list[p]
            if not active[p]:
                assert np.array_equal(got[p, :n], prev_att[p, :n])
                continue
            assert np.allclose(got[p, :n], ref[p, :n], rtol=0, atol=2e-5 * max(1.0, np.abs(ref[p, :n]).max())), (p, n)
            assert np.all(got[p, n:] == 0)


@pytest.mark.parametrize("n_list", [[5, 1, 8, 3], [36, 20, 33], [100, 37, 64, 2], [130, 256], [17, 9, 2, 12, 16]])
def test_relate_exists_fast_path(L, n_list):
    """Un-negated EXISTS/EXISTS predicates on tiles with an absent diagonal take the product path (one exp per element,
    logs of 5-factor products); it must agree with the oracle, also where a factor is clamped (prior = likelihood = log 1,
    which sends the wave back to the general code), with garbage in the padding, and with mixed quantifiers in the batch."""
    rng = np.random.RandomState(sum(n_list) + 11)
    k_list = [1 + (i % 2) for i in range(len(n_list))]
    pq, NS, prior_s, prior_o, tile, _ = _logic_inputs(rng, n_list, k_list)
    P, Q = len(pq), len(n_list)
    n_obj = np.array(n_list, np.int32)
    quant = np.ones((P, 2), np.float32)
    quant[P - 1] = (0.0, 1.0)                                  # one predicate that must not take the fast path
    # clamped factors: an exactly-certain prior meeting an exactly-certain likelihood
    q0, n0 = pq[0], n_list[pq[0]]
    if n0 >= 3:
        prior_s[q0, 1] = 0.0
        prior_o[q0, 2] = 0.0
        tile[0, 1, 2] = 0.0
    # garbage in the padding of the tiles and priors must not matter
    for p in range(P):
        n = n_list[pq[p]]
        tile[p, n:, :] = np.nan
        tile[p, :, n:] = 7.0
    for q, n in enumerate(n_list):
        prior_s[q, n:] = 0.0
        prior_o[q, n:] = 3.0
    neg = np.zeros(P, np.uint8)
    neg[min(1, P - 1)] = 1
    for use_neg in (False, True):
        for want_bits in (3, 1, 2):
            want = np.full(P, want_bits, np.uint8)
            for orient in (0, 1):
                t_in = tile if orient == 0 else np.ascontiguousarray(tile.transpose(0, 2, 1))
                outs = {}
                for da in (True, False):
                    ps, po = L.relate_fwd(dev(prior_s), dev(prior_o), dev(t_in), dev(pq), dev(n_obj), dev(quant[:, 0]), dev(quant[:, 1]),
                                          dev(neg) if use_neg else None, None, dev(want), orient, diag_absent=da)
                    outs[da] = (ps.cpu().numpy(), po.cpu().numpy())
                for p in range(P):
                    q, n = pq[p], n_list[pq[p]]
                    if n < 2:
                        continue
                    r = [orc.relate_block(prior_s[q, :n].astype(dt), prior_o[q, :n].astype(dt), tile[p, :n, :n].astype(dt),
                                          quant[p, 0], quant[p, 1], float(neg[p]) if use_neg else 0.0, use_neg)
                         for dt in (np.float32, np.float64)]
                    for da in (True, False):
                        ps, po = outs[da]
                        if want_bits & 1:
                            gu.check_logprob(ps[p, :n], r[0][0], r[1][0], "fast relate post_s da=%s" % da)
                            assert np.all(ps[p, n:] == 0)
                        if want_bits & 2:
                            gu.check_logprob(po[p, :n], r[0][1], r[1][1], "fast relate post_o da=%s" % da)
                            assert np.all(po[p, n:] == 0)
    # the fused single-posterior kernel shares the product path
    x_att, prev_att = prior_s[pq], prior_o
    got = L.relate_one_fwd(dev(x_att), dev(prev_att), dev(np.ascontiguousarray(tile.transpose(0, 2, 1))), dev(pq), dev(n_obj),
                           dev(np.ones(P, np.float32))).cpu().numpy()
    for p in range(P):
        q, n = pq[p], n_list[pq[p]]
        if n < 2:
            continue
        r = [orc.relate_block(x_att[p, :n].astype(dt), prev_att[q, :n].astype(dt), tile[p, :n, :n].astype(dt), 1.0, 1.0, 0.0, False)
             for dt in (np.float32, np.float64)]
        gu.check_logprob(got[p, :n], r[0][0], r[1][0], "relate_one product path")


@pytest.mark.parametrize("n_list", [[5, 1, 8, 3], [36, 20, 33], [100, 37, 64, 2], [130, 256], [17, 9, 2, 12, 16],
                                    [300, 12, 270]])      # NS = 300 > 256: the plain one-workgroup-per-predicate kernels
def test_relate_negated_and_forall_fast_paths(L, n_list):
    """Negated and FOR_ALL predicates take transcendental-poor forms too (negated EXISTS: product path on 1 - E with the diagonal
    masked; FOR_ALL: plain sums of l' + prior, no exp / log at all when un-negated) and fall back to the clamping general code
    when a term reaches the reference's 1e-20 floor.  Every (negation, quantifier) combination through both Relate kernels against
    the oracle, with planted clamps (1 - E = 0; l' + prior below log 1e-20; E = P = 1) and garbage in the padding."""
    rng = np.random.RandomState(sum(n_list) + 23)
    k_list = [1 + (i % 2) for i in range(len(n_list))]
    pq, NS, prior_s, prior_o, tile, _ = _logic_inputs(rng, n_list, k_list)
    P, Q = len(pq), len(n_list)
    n_obj = np.array(n_list, np.int32)
    q0, n0 = pq[0], n_list[pq[0]]
    if n0 >= 4:                                               # predicate 0 carries every kind of clamp
        tile[0, 1, 2] = 0.0                                   # negated: 1 - E = 0
        prior_s[q0, 1] = 0.0
        prior_o[q0, 2] = 0.0                                  # un-negated EXISTS: E = P = 1
        prior_o[q0, 3] = -40.0                                # FOR_ALL: l' + prior < log(1e-20)
        tile[0, 0, 3] = -20.0
    for p in range(P):
        n = n_list[pq[p]]
        tile[p, n:, :] = np.nan
        tile[p, :, n:] = 7.0
    for q, n in enumerate(n_list):
        prior_s[q, n:] = 0.0
        prior_o[q, n:] = 3.0
    neg_sets = [None, np.ones(P, np.uint8), (np.arange(P) % 2).astype(np.uint8)]
    for neg in neg_sets:
        use_neg = neg is not None
        for qs_v, qo_v in ((1.0, 1.0), (0.0, 0.0), (1.0, 0.0), (0.0, 1.0)):
            quant = np.tile(np.asarray([[qs_v, qo_v]], np.float32), (P, 1))
            refs = {}
            for p in range(P):
                q, n = pq[p], n_list[pq[p]]
                if n >= 2:
                    refs[p] = [orc.relate_block(prior_s[q, :n].astype(dt), prior_o[q, :n].astype(dt), tile[p, :n, :n].astype(dt),
                                                quant[p, 0], quant[p, 1], float(neg[p]) if use_neg else 0.0, use_neg)
                               for dt in (np.float32, np.float64)]
            for orient in (0, 1):
                t_in = tile if orient == 0 else np.ascontiguousarray(tile.transpose(0, 2, 1))
                for da in (True, False):
                    for want_bits in (3, 1, 2):
                        ps, po = L.relate_fwd(dev(prior_s), dev(prior_o), dev(t_in), dev(pq), dev(n_obj), dev(quant[:, 0]), dev(quant[:, 1]),
                                              dev(neg) if use_neg else None, None, dev(np.full(P, want_bits, np.uint8)), orient, diag_absent=da)
                        ps, po = ps.cpu().numpy(), po.cpu().numpy()
                        for p, r in refs.items():
                            n = n_list[pq[p]]
                            what = "neg=%s q=(%g,%g) orient=%d da=%s want=%d p=%d" % (None if neg is None else int(neg[p]), qs_v, qo_v, orient, da, want_bits, p)
                            if want_bits & 1:
                                gu.check_logprob(ps[p, :n], r[0][0], r[1][0], "relate post_s " + what)
                                assert np.all(ps[p, n:] == 0)
                            if want_bits & 2:
                                gu.check_logprob(po[p, :n], r[0][1], r[1][1], "relate post_o " + what)
                                assert np.all(po[p, n:] == 0)
            # the fused single-posterior kernel: x = the object variable (fresh), prev = the subject variable with quantifier qs_v;
            # tiles stored with the summed-out (subject) variable along rows = the reference orientation
            x_att, prev_att = prior_o[pq], prior_s
            got = L.relate_one_fwd(dev(x_att), dev(prev_att), dev(tile), dev(pq), dev(n_obj), dev(quant[:, 0]),
                                   dev(neg) if use_neg else None).cpu().numpy()
            for p, r in refs.items():
                n = n_list[pq[p]]
                gu.check_logprob(got[p, :n], r[0][1], r[1][1], "relate_one neg=%s q_prev=%g p=%d" % (None if neg is None else int(neg[p]), qs_v, p))
                assert np.all(got[p, n:] == 0)


def _pair_ref64(n_list, off, uv, pos, wg, w2, b2, E, be, req_col, req_tile, orient, hid1, hid2, K, NS):
    """float64 restatement of the fused pair MLP (batch_gqa_boxfeatures_pipeline.py:243-281 pair features with the first layer split per
    object, gqa_interpreter_experiments.py:18-36 the MLP, classifier_oracle.py:145-156 LogSigmoid of the requested columns)."""
    Q = len(n_list)
    ref = np.full((K * Q, NS, NS), -30.0)
    for q, n in enumerate(n_list):
        f = off[q]
        for s in range(n):
            for o in range(n):
                if s == o:
                    continue
                x1, y1, w1, h1 = pos[f + s].astype(np.float64)
                x2, y2, w2_, h2 = pos[f + o].astype(np.float64)
                dx, dy = x1 + w1 / 2 - x2 - w2_ / 2, y1 + h1 / 2 - y2 - h2 / 2
                dist = np.sqrt(dx * dx + dy * dy)
                geo = np.array([dist, np.arcsin(dy / max(dist, 1e-10)), np.sign(x2 - x1), np.sign(y2 - y1)])
                z = uv[f + s, :hid1].astype(np.float64) + uv[f + o, hid1:].astype(np.float64) + wg.astype(np.float64) @ geo
                z = np.where(z > 0, z, np.expm1(z))
                h = 1.0 / (1.0 + np.exp(-(w2[:hid2].astype(np.float64) @ z + b2)))
                for k in range(K):
                    c = req_col[k, q]
                    if c >= 0:
                        v = orc._log_sigmoid(np.array([h @ E[c].astype(np.float64) + be[c]]))[0]
                        if orient[k, q]:
                            ref[req_tile[k, q], o, s] = v
                        else:
                            ref[req_tile[k, q], s, o] = v
    return ref


@pytest.mark.parametrize("hid1,hid2,K", [(256, 300, 3), (256, 300, 40), (256, 300, 60), (256, 320, 2), (256, 270, 2), (224, 288, 3),
                                         (64, 200, 2), (32, 12, 2)])
def test_pair_ll_kernels_against_torch(L, hid1, hid2, K):
    """The fused pair MLP kernels (every geometry: 32x32 tiles, 16x16 tiles with 8 wavefronts, the packed occupancy-2 kernels with 16
    and 32 slots per wavefront) against an fp64 restatement: ragged images incl. 1- and 2-object ones, unrequested slots, both tile
    orientations, more request rows than the epilogue can stage in LDS."""
    from dfol_vqa_amd import _lib
    rng = np.random.RandomState(hid1 + hid2 + K)
    n_list = [7, 1, 13, 2, 30, 5]
    Q, O, NS, C = len(n_list), sum(n_list), 32, 50
    off = np.concatenate([[0], np.cumsum(n_list)]).astype(np.int32)
    uv = (rng.uniform(-1, 1, (O, 2 * hid1))).astype(np.float32)
    pos = rng.uniform(0.05, 0.9, (O, 4)).astype(np.float32)
    wg = rng.uniform(-0.5, 0.5, (hid1, 4)).astype(np.float32)
    rows_alloc = (hid2 + 31) // 32 * 32
    w2 = np.zeros((rows_alloc, hid1), np.float32)
    w2[:hid2] = rng.normal(size=(hid2, hid1)).astype(np.float32) / np.sqrt(hid1)
    b2 = rng.normal(size=hid2).astype(np.float32)
    E = (rng.normal(size=(C, hid2)) / np.sqrt(hid2)).astype(np.float32)
    be = rng.normal(size=C).astype(np.float32)
    req_col = rng.randint(0, C, (K, Q)).astype(np.int32)
    req_col[rng.uniform(size=(K, Q)) < 0.2] = -1
    req_col[:, 3] = -1                                         # an image nobody asks about
    req_tile = np.arange(K * Q, dtype=np.int32).reshape(K, Q)
    orient = (rng.uniform(size=(K, Q)) < 0.5).astype(np.uint8)
    ref = _pair_ref64(n_list, off, uv, pos, wg, w2, b2, E, be, req_col, req_tile, orient, hid1, hid2, K, NS)
    args = (dev(uv), hid1, dev(pos), dev(wg))
    tail = (dev(E), dev(be), dev(np.array(n_list, np.int32)), dev(off), max(n_list), dev(req_col), dev(req_tile), dev(orient))
    wanted = np.zeros((K * Q, NS, NS), bool)
    for q, n in enumerate(n_list):
        for k in range(K):
            if req_col[k, q] >= 0:
                wanted[req_tile[k, q], :n, :n] = True
    outs = {}
    t = torch.full((K * Q, NS, NS), -30.0, device="cuda")
    outs["unpacked"] = _lib.pair_ll(*args, dev(w2), dev(b2), *tail, t, hid2=hid2).cpu().numpy()
    if hid1 % 16 == 0:
        packed = _lib.pair_pack_w2(dev(w2), hid2)
        t = torch.full((K * Q, NS, NS), -30.0, device="cuda")
        outs["packed"] = _lib.pair_ll_packed(*args, packed, dev(b2), hid2, *tail, t).cpu().numpy()
        if hid2 > 256:
            t = torch.full((K * Q, NS, NS), -30.0, device="cuda", dtype=torch.bfloat16)
            outs["packed_bf16"] = _lib.pair_ll_packed(*args, packed, dev(b2), hid2, *tail, t).float().cpu().numpy()
    if _lib.pair_split_supported(hid1, hid2):                  # bf16x3 pieces on the bf16 matrix pipes: fp32 results
        split = _lib.pair_pack_w2_split(dev(w2), hid2)
        t = torch.full((K * Q, NS, NS), -30.0, device="cuda")
        outs["split"] = _lib.pair_ll_split(*args, split, dev(b2), hid2, *tail, t).cpu().numpy()
        t = torch.full((K * Q, NS, NS), -30.0, device="cuda", dtype=torch.bfloat16)
        outs["split_bf16"] = _lib.pair_ll_split(*args, split, dev(b2), hid2, *tail, t).float().cpu().numpy()
        h2 = _lib.pair_pack_w2_h2(dev(w2), hid2)                # two fp16 pieces, three products, rows of W2 scaled: fp32 results (round 4 default)
        t = torch.full((K * Q, NS, NS), -30.0, device="cuda")
        outs["h2"] = _lib.pair_ll_h2(*args, h2, dev(b2), hid2, *tail, t).cpu().numpy()
        t = torch.full((K * Q, NS, NS), -30.0, device="cuda", dtype=torch.bfloat16)
        outs["h2_bf16"] = _lib.pair_ll_h2(*args, h2, dev(b2), hid2, *tail, t).float().cpu().numpy()
    for name, got in outs.items():
        tol = 2e-5 if "bf16" not in name else 0.0
        err = np.abs(got - ref)
        if "bf16" in name:
            assert np.all(err[wanted] <= 2.0 ** -8 * np.abs(ref[wanted]) + 1e-6), name
        else:
            assert err[wanted].max() <= tol * max(1.0, np.abs(ref[wanted]).max()), (name, err[wanted].max())
        assert np.all(got[~wanted] == -30.0), name                 # unrequested tiles, padding: untouched
    for name in ("split", "h2"):                               # as close to the exact result as the fp32 matrix pipe
        if name in outs:
            e_split, e_f32 = np.abs(outs[name] - ref)[wanted], np.abs(outs["packed"] - ref)[wanted]
            assert e_split.max() <= 3 * e_f32.max() + 2e-7 and e_split.mean() <= 1.25 * e_f32.mean() + 1e-8, (name, e_split.max(), e_f32.max(),
                                                                                                                e_split.mean(), e_f32.mean())


def test_pair_h2_rows_of_any_scale_and_large_activations(L):
    """The fp16x2 pair kernel's range handling: rows of W2 whose magnitudes span twelve decades (each row is scaled by its own power of
    two in the pack kernel), an all-zero row, and first-layer sums up to ~2000 (activations are split unscaled: fp16 holds them up to
    6e4) - results against float64 as close as the fp32-pipe kernel's."""
    from dfol_vqa_amd import _lib
    rng = np.random.RandomState(77)
    hid1, hid2, K, n_list = 256, 300, 2, [9, 6, 12]
    Q, O, NS, C = len(n_list), sum(n_list), 12, 20
    off = np.concatenate([[0], np.cumsum(n_list)]).astype(np.int32)
    uv = rng.uniform(-1, 1, (O, 2 * hid1)).astype(np.float32)
    uv[:, ::7] *= 300.0                                          # some first-layer sums in the hundreds and thousands
    uv[:, 5::31] *= 1e-4                                         # and some tiny ones
    pos = rng.uniform(0.05, 0.9, (O, 4)).astype(np.float32)
    wg = rng.uniform(-0.5, 0.5, (hid1, 4)).astype(np.float32)
    w2 = np.zeros((320, hid1), np.float32)
    w2[:hid2] = rng.normal(size=(hid2, hid1)).astype(np.float32) / np.sqrt(hid1)
    w2[:hid2] *= (10.0 ** rng.uniform(-9, 3, (hid2, 1))).astype(np.float32)      # row magnitudes from 1e-10 to 1e2
    w2[:hid2, ::7] /= 300.0                                      # (keeps the hidden sums of order one where the activations are large)
    w2[17] = 0.0
    w2[40, 3:] = 0.0                                             # a row with a single entry
    b2 = rng.normal(size=hid2).astype(np.float32)
    E = (rng.normal(size=(C, hid2)) / np.sqrt(hid2)).astype(np.float32)
    be = rng.normal(size=C).astype(np.float32)
    req_col = rng.randint(0, C, (K, Q)).astype(np.int32)
    req_tile = np.arange(K * Q, dtype=np.int32).reshape(K, Q)
    orient = (rng.uniform(size=(K, Q)) < 0.5).astype(np.uint8)
    ref = _pair_ref64(n_list, off, uv, pos, wg, w2, b2, E, be, req_col, req_tile, orient, hid1, hid2, K, NS)
    args = (dev(uv), hid1, dev(pos), dev(wg))
    tail = (dev(E), dev(be), dev(np.array(n_list, np.int32)), dev(off), max(n_list), dev(req_col), dev(req_tile), dev(orient))
    t = torch.full((K * Q, NS, NS), -30.0, device="cuda")
    got = _lib.pair_ll_h2(*args, _lib.pair_pack_w2_h2(dev(w2), hid2), dev(b2), hid2, *tail, t).cpu().numpy()
    t = torch.full((K * Q, NS, NS), -30.0, device="cuda")
    f32 = _lib.pair_ll_packed(*args, _lib.pair_pack_w2(dev(w2), hid2), dev(b2), hid2, *tail, t).cpu().numpy()
    wanted = ref != -30.0
    assert np.isfinite(got).all()
    e_h2, e_f32 = np.abs(got - ref)[wanted], np.abs(f32 - ref)[wanted]
    # (operands of 22 - 23 significand bits: where hundreds of large terms cancel - this test's activations of several hundred - the error
    # may reach a few times the fp32 pipe's; on activations of order one it is BELOW it, test_pair_ll_kernels_against_torch)
    assert e_h2.max() <= 4 * e_f32.max() + 2e-7 and e_h2.mean() <= 1.5 * e_f32.mean() + 1e-8, (e_h2.max(), e_f32.max(), e_h2.mean(), e_f32.mean())
    assert np.all(got[~wanted] == -30.0)


@pytest.mark.parametrize("hid1", [256, 64])
def test_pair_h2_saturation_flag(L, hid1):
    """dfol_pair_ll_h2_f32 with a status word set (dfol_set_range_status): the bound kernel in front of the pair kernel flags
    DFOL_RANGE_PAIR_SATURATED exactly when max_s U[s][k] + max_o V[o][k] + the geometry bound passes 6e4 (units of 1 / ln 2) for some image
    and hidden unit, or a U|V entry is NaN; large NEGATIVE sums (ELU -> -1), an image of one object and a call without a status word are
    not flagged."""
    from dfol_vqa_amd import _lib
    rng = np.random.RandomState(5)
    hid2, K, n_list = 300, 1, [9, 1, 12, 37]
    Q, O, NS, C = len(n_list), sum(n_list), 40, 8
    off = np.concatenate([[0], np.cumsum(n_list)]).astype(np.int32)
    pos = rng.uniform(0.05, 0.9, (O, 4)).astype(np.float32)
    wg = rng.uniform(-0.5, 0.5, (hid1, 4)).astype(np.float32)
    w2 = np.zeros((320, hid1), np.float32)
    w2[:hid2] = rng.normal(size=(hid2, hid1)).astype(np.float32) / np.sqrt(hid1)
    packed = _lib.pair_pack_w2_h2(dev(w2), hid2)
    b2 = dev(rng.normal(size=hid2).astype(np.float32))
    E, be = dev((rng.normal(size=(C, hid2)) / np.sqrt(hid2)).astype(np.float32)), dev(rng.normal(size=C).astype(np.float32))
    req_col, req_tile = dev(rng.randint(0, C, (K, Q)).astype(np.int32)), dev(np.arange(K * Q, dtype=np.int32).reshape(K, Q))
    word = torch.zeros(1, dtype=torch.int32, device="cuda")

    def flagged(uv, with_word=True):
        word.zero_()
        _lib.load().dfol_set_range_status(word.data_ptr() if with_word else None)
        try:
            t = torch.full((K * Q, NS, NS), -30.0, device="cuda")
            _lib.pair_ll_h2(dev(uv), hid1, dev(pos), dev(wg), packed, b2, hid2, E, be, dev(np.array(n_list, np.int32)), dev(off), max(n_list), req_col, req_tile,
                            None, t, uv_prescaled=True)          # (the values below are in the kernel's units of 1 / ln 2)
        finally:
            _lib.load().dfol_set_range_status(None)
        return int(word.item())

    base = rng.uniform(-1, 1, (O, 2 * hid1)).astype(np.float32)
    assert flagged(base) == 0
    neg = base.copy(); neg[:, 3] = -5.0e5                          # ELU(-huge) = -1: nothing saturates
    assert flagged(neg) == 0
    lone = base.copy(); lone[int(off[1])] = 9.0e4                  # the image of ONE object has no pairs
    assert flagged(lone) == 0
    for k in (0, hid1 - 1, hid1 // 2 + 1):
        hot = base.copy()
        hot[int(off[3]) + 5, k] = 4.0e4                            # U[s][k] + V[o][k] = 7e4 > 6e4 for s = 5, o = 20 of the last image
        hot[int(off[3]) + 20, hid1 + k] = 3.0e4
        assert flagged(hot) == _lib.RANGE_PAIR_SATURATED, k
        assert flagged(hot, with_word=False) == 0
        under = hot.copy(); under[int(off[3]) + 20, hid1 + k] = 1.0e4        # 5e4 + geometry (< 4) stays below
        assert flagged(under) == 0, k
        apart = base.copy()                                        # the two large values sit in DIFFERENT images: no pair sees both
        apart[int(off[3]) + 5, k] = 4.0e4
        apart[int(off[2]) + 2, hid1 + k] = 3.0e4
        assert flagged(apart) == 0, k
    nan = base.copy(); nan[int(off[2]) + 7, hid1 + 9] = np.nan
    assert flagged(nan) == _lib.RANGE_PAIR_SATURATED


@pytest.mark.parametrize("n_list", [[8, 3, 5, 1], [40, 33, 17, 8], [100, 104, 64, 2], [130, 256]])
def test_relate_one_bf16_equals_fp32_kernel_on_rounded_tiles(L, n_list):
    """The bf16-tile kernel does the same fp32 arithmetic as the fp32-tile kernel: on tiles that are exactly representable in bf16 the
    two agree to rounding, for EXISTS / FOR_ALL / negated / inactive predicates, ragged images and planted clamped factors."""
    from dfol_vqa_amd import _lib
    rng = np.random.RandomState(sum(n_list) + 3)
    pq, NS, prior_s, prior_o, tile, _ = _logic_inputs(rng, n_list, [1] * len(n_list))
    NS8 = (NS + 7) // 8 * 8
    P = len(pq)
    t32 = np.full((P, NS8, NS8), -30.0, np.float32)
    t32[:, :NS, :NS] = tile
    x = np.zeros((P, NS8), np.float32); x[:, :NS] = prior_s
    pv = np.zeros((P, NS8), np.float32); pv[:, :NS] = prior_o
    if n_list[0] >= 3:
        pv[0, 1] = 0.0
        t32[0, 1, 2] = 0.0                                      # exactly-certain prior meets exactly-certain likelihood: the clamped case
    t16 = dev(t32).to(torch.bfloat16)
    t32r = t16.float()
    n_obj = np.array(n_list, np.int32)
    for quant in (np.ones(P, np.float32), (rng.uniform(size=P) < 0.5).astype(np.float32)):
        for any_neg in (False, True):
            neg = (rng.uniform(size=P) < 0.5).astype(np.uint8) if any_neg else None
            active = np.ones(P, np.uint8)
            if P > 2:
                active[2] = 0
            a = _lib.relate_one_fwd_bf16(dev(x), dev(pv), t16, dev(pq), dev(n_obj), dev(quant), None if neg is None else dev(neg), dev(active))
            b = _lib.relate_one_fwd(dev(x), dev(pv), t32r, dev(pq), dev(n_obj), dev(quant), None if neg is None else dev(neg), dev(active))
            a, b = a.cpu().numpy(), b.cpu().numpy()
            for p in range(P):
                n = n_list[p]
                assert np.allclose(a[p, :n], b[p, :n], rtol=0, atol=3e-5 * max(1.0, np.abs(b[p, :n]).max())), (p, n, any_neg)
                assert np.all(a[p, n:] == 0)


def test_calibration_lstm_cell_equals_torch(L):
    """CalibrationLSTMCell (dfol_lstm_cell_f32: one launch; or gate GEMMs + dfol_lstm_pointwise_f32) == nn.LSTMCell with the same parameters."""
    from dfol_vqa_amd.visual_oracle import CalibrationLSTMCell
    torch.manual_seed(0)
    ref = torch.nn.LSTMCell(318, 50).cuda()
    mine = CalibrationLSTMCell(318, 50).cuda()
    mine.load_state_dict(ref.state_dict())                   # same parameter names
    for rows in (1, 7, 256, 333):
        x = torch.randn(rows, 318, device="cuda")
        h, c = torch.randn(rows, 50, device="cuda") * 0.5, torch.randn(rows, 50, device="cuda")
        with torch.no_grad():
            h1, c1 = ref(x, (h, c))
            h2, c2 = mine(x, (h, c))
        assert torch.allclose(h1, h2, atol=2e-6, rtol=1e-5) and torch.allclose(c1, c2, atol=2e-6, rtol=1e-5)
        # the two-GEMM + pointwise route (what wider cells fall back to) and a strided input view
        with torch.no_grad():
            ig, hg = L.linear_act(x, mine.weight_ih, mine.bias_ih, L.ACT_NONE), L.linear_act(h, mine.weight_hh, mine.bias_hh, L.ACT_NONE)
            h3, c3 = L.lstm_pointwise(ig, hg, c)
            wide = torch.randn(rows, 400, device="cuda")
            wide[:, 40:358] = x
            h4, c4 = mine(wide[:, 40:358], (h, c))
        assert torch.allclose(h1, h3, atol=2e-6, rtol=1e-5) and torch.allclose(c1, c3, atol=2e-6, rtol=1e-5)
        assert torch.equal(h2, h4) and torch.equal(c2, c4)
    # with gradients: forward in one launch that keeps the activated gates, backward on dfol_lstm_cell_bwd_f32 + the dense / TN kernels;
    # every gradient equals torch's own cell's (fp64 as the yardstick), also when only one of (h', c') carries a gradient
    for rows, which in ((4, "both"), (256, "both"), (37, "h"), (37, "c")):
        xs = torch.randn(rows, 318, device="cuda")
        h0, c0 = torch.randn(rows, 50, device="cuda") * 0.5, torch.randn(rows, 50, device="cuda")
        gh, gc = torch.randn(rows, 50, device="cuda"), torch.randn(rows, 50, device="cuda")
        grads = {}
        for tag, cell, dt in (("mine", mine, torch.float32), ("ref", ref, torch.float32), ("ref64", torch.nn.LSTMCell(318, 50).cuda().double(), torch.float64)):
            if tag == "ref64":
                cell.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
            cell.zero_grad()
            x, h, c = (t.detach().to(dt).requires_grad_(True) for t in (xs, h0, c0))
            hy, cy = cell(x, (h, c))
            loss = ((hy * gh.to(dt)).sum() if which != "c" else 0) + ((cy * gc.to(dt)).sum() if which != "h" else 0)
            loss.backward()
            grads[tag] = [x.grad, h.grad, c.grad, cell.weight_ih.grad, cell.weight_hh.grad, cell.bias_ih.grad, cell.bias_hh.grad]
        for gm, gr, g64 in zip(grads["mine"], grads["ref"], grads["ref64"]):
            scale = g64.abs().max().item() + 1e-12
            e_mine, e_ref = (gm.double() - g64).abs().max().item() / scale, (gr.double() - g64).abs().max().item() / scale
            assert e_mine <= 4 * e_ref + 2e-6, (rows, which, e_mine, e_ref)
    # bit-repeatable
    mine.zero_grad()
    x = xs.clone().requires_grad_(True)
    hy, cy = mine(x, (h0, c0))
    (hy.sum() + cy.sum()).backward()
    g1 = [x.grad.clone(), mine.weight_ih.grad.clone(), mine.bias_hh.grad.clone()]
    mine.zero_grad()
    x = xs.clone().requires_grad_(True)
    hy, cy = mine(x, (h0, c0))
    (hy.sum() + cy.sum()).backward()
    assert all(torch.equal(a, b) for a, b in zip(g1, [x.grad, mine.weight_ih.grad, mine.bias_hh.grad]))


def test_lstm_cell_on_tokens_and_attention_output_kernels(L):
    """dfol_lstm_cell_tokens_f32 (the rows [head | table[idx]] built while they are staged) == dfol_calib_features_f32 + dfol_lstm_cell_f32 bit for
    bit, no-op tokens (idx < 0 -> zero rows) included; dfol_attention_modulations_f32 (sixteen lanes per row) == Sigmoid(Linear([fs | bs])) in float64
    to fp32 rounding for the model's widths and others (N > 8 outputs, S not a multiple of 16), missing states as zeros."""
    g = torch.Generator(device="cuda").manual_seed(5)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    w_ih, w_hh, b_ih, b_hh = rnd(200, 318) * 0.05, rnd(200, 50) * 0.1, rnd(200) * 0.1, rnd(200) * 0.1
    wt_ih, wt_hh = w_ih.t().contiguous(), w_hh.t().contiguous()
    head, table = torch.zeros(18, device="cuda"), rnd(11, 300) * 0.2
    head[3], head[17] = 1.0, 1.0
    for rows in (1, 5, 256, 301):
        idx = torch.randint(-1, 11, (rows,), device="cuda", generator=g, dtype=torch.int32)
        h, c = rnd(rows, 50) * 0.5, rnd(rows, 50)
        x = torch.empty(rows, 318, device="cuda")
        L.call("dfol_calib_features_f32", head.data_ptr(), 18, table.data_ptr(), 300, idx.data_ptr(), rows, x.data_ptr(), L._stream())
        assert torch.equal(x, torch.where((idx < 0)[:, None], torch.zeros(1, device="cuda"), torch.cat([head.expand(rows, 18), table[idx.clamp(min=0).long()]], 1)))
        h1, c1 = L.lstm_cell(x, h, c, wt_ih, wt_hh, b_ih, b_hh)
        h2, c2 = L.lstm_cell_tokens(head, table, idx, h, c, wt_ih, wt_hh, b_ih, b_hh)
        assert torch.equal(h1, h2) and torch.equal(c1, c2), rows
        assert (x[idx < 0] == 0).all()
        # against torch's cell in float64
        gates = x.double() @ w_ih.double().t() + b_ih.double() + h.double() @ w_hh.double().t() + b_hh.double()
        i, f, gg, o = gates.chunk(4, 1)
        c64 = torch.sigmoid(f) * c.double() + torch.sigmoid(i) * torch.tanh(gg)
        h64 = torch.sigmoid(o) * torch.tanh(c64)
        assert (h1.double() - h64).abs().max().item() <= 2e-6 and (c1.double() - c64).abs().max().item() <= 4e-6, rows
    for P, S, N in ((256, 50, 4), (1, 50, 4), (77, 23, 11), (300, 64, 1)):
        fs, bs, W, b = rnd(P, S), rnd(P, S), rnd(N, 2 * S) * 0.1, rnd(N)
        for f_, b_ in ((fs, bs), (None, bs), (fs, None)):
            got = L.attention_modulations(f_, b_, W, b)
            z = b.double() + (0 if f_ is None else f_.double() @ W[:, :S].double().t()) + (0 if b_ is None else b_.double() @ W[:, S:].double().t())
            assert (got.double() - torch.sigmoid(z)).abs().max().item() <= 3e-7, (P, S, N)


def test_modulate_backward_kernel_against_autograd(L):
    """dfol_modulate_bwd_f32 (one launch per apply_modulations in the backward of the calibrator phases) against torch autograd through the
    tensor-op restatement of batch_base_types.py:170-179 in float64: attention and modulation gradients, ragged predicates, clamped
    corners (d = 0, d = 1, c = 0, attention at log 1)."""
    from dfol_vqa_amd import ops
    rng = np.random.RandomState(12)
    n_list, k_list = [5, 30, 1, 17], [2, 1, 3, 1]
    pq_h = np.repeat(np.arange(len(n_list)), k_list).astype(np.int32)
    P, NS = len(pq_h), 32
    att = (-np.abs(rng.normal(size=(P, NS))) * 2).astype(np.float32)
    att[0, 0] = 0.0                                            # log 1: log_not at its clamp
    mods = rng.uniform(0.02, 0.98, size=(P, 4)).astype(np.float32)
    mods[1, 3], mods[2, 3], mods[3, 2] = 0.0, 1.0, 0.0         # the clamps of slog(d), slog(1 - d), slog(c)
    g = rng.normal(size=(P, NS)).astype(np.float32)
    pq, n_obj = dev(pq_h), dev(np.asarray(n_list, np.int32))
    ga, gm = L.modulate_bwd(dev(g), dev(att), dev(mods), pq, n_obj)
    a64 = torch.tensor(att, dtype=torch.float64, device="cuda", requires_grad=True)
    m64 = torch.tensor(mods, dtype=torch.float64, device="cuda", requires_grad=True)
    valid = torch.arange(NS, device="cuda").unsqueeze(0) < n_obj.long()[pq.long()].unsqueeze(1)
    out = ops._modulate_reference(a64, m64, valid)
    ra, rm = torch.autograd.grad(out, (a64, m64), torch.tensor(g, dtype=torch.float64, device="cuda"))
    assert torch.allclose(ga.double(), ra, atol=2e-5, rtol=2e-5), (ga.double() - ra).abs().max()
    assert torch.allclose(gm.double(), rm, atol=2e-4, rtol=2e-4), (gm.double() - rm).abs().max()
    assert bool((ga[~valid] == 0).all())
    # and through the autograd Function, bit-repeatable
    a32, m32 = dev(att).requires_grad_(True), dev(mods).requires_grad_(True)
    o1 = ops.modulate(a32, m32, pq, n_obj)
    g1 = torch.autograd.grad(o1, (a32, m32), dev(g))
    o2 = ops.modulate(a32, m32, pq, n_obj)
    g2 = torch.autograd.grad(o2, (a32, m32), dev(g))
    assert torch.equal(g1[0], g2[0]) and torch.equal(g1[1], g2[1]) and torch.equal(g1[0], ga)


def test_find_max_ind_on_device(L):
    """util.find_max_ind (util.py:64-66) as a kernel: ties, thresholds, one-option questions, against the oracle's restatement."""
    from dfol_vqa_amd import _lib
    rng = np.random.RandomState(5)
    counts = [2, 1, 26, 3, 2, 70]
    pq = np.repeat(np.arange(len(counts)), counts)
    lp = np.log(rng.uniform(0.01, 0.9, len(pq))).astype(np.float32)
    lp[3 + 5] = lp[3 + 9] = lp[3:29].max() + 0.01                 # a tie for the maximum of question 2
    lp[0] = lp[1]                                                 # and one in question 0
    seg = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    for thr in (0.0, 0.5, 0.95):
        got = _lib.find_max_ind(dev(lp), dev(seg), thr).cpu().numpy()
        ref = orc.find_max_ind(lp, pq, len(counts), thr)
        assert got.tolist() == ref.tolist(), thr
    assert got.sum() <= len(counts) + 2


def test_torch_custom_ops_opcheck(L):
    """The six core entry points are PyTorch operators (torch.ops.dfol.*: torch_ops.py, custom_op + register_fake + register_autograd):
    torch.library.opcheck validates schema, fake-tensor function, autograd registration and AOT-autograd tracing of each
    (the dispatch site they serve is batch_gqa_interpreter.py:72-78)."""
    from dfol_vqa_amd import torch_ops  # noqa: F401
    rng = np.random.RandomState(9)
    n_list, k_list = [5, 12, 3], [2, 1, 3]
    Q, NS = len(n_list), 12
    pq_h = np.repeat(np.arange(Q), k_list).astype(np.int32)
    P = len(pq_h)
    pq, n_obj = dev(pq_h), dev(np.asarray(n_list, np.int32))
    f = lambda *shape: dev((-np.abs(rng.normal(size=shape))).astype(np.float32))
    neg, want = dev((rng.uniform(size=P) < 0.5).astype(np.uint8)), dev(np.full(P, 3, np.uint8))
    ones = torch.ones(P, device="cuda")
    utils = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_dynamic")
    g = lambda t: t.requires_grad_(True)
    torch.library.opcheck(torch.ops.dfol.filter_fwd.default, (g(f(Q, NS)), g(f(P, NS)), pq, n_obj, neg, None), test_utils=utils)
    torch.library.opcheck(torch.ops.dfol.relate_fwd.default, (g(f(Q, NS)), g(f(Q, NS)), g(f(P, NS, NS)), pq, n_obj, ones, ones.clone(), neg, None, want, 0, False, False),
                          test_utils=utils)
    torch.library.opcheck(torch.ops.dfol.relate_one_fwd.default, (f(P, NS), f(Q, NS), f(P, NS, NS), pq, n_obj, ones, None, None, False), test_utils=utils)
    torch.library.opcheck(torch.ops.dfol.quantify_fwd.default, (g(f(P, NS)), ones, pq, n_obj), test_utils=utils)
    x, w, b = g(dev(rng.normal(size=(40, 24)).astype(np.float32))), g(dev(rng.normal(size=(16, 24)).astype(np.float32) * 0.2)), g(dev(rng.normal(size=16).astype(np.float32)))
    for act in (L.ACT_NONE, L.ACT_SIGMOID, L.ACT_ELU, L.ACT_LOGSIGMOID):
        torch.library.opcheck(torch.ops.dfol.linear_act.default, (x, w, b, int(act)), test_utils=utils)
    # pair_ll mutates its `tiles` argument
    hid1, hid2, D, C = 32, 12, 20, 30
    O = sum(n_list)
    obj_off = dev(np.concatenate([[0], np.cumsum(n_list)]).astype(np.int32))
    uv = dev(rng.normal(size=(O, 2 * hid1)).astype(np.float32))
    pos = dev(rng.uniform(0.05, 0.9, size=(O, 4)).astype(np.float32))
    wg, w2, b2 = dev(rng.normal(size=(hid1, 4)).astype(np.float32) * 0.1), dev(rng.normal(size=(hid2, hid1)).astype(np.float32) * 0.2), dev(rng.normal(size=hid2).astype(np.float32))
    emb_w, emb_b = dev(rng.normal(size=(C, hid2)).astype(np.float32) * 0.3), dev(rng.normal(size=C).astype(np.float32))
    req_col, req_tile = dev(rng.randint(0, C, size=(1, Q)).astype(np.int32)), dev(np.arange(Q, dtype=np.int32).reshape(1, Q))
    tiles = torch.full((Q, NS, NS), -30.0, device="cuda")
    torch.library.opcheck(torch.ops.dfol.pair_ll.default, (uv, hid1, pos, wg, w2, b2, hid2, emb_w, emb_b, n_obj, obj_off, max(n_list), req_col, req_tile, None, tiles, -30.0, 0),
                          test_utils=("test_schema", "test_faketensor"))
    # and the operators' gradients are the HIP backward kernels' (same numbers as the direct call)
    att, ll = g(f(Q, NS)), g(f(P, NS))
    out = torch.ops.dfol.filter_fwd(att, ll, pq, n_obj, neg, None)
    go = torch.ones_like(out)
    ga, gl = torch.autograd.grad(out, (att, ll), go)
    ra, rl = L.filter_bwd(go, ll.detach(), pq, n_obj, neg, None, Q)
    assert torch.equal(ga, ra) and torch.equal(gl, rl)


def test_linear_act_split_block_height_changes_no_bit(L):
    """The bf16x3 dense kernel picks 64-row blocks for products whose 128-row tiling would leave most of the chip idle (the featurizer of
    36-object or shared scenes).  An output element sees the same products in the same order either way, so rows computed inside a small
    batch (64-row blocks) equal the same rows computed inside a large one (128-row blocks) BIT FOR BIT - the property the sharded ==
    single-process equality rests on - for every activation, ragged M / N and 8-byte aligned rows."""
    from dfol_vqa_amd import _lib
    rng = np.random.RandomState(21)
    for (N, K, ld) in ((512, 2048, 2054), (300, 256, 256), (256, 516, 516)):
        big = torch.tensor(rng.normal(size=(70000, ld)).astype(np.float32), device="cuda")
        W = torch.tensor((rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32), device="cuda")
        b = torch.tensor(rng.normal(size=N).astype(np.float32), device="cuda")
        for act in (L.ACT_NONE, L.ACT_SIGMOID, L.ACT_ELU):
            y_big = _lib.linear_act_split(big[:, :K], W, b, act)                  # 547 row blocks of 128
            for m in (37, 1835, 9216):                                           # 64-row blocks
                y_small = _lib.linear_act_split(big[:m, :K], W, b, act)
                assert torch.equal(y_small, y_big[:m]), (N, K, act, m, (y_small - y_big[:m]).abs().max().item())


def test_linear_wide_equals_the_tiled_kernel_bit_for_bit(L):
    """csrc/dfol_dense_wide.hip (one persistent workgroup per CU over 128 rows x all of 256 < N <= 512 columns: the featurizer and the pair
    MLP's stacked first layer) against the tiled kernel on the same rows in chunks small enough that dfol_linear_act_h2_f32 does not forward
    them: equal bit for bit for every activation, both X load widths (16- and 8-byte aligned rows), a k tail (516 = 16 steps + 4), three and
    four column blocks, ragged M, and a batch of fewer blocks than CUs.  And the fp16-range flag rides in this kernel too."""
    from dfol_vqa_amd import _lib
    rng = np.random.RandomState(33)
    lib = _lib.load()
    for (N, K, ld, M) in ((512, 2048, 2054, 25600 + 37), (512, 516, 516, 40000), (384, 132, 136, 33000), (300, 256, 256, 5000), (512, 2048, 2048, 129)):
        X = torch.tensor(rng.normal(size=(M, ld)).astype(np.float32), device="cuda")
        W = torch.tensor((rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32), device="cuda")
        b = torch.tensor(rng.normal(size=N).astype(np.float32), device="cuda")
        assert not lib.dfol_linear_wide_supported(3000, N, K) and (M < 24000 or N <= 384 or M == 40000 or lib.dfol_linear_wide_supported(M, N, K))
        for act in (L.ACT_NONE, L.ACT_SIGMOID, L.ACT_ELU, L.ACT_LOGSIGMOID):
            y = _lib.linear_wide(X[:, :K], W, b, act)
            ref = torch.cat([_lib.linear_act_split(X[m:m + 3000, :K], W, b, act) for m in range(0, M, 3000)])
            assert torch.equal(y, ref), (N, K, ld, act, (y - ref).abs().max().item())
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    lib.dfol_set_range_status(word.data_ptr())
    try:
        _lib.linear_wide(X[:, :K], W, b, L.ACT_NONE)
        assert int(word.item()) == 0
        X[77, 5] = 1.0e5
        _lib.linear_wide(X[:, :K], W, b, L.ACT_NONE)
        assert int(word.item()) & _lib.RANGE_X_OVERFLOW
    finally:
        lib.dfol_set_range_status(None)


@pytest.mark.parametrize("act", [1, 2, 3])
@pytest.mark.parametrize("shape", [(1000, 300), (7, 3), (25600, 512)])
def test_act_bwd_equals_autograd_formulas(act, shape):
    """dfol_act_bwd_f32 == the tensor-op formulas it replaces in the backward of linear_act (and == autograd through the activation)."""
    from dfol_vqa_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(act * 100 + shape[0])
    x = (torch.randn(*shape, device="cuda", generator=g) * 2).requires_grad_(True)
    y = [None, torch.sigmoid, torch.nn.functional.elu, torch.nn.functional.logsigmoid][act](x)
    gy = torch.randn(*shape, device="cuda", generator=g)
    dz = _lib.act_bwd(gy, y.detach().contiguous(), act)
    ref = [None, lambda: gy * y * (1 - y), lambda: gy * torch.where(y > 0, torch.ones_like(y), y + 1), lambda: gy * (1 - torch.exp(y))][act]().detach()
    assert (dz - ref).abs().max().item() <= 1e-6 * max(1.0, ref.abs().max().item())
    (auto,) = torch.autograd.grad(y, x, gy)
    assert (dz - auto).abs().max().item() <= 2e-6 * max(1.0, auto.abs().max().item())


@pytest.mark.parametrize("n_list", [[9, 1, 12, 37], [100, 64], [2]])
def test_pair_train_forward_fused_kernel(L, n_list):
    """dfol_pair_train_fwd_h2_f32 (round 6: the forward of a train step's pair MLP in one launch - Z, pre2, the pair geometry and the first
    reader's logits leave the fused pair kernel's registers) against the kernels it replaces (dfol_pair_hidden1_fwd_f32, the split-operand
    product over Z, dfol_pair_logit_fwd_f32) and against float64: Z and the geometry to fp32 rounding, pre2 and the logits as close to
    float64 as the replaced route."""
    from dfol_vqa_amd import _lib
    rng = np.random.RandomState(sum(n_list) + 11)
    hid1, hid2 = 256, 300
    Q, O = len(n_list), sum(n_list)
    n = np.asarray(n_list, np.int64)
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int32)
    pair_off = np.concatenate([[0], np.cumsum(n * (n - 1))]).astype(np.int64)
    pairs = int(pair_off[-1])
    uv = rng.normal(0, 1.0, (O, 2 * hid1)).astype(np.float32)
    pos = rng.uniform(0.05, 0.9, (O, 4)).astype(np.float32)
    wg = rng.uniform(-0.5, 0.5, (hid1, 4)).astype(np.float32)
    w2 = (rng.normal(size=(hid2, hid1)) / np.sqrt(hid1)).astype(np.float32)
    b2 = rng.normal(size=hid2).astype(np.float32)
    P = sum(1 for x in n_list if x > 1)
    e_rows = (rng.normal(size=(max(P, 1), hid2)) / np.sqrt(hid2)).astype(np.float32)
    req = np.full((1, Q), -1, np.int32)
    req[0, [i for i, x in enumerate(n_list) if x > 1]] = np.arange(P, dtype=np.int32)
    wp = np.zeros((320, hid1), np.float32); wp[:hid2] = w2
    img = _lib.pair_pack_w2_h2(dev(wp), hid2)
    geom = (dev(np.asarray(n_list, np.int32)), dev(off), dev(pair_off)[:Q].contiguous())
    z, pre2, geo, x = _lib.pair_train_fwd_h2(dev(uv) * _lib.LOG2E, hid1, dev(pos), dev(wg), img, dev(b2), hid2, *geom, max(n_list), pairs, dev(e_rows), dev(req))
    z0, geo0 = _lib.pair_hidden1_fwd(dev(uv[:, :hid1].copy()), dev(uv[:, hid1:].copy()), dev(pos), dev(wg), geom[1], geom[2], geom[0], max(n_list), pairs)
    assert torch.equal(geo, geo0)
    assert (z - z0).abs().max().item() <= 2e-6 * max(1.0, z0.abs().max().item())
    # float64 reference of pre2 and the logits from the replaced route's Z
    z64 = z0.double().cpu().numpy()
    p64 = z64 @ w2.astype(np.float64).T + b2
    with _lib.dense_math("f16x2"):
        p0 = _lib.linear_act_split(z0, dev(w2), dev(b2), _lib.ACT_NONE).cpu().numpy()
    e_new, e_old = np.abs(pre2.cpu().numpy() - p64), np.abs(p0 - p64)
    assert e_new.max() <= 4 * e_old.max() + 1e-6, (e_new.max(), e_old.max())
    rowp = np.concatenate([np.full(int(n[i] * (n[i] - 1)), req[0, i]) for i in range(Q)]) if pairs else np.zeros(0, np.int64)
    x64 = (1.0 / (1.0 + np.exp(-p64)) * e_rows.astype(np.float64)[np.maximum(rowp, 0)]).sum(1)
    assert np.abs(x[0].cpu().numpy() - x64).max() <= 2e-5
