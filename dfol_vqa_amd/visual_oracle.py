"""Visual oracle: concept log-likelihoods for the interpreter, in block layout.

Reference: src/nsvqa/nn/vision/base_oracle.py, classifier_oracle.py and the MLP definitions in
src/gqa_interpreter_experiments.py:18-77.  The neural stages run on the exact-fp32 MFMA GEMM with the
activation fused (csrc/dfol_dense.hip); the per-predicate likelihood blocks are gathered from the
cached tables by csrc/dfol_logic.hip.  Parameter names match the reference's state_dict
(`_network.1.weight`, ...) so its checkpoints load with strict=False.
"""

import contextlib
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import ops as L
from .fol_types import TokenType
from .host_util import flatten_list, get_lowered, lower_tokens, segments_of, upload


# ---------------------------------------------------------------------------------------------------
# MLP holders (gqa_interpreter_experiments.py:18-77)
# ---------------------------------------------------------------------------------------------------
def _run_layers(seq, x, out=None):
    """Walk an nn.Sequential of (Dropout, Linear, activation) triples; each triple is one fused GEMM launch.  `out` (inference only): a
    [rows, width] view - any row stride - that the LAST layer writes its result into instead of a new tensor."""
    mods = list(seq)
    last_linear = max([j for j, m in enumerate(mods) if isinstance(m, nn.Linear)] or [-1])
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Dropout):
            if m.training and m.p > 0:
                raise L.DfolError("dropout > 0 in training mode is not built on the HIP path; use dropout: 0")
            i += 1
            continue
        assert isinstance(m, nn.Linear), "unexpected layer %r" % (m,)
        act, step = L.ACT_NONE, 1
        if i + 1 < len(mods):
            nxt = mods[i + 1]
            if isinstance(nxt, nn.ELU):
                act, step = L.ACT_ELU, 2
            elif isinstance(nxt, nn.Sigmoid):
                act, step = L.ACT_SIGMOID, 2
            elif isinstance(nxt, nn.LogSigmoid):
                act, step = L.ACT_LOGSIGMOID, 2
        x = L.linear_act(x, m.weight, m.bias, act, out if i == last_linear else None)
        i += step
    return x


class RegularMLP(nn.Module):
    """[Dropout, Linear, ELU]* + [Dropout, Linear, Sigmoid]  (gqa_interpreter_experiments.py:18-36)."""

    def __init__(self, input_dim, output_dim, layers_config, dropout):
        super(RegularMLP, self).__init__()
        if layers_config is None:
            self._network = None
        else:
            layers, last = [], input_dim
            for width in layers_config:
                layers += [nn.Dropout(dropout), nn.Linear(last, width), nn.ELU()]
                last = width
            layers += [nn.Dropout(dropout), nn.Linear(last, output_dim), nn.Sigmoid()]
            self._network = nn.Sequential(*layers)

    def forward(self, input_tensor, out=None):
        return input_tensor if self._network is None else _run_layers(self._network, input_tensor, out)

    def output_width(self):
        """Columns of forward()'s result, or None for the identity network."""
        if self._network is None:
            return None
        return [m for m in self._network if isinstance(m, nn.Linear)][-1].out_features


class EmbeddingLayer(nn.Module):
    """Dropout + Linear(hidden -> concepts) + LogSigmoid  (gqa_interpreter_experiments.py:60-77)."""

    def __init__(self, input_dim, output_dim, dropout, weights=None, biases=None, freeze_bias=False, cluster_index=None):
        super(EmbeddingLayer, self).__init__()
        if cluster_index is not None:
            raise NotImplementedError("ClusteredLogSoftmax is constructed nowhere in the reference (SURVEY.md §2 row 1)")
        linear = nn.Linear(input_dim, output_dim, bias=not freeze_bias)
        if weights is not None:
            linear.weight = nn.Parameter(weights)
        if biases is not None and not freeze_bias:
            linear.bias = nn.Parameter(biases)
        self._network = nn.Sequential(nn.Dropout(dropout), linear, nn.LogSigmoid())

    def forward(self, input_tensor):
        return _run_layers(self._network, input_tensor)

    @property
    def linear(self):
        return self._network[1]


# ---------------------------------------------------------------------------------------------------
# oracle
# ---------------------------------------------------------------------------------------------------
class OracleBase(nn.Module):
    """base_oracle.py:11-55.  forward() keeps the reference signature; blocks come back with the
    reference's trailing feature dimension: [P, NS, 1] / [P, NS, NS, 1]."""

    def __init__(self, ontology, feature_dim=1):
        super(OracleBase, self).__init__()
        self._feature_dim = feature_dim
        self._ontology = ontology

    def forward(self, token_type, token_list, token_image_map, world, default_log_likelihood=-30, normalized_probability=True):
        if not isinstance(token_list, list):
            token_list = [token_list]
        low = get_lowered(token_list, self._ontology, token_type)
        if isinstance(token_image_map, torch.Tensor):
            host_map = token_image_map.cpu().numpy()
        else:
            host_map = np.asarray(token_image_map)
        res = self.block_likelihood(token_type, low, world.pred_q(token_image_map), host_map, world,
                                    default_log_likelihood, normalized_probability)
        return res.unsqueeze(-1)

    def block_likelihood(self, token_type, low, pred_q, pred_q_host, world, default_log_likelihood=-30,
                         normalized_probability=True, orientation=L.TILE_SUBJECT_ROWS):
        raise NotImplementedError

    def get_embedding(self, tokens, meta_data, device):       # base_oracle.py:45-55
        try:
            ind = [meta_data['index'][t] for t in tokens]
            emb = meta_data['embedding']
            if emb.is_cuda:                                  # (an index list would be uploaded, synchronously, on every call)
                return emb.index_select(0, upload(np.asarray(ind, np.int64), emb.device))
            return emb[ind, :]
        except (KeyError, TypeError):
            return torch.from_numpy(self._ontology.get_embeddings(tokens)).float().to(device)


class _LSTMCellFn(torch.autograd.Function):
    """nn.LSTMCell with the forward in ONE launch (dfol_lstm_cell_train_f32, which keeps the activated gates) and the backward on this
    library's kernels: dfol_lstm_cell_bwd_f32 (the pointwise stage), then the four products of d_gates with the weights / inputs on the
    dense and the TN weight-gradient kernels (deterministic).  Trains the attention calibrator (cur6-7)."""

    @staticmethod
    def forward(ctx, x, h, c, w_ih, w_hh, b_ih, b_hh, w_ih_t, w_hh_t):
        hy, cy, gates = L.lstm_cell_train(x, h, c, w_ih_t, w_hh_t, None if b_ih is None else b_ih.detach(), None if b_hh is None else b_hh.detach())
        ctx.save_for_backward(x, h, c, cy, gates, w_ih_t, w_hh_t)
        ctx.has_bias = b_ih is not None
        return hy, cy

    @staticmethod
    def backward(ctx, d_hy, d_cy):
        x, h, c, cy, gates, w_ih_t, w_hh_t = ctx.saved_tensors
        dg, dc = L.lstm_cell_bwd(gates, c, cy, None if d_hy is None else d_hy.contiguous(), None if d_cy is None else d_cy.contiguous())
        need = ctx.needs_input_grad
        dx = L.linear_act(dg, w_ih_t, None, L.ACT_NONE) if need[0] else None            # d_gates @ W_ih   (w_ih_t = W_ih^T, so w_ih_t^T = W_ih)
        dh = L.linear_act(dg, w_hh_t, None, L.ACT_NONE) if need[1] else None
        dwi = dwh = db = None
        if need[3]:
            dwi = _lib.linear_wgrad(dg, x if x.stride(-1) == 1 else x.contiguous(), bias=ctx.has_bias)
            if ctx.has_bias:
                dwi, db = dwi
        if need[4]:
            dwh = _lib.linear_wgrad(dg, h if h.stride(-1) == 1 else h.contiguous())
        if ctx.has_bias and db is None and (need[5] or need[6]):
            db = dg.sum(0)
        return dx, dh, (dc if need[2] else None), dwi, dwh, (db if need[5] else None), (db if need[6] else None), None, None


class CalibrationLSTMCell(nn.LSTMCell):
    """nn.LSTMCell (same parameters and state_dict names) whose forward is ONE launch (dfol_lstm_cell_f32: both gate products, the
    biases and the pointwise stage; the two small GEMMs [Q, 318] x [318, 200], [Q, 50] x [50, 200] take 54 us each in the vendor BLAS
    and 15 us each as two launches of the dense kernel, and a calibrated forward runs eight cells).  With gradients enabled the same
    launch keeps the activated gates and the backward runs on this library's kernels (_LSTMCellFn; DFOL_LSTM_BWD=torch: torch's own
    cell, for A/B runs)."""

    def _transposed(self):
        key = (self.weight_ih._version, self.weight_hh._version, self.weight_ih.data_ptr(), self.weight_hh.data_ptr())
        if getattr(self, "_wt", (None,))[0] != key:              # transposed copies, once per weight version
            self._wt = (key, self.weight_ih.detach().t().contiguous(), self.weight_hh.detach().t().contiguous())
        return L.keep_alive(self._wt)

    def forward(self, x, state=None):
        fits = state is not None and x.is_cuda and x.dtype == torch.float32 and x.stride(-1) == 1 and state[0].stride(-1) == 1 and \
            4 * (x.shape[1] + 5 * state[0].shape[1]) * 4 <= 65536
        grads = torch.is_grad_enabled() and (x.requires_grad or (state is not None and (state[0].requires_grad or state[1].requires_grad)) or
                                             any(p.requires_grad for p in self.parameters()))
        if grads and fits and os.environ.get("DFOL_LSTM_BWD", "hip") != "torch":
            wt = self._transposed()
            h, c = state
            return _LSTMCellFn.apply(x, h.contiguous(), c.contiguous(), self.weight_ih, self.weight_hh, self.bias_ih, self.bias_hh, wt[1], wt[2])
        if state is None or not x.is_cuda or grads:
            return super(CalibrationLSTMCell, self).forward(x, state)
        h, c = state
        if fits:
            wt = self._transposed()
            return L.lstm_cell(x, h, c.contiguous(), wt[1], wt[2], self.bias_ih, self.bias_hh)          # one launch
        ig = L.linear_act(x.contiguous(), self.weight_ih, self.bias_ih, L.ACT_NONE)
        hg = L.linear_act(h.contiguous(), self.weight_hh, self.bias_hh, L.ACT_NONE)
        return L.lstm_pointwise(ig, hg, c.contiguous())


class _TallLinear(torch.autograd.Function):
    """y = x W^T + b for a very tall x ([rows, K] with millions of rows: one row per object pair).  Forward and grad_x run on the
    bf16x3 split kernel (fp32 results); the weight gradient g^T x contracts over the rows, a shape for which the vendor library picks
    a slow kernel (7.5 ms for [300 x 2.5M] x [2.5M x 256]): it runs on the TN kernel of csrc/dfol_dense_wgrad.hip."""

    @staticmethod
    def _split_ok(x2, k):
        if x2.dtype == torch.bfloat16:                       # bf16-stored activations exist in the bf16 mode only, and only this kernel takes them
            return x2.is_cuda and x2.is_contiguous() and k % 4 == 0 and _lib._dense_math() == "bf16"
        # (any row count since round 6: the last batch of an epoch, or 20 questions per GPU on small scenes, used to leave for the vendor GEMM below 4096 rows)
        return x2.is_cuda and x2.is_contiguous() and k % 4 == 0 and x2.shape[0] >= 1 and x2.dtype == torch.float32 and _lib._dense_math() != "f32"

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.math = _lib._dense_math()
        x2 = x.reshape(-1, x.shape[-1])
        if _TallLinear._split_ok(x2, x2.shape[1]):          # fp32 results from the bf16 matrix pipe (csrc/dfol_dense_split.hip)
            w = weight.detach()
            _lib.note("tall_linear")
            return L.linear_act_split(x2, w if w.is_contiguous() else w.contiguous(), None if bias is None else bias.detach(),
                                      L.ACT_NONE).view(*x.shape[:-1], weight.shape[0])
        if x.is_cuda:
            _lib.fallback("_TallLinear.forward", "rows %d, K %d %% 4 != 0, a non-contiguous input or dense math 'f32'" % (x2.shape[0], x2.shape[1]))
        if x.dtype != weight.dtype:                          # (a bf16-stored input the kernel cannot take: compute in fp32, store as it came)
            return nn.functional.linear(x.to(weight.dtype), weight, bias).to(x.dtype)
        return nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        with _lib.dense_math(ctx.math):
            return _TallLinear._backward(ctx, g)

    @staticmethod
    def _backward(ctx, g):
        x, weight = ctx.saved_tensors
        if g.dtype != x.dtype:                               # (bf16-stored activations: the gradient arrives in the output's dtype already)
            g = g.to(x.dtype)
        g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
        gx = None
        if ctx.needs_input_grad[0]:
            g2c = g2 if g2.is_contiguous() else g2.contiguous()
            if _TallLinear._split_ok(g2c, g2c.shape[1]):
                gx = L.linear_act_split(g2c, weight.detach(), None, L.ACT_NONE, transpose_w=True).view_as(x)
            else:
                if g2.is_cuda:
                    _lib.fallback("_TallLinear.backward (input gradient)", "rows %d, width %d %% 4 != 0 or dense math 'f32'" % (g2c.shape[0], g2c.shape[1]))
                gx = (g2.to(weight.dtype) @ weight).to(x.dtype).view_as(x)
        gw = gb = None
        if ctx.needs_input_grad[1]:
            if g2.is_cuda and g2.dtype in (torch.float32, torch.bfloat16):
                # dY^T X over millions of rows: the deterministic TN kernel (csrc/dfol_dense_wgrad.hip); the bias gradient is the
                # column sums of the dY rows it loads anyway (a separate sum over 3 GB costs 0.75 ms)
                gw = L.linear_wgrad(g2 if g2.stride(-1) == 1 else g2.contiguous(), x2 if x2.stride(-1) == 1 else x2.contiguous(),
                                    bias=ctx.needs_input_grad[2])
                if ctx.needs_input_grad[2]:
                    gw, gb = gw
            else:
                rows = g2.shape[0]
                S = 64                                        # (CPU tensors: the gloo tests of the data-parallel step)
                while S > 1 and rows % S:
                    S //= 2
                gw = torch.bmm(g2.view(S, rows // S, -1).transpose(1, 2), x2.view(S, rows // S, -1)).sum(0) if S > 1 else g2.t() @ x2
        if gb is None and ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return gx, gw, gb


_PAIR_STREAMS = {}


def _pair_side_stream(dev):
    """The side stream of the pair branch in training (None: not used - no gradients, a CPU device, or DFOL_TRAIN_PAIR_STREAM=0)."""
    dev = torch.device(dev)
    if dev.type != "cuda" or not torch.is_grad_enabled() or os.environ.get("DFOL_TRAIN_PAIR_STREAM", "1") == "0":
        return None
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _PAIR_STREAMS.get(key)
    if s is None:
        s = _PAIR_STREAMS[key] = torch.cuda.Stream(device=dev)
    return s


class _FusedHidden1(torch.autograd.Function):
    """z = ELU(U[s] + V[o] + Wg geo(s, o)) over the ordered pairs of a scene, one kernel forward and one backward
    (csrc/dfol_pair_train.hip) instead of two gathers, the geometry features, a tall K = 4 product, two adds and the ELU - and, in the
    backward, instead of the scatter-adds of the gathers (the largest single item of a ragged train step)."""

    @staticmethod
    def forward(ctx, U, V, Wg, pos, world, store=torch.float32, fwd=None):
        """fwd (round 6): None, or a dict {"w2h": the second layer's dfol_pair_pack_w2_f16x2 image, "b2", "hid2", "first"} - the WHOLE forward of the
        pair MLP then runs as one launch of the fused pair kernel (dfol_pair_train_fwd_h2_f32: Z, pre2, the geometry and the first reader's logits
        leave its registers; Z is not read back for the second layer) and pre2 / the logits are left in fwd["out"] for _PairTrunk."""
        ctx.joined = V is None                                  # U is [O, 2 HID1]: U | V, one product of the object features
        UV = U
        if ctx.joined:
            h = U.shape[1] // 2
            U, V = U[:, :h], U[:, h:]
        max_n = max(world._n_list)
        if fwd is not None:
            first = fwd["first"]
            e_rows = req = None
            if first is not None and len(first) > 2:
                extras = first[3] if len(first) > 3 else []
                req_host = np.full((1 + len(extras), world._batch_size), -1, np.int32)
                req_host[0, np.asarray(first[2], np.int64)] = np.arange(len(first[2]), dtype=np.int32)
                e_rows = first[0]
                if extras:                                        # the other readers' embedding rows, behind the first reader's
                    P0 = e_rows.shape[0]
                    e_rows = torch.cat([e_rows] + [first[4].index_select(0, upload(c, U.device)) for c in extras], 0)
                    for i, c in enumerate(extras):
                        req_host[1 + i, :] = P0 + i * world._batch_size + np.arange(world._batch_size, dtype=np.int32)
                req = upload(req_host, U.device)
            z, pre2, geo, x = L.pair_train_fwd_h2(UV.detach() * L.LOG2E, h, pos, Wg.detach(), fwd["w2h"], fwd["b2"], fwd["hid2"], world._img_n_obj,
                                                  world._obj_off, world._pair_off, max_n, world._pair_num, e_rows, req)
            fwd["out"] = (pre2, x)
        else:
            z, geo = L.pair_hidden1_fwd(U, V, pos, Wg, world._obj_off, world._pair_off, world._n_obj, max_n, world._pair_num, store)
        # (the world itself must not hang on the graph: world -> cached activations -> graph -> world would be a reference cycle that
        # only the garbage collector frees, 3 GB per step)
        # fp32 storage: the backward rebuilds z from U, V, Wg and the geometry instead of reading it (dfol_pair_hidden1_bwd_recompute_f32)
        ctx.recompute = store == torch.float32 and L.hidden1_recompute((U, V, Wg), z, max_n, z.shape[1])
        ctx.save_for_backward(z, geo, world._obj_off, world._pair_off, world._n_obj, *((U, V, Wg) if ctx.recompute else ()))
        ctx.max_n, ctx.total_obj = max_n, U.shape[0]
        return z

    @staticmethod
    def backward(ctx, dz):
        z, geo, obj_off, pair_off, n_obj = ctx.saved_tensors[:5]
        du, dv, dwg = L.pair_hidden1_bwd(dz.contiguous(), z, geo, obj_off, pair_off, n_obj, ctx.max_n, ctx.total_obj, ctx.joined,
                                         uvw=ctx.saved_tensors[5:] if ctx.recompute else None)
        return du, dv, dwg, None, None, None, None


class _FusedLogit(torch.autograd.Function):
    """x[r] = Sigmoid(pre2[r]) . E[p(r)] + be[p(r)] for the rows of each predicate's image: Sigmoid, the embedding product and its row
    sum in one pass, and one pass for the three gradients (csrc/dfol_pair_train.hip); the hidden activations are never stored."""

    @staticmethod
    def forward(ctx, pre2, e_rows, be_rows, pred_off, max_rows):
        ctx.save_for_backward(pre2, e_rows, pred_off)
        return L.pair_logit_fwd(pre2, e_rows, be_rows, pred_off, max_rows)

    @staticmethod
    def backward(ctx, dx):
        pre2, e_rows, pred_off = ctx.saved_tensors
        dp2, de, dbe = L.pair_logit_bwd(dx.contiguous(), pre2, e_rows, pred_off, need_bias=ctx.needs_input_grad[2])
        return dp2, de, dbe, None, None


class _PairTrunk(torch.autograd.Function):
    """pre2 = z W2^T + b2 for every pair row, as the ROOT of a deferred backward: the logit layers that read pre2 (`_HeadUse`, one per relation
    operator group) do not send a [pairs, HID2] gradient back through autograd.  Each of them runs, in ITS backward, the two products that
    consume dpre2 (dz (+)= dpre2 W2, dW2 = dpre2^T z) with dpre2 rebuilt on the fly from pre2 (csrc/dfol_dense_tall.hip /
    dfol_dense_split.hip LsProducer, csrc/dfol_dense_wgrad.hip pair_wgrad_fused_kernel) - the weight-gradient pass also yields the reader's
    own sums (dE, dbe) - leaves the results in `state`, which it shares with this node, and returns a zero for the one-element `token`
    that ties it to this node.  Autograd runs this backward after all of them: it hands out what they accumulated.  3 GB written and 9 GB
    read less per step at 256 x 100 objects."""

    @staticmethod
    def forward(ctx, z, weight, bias, state, first, pre=None):
        """first: None, or the first reader's (embedding rows [P, HID2] (detached), row -> embedding row [pairs] int32): its logit layer's
        forward then comes out of the product's epilogue as partial sums [slots, pairs] (no second pass over pre2).
        pre: (pre2, x_part or None) when the fused forward kernel (_FusedHidden1 with `fwd`) has computed the product already."""
        w = weight.detach()
        w = w if w.is_contiguous() else w.contiguous()
        b = None if bias is None else bias.detach()
        f16x2 = _lib._dense_math() == "f16x2"
        if pre is not None:
            pre2, x_part = pre[0], (z.new_zeros(0) if pre[1] is None else pre[1])
        elif z.dtype == torch.bfloat16:                         # the bf16 mode's stored activations: the persistent product over them (the logit layer keeps its own pass: its
            pre2, x_part = L.linear_tall_h2(z, w, b)[0], z.new_zeros(0, dtype=torch.float32)      # epilogue form is the slower one at half the bytes)
        elif f16x2 and L.linear_tall_supported(z.shape[0], w.shape[0], w.shape[1]):
            # one persistent workgroup per CU over the row blocks (csrc/dfol_dense_tall.hip): the same bits, 1.9 ms against 2.6 at 256 x 100 objects
            pre2, x_part = L.linear_tall_h2(z, w, b, *((first[1], first[0]) if first is not None else ()))
            x_part = z.new_zeros(0) if x_part is None else x_part
        elif first is not None and f16x2:
            pre2, x_part = L.linear_logit_h2(z, w, b, first[1], first[0])
        else:
            pre2, x_part = L.linear_act_split(z, w, b, L.ACT_NONE), z.new_zeros(0)
        state.update(z=z.detach(), w=w, pre2=pre2, need_dz=ctx.needs_input_grad[0], need_dw=ctx.needs_input_grad[1],
                     need_db=bias is not None and ctx.needs_input_grad[2], dz=None, dw=None, db=None)
        ctx.state, ctx.shapes = state, (z.shape, weight.shape)
        ctx.mark_non_differentiable(pre2, x_part)
        ctx.set_materialize_grads(False)                     # (or the engine fills a [pairs, HID2] zero gradient for pre2 on the way in: 3 GB, 0.4 - 0.6 ms)
        return pre2, z.new_zeros(1), x_part

    @staticmethod
    def backward(ctx, _g_pre2, _g_token, _g_part):
        st = ctx.state
        jobs = st.get("dz_jobs")
        if jobs:                                             # the readers' dZ shares in one pass (several readers: see _HeadUse.backward)
            if len(jobs) == 1:
                dxj, ej, rpj = jobs[0]
                pred_off = None
                st["dz"], _ = L.pair_head_products(dxj, st["pre2"], st["z"], st["w"], ej, pred_off, rpj, True, False, dz_out=st["dz"])
            else:
                st["dz"] = L.pair_dz_tall_multi([j[0] for j in jobs], st["pre2"], [j[1] for j in jobs], jobs[0][2], st["w"], dz_out=st["dz"])
                _lib.note("pair_dz_multi")
        dz, dw, db = st["dz"], st["dw"], st["db"]
        dev = st["w"].device
        if st["need_dz"] and dz is None:
            dz = torch.zeros(ctx.shapes[0], dtype=st["z"].dtype, device=dev)
        if st["need_dw"] and dw is None:
            dw = torch.zeros(ctx.shapes[1], dtype=torch.float32, device=dev)
        if st["need_db"] and db is None:
            db = torch.zeros(ctx.shapes[1][0], dtype=torch.float32, device=dev)
        st.clear()                                           # (the activations must not outlive the step through this dictionary)
        return dz, dw, db, None, None, None


class _HeadUse(torch.autograd.Function):
    """x[r] = Sigmoid(pre2[r]) . E[p(r)] + be[p(r)] (the forward of `_FusedLogit`) on a `_PairTrunk`; the backward does this reader's share of
    the trunk's backward (see there).  sums_ok: every predicate owns at least 64 pair rows or none - the embedding rows' gradients then come
    out of the weight-gradient pass; otherwise (and when the second layer's weights do not train) from a pass of their own over pre2."""

    @staticmethod
    def forward(ctx, token, pre2, e_rows, be_rows, pred_off, row_pred, max_rows, state, x_part, sums_ok):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(pre2, e_rows, pred_off, row_pred)
        ctx.state, ctx.sums_ok = state, bool(sums_ok)
        state["readers"] = state.get("readers", 0) + 1            # (several: their dZ shares are taken in ONE pass by the trunk's backward)
        if x_part is not None:                                # the trunk's epilogue has this reader's partial sums already
            return x_part.sum(0) + be_rows.index_select(0, row_pred)
        return L.pair_logit_fwd(pre2, e_rows, be_rows, pred_off, max_rows)

    @staticmethod
    def backward(ctx, dx):
        pre2, e_rows, pred_off, row_pred = ctx.saved_tensors
        st = ctx.state
        if dx is None:                                        # (this reader's logits did not reach the loss)
            return (None,) * 10
        dx = dx.contiguous()
        e_rows = e_rows if e_rows.is_contiguous() else e_rows.contiguous()
        need_be = ctx.needs_input_grad[3]
        if pre2.dtype == torch.bfloat16:
            return _HeadUse._backward_bf16(ctx, st, dx, pre2, e_rows, pred_off, row_pred, need_be)
        mode = os.environ.get("DFOL_HEAD_SUMS", "auto")       # "1" / "0": always / never take the sums from the weight-gradient pass (where it can)
        # several readers of the trunk (relate hops of a program, option slots): their dZ shares - a pass over pre2 each and, from the second on,
        # a read-modify-write of dZ - are left to ONE pass in the trunk's backward (_PairTrunk.backward, dfol_pair_dz_tall_multi_f32); the readers
        # of a trunk share one row -> predicate map when each covers every pair row of the batch (P predicates = the batch's questions)
        defer = (st["need_dz"] and st.get("readers", 1) >= 2 and os.environ.get("DFOL_DZ_MULTI", "1") != "0" and pre2.dtype == torch.float32
                 and e_rows.shape[0] == st.get("dz_P", e_rows.shape[0]) and int(pred_off.shape[0]) - 1 == e_rows.shape[0]
                 and L.linear_tall_supported(pre2.shape[0], st["w"].shape[1], pre2.shape[1]) and row_pred is not None and row_pred.shape[0] == pre2.shape[0]
                 and row_pred.data_ptr() in st.get("allq", ()))
        need_dz_now = st["need_dz"] and not defer
        if defer:
            st["dz_P"] = e_rows.shape[0]
            st.setdefault("dz_jobs", []).append((dx, e_rows, row_pred))
            _lib.note("head_use_dz_deferred")
        if ctx.sums_ok and st["need_dw"] and pre2.shape[1] % 3 == 0 and mode != "0" and (mode == "1" or pre2.shape[0] >= (1 << 20)):
            _lib.note("head_use_backward_sums")
            dz, dw, de, dbe, db2 = L.pair_head_products(dx, pre2, st["z"], st["w"], e_rows, pred_off, row_pred, need_dz_now, True, dz_out=st["dz"],
                                                        sums=True, need_bias=need_be)
        else:
            _lib.note("head_use_backward")
            de, dbe, db2p = L.pair_head_sums(dx, pre2, e_rows, pred_off, need_bias=need_be)
            db2 = db2p.sum(0) if st["need_db"] else None
            dz, dw = L.pair_head_products(dx, pre2, st["z"], st["w"], e_rows, pred_off, row_pred, need_dz_now, st["need_dw"], dz_out=st["dz"])
        if need_dz_now:
            st["dz"] = dz                                     # (a later reader adds into it)
        if st["need_dw"]:
            st["dw"] = dw if st["dw"] is None else st["dw"] + dw
        if st["need_db"]:
            st["db"] = db2 if st["db"] is None else st["db"] + db2
        return dx.new_zeros(1), None, de, dbe, None, None, None, None, None, None

    @staticmethod
    def _backward_bf16(ctx, st, dx, pre2, e_rows, pred_off, row_pred, need_be):
        """The bf16 mode (bf16-stored pre2 / Z): dz in bfloat16 from the persistent product with dpre2 rebuilt and rounded as the materialised
        route stores it, dW2 and the sums from one pass; predicates too small for that pass go the materialised way, locally."""
        with _lib.dense_math("bf16"):
            if ctx.sums_ok and st["need_dw"]:
                dz = L.pair_dz_tall_bf16(dx, pre2, e_rows, row_pred, st["w"], dz_out=st["dz"]) if st["need_dz"] else None
                dw, de, dbe, db2 = L.pair_wgrad_sums_bf16(dx, pre2, st["z"], e_rows, pred_off, row_pred, need_bias=need_be)
            else:
                dp2, de, dbe = L.pair_logit_bwd(dx, pre2, e_rows, pred_off, need_bias=need_be)
                dz = None
                if st["need_dz"]:
                    dz = L.linear_act_split(dp2, st["w"], None, L.ACT_NONE, transpose_w=True)
                    dz = dz if st["dz"] is None else st["dz"] + dz
                dw = db2 = None
                if st["need_dw"]:
                    dw, db2 = L.linear_wgrad(dp2, st["z"], bias=True)
                elif st["need_db"]:
                    db2 = dp2.float().sum(0)
        if st["need_dz"]:
            st["dz"] = dz
        if st["need_dw"]:
            st["dw"] = dw if st["dw"] is None else st["dw"] + dw
        if st["need_db"]:
            st["db"] = db2 if st["db"] is None else st["db"] + db2
        return dx.new_zeros(1), None, de, dbe, None, None, None, None, None, None


def _concept_plan(cols, device, cache):
    """Fixed-order combination plan for per-predicate gradient rows of the embedding layer: predicates naming the same concept are
    summed in predicate order (stable sort), the unique concepts then receive one row each - no atomics, repeatable bit for bit.
    -> (order [V] int32: predicate of each sorted slot, seg_off [U+1] int32, ucols [U] int64), device tensors cached by content."""
    cols = np.asarray(cols, np.int64)
    key = ("concepts", str(device), cols.tobytes())
    hit = cache.get(key)
    if hit is None:
        valid = np.nonzero(cols >= 0)[0]
        order = valid[np.argsort(cols[valid], kind="stable")]
        sc = cols[order]
        ucols, start = np.unique(sc, return_index=True)
        seg = np.concatenate([start, [len(sc)]]).astype(np.int32)
        hit = (torch.as_tensor(order.astype(np.int32)).to(device), torch.as_tensor(seg).to(device), torch.as_tensor(ucols.astype(np.int64)).to(device))
        cache[key] = hit
    return hit


def _combine_concept_rows(rows, plan, out_shape, leaf=None):
    """rows [P, ...] per predicate -> gradient of `out_shape` (one row per concept), deterministic: one launch (dfol_concept_rows_f32).
    leaf: the weight itself.  When it carries a persistent fp32 gradient (the views of parallel.GradBucket; any preallocated .grad), the rows
    are added straight into it and None is returned - the dense, mostly zero [concepts, width] intermediate (2.8 MB for the embedding
    layer: a fill, and an add of the whole matrix by autograd's accumulation, per call and four calls per step) never exists.  The sum per
    concept is formed in predicate order before it meets the gradient, as autograd's accumulation of the dense form did."""
    order, seg_off, ucols = plan
    direct = leaf is not None and DIRECT_GRAD[0] and leaf.is_leaf and leaf.grad is not None and leaf.grad.dtype == torch.float32 \
        and leaf.grad.is_contiguous() and rows.is_cuda and tuple(leaf.grad.shape) == tuple(out_shape)
    if direct:
        if order.numel():
            L.concept_rows(rows, order, seg_off, ucols, leaf.grad, True)
        return None
    out = torch.zeros(out_shape, dtype=rows.dtype, device=rows.device)
    if order.numel() == 0:
        return out
    if rows.is_cuda:
        return L.concept_rows(rows, order, seg_off, ucols, out, False)
    flat = rows.reshape(rows.shape[0], -1)
    summed = L.segment_sum_rows(L.gather_rows(flat, order), seg_off)
    out.reshape(out_shape[0], -1).index_copy_(0, ucols, summed)          # unique rows: no accumulation, no atomics
    return out


# Off unless a caller that owns the parameters' gradients switches it on around ITS backward pass (training.train_batch / GraphedTrainStep with a
# gradient bucket): torch.autograd.grad(...) with respect to such a weight, or a hook on its gradient, must keep seeing the gradient as a
# returned value.  DFOL_DIRECT_GRAD=0 keeps it off everywhere (A/B runs).
DIRECT_GRAD = [False]


class direct_grad(object):
    """with direct_grad(): backward passes inside add embedding-layer gradient rows straight into the weights' persistent .grad."""

    def __init__(self, on=True):
        self._on = bool(on) and os.environ.get("DFOL_DIRECT_GRAD", "1") != "0"

    def __enter__(self):
        self._old, DIRECT_GRAD[0] = DIRECT_GRAD[0], self._on

    def __exit__(self, *exc):
        DIRECT_GRAD[0] = self._old
        return False


class _AttrLL(torch.autograd.Function):
    """[P, NS] blocks LogSigmoid(hidden[o] . E[col_p] + be[col_p]) of the requested attribute columns: the inference kernel forward,
    three small HIP kernels backward (csrc/dfol_logic_bwd.hip), no gathers / scatter-adds, deterministic."""

    @staticmethod
    def forward(ctx, hidden, emb_w, emb_b, obj_off, pred_q, pred_col, NS, plan):
        ctx.save_for_backward(hidden, emb_w, emb_b if emb_b is not None else hidden.new_empty(0), obj_off, pred_q, pred_col)
        ctx.plan, ctx.has_bias = plan, emb_b is not None
        return L.attr_ll(hidden.detach(), emb_w.detach(), None if emb_b is None else emb_b.detach(), obj_off, pred_q, pred_col, NS, -30.0)

    @staticmethod
    def backward(ctx, g):
        hidden, emb_w, emb_b, obj_off, pred_q, pred_col = ctx.saved_tensors
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        d_hidden, dE, db = L.attr_ll_bwd(g.contiguous(), hidden, emb_w, emb_b if ctx.has_bias else None, obj_off, pred_q, pred_col,
                                         ctx.needs_input_grad[0], ctx.needs_input_grad[1], need_b)
        gw = _combine_concept_rows(dE, ctx.plan, emb_w.shape, emb_w) if ctx.needs_input_grad[1] else None
        gb = _combine_concept_rows(db, ctx.plan, emb_b.shape, emb_b) if need_b else None
        return d_hidden, gw, gb, None, None, None, None, None


class _EmbRows(torch.autograd.Function):
    """(E[cols], be[cols]) with a deterministic backward: index_select's backward is an atomic scatter-add."""

    @staticmethod
    def forward(ctx, emb_w, emb_b, cols_dev, plan):
        ctx.plan, ctx.shapes = plan, (emb_w.shape, emb_b.shape)
        ctx.save_for_backward(emb_w, emb_b)                  # (the leaves: their persistent gradients can take the rows directly)
        return emb_w.index_select(0, cols_dev), emb_b.index_select(0, cols_dev)

    @staticmethod
    def backward(ctx, gw_rows, gb_rows):
        emb_w, emb_b = ctx.saved_tensors
        gw = _combine_concept_rows(gw_rows.contiguous(), ctx.plan, ctx.shapes[0], emb_w) if ctx.needs_input_grad[0] else None
        gb = _combine_concept_rows(gb_rows.contiguous(), ctx.plan, ctx.shapes[1], emb_b) if ctx.needs_input_grad[1] else None
        return gw, gb, None, None


class ClassifierOracle(OracleBase):
    """classifier_oracle.py:11-156 with cached tables (the only mode the reference's experiments use:
    gqa_interpreter_experiments.py:209-210 builds it with cached=True)."""

    def __init__(self, ontology, attribute_network, relation_network, embedding_network, normalize=False, cached=False):
        super(ClassifierOracle, self).__init__(ontology, feature_dim=1)
        self._attribute_network = attribute_network
        self._relation_network = relation_network
        self._embedding_network = embedding_network
        self._normalize = normalize
        self._cached = cached
        self._needed_columns = True       # compute only the likelihood columns a program asks for (when the MLP shape allows)
        self._split_cache = None
        self._index_cache = _lib.LRUCache(64)
        # storage type of prefetched relation tiles: torch.bfloat16 halves the tile stream of the Relate kernel (BASELINE configs[4]);
        # an opt-in whose results differ from the reference by the rounding of the stored likelihoods.  Used when the scene's padded
        # width is a multiple of 8 and the operator runs on the fused single-posterior kernel; everything else stays fp32.
        self._tile_dtype = torch.float32

    # ---- a3: the cached tables (classifier_oracle.py:145-156) ------------------------------------------
    def _relation_embedding(self):
        """Rows of the embedding layer that are relations ([:, relation_index] commutes with the GEMM)."""
        lin = self._embedding_network.linear
        idx = torch.as_tensor(self._ontology._relation_index, dtype=torch.int64, device=lin.weight.device)
        live = torch.is_grad_enabled() and lin.weight.requires_grad      # training: the relation rows receive gradient too
        w = (lin.weight if live else lin.weight.detach()).index_select(0, idx).contiguous()
        b = None if lin.bias is None else (lin.bias if live else lin.bias.detach()).index_select(0, idx).contiguous()
        return w, b

    def compute_all_log_likelihood_2(self, object_features, pair_object_features):
        if self._embedding_network is None or self._attribute_network is None:
            attr_output = object_features
        else:
            attr_output = self._embedding_network(self._attribute_network(object_features))
        if self._embedding_network is None or self._relation_network is None or pair_object_features is None:
            rel_output = pair_object_features
        else:
            h = self._relation_network(pair_object_features)
            w, b = self._relation_embedding()
            rel_output = L.linear_act(h, w, b, L.ACT_LOGSIGMOID)      # only the 333 relation columns are computed
        return attr_output, rel_output

    # ---- needed-columns mode ---------------------------------------------------------------------------
    def supports_needed_columns(self):
        """The fused path needs the classifier-oracle shape of the reference's configs: a relation MLP with exactly one
        hidden layer (Linear, ELU, Linear, Sigmoid), hidden <= 256 (multiple of 4), embedding input <= 320."""
        if not (self._needed_columns and self._cached):
            return False
        if self._attribute_network is None or self._relation_network is None or self._embedding_network is None:
            return False
        rel = self._relation_network._network
        if rel is None or self._attribute_network._network is None:
            return False
        lins = [m for m in rel if isinstance(m, nn.Linear)]
        acts = [m for m in rel if isinstance(m, (nn.ELU, nn.Sigmoid))]
        if len(lins) != 2 or len(acts) != 2 or not isinstance(acts[0], nn.ELU) or not isinstance(acts[1], nn.Sigmoid):
            return False
        if any(isinstance(m, nn.Dropout) and m.training and m.p > 0 for m in rel):
            return False
        hid1, hid2 = lins[0].out_features, lins[1].out_features
        return hid1 % 4 == 0 and hid1 <= 256 and hid2 <= 320 and (lins[0].in_features - 4) % 2 == 0 and lins[1].bias is not None

    def _split_first_layer(self):
        """W1 [HID1, 2D+4] -> stacked per-object weight [2 HID1, D] (+ bias [b1, 0]) and the geometry columns [HID1, 4].
        W1 [obj_s, obj_o, geo] + b1 = (W1a obj_s + b1) + W1b obj_o + Wg geo: exact up to fp32 reassociation."""
        lin = [m for m in self._relation_network._network if isinstance(m, nn.Linear)][0]
        # The fp16x2 pair kernel (csrc/dfol_pair_h2.hip) takes U | V in units of ln 2, i.e. multiplied by log2(e): its ELU then needs no multiply in
        # front of the hardware exponential (a build tick of that kernel is paced by its instruction count).  The factor goes into the stacked
        # weight and bias here, once per weight version; the kernel scales the geometry columns itself and its pack kernel folds ln 2 into W2.
        scaled = self._pair_kind() == "f16x2"
        key = (lin.weight.data_ptr(), lin.weight._version, None if lin.bias is None else lin.bias._version, scaled)
        if self._split_cache is None or self._split_cache[0] != key:
            w = lin.weight.detach()
            hid1, D = w.shape[0], (w.shape[1] - 4) // 2
            wuv = torch.cat([w[:, :D], w[:, D:2 * D]], 0).contiguous()
            b1 = lin.bias.detach() if lin.bias is not None else torch.zeros(hid1, device=w.device)
            buv = torch.cat([b1, torch.zeros_like(b1)]).contiguous()
            if scaled:
                wuv, buv = wuv * L.LOG2E, buv * L.LOG2E
            wg = w[:, 2 * D:2 * D + 4].contiguous()
            self._split_cache = (key, wuv, buv, wg, hid1, D)
        return L.keep_alive(self._split_cache)[1:]

    def _pair_kind(self):
        """Which fused pair kernel evaluates this oracle's relation tiles: "f16x2", "bf16x3", "packed" (fp32 pipe, packed W2) or "plain"."""
        packed = self._padded_second_layer()[3]
        if isinstance(packed, tuple):
            return packed[0]
        return "plain" if packed is None else "packed"

    def _padded_second_layer(self):
        """W2 zero-padded to a multiple of 32 rows, so the fused pair kernel's main loop needs no bounds checks."""
        lin = [m for m in self._relation_network._network if isinstance(m, nn.Linear)][1]
        key = (lin.weight.data_ptr(), lin.weight._version, lin.bias._version, L.pair_math())
        cache = getattr(self, "_w2_cache", None)
        if cache is None or cache[0] != key:
            w = lin.weight.detach()
            rows = (w.shape[0] + 31) // 32 * 32
            wp = torch.zeros(rows, w.shape[1], dtype=w.dtype, device=w.device)
            wp[:w.shape[0]] = w
            packed = None
            if w.shape[1] % L.PACKED_W2_CHUNK == 0 and w.shape[1] <= 256 and w.shape[0] <= L.PACKED_W2_ROWS \
                    and os.environ.get("DFOL_PAIR_PACKED", "1") != "0":
                packed = L.pair_pack_w2(wp, w.shape[0])      # the layout of the occupancy-2 pair kernel (csrc/dfol_pair.hip)
                # full-size second layer: fp32 results from the fp16 matrix pipe (two fp16 pieces per operand, three products:
                # csrc/dfol_pair_h2.hip) or, DFOL_PAIR_MATH=bf16x3, from the bf16 pipe (three pieces, six products: csrc/dfol_pair_split.hip);
                # DFOL_PAIR_MATH=f32 keeps the fp32 matrix pipe
                if L.pair_split_supported(w.shape[1], w.shape[0]) and L.pair_math() != "f32":
                    packed = ("f16x2", L.pair_pack_w2_h2(wp, w.shape[0])) if L.pair_math() == "f16x2" else ("bf16x3", L.pair_pack_w2_split(wp, w.shape[0]))
            self._w2_cache = (key, wp, lin.bias.detach().contiguous(), w.shape[0], packed)
        return L.keep_alive(self._w2_cache)[1:]

    def prepare_scene(self, world, obj, train=False):
        """Hidden activations of a scene: attribute hidden [O, H] and the per-object halves of the pair MLP's first layer.
        train=True: gradients must reach the oracle / featurizer weights; the requested columns are then evaluated by
        differentiable tensor ops (library GEMMs) instead of the fused forward-only kernels, still without the full tables."""
        world._lazy = self
        world._obj = obj
        world._train = bool(train)
        world._hidden_attr = self._attribute_network(obj)
        world._attr_table = None
        world._rel_table = None
        world._pair_h = None
        world._pair_pre2 = None
        world._pair_head = world._pair_z = None
        if train:
            return
        wuv, buv, wg, hid1, D = self._split_first_layer()
        assert obj.shape[1] == D, "object feature width does not match the relation network"
        world._uv = L.linear_act(obj, wuv, buv, L.ACT_NONE)

    # ---- needed columns with gradients (training of the oracle, trainer.py:429-442) ------------------------------
    def _fused_training(self, world):
        """The fused training kernels apply (widths, image sizes) and are not switched off (DFOL_TRAIN_FUSED=0)."""
        lin1, lin2 = [m for m in self._relation_network._network if isinstance(m, nn.Linear)]
        return os.environ.get("DFOL_TRAIN_FUSED", "1") != "0" and world._pair_num > 0 and \
            L.pair_train_supported(lin1.weight.shape[0], lin2.weight.shape[0], max(world._n_list))

    def _why_not_fused(self, world):
        lin1, lin2 = [m for m in self._relation_network._network if isinstance(m, nn.Linear)]
        if os.environ.get("DFOL_TRAIN_FUSED", "1") == "0":
            return "DFOL_TRAIN_FUSED=0"
        return "the fused pair kernels take hidden widths 16..1024 with width / 4 a power of two, an embedding input <= 512 and images of at most " \
               "16 * 4096 / width objects; this network is %d -> %d with up to %d objects per image" % (lin1.weight.shape[0], lin2.weight.shape[0],
                                                                                                     max(world._n_list) if world._n_list else 0)

    def _pair_pre2_autograd(self, world, first=None):
        """pre2 = W2 ELU(W1 [obj_s, obj_o, geo] + b1) + b2 for every ordered pair [pairs, HID2] (the hidden layer before its Sigmoid),
        with the first layer split per object exactly as the fused inference kernel does; one evaluation per scene.
        first: the reader that asks first, as (embedding rows, row -> embedding row): see _PairTrunk.forward."""
        if getattr(world, "_pair_pre2", None) is None:
            lin1, lin2 = [m for m in self._relation_network._network if isinstance(m, nn.Linear)]
            obj = world._obj
            D = (lin1.weight.shape[1] - 4) // 2
            assert obj.shape[1] == D, "object feature width does not match the relation network"
            pos = obj[:, D - 4:].detach()                       # batch_gqa_boxfeatures_pipeline.py:263-279
            hid1 = lin1.weight.shape[0]
            fused = self._fused_training(world) and hid1 % 4 == 0
            if fused and obj.is_cuda and obj.dtype == torch.float32 and os.environ.get("DFOL_TRAIN_UV_JOINED", "1") != "0":
                # U | V as ONE product of the object features ([2 HID1, D] weight: the subject block over the object block; the bias
                # belongs to U alone): the features are read once forward, their gradient is one K = 2 HID1 product backward instead
                # of two products and an add, and the pair layer's backward writes dU | dV into one buffer
                wuv = torch.cat([lin1.weight[:, :D], lin1.weight[:, D:2 * D]], 0)
                buv = torch.cat([lin1.bias, lin1.bias.new_zeros(hid1)])
                U, V = L.linear_act(obj, wuv, buv, L.ACT_NONE), None
            elif obj.is_cuda and obj.dtype == torch.float32:
                U = L.linear_act(obj, lin1.weight[:, :D], lin1.bias, L.ACT_NONE)          # forward and both gradients on the HIP kernels
                V = L.linear_act(obj, lin1.weight[:, D:2 * D], None, L.ACT_NONE)
            else:
                U = nn.functional.linear(obj, lin1.weight[:, :D], lin1.bias)
                V = nn.functional.linear(obj, lin1.weight[:, D:2 * D])
            if not fused and obj.is_cuda:
                _lib.fallback("relation network training (first layer: gathers + torch ELU)", self._why_not_fused(world))
            if fused:
                # bf16 mode: Z, pre2 and their gradients live in bfloat16 (what autocast stores; half the bytes of the step's streams and
                # of the two tall products' operands) - csrc/dfol_pair_train.hip, dfol_linear_act_bf16_bf16, dfol_linear_wgrad_bias_bf16_bf16
                store = torch.bfloat16 if L.bf16_store(lin1.weight.shape[0], lin2.weight.shape[0]) else torch.float32
                _lib.note("fused_hidden1")
                fwd = self._fused_forward(world, U, V, store, lin1, lin2, first)
                z = _FusedHidden1.apply(U.contiguous(), None if V is None else V.contiguous(), lin1.weight[:, 2 * D:2 * D + 4].contiguous(),
                                        pos, world, store, fwd)
            else:
                s_idx, o_idx = world.pair_index()
                ps, po = pos.index_select(0, s_idx), pos.index_select(0, o_idx)
                dx = ps[:, 0] + ps[:, 2] / 2.0 - po[:, 0] - po[:, 2] / 2.0
                dy = ps[:, 1] + ps[:, 3] / 2.0 - po[:, 1] - po[:, 3] / 2.0
                dist = torch.sqrt(dx * dx + dy * dy)
                geo = torch.stack([dist, torch.asin(dy / dist.clamp(min=1e-10)), torch.sign(po[:, 0] - ps[:, 0]),
                                   torch.sign(po[:, 1] - ps[:, 1])], 1)
                z = nn.functional.elu(U.index_select(0, s_idx) + V.index_select(0, o_idx)
                                      + _TallLinear.apply(geo, lin1.weight[:, 2 * D:2 * D + 4], None))
            world._pair_head, world._pair_z = None, z
            if self._head_fused(world, z, lin1, lin2):
                # the head's backward without dpre2 in memory: pre2 comes out of the trunk node, its readers register with `uses`
                state = {}
                _lib.note("pair_trunk")
                pre = None
                if fused and fwd is not None:
                    pre = fwd.pop("out")
                    _lib.note("pair_forward_fused")
                pre2, token, x_part = _PairTrunk.apply(z, lin2.weight, lin2.bias, state, first, pre)
                world._pair_head = (token, state, x_part)
                world._pair_x_rows = pre is not None             # x_part: one row of raw logits per reader (the fused forward) / partial sums of one reader
                world._pair_pre2 = pre2
            else:
                world._pair_pre2 = _TallLinear.apply(z, lin2.weight, lin2.bias)
        return world._pair_pre2

    def _fused_forward(self, world, U, V, store, lin1, lin2, first):
        """The fused forward kernel's operands when the whole pair MLP's forward can run as one launch (round 6), else None: fp32
        storage on the two-piece fp16 pipes, the joined U | V product, widths the fused pair kernel takes (HID1 <= 256, 256 < HID2 <= 320), the
        deferred head (whose readers take pre2 from the trunk) and a backward that rebuilds Z from U | V instead of reading it."""
        if os.environ.get("DFOL_TRAIN_FWD_FUSED", "1") == "0" or V is not None or store != torch.float32 or not U.is_cuda:
            return None
        hid1, hid2 = lin1.weight.shape[0], lin2.weight.shape[0]
        if _lib._dense_math() != "f16x2" or L.pair_math() != "f16x2" or not L.pair_split_supported(hid1, hid2):
            return None
        if os.environ.get("DFOL_TRAIN_HEAD_FUSED", "1") == "0" or not L.pair_head_fused_supported(hid1, hid2) or not torch.is_grad_enabled():
            return None
        if not bool(L.load().dfol_pair_hidden1_bwd_recompute_supported(int(max(world._n_list)), int(hid1))) or os.environ.get("DFOL_H1B_RECOMPUTE", "1") == "0":
            return None
        packed = self._padded_second_layer()[3]
        if not (isinstance(packed, tuple) and packed[0] == "f16x2"):
            return None
        return {"w2h": packed[1], "b2": lin2.bias.detach().contiguous(), "hid2": int(hid2), "first": first}

    def _head_fused(self, world, z, lin1, lin2):
        """The deferred head backward applies: fused training kernels, fp32-stored activations on the split-operand pipes, widths the one-workgroup
        weight-gradient kernel takes, and not switched off (DFOL_TRAIN_HEAD_FUSED=0)."""
        if os.environ.get("DFOL_TRAIN_HEAD_FUSED", "1") == "0" or not (self._fused_training(world) and z.is_cuda and torch.is_grad_enabled()):
            return False
        if not L.pair_head_fused_supported(lin1.weight.shape[0], lin2.weight.shape[0]):
            return False
        if z.dtype == torch.bfloat16:
            # the bf16 mode with bf16-stored activations: built and pinned (tests), but NOT faster than the materialised route there - at half
            # the bytes the two products that rebuild dpre2 are bound by its arithmetic, done twice (6.98 against 7.01 ms per step): opt-in
            return os.environ.get("DFOL_TRAIN_HEAD_FUSED_BF16", "0") == "1" and _lib._dense_math() == "bf16" and lin2.weight.shape[0] % 4 == 0 and \
                L.linear_tall_supported(z.shape[0], lin2.weight.shape[0], lin2.weight.shape[1]) and \
                L.linear_tall_supported(z.shape[0], lin2.weight.shape[1], lin2.weight.shape[0])
        return z.dtype == torch.float32 and _lib._dense_math() in ("f16x2", "bf16x3") and z.shape[0] >= 1

    def _pair_hidden_autograd(self, world):
        """h = Sigmoid(pre2) [pairs, HID2]; shared by all relation operators of the scene."""
        if world._pair_h is None:
            pre2 = self._pair_pre2_autograd(world)
            if getattr(world, "_pair_head", None) is not None:
                # the trunk's pre2 carries no autograd edge of its own (its readers register with the trunk); a reader that cannot - rows
                # gathered out of order, no-op tokens - gets a second, ordinary evaluation of the layer from the kept first hidden layer
                lin1, lin2 = [m for m in self._relation_network._network if isinstance(m, nn.Linear)]
                _lib.note("pair_second_evaluation")
                pre2 = _TallLinear.apply(world._pair_z, lin2.weight, lin2.bias)
            world._pair_h = torch.sigmoid(pre2.float())
        return world._pair_h

    def _pair_hidden_dense(self, world, n):
        """The same hidden layer for a batch whose images all have n objects, as [Q, n, n, HID2] over ALL (s, o) slots: the two
        per-object halves broadcast instead of being gathered per pair, so the backward is a reduction, not a scatter-add."""
        if world._pair_h is None:
            lin1, lin2 = [m for m in self._relation_network._network if isinstance(m, nn.Linear)]
            obj, Q = world._obj, world._batch_size
            D = (lin1.weight.shape[1] - 4) // 2
            assert obj.shape[1] == D, "object feature width does not match the relation network"
            if obj.is_cuda:
                _lib.fallback("relation network training (uniform batch: dense torch ops)", self._why_not_fused(world))
            U = nn.functional.linear(obj, lin1.weight[:, :D], lin1.bias).view(Q, n, 1, -1)
            V = nn.functional.linear(obj, lin1.weight[:, D:2 * D]).view(Q, 1, n, -1)
            pos = obj[:, D - 4:].detach().view(Q, n, 4)           # batch_gqa_boxfeatures_pipeline.py:263-279
            ps, po = pos[:, :, None, :], pos[:, None, :, :]
            dx = ps[..., 0] + ps[..., 2] / 2.0 - po[..., 0] - po[..., 2] / 2.0
            dy = ps[..., 1] + ps[..., 3] / 2.0 - po[..., 1] - po[..., 3] / 2.0
            dist = torch.sqrt(dx * dx + dy * dy)
            geo = torch.stack([dist, torch.asin(dy / dist.clamp(min=1e-10)), torch.sign(po[..., 0] - ps[..., 0]),
                               torch.sign(po[..., 1] - ps[..., 1])], -1)
            z = nn.functional.elu(U + V + _TallLinear.apply(geo, lin1.weight[:, 2 * D:2 * D + 4], None))
            world._pair_h = torch.sigmoid(_TallLinear.apply(z, lin2.weight, lin2.bias))
        return world._pair_h

    def _relation_tiles_dense(self, world, low, pred_q_host, n):
        emb = self._embedding_network.linear
        dev = world._device
        full = self._relation_full_columns(low.cols)
        pq = np.asarray(list(pred_q_host), np.int64)
        P, NS = len(pq), world._NS
        h = self._pair_hidden_dense(world, n)
        eye = torch.eye(n, dtype=torch.bool, device=dev)
        valid = full >= 0
        cols = upload(np.where(valid, full, 0).astype(np.int64), dev)
        # one pass over all predicates: each reads its image's [n, n, H] slab against its own embedding row (the per-concept loop
        # of the ragged form would scatter-add into the 3 GB hidden gradient once per concept)
        hq = h if (P == world._batch_size and np.array_equal(pq, np.arange(P))) else h.index_select(0, upload(pq, dev))
        x = (hq * emb.weight.index_select(0, cols)[:, None, None, :]).sum(-1) + emb.bias.index_select(0, cols)[:, None, None]
        tiles = nn.functional.logsigmoid(x).masked_fill(eye, -30.0)                     # self-relations stay absent
        if not valid.all():
            tiles = tiles.masked_fill(upload(~valid, dev)[:, None, None], -30.0)
        if NS != n:
            tiles = nn.functional.pad(tiles, (0, NS - n, 0, NS - n), value=-30.0)
        return tiles

    def _relation_tiles_autograd(self, world, low, pred_q_host):
        """[P, NS, NS] tiles (subjects along rows) of the requested relation columns, differentiable."""
        fused = self._fused_training(world)
        if world._pair_num > 0 and min(world._n_list) == max(world._n_list) and not fused:
            return self._relation_tiles_dense(world, low, pred_q_host, world._n_list[0])
        emb = self._embedding_network.linear
        dev = world._device
        full = self._relation_full_columns(low.cols)
        pq = np.asarray(list(pred_q_host), np.int64)
        P, NS = len(pq), world._NS
        n = np.asarray(world._n_list, np.int64)
        pair_off = np.concatenate([[0], np.cumsum(n * (n - 1))])
        flat = torch.full((P * NS * NS,), -30.0, dtype=torch.float32, device=dev)
        if world._pair_num == 0:
            return flat.view(P, NS, NS)
        preds = np.nonzero(full >= 0)[0]
        if len(preds) == 0 or (n[pq[preds]] * (n[pq[preds]] - 1)).sum() == 0:
            return flat.view(P, NS, NS)
        groups = [preds]
        keep_pred = None
        if fused and len(preds) < P and P == world._batch_size and np.array_equal(pq, np.arange(P)):
            # no-op tokens (questions whose program has no operator at this step - the normal case of a batch of programs of differing lengths):
            # the fused logit kernels want every pair row of the batch under exactly one predicate, in order.  So the idle questions ride
            # along under a borrowed concept - their logits are computed and dropped, their rows carry a zero gradient (what the kernels'
            # contract calls "a row without a gradient") - instead of sending this reader through tensor ops: a gather of the [pairs, HID2]
            # hidden layer, a second evaluation of the trunk to hang its gradient on and torch's atomic index_select backward (34.7 ms per step
            # at 256 questions x 100 objects with 1..3 hops, against 9.4 for the aligned program: tools/lab/time_train_ragged_hops.py)
            keep_pred = np.zeros(P, bool)
            keep_pred[preds] = True
            full = full.copy()
            full[~keep_pred] = full[preds[0]]
            groups = [np.arange(P)]
            _lib.note("idle_questions_ride_along")
        if fused and len(preds) > world._batch_size:
            # several predicates per question (choose_rel's options): the j-th predicates of all questions form a group in which every
            # pair row belongs to exactly one predicate, in order - the shape the fused logit kernels take (no gathers of the
            # [pairs, HID2] hidden layer, no atomic scatter-adds in the backward)
            slot, seen = np.zeros(len(preds), np.int64), {}
            for i, qq in enumerate(pq[preds]):
                slot[i] = seen.get(int(qq), 0)
                seen[int(qq)] = slot[i] + 1
            groups = [preds[slot == j] for j in range(int(slot.max()) + 1)]
        for grp in groups:
            dst, val = self._relation_group_autograd(world, full, pq, grp, n, pair_off, NS, fused)
            if keep_pred is not None:                             # (the idle questions' rows: computed, not used)
                rows = np.repeat(keep_pred[grp], n[pq[grp]] * (n[pq[grp]] - 1))
                if not rows.all():
                    sel = upload(np.nonzero(rows)[0].astype(np.int64), dev)
                    dst, val = dst.index_select(0, sel), val.index_select(0, sel)
            flat = flat.index_put((dst,), val)
        return flat.view(P, NS, NS)

    def _relation_group_autograd(self, world, full, pq, preds, n, pair_off, NS, fused):
        """LogSigmoid logits of the predicates `preds` over the ordered pairs of their images and their positions in the flat tile tensor."""
        emb = self._embedding_network.linear
        dev = world._device
        q = pq[preds]
        cnt = n[q] * (n[q] - 1)
        key = ("rel", str(dev), tuple(world._n_list), NS, preds.tobytes(), q.tobytes())
        hit = self._index_cache.get(key)
        if hit is None:                                       # gather / scatter indices depend on the batch shape only: upload once
            rep = np.repeat(np.arange(len(preds)), cnt)
            k = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            nq = n[q][rep]
            s_, o_ = k // (nq - 1), k % (nq - 1)
            o_ = o_ + (o_ >= s_)                              # pairs are row-major in s with the diagonal left out (util.py:87-103)
            src = pair_off[q][rep] + k
            identity = len(src) == world._pair_num and np.array_equal(src, np.arange(len(src)))
            hit = (None if identity else torch.as_tensor(src).to(dev), torch.as_tensor(preds[rep] * (NS * NS) + s_ * NS + o_).to(dev),
                   torch.as_tensor(rep).to(dev), torch.as_tensor(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)).to(dev), int(cnt.max()),
                   torch.as_tensor(rep.astype(np.int32)).to(dev) if identity else None, bool((cnt[cnt > 0] >= 64).all()))
            self._index_cache[key] = hit
        src, dst, rep, pred_off, max_rows, rep32, sums_ok = hit
        cols = upload(full[preds].astype(np.int64), dev)
        if fused and src is None:
            # every pair row belongs to exactly one predicate, in order: Sigmoid, embedding product and row sum in one kernel
            plan = _concept_plan(full[preds], dev, self._index_cache)
            e_rows, be_rows = _EmbRows.apply(emb.weight, emb.bias, cols, plan)
            fresh = getattr(world, "_pair_pre2", None) is None        # this reader creates the trunk: its logits come out of the trunk's epilogue
            # The pair branch (U | V product, the pair MLP, this reader's logits) on a SIDE stream: autograd runs a node's backward on the stream
            # its forward ran on, so the branch's backward - the step's three large kernels - runs beside the attribute branch's backward (a chain
            # of ~60 small launches with most CUs idle) instead of in front of it.  Same kernels, same arithmetic, no shared accumulators (the
            # two branches' concept rows are different rows of the embedding gradient); joined right here in the forward.
            side = _pair_side_stream(dev)
            cur = torch.cuda.current_stream(dev) if side is not None else None
            if side is not None:
                side.wait_stream(cur)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                own = np.asarray(full[preds], np.int64)
                extras = []
                if fresh and len(q) == world._batch_size and np.array_equal(q, np.arange(len(q))):
                    extras = [c for c in getattr(world, "_train_rel_readers", []) if not np.array_equal(c, own)]
                pre2 = self._pair_pre2_autograd(world, first=(e_rows.detach().contiguous(), rep32, q, extras, emb.weight.detach()))
                head = getattr(world, "_pair_head", None)
                _lib.note("emb_rows")
                if head is not None:                      # the deferred backward: no [pairs, HID2] gradient between this layer and the trunk
                    x_part = None
                    if head[2] is not None and head[2].numel() > 0 and not getattr(world, "_pair_x_rows", False):
                        x_part = head[2] if fresh else None      # (the tall product's epilogue: the first reader's partial sums)
                    elif head[2] is not None and head[2].numel() > 0:
                        if fresh:
                            world._pair_x_slots = {own.tobytes(): 0, **{c.tobytes(): 1 + i for i, c in enumerate(extras)}} if head[2].shape[0] == 1 + len(extras) else {own.tobytes(): 0}
                            x_part = head[2][0:1]
                        else:                             # a later reader whose logits the trunk's one launch left behind
                            slot = getattr(world, "_pair_x_slots", {}).get(own.tobytes())
                            if slot is not None and slot < head[2].shape[0] and len(q) == world._batch_size:
                                x_part = head[2][slot:slot + 1]
                                _lib.note("head_use_logits_from_trunk")
                    _lib.note("head_use")
                    if len(q) == world._batch_size and np.array_equal(q, np.arange(len(q))):
                        # (a reader with one predicate per question, in order: all such readers map the pair rows to predicates the same way -
                        # what lets the trunk take their dZ shares in one pass, _HeadUse.backward)
                        head[1].setdefault("allq", set()).add(rep32.data_ptr())
                    x = _HeadUse.apply(head[0], pre2, e_rows, be_rows, pred_off, rep32, max_rows, head[1], x_part, sums_ok)
                else:
                    _lib.note("fused_logit")
                    x = _FusedLogit.apply(pre2, e_rows, be_rows, pred_off, max_rows)
            if side is not None:
                cur.wait_stream(side)
                x.record_stream(cur)                      # (allocated on the side stream, read on this one)
                if torch.is_tensor(pre2):
                    pre2.record_stream(cur)
                _lib.note("pair_branch_side_stream")
        else:
            if fused:
                _lib.note("logit_rows_gathered")          # rows out of order / no-op tokens: the fused first layer, torch ops for this reader's logits
            elif torch.device(world._device).type == "cuda":
                _lib.fallback("relation network training (logit layer: torch ops)", self._why_not_fused(world))
            # one pass over all predicates (a loop over concepts would scatter-add into the hidden gradient once per concept; a
            # matrix-vector product would go to rocBLAS gemv, whose backward on a [2.5M, 300] operand takes 11 ms)
            h = self._pair_hidden_autograd(world)
            e_rows = emb.weight.index_select(0, cols).index_select(0, rep)
            x = ((h if src is None else h.index_select(0, src)) * e_rows).sum(1) + emb.bias.index_select(0, cols).index_select(0, rep)
        return dst, nn.functional.logsigmoid(x)

    def _attr_ll_autograd(self, world, low, pred_q_host):
        """[P, NS] blocks of the requested attribute columns, differentiable."""
        emb = self._embedding_network.linear
        dev = world._device
        cols = np.asarray(low.cols, np.int64)
        pq = np.asarray(list(pred_q_host), np.int64)
        P, NS = len(pq), world._NS
        if os.environ.get("DFOL_TRAIN_FUSED", "1") != "0" and world._hidden_attr.shape[1] <= 512 and emb.bias is not None and \
                (len(pq) < 2 or bool(np.all(pq[1:] >= pq[:-1]))):
            plan = _concept_plan(cols, dev, self._index_cache)
            _lib.note("attr_ll_fused")
            return _AttrLL.apply(world._hidden_attr, emb.weight, emb.bias, world._obj_off, upload(pq.astype(np.int32), dev),
                                 low.on(dev)[0], NS, plan)
        if torch.device(dev).type == "cuda":
            _lib.fallback("attribute columns in training (gathers + torch ops)",
                          "DFOL_TRAIN_FUSED=0, a hidden width above 512, an embedding layer without bias or an unsorted predicate -> question map")
        n = np.asarray(world._n_list, np.int64)
        obj_off = np.concatenate([[0], np.cumsum(n)])
        flat = torch.full((P * NS,), -30.0, dtype=torch.float32, device=dev)
        preds = np.nonzero(cols >= 0)[0]
        if len(preds) == 0:
            return flat.view(P, NS)
        q = pq[preds]
        cnt = n[q]
        rep = np.repeat(np.arange(len(preds)), cnt)
        k = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        src = torch.as_tensor(obj_off[q][rep] + k).to(dev)
        dst = torch.as_tensor(preds[rep] * NS + k).to(dev)
        col = torch.as_tensor(cols[preds][rep]).to(dev)
        x = (world._hidden_attr.index_select(0, src) * emb.weight.index_select(0, col)).sum(1) + emb.bias.index_select(0, col)
        return flat.index_put((dst,), nn.functional.logsigmoid(x)).view(P, NS)

    def materialize_tables(self, world):
        """Full cached tables exactly as the reference builds them (only when a caller reads them)."""
        if getattr(world, "_shared", False):
            raise L.DfolError("the full cached tables of a shared-scene batch are not built (they are per question in the reference's layout): "
                              "collate with share_scenes=False to read world._attribute_features / _relation_features")
        lazy, world._lazy = world._lazy, None
        try:
            pair = None
            if world._pair_num > 0:
                D = world._obj.shape[1]
                pair = L.pair_features(world._obj, D, world._obj_off, world._pair_off, world._batch_size, max(world._n_list), world._pair_num,
                                       pair_index=world.pair_index)
            a, r = self.compute_all_log_likelihood_2(world._obj, pair)
            world._attr_table, world._rel_table = a, {'features': r, 'index': None}
        finally:
            world._lazy = lazy

    def _relation_full_columns(self, cols333):
        idx = np.asarray(self._ontology._relation_index, np.int32)
        c = np.asarray(cols333, np.int64)
        return np.where(c >= 0, idx[np.maximum(c, 0)], -1).astype(np.int32)

    def _launch_pairs(self, world, req_col, req_tile, tiles, req_orient=None):
        _, _, wg, hid1, D = self._split_first_layer()
        w2p, b2, hid2, packed = self._padded_second_layer()
        emb = self._embedding_network.linear
        dev = world._device
        up = lambda a: a if isinstance(a, torch.Tensor) else torch.as_tensor(a).to(dev)
        rc, rt = up(req_col), up(req_tile)
        ro = None if req_orient is None else up(req_orient)
        if isinstance(packed, tuple) and packed[0] == "f16x2":
            L.pair_ll_h2(world._uv, hid1, world._obj[:, D - 4:], wg, packed[1], b2, hid2, emb.weight, emb.bias, world._img_n_obj,
                         world._obj_off, max(world._n_list), rc, rt, ro, tiles, -30.0, uv_prescaled=True)
        elif isinstance(packed, tuple):
            L.pair_ll_split(world._uv, hid1, world._obj[:, D - 4:], wg, packed[1], b2, hid2, emb.weight, emb.bias, world._img_n_obj,
                            world._obj_off, max(world._n_list), rc, rt, ro, tiles, -30.0)
        elif packed is not None:
            L.pair_ll_packed(world._uv, hid1, world._obj[:, D - 4:], wg, packed, b2, hid2, emb.weight, emb.bias, world._img_n_obj,
                             world._obj_off, max(world._n_list), rc, rt, ro, tiles, -30.0)
        else:
            L.pair_ll(world._uv, hid1, world._obj[:, D - 4:], wg, w2p, b2, emb.weight, emb.bias, world._img_n_obj,
                      world._obj_off, max(world._n_list), rc, rt, ro, tiles, -30.0, hid2=hid2)

    def _new_tiles(self, world, count, dtype=torch.float32):
        # the pair kernels write the real ordered pairs (the bf16x3 kernel not even the diagonal): absent everywhere else
        return torch.full((count, world._NS, world._NS), -30.0, dtype=dtype, device=world._device)

    def _collect_relation_readers(self, world, program_batch):
        """Training: the embedding columns every relate / verify_rel of the program batch will ask the pair branch for (one per question; an idle
        question borrows the first valid one, as _relation_tiles_autograd does), so that the fused forward kernel leaves ALL readers' raw logits behind
        in its one launch - a reader that finds its columns there skips its own pass over pre2 (dfol_pair_logit_fwd_f32: 0.58 ms at 256 x 100 objects)."""
        world._train_rel_readers = []
        if os.environ.get("DFOL_TRAIN_ALL_LOGITS", "1") == "0":
            return
        Q = world._batch_size
        for ob in program_batch._op_batch_list:
            if ob._op_name in ("relate", "verify_rel") and ob._arguments:
                low = get_lowered(ob._arguments[0], self._ontology, TokenType.RELATION)
                if not (low.any_valid and len(low.cols) == Q):
                    continue
                full = self._relation_full_columns(low.cols)
                ok = full >= 0
                if not ok.any():
                    continue
                full = full.copy()
                full[~ok] = full[np.nonzero(ok)[0][0]]
                if not any(np.array_equal(full, c) for c in world._train_rel_readers) and len(world._train_rel_readers) < 8:
                    world._train_rel_readers.append(full.astype(np.int64))

    def prefetch_relations(self, world, program_batch, fused=True):
        """One fused pair-kernel launch for every relation operator of the program batch (relate / verify_rel /
        choose_rel): the pair MLP's hidden layer is then evaluated once per object pair, whatever the number of hops."""
        if world._lazy is not None and getattr(world, "_train", False):
            self._collect_relation_readers(world, program_batch)
        if world._lazy is None or getattr(world, "_train", False):
            return
        Q = world._batch_size
        entries = []                                   # (lowered tokens, predicate -> question, per-predicate orientation)
        for ob in program_batch._op_batch_list:
            if not ob._arguments:
                continue
            if ob._op_name in ("relate", "verify_rel"):
                toks = ob._arguments[0]
                low = get_lowered(toks, self._ontology, TokenType.RELATION)
                if getattr(toks, "lowered", None) is None:
                    toks.lowered, toks.lowered_type = low, TokenType.RELATION
                if low.any_valid and len(low.cols) == Q:
                    # the operator keeps the posterior of the freshly selected variable: store every tile with the
                    # OTHER (summed-out) variable along rows, so that posterior is a column reduction
                    # (fused=False: the attentions carry gradients, every relate goes through the generic two-posterior cell,
                    # which reads subject-row tiles)
                    orient = np.asarray([L.TILE_OBJECT_ROWS if (f and fused) else L.TILE_SUBJECT_ROWS for f in ob._arguments[1]], np.uint8)
                    entries.append((low, np.arange(Q), orient, toks))
            elif ob._op_name == "choose_rel":
                flat, batch_index = flatten_list(ob._arguments[0])
                low = lower_tokens(flat, self._ontology, TokenType.RELATION)
                ob._arguments[0].flat_lowered = low    # GQAChooseRelBatch hands it to RelateBatch
                if low.any_valid:
                    entries.append((low, np.asarray(batch_index, np.int64), np.zeros(len(flat), np.uint8), None))
        if not entries:
            return
        total = sum(len(e[0].cols) for e in entries)
        # bf16 storage only when every consumer is the fused single-posterior kernel (relate / verify_rel), which reads it directly
        bf16 = self._tile_dtype == torch.bfloat16 and world._NS % 8 == 0 and self._padded_second_layer()[3] is not None and \
            self._padded_second_layer()[2] > 256 and all(ob._op_name != "choose_rel" for ob in program_batch._op_batch_list)
        if world._shared:
            return self._prefetch_relations_shared(world, program_batch, entries, torch.bfloat16 if bf16 else torch.float32, fused)
        tiles = self._new_tiles(world, total, torch.bfloat16 if bf16 else torch.float32)
        # the request arrays depend on the program batch only: build and upload them once per batch (a pageable upload
        # synchronises the stream), keyed by what else they depend on
        key = (id(self), str(world._device), Q, tuple((e[0].cols.tobytes(), e[0].valid.tobytes(), np.asarray(e[1]).tobytes(), e[2].tobytes()) for e in entries))
        plan = getattr(program_batch, "_dfol_rel_plan", None)
        if plan is None or plan[0] != key:
            rows_col, rows_tile, rows_orient, invalid, base = [], [], [], [], 0
            for low, pq, orient, _ in entries:
                P = len(pq)
                pq = np.asarray(pq, np.int64)
                slot = np.zeros(P, np.int64)               # j-th predicate of its question, in predicate order
                if P > 1 and not (P == Q and pq[0] == 0 and pq[-1] == Q - 1 and (np.diff(pq) == 1).all()):       # (one predicate per question: all zeros)
                    order = np.argsort(pq, kind="stable")
                    sq = pq[order]
                    start = np.flatnonzero(np.concatenate([[True], sq[1:] != sq[:-1]]))
                    slot[order] = np.arange(P) - np.repeat(start, np.diff(np.concatenate([start, [P]])))
                K = int(slot.max()) + 1 if P else 1
                col = np.full((K, Q), -1, np.int32)
                til = np.zeros((K, Q), np.int32)
                ori = np.zeros((K, Q), np.uint8)
                col[slot, pq] = self._relation_full_columns(low.cols)
                til[slot, pq] = base + np.arange(P, dtype=np.int32)
                ori[slot, pq] = orient
                rows_col.append(col), rows_tile.append(til), rows_orient.append(ori)
                if not low.all_valid:
                    invalid.append(base + np.nonzero(low.valid == 0)[0])
                base += P
            dev = world._device
            # (through the pinned staging ring: a pageable torch.as_tensor(...).to(dev) makes the host wait for the stream, four times per batch)
            plan = (key, upload(np.concatenate(rows_col), dev), upload(np.concatenate(rows_tile), dev), upload(np.concatenate(rows_orient), dev),
                    upload(np.concatenate(invalid), dev) if invalid else None)
            program_batch._dfol_rel_plan = plan
        _, req_col, req_tile, req_orient, invalid = plan
        base = 0
        for low, pq, orient, toks in entries:
            P = len(pq)
            self._remember_tiles(world, low, toks, (tiles[base:base + P], orient, fused))
            base += P
        self._launch_pairs(world, req_col, req_tile, tiles, req_orient)

    def _shared_requests(self, world, items):
        """Shared scenes: the distinct (scene, relation column, orientation) triples among `items` = [(full columns, predicate ->
        question, orientation)], as pair-kernel request arrays [K', scenes] over the IMAGE-level geometry, plus, per item, the index of
        every predicate's tile among the distinct ones (`U` = the extra all-absent tile for no-op tokens)."""
        from . import native_plan
        return native_plan.shared_requests(world._q_img, len(world._img_n_list), items)     # (one definition: the native executor's lowering builds the same arrays)

    def _prefetch_relations_shared(self, world, program_batch, entries, dtype, fused):
        """prefetch_relations for a batch whose questions share scenes: one tile per distinct (scene, concept, orientation) from the pair
        kernel, then every operator's per-predicate tiles are row gathers of those (40 KB per predicate, against 0.16 MFLOP per object
        pair for computing a tile again)."""
        dev = world._device
        key = (id(self), str(dev), world._q_img.tobytes(), tuple((e[0].cols.tobytes(), np.asarray(e[1]).tobytes(), e[2].tobytes()) for e in entries))
        plan = getattr(program_batch, "_dfol_rel_plan", None)
        if plan is None or plan[0] != key:
            U, col, til, ori, maps = self._shared_requests(world, [(self._relation_full_columns(e[0].cols), np.asarray(e[1], np.int64), e[2]) for e in entries])
            up = lambda a: torch.as_tensor(a).to(dev)
            plan = (key, U, up(col), up(til), up(ori), [up(m) for m in maps])
            program_batch._dfol_rel_plan = plan
        _, U, req_col, req_tile, req_orient, maps = plan
        tiles = torch.full((U + 1, world._NS, world._NS), -30.0, dtype=dtype, device=dev)      # tile U: all absent (no-op tokens)
        if U:
            self._launch_pairs(world, req_col, req_tile, tiles, req_orient)
        for (low, pq, orient, toks), m in zip(entries, maps):
            self._remember_tiles(world, low, toks, (tiles.index_select(0, m), orient, fused))

    def prefetch_attributes(self, world, program_batch):
        """One attribute-column launch for the simple attribute token lists of the program batch (select names, filter attributes,
        relate's object names: one token per question) instead of one per operator; the operators pick their block up by the identity
        of the lowered token list, which is memoised by content."""
        if world._lazy is None or getattr(world, "_train", False):
            return
        Q = world._batch_size
        lows = []
        for ob in program_batch._op_batch_list:
            at = {"select": 0, "filter": 0, "relate": 2}.get(ob._op_name)
            if at is None or not ob._arguments or at >= len(ob._arguments) or ob._arguments[at] is None:
                continue
            toks = ob._arguments[at]
            try:
                low = get_lowered(toks, self._ontology, TokenType.ATTRIBUTE)
            except Exception:                                   # an unknown token: let the operator raise where the reference does
                return
            if len(low.cols) == Q and low.any_valid and all(low is not other for other in lows):
                lows.append(low)
        if len(lows) < 2:
            return
        dev = world._device
        emb = self._embedding_network.linear
        cols = upload(np.concatenate([low.cols for low in lows]), dev)
        pred_img = upload(np.tile(world._q_img.astype(np.int32), len(lows)), dev)          # predicate -> scene (= question without sharing)
        ll = L.attr_ll(world._hidden_attr, emb.weight, emb.bias, world._obj_off, pred_img, cols, world._NS, -30.0)
        world._attr_blocks = {id(low): ll[i * Q:(i + 1) * Q] for i, low in enumerate(lows)}

    @staticmethod
    def _remember_tiles(world, low, toks, entry):
        """Prefetched tiles are found again by the OPERATOR's own token list (two relate operators of a batch may name the same relations -
        lowered token lists are memoised by content, so they then share one `low` - with different subject flags, i.e. other orientations:
        keyed by `low` alone the first would read the second's tiles) and, for readers that only have the lowered list, by `low`."""
        if toks is not None:
            world._rel_tiles[("op", id(toks))] = entry
        world._rel_tiles[id(low)] = entry

    def oriented_tiles(self, world, low, tokens=None):
        """Prefetched tiles of a relate operator, each stored with its summed-out variable along rows (or None)."""
        if world._lazy is None:
            return None
        hit = world._rel_tiles.get(("op", id(tokens))) if tokens is not None else None
        if hit is None:
            hit = world._rel_tiles.get(id(low))
        return None if (hit is None or not hit[2]) else hit[0]

    def _relation_tiles_now(self, world, low, pred_q_host):
        """Relation tiles for one token list outside the prefetch (e.g. choose_rel's flattened option list)."""
        pq = np.asarray(list(pred_q_host), np.int64)
        P, Q = len(pq), world._batch_size
        if world._shared:
            U, col, til, ori, maps = self._shared_requests(world, [(self._relation_full_columns(low.cols), pq, np.zeros(P, np.uint8))])
            tiles = torch.full((U + 1, world._NS, world._NS), -30.0, dtype=torch.float32, device=world._device)
            if U:
                self._launch_pairs(world, col, til, tiles, ori)
            return tiles.index_select(0, upload(maps[0], world._device))
        slot = np.zeros(P, np.int64)                      # j-th predicate of its question
        seen = {}
        for p, q in enumerate(pq):
            slot[p] = seen.get(int(q), 0)
            seen[int(q)] = slot[p] + 1
        K = int(slot.max()) + 1
        req_col = np.full((K, Q), -1, np.int32)
        req_tile = np.zeros((K, Q), np.int32)
        full = self._relation_full_columns(low.cols)
        req_col[slot, pq] = full
        req_tile[slot, pq] = np.arange(P, dtype=np.int32)
        tiles = self._new_tiles(world, P)                 # (no-op tokens request nothing: their tiles stay absent)
        self._launch_pairs(world, req_col, req_tile, tiles)
        return tiles

    # ---- a4 / a5: per-predicate blocks (classifier_oracle.py:44-137) -------------------------------------
    def block_likelihood(self, token_type, low, pred_q, pred_q_host, world, default_log_likelihood=-30,
                         normalized_probability=True, orientation=L.TILE_SUBJECT_ROWS):
        if not self._cached:
            raise NotImplementedError("only the cached-table oracle of the reference's experiments is built")
        dev = world._device
        cols, _, _ = low.on(dev)
        if world._lazy is not None and float(default_log_likelihood) == -30.0:
            return self._block_likelihood_needed(token_type, low, cols, pred_q, pred_q_host, world, normalized_probability, orientation)
        if token_type == TokenType.ATTRIBUTE:
            gather = lambda c, pq: L.attr_gather(world._attribute_features, world._obj_off, pq, c, world._NS,
                                                 float(default_log_likelihood))
        else:
            table = world._relation_features['features']
            if table is None:                                   # no image has two objects
                table = torch.zeros(1, 1, dtype=torch.float32, device=dev)
            gather = lambda c, pq: L.rel_gather(table, world._pair_off, world._n_obj, pq, c, world._NS, orientation,
                                                float(default_log_likelihood))
        if not (self._normalize and normalized_probability):
            return gather(cols, pred_q)
        valid = low.valid.astype(bool)
        seg = segments_of(np.asarray(pred_q_host)[valid])        # clusters of the compressed list (:23, :72, :124)
        if len(seg) - 1 == int(valid.sum()):                     # all singletons: cluster_map is None (:27-28)
            return gather(cols, pred_q)
        if low.all_valid:
            ll = gather(cols, pred_q)
            return L.option_normalize_(ll, upload(seg, dev), pred_q, world._n_obj, world._NS)
        # no-op tokens inside an option list: normalise the compressed list, then put default blocks back
        keep = upload(np.nonzero(valid)[0], dev)
        pq_c = pred_q.index_select(0, keep).contiguous()
        ll_c = gather(cols.index_select(0, keep).contiguous(), pq_c)
        ll_c = L.option_normalize_(ll_c, upload(seg, dev), pq_c, world._n_obj, world._NS)
        ll = torch.full((len(low.cols),) + tuple(ll_c.shape[1:]), float(default_log_likelihood), dtype=torch.float32, device=dev)
        ll.index_copy_(0, keep, ll_c)
        return ll

    def _block_likelihood_needed(self, token_type, low, cols, pred_q, pred_q_host, world, normalized_probability, orientation):
        """The same blocks as the cached-table gathers, computed from the hidden activations for the requested columns only."""
        dev = world._device
        emb = self._embedding_network.linear
        if getattr(world, "_train", False):
            assert orientation == L.TILE_SUBJECT_ROWS
            ll = self._attr_ll_autograd(world, low, pred_q_host) if token_type == TokenType.ATTRIBUTE \
                else self._relation_tiles_autograd(world, low, pred_q_host)
        elif token_type == TokenType.ATTRIBUTE:
            ll = getattr(world, "_attr_blocks", {}).get(id(low)) if pred_q is world._ident else None      # prefetch_attributes
            if ll is None:
                ll = L.attr_ll(world._hidden_attr, emb.weight, emb.bias, world._obj_off, world.pred_img(pred_q), cols, world._NS, -30.0)
        else:
            assert orientation == L.TILE_SUBJECT_ROWS
            hit = world._rel_tiles.get(id(low))
            if hit is not None and not hit[1].any():          # prefetched, and every tile already has subjects along rows
                ll = hit[0].float() if hit[0].dtype != torch.float32 else \
                    (hit[0].clone() if (self._normalize and normalized_probability) else hit[0])
            else:
                ll = self._relation_tiles_now(world, low, pred_q_host)
        if not (self._normalize and normalized_probability):
            return ll
        valid = low.valid.astype(bool)
        seg = segments_of(np.asarray(list(pred_q_host))[valid])
        if len(seg) - 1 == int(valid.sum()):
            return ll
        if low.all_valid:
            return L.option_normalize_(ll, upload(seg, dev), pred_q, world._n_obj, world._NS)
        keep = upload(np.nonzero(valid)[0], dev)
        pq_c = pred_q.index_select(0, keep).contiguous()
        ll_c = ll.index_select(0, keep).contiguous()
        ll_c = L.option_normalize_(ll_c, upload(seg, dev), pq_c, world._n_obj, world._NS)
        ll.index_copy_(0, keep, ll_c)
        return ll
