"""Autograd-aware front of the C-ABI wrappers.

Every function here has the signature of its namesake in `_lib`.  Under `torch.no_grad()` (inference) the call
goes straight to the kernel; when a floating-point input requires grad, the core entry points (filter, relate, quantify,
linear_act) go through the registered PyTorch operators `torch.ops.dfol.*` (torch_ops.py: custom_op + register_autograd +
register_fake), the small glue through a `torch.autograd.Function`; either way the backward launches the HIP backward
kernels (`csrc/dfol_logic_bwd.hip`).  The backward of the tiny per-question
vector glue (gate, segment reductions, and/or/not, compare) and of the dense layers is written with torch tensor ops /
library GEMMs on the GPU — plumbing, as the design notes say.
"""

import os

import torch

from . import _lib
from . import torch_ops  # noqa: F401  (registers torch.ops.dfol.*)
from ._lib import *  # noqa: F401,F403  (constants, DfolError, non-differentiable wrappers)
from ._lib import DfolError  # noqa: F401

_EPS = 1e-20


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _dpnot(x, alpha):
    """d/dx log(max(alpha + (1 - 2 alpha) e^x, eps)); zero where the clamp is active."""
    e = torch.exp(x)
    d = alpha + (1 - 2 * alpha) * e
    return torch.where(d > _EPS, (1 - 2 * alpha) * e / d.clamp_min(_EPS), torch.zeros_like(d))


# ---- filter ------------------------------------------------------------------------------------------
def filter_fwd(att_in, ll, pred_q, n_obj, neg=None, active=None):
    if _needs_grad(att_in, ll):
        return torch.ops.dfol.filter_fwd(att_in, ll, pred_q, n_obj, neg, active)
    return _lib.filter_fwd(att_in, ll, pred_q, n_obj, neg, active)


# ---- relate ------------------------------------------------------------------------------------------
def relate_fwd(prior_s, prior_o, tile, pred_q, n_obj, quant_s, quant_o, neg=None, active=None, want=None,
               orientation=_lib.TILE_SUBJECT_ROWS, lone_forall_identity=False, need_s=True, need_o=True, diag_absent=False):
    if _needs_grad(prior_s, prior_o, tile):
        return torch.ops.dfol.relate_fwd(prior_s, prior_o, tile, pred_q, n_obj, quant_s, quant_o, neg, active, want, int(orientation),
                                         bool(lone_forall_identity), bool(diag_absent))
    return _lib.relate_fwd(prior_s, prior_o, tile, pred_q, n_obj, quant_s, quant_o, neg, active, want, orientation, lone_forall_identity,
                           need_s, need_o, diag_absent)


# ---- quantify ----------------------------------------------------------------------------------------
def quantify_fwd(att, quant, pred_q, n_obj):
    if _needs_grad(att):
        return torch.ops.dfol.quantify_fwd(att, quant, pred_q, n_obj)
    return _lib.quantify_fwd(att, quant, pred_q, n_obj)


def quantify_hard(att, quant, pred_q, n_obj, total_obj):
    """hard_mode aggregation; the reference only uses it when answering (give_answer and hard_mode), so it carries no gradient."""
    return _lib.quantify_hard(att.detach(), quant, pred_q, n_obj, total_obj)


# ---- gathers and option normalisation ----------------------------------------------------------------
class _AttrGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, obj_off, pred_q, pred_col, NS, default_ll):
        ctx.save_for_backward(obj_off, pred_q, pred_col)
        ctx.shape = tuple(table.shape)
        return _lib.attr_gather(table, obj_off, pred_q, pred_col, NS, default_ll)

    @staticmethod
    def backward(ctx, g):
        obj_off, pred_q, pred_col = ctx.saved_tensors
        return _lib.attr_gather_bwd(g.contiguous(), obj_off, pred_q, pred_col, ctx.shape), None, None, None, None, None


def attr_gather(table, obj_off, pred_q, pred_col, NS, default_ll=-30.0):
    if _needs_grad(table):
        return _AttrGather.apply(table.contiguous(), obj_off, pred_q, pred_col, NS, default_ll)
    return _lib.attr_gather(table, obj_off, pred_q, pred_col, NS, default_ll)


class _RelGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, pair_off, n_obj, pred_q, pred_col, NS, orientation, default_ll):
        ctx.save_for_backward(pair_off, n_obj, pred_q, pred_col)
        ctx.meta = (tuple(table.shape), orientation)
        return _lib.rel_gather(table, pair_off, n_obj, pred_q, pred_col, NS, orientation, default_ll)

    @staticmethod
    def backward(ctx, g):
        pair_off, n_obj, pred_q, pred_col = ctx.saved_tensors
        shape, orientation = ctx.meta
        return _lib.rel_gather_bwd(g.contiguous(), pair_off, n_obj, pred_q, pred_col, orientation, shape), None, None, None, None, None, None, None


def rel_gather(table, pair_off, n_obj, pred_q, pred_col, NS, orientation=_lib.TILE_SUBJECT_ROWS, default_ll=-30.0):
    if _needs_grad(table):
        return _RelGather.apply(table.contiguous(), pair_off, n_obj, pred_q, pred_col, NS, orientation, default_ll)
    return _lib.rel_gather(table, pair_off, n_obj, pred_q, pred_col, NS, orientation, default_ll)


class _OptionNormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ll, seg_off, pred_q, n_obj, NS):
        y = _lib.option_normalize_(ll.clone(), seg_off, pred_q, n_obj, NS)
        ctx.save_for_backward(y, seg_off, pred_q, n_obj)
        ctx.NS = NS
        return y

    @staticmethod
    def backward(ctx, g):
        y, seg_off, pred_q, n_obj = ctx.saved_tensors
        return _lib.option_normalize_bwd(g.contiguous(), y, seg_off, pred_q, n_obj, ctx.NS), None, None, None, None


def option_normalize_(ll, seg_off, pred_q, n_obj, NS):
    """In place for inference; under autograd the normalised copy is returned (callers use the return value)."""
    if _needs_grad(ll):
        return _OptionNormalize.apply(ll, seg_off, pred_q, n_obj, NS)
    return _lib.option_normalize_(ll, seg_off, pred_q, n_obj, NS)


# ---- small per-question glue: forward = HIP kernel, backward = a few tensor ops ----------------------
class _Gate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_att, y_att, x_quant, y_quant, g):
        ctx.save_for_backward(g)
        att, quant = _lib.gate(x_att, y_att, x_quant, y_quant, g)
        ctx.mark_non_differentiable(quant)
        return att, quant

    @staticmethod
    def backward(ctx, g_att, _g_quant):
        (g,) = ctx.saved_tensors
        sel = (g > 0).unsqueeze(1)
        zero = torch.zeros_like(g_att)
        return torch.where(sel, g_att, zero), torch.where(sel, zero, g_att), None, None, None


def gate(x_att, y_att, x_quant, y_quant, g):
    if _needs_grad(x_att, y_att):
        return _Gate.apply(x_att, y_att, x_quant, y_quant, g)
    return _lib.gate(x_att, y_att, x_quant, y_quant, g)


class _SegmentSumRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, seg_off):
        ctx.save_for_backward(seg_off)
        ctx.P = src.shape[0]
        return _lib.segment_sum_rows(src, seg_off)

    @staticmethod
    def backward(ctx, g):
        (seg_off,) = ctx.saved_tensors
        counts = (seg_off[1:] - seg_off[:-1]).to(torch.int64)
        idx = torch.repeat_interleave(torch.arange(counts.numel(), device=g.device), counts).to(torch.int32)
        return _lib.gather_rows(g.contiguous(), idx), None


def segment_sum_rows(src, seg_off):
    if _needs_grad(src):
        return _SegmentSumRows.apply(src, seg_off)
    return _lib.segment_sum_rows(src, seg_off)


class _SegmentOr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lp, seg_off, as_written=False):
        ctx.save_for_backward(lp, seg_off)
        return _lib.segment_or(lp, seg_off, as_written)

    @staticmethod
    def backward(ctx, g):
        lp, seg_off = ctx.saved_tensors
        counts = (seg_off[1:] - seg_off[:-1]).to(torch.int64)
        idx = torch.repeat_interleave(torch.arange(counts.numel(), device=g.device), counts)
        one = torch.ones((), device=lp.device)
        inner = torch.log((1 - torch.exp(lp)).clamp_min(_EPS))
        s = _lib.segment_sum_rows(inner.unsqueeze(1).contiguous(), seg_off).squeeze(1)     # (index_add_ would be atomic: not repeatable)
        return (g * _dpnot(s, one))[idx] * _dpnot(lp, one), None, None


def segment_or(lp, seg_off, as_written=False):
    if _needs_grad(lp):
        return _SegmentOr.apply(lp, seg_off, as_written)
    return _lib.segment_or(lp, seg_off, as_written)


class _Implication(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prior, x, pred_q, n_obj):
        ctx.save_for_backward(prior, x, pred_q, n_obj)
        return _lib.implication(prior, x, pred_q, n_obj)

    @staticmethod
    def backward(ctx, g):
        prior, x, pred_q, n_obj = ctx.saved_tensors
        one = torch.ones((), device=x.device)
        pq = pred_q.to(torch.int64)
        valid = torch.arange(x.shape[1], device=x.device).unsqueeze(0) < n_obj.to(torch.int64)[pq].unsqueeze(1)
        m = torch.log((1 - torch.exp(x)).clamp_min(_EPS))
        z = prior[pq] + m
        dz = torch.where(valid, g * _dpnot(z, one), torch.zeros_like(g))
        g_prior = _lib.reduce_by_question(dz.contiguous(), pred_q, None, prior.shape[0])    # deterministic (no atomics)
        return g_prior, dz * _dpnot(x, one), None, None


def implication(prior, x, pred_q, n_obj):
    if _needs_grad(prior, x):
        return _Implication.apply(prior, x, pred_q, n_obj)
    return _lib.implication(prior, x, pred_q, n_obj)


class _Logic(torch.autograd.Function):
    @staticmethod
    def forward(ctx, op, a, b):
        ctx.op = op
        ctx.save_for_backward(a, b if b is not None else a.new_empty(0))
        return _lib.logic(op, a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        if ctx.op == _lib.LOGIC_AND:
            return None, g, g
        one = torch.ones((), device=a.device)
        if ctx.op == _lib.LOGIC_NOT:
            return None, g * _dpnot(a, one), None
        ea, eb = torch.exp(a), torch.exp(b)
        x = 1 - (1 - ea) * (1 - eb)
        live = x > _EPS
        xs = x.clamp_min(_EPS)
        zero = torch.zeros_like(g)
        return None, torch.where(live, g * ea * (1 - eb) / xs, zero), torch.where(live, g * eb * (1 - ea) / xs, zero)


def logic(op, a, b=None):
    if _needs_grad(a, b):
        return _Logic.apply(op, a, b)
    return _lib.logic(op, a, b)


class _Compare(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lp1, lp2, is_less):
        ctx.save_for_backward(lp1, lp2, is_less)
        return _lib.compare(lp1, lp2, is_less)

    @staticmethod
    def backward(ctx, g):
        lp1, lp2, is_less = ctx.saved_tensors
        st = torch.stack([lp1, lp2], 1)
        ls = torch.log_softmax(st, 1)
        gl = g * _dpnot(ls, is_less.unsqueeze(1))
        gst = gl - torch.exp(ls) * gl.sum(1, keepdim=True)
        return gst[:, 0], gst[:, 1], None


def compare(lp1, lp2, is_less):
    if _needs_grad(lp1, lp2):
        return _Compare.apply(lp1, lp2, is_less)
    return _lib.compare(lp1, lp2, is_less)


# ---- dense layers: forward = fused MFMA kernel, backward = the same kernels + the TN weight-gradient kernel ------
def linear_act(x, weight, bias, act, out=None):
    if out is None and _needs_grad(x, weight, bias):
        return torch.ops.dfol.linear_act(x, weight, bias, int(act))
    return _lib.linear_act(x, weight, bias, act, out)


class _PairFeatures(torch.autograd.Function):
    """[obj_s, obj_o, geometry]: the gradient flows back to the two object rows (the geometry comes from the raw boxes).  Only the
    full-table compatibility dataflow (`_needed_columns = False`) builds the [pairs, 1036] matrix; its backward is torch's atomic
    index_add_ - the one place the deterministic-backward guarantee does not cover."""

    @staticmethod
    def forward(ctx, obj, D, obj_off, pair_off, Q, max_n, pairs, ind_s, ind_o):
        ctx.save_for_backward(ind_s, ind_o)
        ctx.meta = (tuple(obj.shape), D)
        return _lib.pair_features(obj, D, obj_off, pair_off, Q, max_n, pairs)

    @staticmethod
    def backward(ctx, g):
        ind_s, ind_o = ctx.saved_tensors
        shape, D = ctx.meta
        g_obj = torch.zeros(shape, dtype=g.dtype, device=g.device)
        g_obj[:, :D].index_add_(0, ind_s, g[:, :D])
        g_obj[:, :D].index_add_(0, ind_o, g[:, D:2 * D])
        return g_obj, None, None, None, None, None, None, None, None


def pair_features(obj, D, obj_off, pair_off, Q, max_n, pairs, pair_index=None):
    if _needs_grad(obj):
        ind_s, ind_o = pair_index()
        return _PairFeatures.apply(obj, D, obj_off, pair_off, Q, max_n, pairs, ind_s, ind_o)
    return _lib.pair_features(obj, D, obj_off, pair_off, Q, max_n, pairs)


def _modulate_reference(att, mods, valid):
    """apply_modulations (batch_base_types.py:170-179) in tensor ops; used for the backward of the HIP kernel."""
    alpha, beta, c, d = (mods[:, k].unsqueeze(1) * (10.0 if k < 3 else 1.0) for k in range(4))
    slog = lambda x: torch.log(x.clamp_min(_EPS))
    t = alpha * att + slog(c) + slog(d)
    u = beta * slog(1 - torch.exp(att)) + slog(1 - d)
    return torch.where(valid, t - slog(torch.exp(u) + torch.exp(t)), torch.zeros_like(att))


class _Modulate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, att, mods, pred_q, n_obj):
        ctx.save_for_backward(att, mods, pred_q, n_obj)
        return _lib.modulate(att, mods.contiguous(), pred_q, n_obj)

    @staticmethod
    def backward(ctx, g):
        att, mods, pred_q, n_obj = ctx.saved_tensors
        if os.environ.get("DFOL_MODULATE_BWD", "hip") == "torch":      # A/B: autograd through the tensor-op restatement (~70 launches)
            valid = torch.arange(att.shape[1], device=att.device).unsqueeze(0) < n_obj.to(torch.int64)[pred_q.to(torch.int64)].unsqueeze(1)
            with torch.enable_grad():
                a = att.detach().requires_grad_(True)
                m = mods.detach().requires_grad_(True)
                out = _modulate_reference(a, m, valid)
                ga, gm = torch.autograd.grad(out, (a, m), g, allow_unused=True)
            return ga, gm, None, None
        ga, gm = _lib.modulate_bwd(g.contiguous(), att.contiguous(), mods.contiguous(), pred_q, n_obj)      # one launch, deterministic
        return ga, gm, None, None


def modulate(att, mods, pred_q, n_obj):
    if _needs_grad(att, mods):
        return _Modulate.apply(att, mods, pred_q, n_obj)
    return _lib.modulate(att, mods.contiguous(), pred_q, n_obj)
