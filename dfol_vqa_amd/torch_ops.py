"""The core entry points as PyTorch custom operators: torch.ops.dfol.{filter_fwd, relate_fwd, relate_one_fwd, quantify_fwd, linear_act,
pair_ll} (+ their backward operators), registered with torch.library.custom_op / register_fake / register_autograd around the C-ABI calls.

SURVEY.md 8(b) asks for ops the dispatcher can see (`TORCH_LIBRARY(dfol, ...)` + `register_autograd`); the reference's dispatch site they
slot under is `self._ops[name](...)` in batch_gqa_interpreter.py:72-78.  The kernels stay behind the C ABI (include/dfol_vqa.h) - no torch
types cross it - and these registrations are the PyTorch-side plumbing: schema, fake-tensor (shape) functions so the ops trace under
torch.compile / AOT autograd, and autograd formulas whose backward launches the HIP backward kernels (which are operators themselves, so
the backward traces as well).  `dfol_vqa_amd.ops` routes every gradient-carrying call through these operators; inference under
torch.no_grad() keeps calling `_lib` directly (a custom-op dispatch costs ~10 us per launch, which an eager 13-launch step would feel).
`tests/test_kernels_gpu.py::test_torch_custom_ops_opcheck` runs torch.library.opcheck on all six.
"""

import os
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib

_op = torch.library.custom_op


# ---- filter (batch_base_ops.py:62-151, arity 1) ----------------------------------------------------------------------------------------
@_op("dfol::filter_fwd", mutates_args=())
def filter_fwd(att_in: Tensor, ll: Tensor, pred_q: Tensor, n_obj: Tensor, neg: Optional[Tensor], active: Optional[Tensor]) -> Tensor:
    return _lib.filter_fwd(att_in, ll, pred_q, n_obj, neg, active)


@filter_fwd.register_fake
def _(att_in, ll, pred_q, n_obj, neg, active):
    return torch.empty_like(ll, memory_format=torch.contiguous_format)


@_op("dfol::filter_bwd", mutates_args=())
def filter_bwd(g_out: Tensor, ll: Tensor, pred_q: Tensor, n_obj: Tensor, neg: Optional[Tensor], active: Optional[Tensor], Q: int,
               need_prior: bool, need_ll: bool) -> Tuple[Tensor, Tensor]:
    """A gradient that is not needed comes back as an empty tensor (an operator cannot return None)."""
    g_prior, g_ll = _lib.filter_bwd(g_out, ll, pred_q, n_obj, neg, active, Q, need_prior, need_ll)
    return (g_prior if need_prior else ll.new_empty(0)), (g_ll if need_ll else ll.new_empty(0))


@filter_bwd.register_fake
def _(g_out, ll, pred_q, n_obj, neg, active, Q, need_prior, need_ll):
    return (ll.new_empty(Q, ll.shape[1]) if need_prior else ll.new_empty(0)), \
        (torch.empty_like(ll, memory_format=torch.contiguous_format) if need_ll else ll.new_empty(0))


def _filter_setup(ctx, inputs, output):
    att_in, ll, pred_q, n_obj, neg, active = inputs
    ctx.save_for_backward(ll, pred_q, n_obj, neg, active)
    ctx.Q = att_in.shape[0]


def _filter_backward(ctx, g):
    ll, pred_q, n_obj, neg, active = ctx.saved_tensors
    need_prior, need_ll = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    g_prior, g_ll = torch.ops.dfol.filter_bwd(g.contiguous(), ll, pred_q, n_obj, neg, active, ctx.Q, need_prior, need_ll)
    return (g_prior if need_prior else None), (g_ll if need_ll else None), None, None, None, None


filter_fwd.register_autograd(_filter_backward, setup_context=_filter_setup)


# ---- relate, both posteriors (batch_base_ops.py:62-151, arity 2) ------------------------------------------------------------------------
@_op("dfol::relate_fwd", mutates_args=())
def relate_fwd(prior_s: Tensor, prior_o: Tensor, tile: Tensor, pred_q: Tensor, n_obj: Tensor, quant_s: Tensor, quant_o: Tensor,
               neg: Optional[Tensor], active: Optional[Tensor], want: Optional[Tensor], orientation: int, lone_forall_identity: bool,
               diag_absent: bool) -> Tuple[Tensor, Tensor]:
    ps, po = _lib.relate_fwd(prior_s, prior_o, tile, pred_q, n_obj, quant_s, quant_o, neg, active, want, orientation, lone_forall_identity,
                             diag_absent=diag_absent)
    return ps, po


@relate_fwd.register_fake
def _(prior_s, prior_o, tile, pred_q, n_obj, quant_s, quant_o, neg, active, want, orientation, lone_forall_identity, diag_absent):
    P, NS = tile.shape[0], tile.shape[1]
    return tile.new_empty(P, NS), tile.new_empty(P, NS)


@_op("dfol::relate_bwd", mutates_args=())
def relate_bwd(prior_s: Tensor, prior_o: Tensor, tile: Tensor, pred_q: Tensor, n_obj: Tensor, quant_s: Tensor, quant_o: Tensor,
               neg: Optional[Tensor], active: Optional[Tensor], g_post_s: Tensor, g_post_o: Tensor, orientation: int,
               lone_forall_identity: bool, need_prior: bool, need_tile: bool) -> Tuple[Tensor, Tensor, Tensor]:
    g_ps, g_po, g_tile = _lib.relate_bwd(prior_s, prior_o, tile, pred_q, n_obj, quant_s, quant_o, neg, active, g_post_s, g_post_o, orientation,
                                         lone_forall_identity, need_prior, need_tile)
    e = tile.new_empty(0)
    return (g_ps if need_prior else e), (g_po if need_prior else tile.new_empty(0)), (g_tile if need_tile else tile.new_empty(0))


@relate_bwd.register_fake
def _(prior_s, prior_o, tile, pred_q, n_obj, quant_s, quant_o, neg, active, g_post_s, g_post_o, orientation, lone_forall_identity, need_prior, need_tile):
    c = torch.contiguous_format
    return (torch.empty_like(prior_s, memory_format=c) if need_prior else tile.new_empty(0)), \
        (torch.empty_like(prior_o, memory_format=c) if need_prior else tile.new_empty(0)), \
        (torch.empty_like(tile, memory_format=c) if need_tile else tile.new_empty(0))


def _relate_setup(ctx, inputs, output):
    prior_s, prior_o, tile, pred_q, n_obj, quant_s, quant_o, neg, active, want, orientation, lone, diag_absent = inputs
    ctx.save_for_backward(prior_s, prior_o, tile, pred_q, n_obj, quant_s, quant_o, neg, active, want)
    ctx.orientation, ctx.lone = orientation, lone


def _relate_backward(ctx, gs, go):
    prior_s, prior_o, tile, pred_q, n_obj, quant_s, quant_o, neg, active, want = ctx.saved_tensors
    gs, go = gs.contiguous(), go.contiguous()
    if want is not None:                                     # a posterior the forward did not produce carries no gradient
        gs = gs * ((want & 1) > 0).to(gs.dtype).unsqueeze(1)
        go = go * ((want & 2) > 0).to(go.dtype).unsqueeze(1)
    need_prior, need_tile = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1]), bool(ctx.needs_input_grad[2])
    g_ps, g_po, g_tile = torch.ops.dfol.relate_bwd(prior_s, prior_o, tile, pred_q, n_obj, quant_s, quant_o, neg, active, gs, go, ctx.orientation, ctx.lone,
                                                   need_prior, need_tile)
    return (g_ps if ctx.needs_input_grad[0] else None), (g_po if ctx.needs_input_grad[1] else None), (g_tile if need_tile else None), \
        None, None, None, None, None, None, None, None, None, None


relate_fwd.register_autograd(_relate_backward, setup_context=_relate_setup)


# ---- relate, the one posterior GQARelateBatch keeps (batch_gqa_ops.py:364-371); inference only ------------------------------------------
@_op("dfol::relate_one_fwd", mutates_args=())
def relate_one_fwd(x_att: Tensor, prev_att: Tensor, tile: Tensor, pred_q: Tensor, n_obj: Tensor, quant_prev: Tensor, neg: Optional[Tensor],
                   active: Optional[Tensor], lone_forall_identity: bool) -> Tensor:
    fn = _lib.relate_one_fwd_bf16 if tile.dtype == torch.bfloat16 else _lib.relate_one_fwd
    return fn(x_att, prev_att, tile, pred_q, n_obj, quant_prev, neg, active, lone_forall_identity)


@relate_one_fwd.register_fake
def _(x_att, prev_att, tile, pred_q, n_obj, quant_prev, neg, active, lone_forall_identity):
    return x_att.new_empty(tile.shape[0], tile.shape[1])


# ---- quantifier aggregation (batch_base_types.py:103-125) -----------------------------------------------------------------------------
@_op("dfol::quantify_fwd", mutates_args=())
def quantify_fwd(att: Tensor, quant: Tensor, pred_q: Tensor, n_obj: Tensor) -> Tensor:
    return _lib.quantify_fwd(att, quant, pred_q, n_obj)


@quantify_fwd.register_fake
def _(att, quant, pred_q, n_obj):
    return att.new_empty(att.shape[0])


@_op("dfol::quantify_bwd", mutates_args=())
def quantify_bwd(g_lp: Tensor, att: Tensor, quant: Tensor, pred_q: Tensor, n_obj: Tensor) -> Tensor:
    return _lib.quantify_bwd(g_lp, att, quant, pred_q, n_obj)


@quantify_bwd.register_fake
def _(g_lp, att, quant, pred_q, n_obj):
    return torch.empty_like(att, memory_format=torch.contiguous_format)


def _quantify_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _quantify_backward(ctx, g):
    att, quant, pred_q, n_obj = ctx.saved_tensors
    return torch.ops.dfol.quantify_bwd(g.contiguous(), att, quant, pred_q, n_obj), None, None, None


quantify_fwd.register_autograd(_quantify_backward, setup_context=_quantify_setup)


# ---- dense layer with fused activation (gqa_interpreter_experiments.py:18-77) ------------------------------------------------------------
@_op("dfol::linear_act", mutates_args=())
def linear_act(x: Tensor, weight: Tensor, bias: Optional[Tensor], act: int) -> Tensor:
    return _lib.linear_act(x, weight, bias, act)


@linear_act.register_fake
def _(x, weight, bias, act):
    return x.new_empty(x.shape[0], weight.shape[0])


@_op("dfol::linear_gradx", mutates_args=())
def linear_gradx(dz: Tensor, weight: Tensor) -> Tensor:
    return _lib.linear_gradx(dz, weight)


@linear_gradx.register_fake
def _(dz, weight):
    return dz.new_empty(dz.shape[0], weight.shape[1])


@_op("dfol::linear_wgrad", mutates_args=())
def linear_wgrad(dz: Tensor, x: Tensor) -> Tuple[Tensor, Tensor]:
    dw, db = _lib.linear_wgrad(dz, x if x.stride(-1) == 1 else x.contiguous(), bias=True)
    return dw, db


@linear_wgrad.register_fake
def _(dz, x):
    return dz.new_empty(dz.shape[1], x.shape[1]), dz.new_empty(dz.shape[1])


@_op("dfol::act_bwd", mutates_args=())
def act_bwd(g: Tensor, y: Tensor, act: int) -> Tensor:
    return _lib.act_bwd(g, y, act)


@act_bwd.register_fake
def _(g, y, act):
    return torch.empty_like(y)


def _linear_setup(ctx, inputs, output):
    x, weight, bias, act = inputs
    ctx.save_for_backward(x, weight, output)
    ctx.act, ctx.has_bias = act, bias is not None
    ctx.math = _lib._dense_math()                            # the backward products run in the forward's arithmetic


def _linear_backward(ctx, g):
    x, weight, y = ctx.saved_tensors
    if ctx.act != _lib.ACT_NONE and g.is_cuda and g.dtype == torch.float32 and y.is_contiguous() and g.shape == y.shape and \
            os.environ.get("DFOL_ACT_BWD", "hip") != "torch":
        dz = torch.ops.dfol.act_bwd(g, y, ctx.act)           # (the same formulas as below, one launch instead of three to five)
    elif ctx.act == _lib.ACT_SIGMOID:
        dz = g * y * (1 - y)
    elif ctx.act == _lib.ACT_ELU:
        dz = g * torch.where(y > 0, torch.ones_like(y), y + 1)
    elif ctx.act == _lib.ACT_LOGSIGMOID:
        dz = g * (1 - torch.exp(y))
    else:
        dz = g
    dz = dz.contiguous()
    gx = gw = gb = None
    with _lib.dense_math(ctx.math):
        if ctx.needs_input_grad[0]:
            gx = torch.ops.dfol.linear_gradx(dz, weight.detach())
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = torch.ops.dfol.linear_wgrad(dz, x)      # (the bias gradient comes from the same pass over dz)
    return gx, (gw if ctx.needs_input_grad[1] else None), (gb if ctx.has_bias and ctx.needs_input_grad[2] else None), None


linear_act.register_autograd(_linear_backward, setup_context=_linear_setup)


# ---- fused pair MLP -> requested relation tiles (classifier_oracle.py:145-156 for the needed columns); inference only -------------------
@_op("dfol::pair_ll", mutates_args=("tiles",))
def pair_ll(uv: Tensor, hid1: int, pos: Tensor, wg: Tensor, w2: Tensor, b2: Tensor, hid2: int, emb_w: Tensor, emb_b: Optional[Tensor],
            n_obj: Tensor, obj_off: Tensor, max_n: int, req_col: Tensor, req_tile: Tensor, req_orient: Optional[Tensor], tiles: Tensor,
            default_ll: float, packing: int) -> None:
    """packing: 0 = w2 is the padded [rows, HID1] matrix (dfol_pair_ll_f32), 1 = dfol_pair_pack_w2_f32's image, 2 = dfol_pair_pack_w2_bf16x3's."""
    if packing == 2:
        _lib.pair_ll_split(uv, hid1, pos, wg, w2, b2, hid2, emb_w, emb_b, n_obj, obj_off, max_n, req_col, req_tile, req_orient, tiles, default_ll)
    elif packing == 1:
        _lib.pair_ll_packed(uv, hid1, pos, wg, w2, b2, hid2, emb_w, emb_b, n_obj, obj_off, max_n, req_col, req_tile, req_orient, tiles, default_ll)
    else:
        _lib.pair_ll(uv, hid1, pos, wg, w2, b2, emb_w, emb_b, n_obj, obj_off, max_n, req_col, req_tile, req_orient, tiles, default_ll, hid2=hid2)


@pair_ll.register_fake
def _(uv, hid1, pos, wg, w2, b2, hid2, emb_w, emb_b, n_obj, obj_off, max_n, req_col, req_tile, req_orient, tiles, default_ll, packing):
    return None


CORE_OPS = ("filter_fwd", "relate_fwd", "relate_one_fwd", "quantify_fwd", "linear_act", "pair_ll")
