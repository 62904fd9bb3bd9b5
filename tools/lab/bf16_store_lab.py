#!/usr/bin/env python3
"""Times the bf16-storage kernels of the pair MLP's train step at 256 x 100 objects (DFOL_LIB selects a library variant)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dfol_vqa_amd import _lib

DEV = torch.device("cuda", 0)
BF = torch.bfloat16
Q, n, H1, H2 = 256, int(os.environ.get("LAB_N", "100")), 256, 300
pairs = Q * n * (n - 1)
g = torch.Generator(device=DEV).manual_seed(0)


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


n_obj = torch.full((Q,), n, dtype=torch.int32, device=DEV)
obj_off = (torch.arange(Q, device=DEV, dtype=torch.int32) * n)
pair_off = (torch.arange(Q, device=DEV, dtype=torch.int64) * n * (n - 1))
pred_off = torch.cat([pair_off, torch.tensor([pairs], device=DEV, dtype=torch.int64)])
U = torch.randn(Q * n, H1, device=DEV, generator=g)
V = torch.randn(Q * n, H1, device=DEV, generator=g)
pos = torch.rand(Q * n, 4, device=DEV, generator=g)
Wg = torch.randn(H1, 4, device=DEV, generator=g) * 0.3
W2 = torch.randn(H2, H1, device=DEV, generator=g) / 16
b2 = torch.randn(H2, device=DEV, generator=g)
E = torch.randn(Q, H2, device=DEV, generator=g)
be = torch.randn(Q, device=DEV, generator=g)
dx = torch.randn(pairs, device=DEV, generator=g)
which = sys.argv[1:] or ["all"]
on = lambda k: "all" in which or k in which
with _lib.dense_math("bf16"):
    z, geo = _lib.pair_hidden1_fwd(U, V, pos, Wg, obj_off, pair_off, n_obj, n, pairs, store=BF)
    p2 = _lib.linear_act_split(z, W2, b2, 0)
    dp2, _, _ = _lib.pair_logit_bwd(dx, p2, E, pred_off)
    dz = _lib.linear_act_split(dp2, W2, None, 0, transpose_w=True)
    rows = []
    if on("h1f"): rows.append(("hidden1_fwd", timed(lambda: _lib.pair_hidden1_fwd(U, V, pos, Wg, obj_off, pair_off, n_obj, n, pairs, store=BF)), pairs * H1 * 2))
    if on("gemm"): rows.append(("gemm fwd Z->pre2", timed(lambda: _lib.linear_act_split(z, W2, b2, 0)), pairs * (H1 + H2) * 2))
    if on("lgf"): rows.append(("logit_fwd", timed(lambda: _lib.pair_logit_fwd(p2, E, be, pred_off, n * (n - 1))), pairs * H2 * 2))
    if on("lgb"): rows.append(("logit_bwd", timed(lambda: _lib.pair_logit_bwd(dx, p2, E, pred_off)), pairs * H2 * 4))
    if on("gemm"): rows.append(("gemm dpre2->dZ", timed(lambda: _lib.linear_act_split(dp2, W2, None, 0, transpose_w=True)), pairs * (H1 + H2) * 2))
    if on("wgrad"): rows.append(("wgrad", timed(lambda: _lib.linear_wgrad(dp2, z, bias=True)), pairs * (H1 + H2) * 2))
    if on("h1b"): rows.append(("hidden1_bwd", timed(lambda: _lib.pair_hidden1_bwd(dz, z, geo, obj_off, pair_off, n_obj, n, Q * n)), pairs * H1 * 4))
for name, ms, nbytes in rows:
    print("%-18s %.3f ms  %.2f TB/s (algorithmic %.2f GB)" % (name, ms, nbytes / ms / 1e9, nbytes / 1e9))
print("sum %.3f ms   lib %s" % (sum(r[1] for r in rows), os.environ.get("DFOL_LIB", "default")))
