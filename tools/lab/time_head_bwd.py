"""Times the pair MLP head's backward at the north-star train shape (256 images x 100 objects: 2,534,400 pair rows, HID1 = 256,
HID2 = 300): the materialised route (logit_bwd -> dpre2 -> input-gradient product, weight gradient) against the three kernels that
rebuild dpre2 on the fly.  HIP events around each entry point, 10 launches each."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dfol_vqa_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
Q, n, H1, H2 = 256, int(os.environ.get("N_OBJ", "100")), 256, 300
per = n * (n - 1)
M = Q * per
g = torch.Generator(device=dev).manual_seed(1)
p2 = torch.randn(M, H2, device=dev, generator=g) * 2
z = torch.nn.functional.elu(torch.randn(M, H1, device=dev, generator=g))
w2 = torch.randn(H2, H1, device=dev, generator=g) / 16
E = torch.randn(Q, H2, device=dev, generator=g) * 0.1
dx = torch.randn(M, device=dev, generator=g) * 1e-3
pred_off = torch.arange(Q + 1, device=dev, dtype=torch.int64) * per
rep = torch.arange(Q, device=dev, dtype=torch.int32).repeat_interleave(per)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


names = ["dfol_pair_logit_bwd_f32", "dfol_linear_act_split_f32", "dfol_linear_wgrad_bias_f32", "dfol_pair_logit_bwd_sums_f32", "dfol_pair_dz_fused_f32",
         "dfol_pair_wgrad_fused_f32"]


def materialised():
    dp2, de, dbe = _lib.pair_logit_bwd(dx, p2, E, pred_off)
    dz = _lib.linear_act_split(dp2, w2, None, _lib.ACT_NONE, transpose_w=True)
    dw, db = _lib.linear_wgrad(dp2, z, bias=True)
    return dz, dw, db, de, dbe


def fused():
    return _lib.pair_head_bwd(dx, p2, z, w2, E, pred_off, rep)


for name, fn in (("materialised", materialised), ("fused", fused)):
    total = timed(fn)
    _lib._timed = {k: [] for k in names}
    fn()
    torch.cuda.synchronize()
    parts = {k: v[0][0].elapsed_time(v[0][1]) for k, v in _lib._timed.items() if v}
    _lib._timed = None
    print("%-13s %.3f ms  " % (name, total) + "  ".join("%s %.3f" % (k.replace("dfol_", "").replace("_f32", ""), v) for k, v in parts.items()))
a, b = materialised(), fused()
for tag, x, y in zip(("dz", "dw", "db2", "de", "dbe"), a, b):
    print(tag, "max |a - b| / max |a| = %.3g" % float((x - y).abs().max() / x.abs().max()))
