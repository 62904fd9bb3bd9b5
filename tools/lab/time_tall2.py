"""Tiled against persistent form of the train step's two tall products (DFOL_TALL=0 / 1), HIP events, best of 5."""
import os
import sys

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
b2 = torch.randn(H2, device=dev, generator=g)
E = torch.randn(Q, H2, device=dev, generator=g) * 0.1
dx = torch.randn(M, device=dev, generator=g) * 1e-3
pred_off = torch.arange(Q + 1, device=dev, dtype=torch.int64) * per
rep = torch.arange(Q, device=dev, dtype=torch.int32).repeat_interleave(per)
names = ["dfol_linear_act_h2_f32", "dfol_linear_logit_h2_f32", "dfol_linear_tall_h2_f32", "dfol_pair_dz_fused_f32", "dfol_pair_dz_tall_f32"]


def run():
    _lib.linear_act_split(z, w2, b2, _lib.ACT_NONE)
    _lib.linear_logit_h2(z, w2, b2, rep, E)
    _lib.linear_tall_h2(z, w2, b2)
    _lib.linear_tall_h2(z, w2, b2, rep, E)
    for t in ("0", "1"):
        os.environ["DFOL_TALL"] = t
        _lib.pair_head_products(dx, p2, z, w2, E, pred_off, rep, need_dw=False)


run()
_lib._timed = {k: [] for k in names}
for _ in range(5):
    run()
torch.cuda.synchronize()
for k, v in _lib._timed.items():
    ts = [a.elapsed_time(b) for a, b in v]
    per_call = len(ts) // 5
    print(k, " ".join("%.3f" % min(ts[i::per_call]) for i in range(per_call)))
