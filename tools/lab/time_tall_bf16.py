"""bf16 mode: tiled against persistent form of the train step's two tall products over bf16-stored activations (256 x 100 objects)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dfol_vqa_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
Q, n, H1, H2 = 256, 100, 256, 300
per = n * (n - 1)
M = Q * per
g = torch.Generator(device=dev).manual_seed(1)
p2 = (torch.randn(M, H2, device=dev, generator=g) * 2).to(torch.bfloat16)
z = torch.nn.functional.elu(torch.randn(M, H1, device=dev, generator=g)).to(torch.bfloat16)
w2 = torch.randn(H2, H1, device=dev, generator=g) / 16
b2 = torch.randn(H2, device=dev, generator=g)
E = torch.randn(Q, H2, device=dev, generator=g) * 0.1
dx = torch.randn(M, device=dev, generator=g) * 1e-3
pred_off = torch.arange(Q + 1, device=dev, dtype=torch.int64) * per
rep = torch.arange(Q, device=dev, dtype=torch.int32).repeat_interleave(per)
names = ["dfol_linear_act_bf16_bf16", "dfol_pair_logit_fwd_bf16", "dfol_linear_tall_bf16_bf16", "dfol_pair_logit_bwd_bf16", "dfol_pair_dz_tall_bf16"]


def run():
    with _lib.dense_math("bf16"):
        y = _lib.linear_act_split(z, w2, b2, _lib.ACT_NONE)
        _lib.pair_logit_fwd(y, E, None, pred_off, per)
        _lib.linear_tall_h2(z, w2, b2)
        _lib.linear_tall_h2(z, w2, b2, rep, E)
        dp2, _, _ = _lib.pair_logit_bwd(dx, p2, E, pred_off)
        _lib.linear_act_split(dp2, w2, None, _lib.ACT_NONE, transpose_w=True)
        _lib.pair_dz_tall_bf16(dx, p2, E, rep, w2)


run()
_lib._timed = {k: [] for k in names}
for _ in range(5):
    run()
torch.cuda.synchronize()
for k, v in _lib._timed.items():
    ts = [a.elapsed_time(b) for a, b in v]
    per_call = len(ts) // 5
    print(k, " ".join("%.3f" % min(ts[i::per_call]) for i in range(per_call)))
