"""The head's backward with the sums from their own pass against the sums from the weight-gradient pass (256 x 100 objects)."""
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
p2 = torch.randn(M, H2, device=dev, generator=g) * 2
z = torch.nn.functional.elu(torch.randn(M, H1, device=dev, generator=g))
w2 = torch.randn(H2, H1, device=dev, generator=g) / 16
E = torch.randn(Q, H2, device=dev, generator=g) * 0.1
dx = torch.randn(M, device=dev, generator=g) * 1e-3
pred_off = torch.arange(Q + 1, device=dev, dtype=torch.int64) * per
rep = torch.arange(Q, device=dev, dtype=torch.int32).repeat_interleave(per)
names = ["dfol_pair_logit_bwd_sums_f32", "dfol_pair_dz_tall_f32", "dfol_pair_wgrad_fused_f32", "dfol_pair_wgrad_fused_sums_f32"]
for sums in (False, True):
    for _ in range(2):
        _lib.pair_head_bwd(dx, p2, z, w2, E, pred_off, rep, sums=sums)
    _lib._timed = {k: [] for k in names}
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        _lib.pair_head_bwd(dx, p2, z, w2, E, pred_off, rep, sums=sums)
    e.record()
    torch.cuda.synchronize()
    print("sums fused" if sums else "sums apart", "%.3f ms per backward: " % (s.elapsed_time(e) / 5),
          "  ".join("%s %.3f" % (k.replace("dfol_pair_", "").replace("_f32", ""), min(a.elapsed_time(b) for a, b in v)) for k, v in _lib._timed.items() if v))
    _lib._timed = None
