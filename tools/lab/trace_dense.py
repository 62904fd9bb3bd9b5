"""clock64 stamps of one workgroup of the tall split-kernel products (build with -DDFOL_DENSE_TRACE: tools/lab/build_variant.sh
dtrace dfol_dense_split.hip -DDFOL_DENSE_TRACE; DFOL_LIB=build/lib_dtrace.so).  Slots: 60 start, 4 ks + {0 store, 1 stored, 2 past the
barrier, 3 MFMAs done}, 61 loop done, 62 epilogue done."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dfol_vqa_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
M, H1, H2 = 256 * 9900, 256, 300
g = torch.Generator(device=dev).manual_seed(1)
z = torch.randn(M, H1, device=dev, generator=g)
w2 = torch.randn(H2, H1, device=dev, generator=g) / 16
p2 = torch.randn(M, H2, device=dev, generator=g)
E = torch.randn(256, H2, device=dev, generator=g) * 0.1
dx = torch.randn(M, device=dev, generator=g) * 1e-3
rep = torch.arange(256, device=dev, dtype=torch.int32).repeat_interleave(9900)
pred_off = torch.arange(257, device=dev, dtype=torch.int64) * 9900
lib = _lib.load()
buf = (ctypes.c_longlong * 256)()


def show(tag, ksteps):
    torch.cuda.synchronize()
    lib.dfol_dense_trace_read(buf)
    for w in range(4):
        t = [buf[w * 64 + i] for i in range(64)]
        t0 = t[60]
        steps = ["%d/%d/%d" % (t[4 * k + 1] - t[4 * k], t[4 * k + 2] - t[4 * k + 1], t[4 * k + 3] - t[4 * k + 2]) for k in range(ksteps)]
        print("%s wave %d: prologue->step0 %d | store/barrier/mfma per step: %s | between steps %s | loop %d epilogue %d" % (
            tag, w, t[0] - t0, " ".join(steps), " ".join(str(t[4 * (k + 1)] - t[4 * k + 3]) for k in range(ksteps - 1)), t[61] - t0, t[62] - t[61]))


for _ in range(2):
    y = _lib.linear_act_split(z, w2, None, _lib.ACT_NONE)
show("forward z W2^T (K = 256)", 8)
for _ in range(2):
    out = _lib.pair_head_bwd(dx, p2, z, w2, E, pred_off, rep)
show("dz fused (K = 300)", 10)
