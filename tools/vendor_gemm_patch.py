"""A/B tooling, NOT part of the product: routes the backward products of the dense layers (input gradient dz @ W, weight gradient dz^T x)
through the vendor BLAS by monkeypatching the two ctypes wrappers, to time this library's kernels against it.  The package itself has
one backend (the HIP kernels); round 3's in-package switch DFOL_TRAIN_GEMM=torch is gone.

    python -c "import tools.vendor_gemm_patch as v; v.install(); import bench; bench.main(['--mode', 'train', '--graph', '0'])"
"""
import torch


def install():
    from dfol_vqa_amd import _lib

    def linear_gradx(dz, weight):
        return (dz.to(weight.dtype) @ weight.detach()).to(dz.dtype)

    def linear_wgrad(dy, x, bias=False):
        rows = dy.shape[0]
        S = 64
        while S > 1 and rows % S:
            S //= 2
        d32, x32 = dy.float(), x.float()
        dw = torch.bmm(d32.view(S, rows // S, -1).transpose(1, 2), x32.view(S, rows // S, -1)).sum(0) if S > 1 and rows else d32.t() @ x32
        return (dw, d32.sum(0)) if bias else dw

    _lib.linear_gradx = linear_gradx
    _lib.linear_wgrad = linear_wgrad
