#!/usr/bin/env python3
"""Micro-benchmarks of the HBM-bound logic kernels and the MFMA GEMM (run on the GPU box).

    python tools/bench_kernels.py [--preds 65536] [--n 100]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfol_vqa_amd import _lib as L  # noqa: E402

HBM_PEAK = 8.0e12
F32_MFMA_PEAK = 157.3e12


def timeit(fn, iters=20, warmup=10):                  # (3 warm-up launches left the first row 8 % slow: clocks not ramped)
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preds", type=int, default=65536)
    ap.add_argument("--n", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    P, N = args.preds, args.n
    NS = (N + 3) // 4 * 4
    out = []
    g = torch.Generator(device=dev).manual_seed(0)
    p = torch.rand(P, NS, NS, device=dev, generator=g)
    tile = torch.log(torch.where(torch.rand(P, NS, NS, device=dev, generator=g) < 0.1, 0.5 + 0.5 * p, 0.05 * p).clamp_min(1e-5))
    del p
    tile.diagonal(dim1=1, dim2=2).fill_(-30.0)              # self-relations hold the absent value, as the oracle writes them
    prior = torch.log(torch.rand(P, NS, device=dev, generator=g).clamp_min(1e-3)) * 0.3
    pq = torch.arange(P, dtype=torch.int32, device=dev)
    n_obj = torch.full((P,), N, dtype=torch.int32, device=dev)
    ones = torch.ones(P, device=dev)
    for label, ns, no, bytes_per in (("relate_both", True, True, 4 * N * N + 16 * N), ("relate_one_colsum", False, True, 4 * N * N + 12 * N),
                                     ("relate_one_rowsum", True, False, 4 * N * N + 12 * N)):
        for da in (True, False):
            t = timeit(lambda: L.relate_fwd(prior, prior, tile, pq, n_obj, ones, ones, need_s=ns, need_o=no, diag_absent=da))
            out.append({"kernel": label + ("" if da else " (general path)"), "P": P, "N": N, "ms": t * 1e3, "GBps": bytes_per * P / t / 1e9,
                        "frac_hbm_peak": bytes_per * P / t / HBM_PEAK})
    t = timeit(lambda: L.relate_one_fwd(prior, prior, tile, pq, n_obj, ones))
    b1 = 4 * N * N + 12 * N
    out.append({"kernel": "relate_one", "P": P, "N": N, "ms": t * 1e3, "GBps": b1 * P / t / 1e9, "frac_hbm_peak": b1 * P / t / HBM_PEAK})
    # the other predicate kinds (negated, FOR_ALL): what used to be the transcendental-bound "general path"
    zeros = torch.zeros(P, device=dev)
    negs = torch.ones(P, dtype=torch.uint8, device=dev)
    for label, q, ng in (("negated EXISTS", ones, negs), ("FOR_ALL", zeros, None), ("negated FOR_ALL", zeros, negs)):
        t = timeit(lambda: L.relate_one_fwd(prior, prior, tile, pq, n_obj, q, ng))
        out.append({"kernel": "relate_one (%s)" % label, "P": P, "N": N, "ms": t * 1e3, "GBps": b1 * P / t / 1e9, "frac_hbm_peak": b1 * P / t / HBM_PEAK})
        b2 = 4 * N * N + 16 * N
        t = timeit(lambda: L.relate_fwd(prior, prior, tile, pq, n_obj, q, q, ng, diag_absent=True))
        out.append({"kernel": "relate_both (%s)" % label, "P": P, "N": N, "ms": t * 1e3, "GBps": b2 * P / t / 1e9, "frac_hbm_peak": b2 * P / t / HBM_PEAK})
    # BASELINE configs[4]: 256-object tiles, fp32 vs bf16 storage (algorithmic bytes 4N^2 + 12N vs 2N^2 + 12N per predicate)
    if N == 100:
        P2, N2 = max(P // 8, 256), 256
        g2 = torch.Generator(device=dev).manual_seed(1)
        t32 = torch.log(torch.where(torch.rand(P2, N2, N2, device=dev, generator=g2) < 0.1, 0.5 + 0.5 * torch.rand(P2, N2, N2, device=dev, generator=g2),
                                    0.05 * torch.rand(P2, N2, N2, device=dev, generator=g2)).clamp_min(1e-5))
        t32.diagonal(dim1=1, dim2=2).fill_(-30.0)
        pr2 = torch.log(torch.rand(P2, N2, device=dev, generator=g2).clamp_min(1e-3)) * 0.3
        pq2 = torch.arange(P2, dtype=torch.int32, device=dev)
        no2 = torch.full((P2,), N2, dtype=torch.int32, device=dev)
        on2 = torch.ones(P2, device=dev)
        t = timeit(lambda: L.relate_one_fwd(pr2, pr2, t32, pq2, no2, on2))
        b = 4 * N2 * N2 + 12 * N2
        out.append({"kernel": "relate_one (fp32 tiles)", "P": P2, "N": N2, "ms": t * 1e3, "GBps": b * P2 / t / 1e9, "frac_hbm_peak": b * P2 / t / HBM_PEAK})
        ref = L.relate_one_fwd(pr2, pr2, t32, pq2, no2, on2)
        t16 = t32.to(torch.bfloat16)
        del t32
        t = timeit(lambda: L.relate_one_fwd_bf16(pr2, pr2, t16, pq2, no2, on2))
        b = 2 * N2 * N2 + 12 * N2
        got = L.relate_one_fwd_bf16(pr2, pr2, t16, pq2, no2, on2)
        out.append({"kernel": "relate_one (bf16 tiles)", "P": P2, "N": N2, "ms": t * 1e3, "GBps": b * P2 / t / 1e9, "frac_hbm_peak": b * P2 / t / HBM_PEAK,
                    "max_abs_diff_vs_fp32_tiles": (got - ref).abs().max().item()})
        del t16
    t = timeit(lambda: tile.sum())
    out.append({"kernel": "torch.sum(tile) (streaming read reference)", "ms": t * 1e3, "GBps": tile.numel() * 4 / t / 1e9})
    dst = torch.empty_like(tile[: P // 2])
    t = timeit(lambda: dst.copy_(tile[: P // 2]))
    out.append({"kernel": "torch copy half (read+write reference)", "ms": t * 1e3, "GBps": 2 * dst.numel() * 4 / t / 1e9})
    del dst
    ll = tile[:, 0, :].contiguous()
    t = timeit(lambda: L.filter_fwd(prior, ll, pq, n_obj))
    out.append({"kernel": "filter", "P": P, "N": N, "ms": t * 1e3, "GBps": 12 * N * P / t / 1e9, "frac_hbm_peak": 12 * N * P / t / HBM_PEAK})
    t = timeit(lambda: L.quantify_fwd(prior, ones, pq, n_obj))
    out.append({"kernel": "quantify", "P": P, "N": N, "ms": t * 1e3, "GBps": (4 * N + 4) * P / t / 1e9})
    del tile, ll
    # the small kernels on HBM-sized inputs (>= 512 MB of blocks, beyond the 256 MiB Infinity Cache): 79 MB of blocks sit in the cache
    PF = 1 << 19
    g3 = torch.Generator(device=dev).manual_seed(3)
    llf = torch.log(torch.rand(PF, NS, device=dev, generator=g3).clamp_min(1e-5))
    prf = torch.log(torch.rand(PF, NS, device=dev, generator=g3).clamp_min(1e-3)) * 0.3
    pqf = torch.arange(PF, dtype=torch.int32, device=dev)
    nof = torch.full((PF,), N, dtype=torch.int32, device=dev)
    t = timeit(lambda: L.filter_fwd(prf, llf, pqf, nof))
    out.append({"kernel": "filter (HBM-sized)", "P": PF, "N": N, "MB": 12 * N * PF / 1e6, "ms": t * 1e3, "GBps": 12 * N * PF / t / 1e9,
                "frac_hbm_peak": 12 * N * PF / t / HBM_PEAK})
    del llf
    PQ = 1 << 21
    prq = torch.log(torch.rand(PQ, NS, device=dev, generator=g3).clamp_min(1e-3)) * 0.3
    pqq = torch.arange(PQ, dtype=torch.int32, device=dev)
    noq = torch.full((PQ,), N, dtype=torch.int32, device=dev)
    onq = torch.ones(PQ, device=dev)
    t = timeit(lambda: L.quantify_fwd(prq, onq, pqq, noq))
    out.append({"kernel": "quantify (HBM-sized)", "P": PQ, "N": N, "MB": (4 * N + 4) * PQ / 1e6, "ms": t * 1e3, "GBps": (4 * N + 4) * PQ / t / 1e9,
                "frac_hbm_peak": (4 * N + 4) * PQ / t / HBM_PEAK})
    del prq, prf, pqq, noq, onq, pqf, nof
    for (M, Nn, K, act) in ((25600, 512, 2048, 1), (25600, 512, 516, 0), (25600, 256, 516, 2), (25600, 300, 256, 1), (9216, 512, 2048, 1), (9216, 512, 516, 0),
                              (9216, 256, 516, 2), (9216, 300, 256, 1), (9216, 2335, 300, 3)):
        x = torch.rand(M, K, device=dev) - 0.5
        w = torch.rand(Nn, K, device=dev) - 0.5
        b = torch.rand(Nn, device=dev)
        y = torch.empty(M, Nn, device=dev)
        t = timeit(lambda: L.linear_act(x, w, b, act, y), iters=10)
        fl = 2.0 * M * Nn * K
        out.append({"kernel": "linear_act" + ("" if os.environ.get("DFOL_DENSE_MATH") == "f32" else " (auto: the split kernel when large)"), "M": M, "N": Nn, "K": K,
                    "act": act, "ms": t * 1e3, "TFLOPs": fl / t / 1e12, "frac_f32_mfma_peak": fl / t / F32_MFMA_PEAK})
        if K % 4 == 0:
            for math, pieces in (("f16x2", 3), ("bf16x3", 6)):
                with L.dense_math(math):
                    t = timeit(lambda: L.linear_act_split(x, w, b, act, y), iters=10)
                out.append({"kernel": "linear_act_split " + math, "M": M, "N": Nn, "K": K, "act": act, "ms": t * 1e3, "TFLOPs": fl / t / 1e12,
                            "frac_16bit_mfma_peak_executed": pieces * fl / t / 2.5e15})
    # the two tall products of a train step's pair MLP (one row per ordered object pair): forward with the Sigmoid, input gradient
    for (M, Nn, K, act) in ((256 * N * (N - 1), 300, 256, 1), (256 * N * (N - 1), 256, 300, 0)):
        x = torch.rand(M, K, device=dev) - 0.5
        w = torch.rand(Nn, K, device=dev) - 0.5
        b = torch.rand(Nn, device=dev)
        y = torch.empty(M, Nn, device=dev)
        fl = 2.0 * M * Nn * K
        for math, pieces in (("f16x2", 3), ("bf16x3", 6)):
            with L.dense_math(math):
                t = timeit(lambda: L.linear_act_split(x, w, b, act, y), iters=5)
            out.append({"kernel": "linear_act_split " + math, "M": M, "N": Nn, "K": K, "act": act, "ms": t * 1e3, "TFLOPs": fl / t / 1e12,
                        "frac_16bit_mfma_peak_executed": pieces * fl / t / 2.5e15})
        del x, y
    # weight-gradient kernel (dW = dY^T X, csrc/dfol_dense_wgrad.hip) against the library's product, training shapes
    for (M, Nn, K) in ((256 * N * (N - 1), 300, 256), (256 * N, 512, 2048), (256 * N, 256, 516), (256 * N, 300, 256)):
        dy = torch.rand(M, Nn, device=dev) - 0.5
        x = torch.rand(M, K, device=dev) - 0.5
        t = timeit(lambda: L.linear_wgrad(dy, x), iters=5)
        tl = timeit(lambda: dy.t() @ x, iters=5)
        fl = 2.0 * M * Nn * K
        out.append({"kernel": "linear_wgrad (TN, bf16x3 unless DFOL_WGRAD_MATH=f32)", "M": M, "N": Nn, "K": K, "ms": t * 1e3, "TFLOPs": fl / t / 1e12,
                    "frac_f32_mfma_peak": fl / t / F32_MFMA_PEAK, "library_ms": tl * 1e3})
        del dy, x
    # fused pair kernel at the bench shape: Q images of N objects, one requested relation column each
    Q, HID1, HID2, C = 256, 256, 300, 2335
    O = Q * N
    uv = torch.rand(O, 2 * HID1, device=dev) - 0.5
    pos = torch.rand(O, 4, device=dev)
    wg = torch.rand(HID1, 4, device=dev) - 0.5
    w2 = torch.zeros(320, HID1, device=dev)
    w2[:HID2] = (torch.rand(HID2, HID1, device=dev) - 0.5) * 0.2
    b2 = torch.rand(HID2, device=dev) - 0.5
    E = (torch.rand(C, HID2, device=dev) - 0.5) * 0.2
    be = torch.rand(C, device=dev) - 2
    n_o = torch.full((Q,), N, dtype=torch.int32, device=dev)
    off = (torch.arange(Q + 1, device=dev) * N).to(torch.int32)
    for K in (1, 2):
        req_col = torch.randint(0, C, (K, Q), dtype=torch.int32, device=dev)
        req_tile = torch.arange(K * Q, dtype=torch.int32, device=dev).reshape(K, Q)
        tiles = torch.full((K * Q, NS, NS), -30.0, device=dev)      # the split / fp16x2 pair kernels write the ordered pairs only: the diagonal keeps the fill
        t = timeit(lambda: L.pair_ll(uv, HID1, pos, wg, w2, b2, E, be, n_o, off, N, req_col, req_tile, None, tiles, hid2=HID2), iters=10)
        fl = 2.0 * Q * N * (N - 1) * (4 * HID1 + HID1 * HID2 + HID2 * K)
        out.append({"kernel": "pair_ll", "Q": Q, "N": N, "K": K, "ms": t * 1e3, "TFLOPs": fl / t / 1e12, "frac_f32_mfma_peak": fl / t / F32_MFMA_PEAK})
        ref = tiles.clone()
        w2p = L.pair_pack_w2(w2, HID2)
        tiles.zero_()
        t = timeit(lambda: L.pair_ll_packed(uv, HID1, pos, wg, w2p, b2, HID2, E, be, n_o, off, N, req_col, req_tile, None, tiles), iters=10)
        out.append({"kernel": "pair_ll_packed", "Q": Q, "N": N, "K": K, "ms": t * 1e3, "TFLOPs": fl / t / 1e12,
                    "frac_f32_mfma_peak": fl / t / F32_MFMA_PEAK, "max_abs_diff_vs_pair_ll": (tiles[:, :N, :N] - ref[:, :N, :N]).abs().max().item()})
        w2s = L.pair_pack_w2_split(w2, HID2)
        tiles.fill_(-30.0)                                  # (this kernel leaves the diagonal to the caller's fill)
        t = timeit(lambda: L.pair_ll_split(uv, HID1, pos, wg, w2s, b2, HID2, E, be, n_o, off, N, req_col, req_tile, None, tiles), iters=10)
        out.append({"kernel": "pair_ll_split", "Q": Q, "N": N, "K": K, "ms": t * 1e3, "TFLOPs": fl / t / 1e12,
                    "frac_bf16_mfma_peak_executed": 6 * fl / t / 2.5e15,
                    "max_abs_diff_vs_pair_ll": (tiles[:, :N, :N] - ref[:, :N, :N]).abs().max().item()})
        w2h = L.pair_pack_w2_h2(w2, HID2)
        tiles.fill_(-30.0)
        t = timeit(lambda: L.pair_ll_h2(uv, HID1, pos, wg, w2h, b2, HID2, E, be, n_o, off, N, req_col, req_tile, None, tiles), iters=10)
        out.append({"kernel": "pair_ll_h2", "Q": Q, "N": N, "K": K, "ms": t * 1e3, "TFLOPs": fl / t / 1e12,
                    "frac_f16_mfma_peak_executed": 3 * fl / t / 2.5e15,
                    "max_abs_diff_vs_pair_ll": (tiles[:, :N, :N] - ref[:, :N, :N]).abs().max().item()})
    for r in out:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
