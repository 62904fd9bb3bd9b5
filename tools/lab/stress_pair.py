"""Repeatability stress of the fp16x2 pair kernel for a library build: N launches at the bench shape and on a ragged batch, every result
compared bit for bit with the first.  usage: DFOL_LIB=build/lib_x.so python tools/lab/stress_pair.py [launches]"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from dfol_vqa_amd import _lib as L
n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.manual_seed(0)
HID1, HID2, C = 256, 300, 333
dev = 'cuda'
for name, n_list, K in (("bench 256x100", [100] * 256, 2), ("ragged small", [7, 1, 13, 2, 30, 5] * 40, 3), ("ragged mid", [36, 20, 33, 64, 17] * 50, 1)):
    Q, O, N = len(n_list), sum(n_list), max(n_list)
    NS = (N + 3) // 4 * 4
    uv = torch.randn(O, 2 * HID1, device=dev) * 0.5
    pos = torch.rand(O, 4, device=dev) * 0.5 + 0.05
    wg = torch.randn(HID1, 4, device=dev) * 0.3
    w2 = torch.zeros(320, HID1, device=dev); w2[:HID2] = torch.randn(HID2, HID1, device=dev) / 16
    b2 = torch.randn(HID2, device=dev); E = torch.randn(C, HID2, device=dev) / 17; be = torch.randn(C, device=dev)
    n_o = torch.tensor(n_list, dtype=torch.int32, device=dev)
    off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.tensor(n_list).cumsum(0)]).to(torch.int32).to(dev)
    req_col = torch.randint(0, C, (K, Q), dtype=torch.int32, device=dev); req_tile = torch.arange(K * Q, dtype=torch.int32, device=dev).view(K, Q)
    w2s = L.pair_pack_w2_h2(w2, HID2)
    first, bad, worst, outs = None, 0, 0.0, []
    for _ in range(n_rep):
        tiles = torch.full((K * Q, NS, NS), -30.0, device=dev)
        L.pair_ll_h2(uv, HID1, pos, wg, w2s, b2, HID2, E, be, n_o, off, N, req_col, req_tile, None, tiles)
        outs.append(tiles[:8].clone())
        if first is None:
            first = tiles.clone()
        elif not torch.equal(first, tiles):
            bad += 1
            worst = max(worst, float((first - tiles).abs().max()))
    vs_last = sum(int(not torch.equal(o, outs[-1])) for o in outs[:-1])
    print("%-14s %d launches: %d differ from the first (max |diff| %.3g), %d from the last (first 8 tiles)%s" % (
        name, n_rep, bad, worst, vs_last, "" if torch.isfinite(first).all() else "  NON-FINITE VALUES"))
