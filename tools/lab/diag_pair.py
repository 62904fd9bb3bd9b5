"""Where do two launches of the fp16x2 pair kernel differ?  (lab; DFOL_LIB selects the build)"""
import sys, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from dfol_vqa_amd import _lib as L
torch.manual_seed(0)
HID1, HID2, C, K = 256, 300, 333, 1
dev = 'cuda'
n_list = [100] * 256
Q, O, N = len(n_list), sum(n_list), max(n_list)
NS = 100
uv = torch.randn(O, 2 * HID1, device=dev) * 0.5
pos = torch.rand(O, 4, device=dev) * 0.5 + 0.05
wg = torch.randn(HID1, 4, device=dev) * 0.3
w2 = torch.zeros(320, HID1, device=dev); w2[:HID2] = torch.randn(HID2, HID1, device=dev) / 16
b2 = torch.randn(HID2, device=dev); E = torch.randn(C, HID2, device=dev) / 17; be = torch.randn(C, device=dev)
n_o = torch.tensor(n_list, dtype=torch.int32, device=dev)
off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.tensor(n_list).cumsum(0)]).to(torch.int32).to(dev)
req_col = torch.randint(0, C, (K, Q), dtype=torch.int32, device=dev); req_tile = torch.arange(K * Q, dtype=torch.int32, device=dev).view(K, Q)
w2s = L.pair_pack_w2_h2(w2, HID2)
outs = []
for _ in range(12):
    tiles = torch.full((K * Q, NS, NS), -30.0, device=dev)
    L.pair_ll_h2(uv, HID1, pos, wg, w2s, b2, HID2, E, be, n_o, off, N, req_col, req_tile, None, tiles)
    outs.append(tiles.cpu().numpy())
ref = outs[-1]
for i, o in enumerate(outs[:-1]):
    d = o != ref
    if not d.any():
        continue
    t, r, c = np.nonzero(d)
    # ordered-pair slot of (r, c) in its image: e = r * (N - 1) + (c - (c > r)); workgroup = e // 256, wave = (e % 256) // 32, slot in wave = e % 32
    e = r * (N - 1) + (c - (c > r))
    print("launch %d: %d elements differ in %d tiles; max |diff| %.3g" % (i, d.sum(), len(set(t)), np.abs(o - ref).max()))
    wg_ids = sorted(set(zip(t.tolist(), (e // 256).tolist())))
    print("   workgroups (tile, wg): %d distinct; first %s" % (len(wg_ids), wg_ids[:6]))
    waves = sorted(set(((e % 256) // 32).tolist()))
    print("   waves hit: %s; slots-in-wave hit: %s" % (waves, sorted(set((e % 32).tolist()))[:40]))
    per_wg = {}
    for tt, ee in zip(t.tolist(), e.tolist()):
        per_wg.setdefault((tt, ee // 256), []).append(ee % 256)
    k0 = wg_ids[0]
    print("   in workgroup %s: slots %s" % (k0, sorted(per_wg[k0])))
