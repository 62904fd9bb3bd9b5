"""clock64 stamps of the persistent pair kernel (build csrc with EXTRA=-DDFOL_PAIR_TRACE): per task and half, the tick lengths."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from dfol_vqa_amd import _lib as L
torch.manual_seed(0)
Q, N, HID1, HID2, C, K = 256, 100, 256, 300, 333, 1
O = Q * N
dev = 'cuda'
uv = torch.randn(O, 2 * HID1, device=dev) * 0.5
pos = torch.rand(O, 4, device=dev) * 0.5 + 0.05
wg = torch.randn(HID1, 4, device=dev) * 0.3
w2 = torch.zeros(320, HID1, device=dev); w2[:HID2] = torch.randn(HID2, HID1, device=dev) / 16
b2 = torch.randn(HID2, device=dev); E = torch.randn(C, HID2, device=dev) / 17; be = torch.randn(C, device=dev)
n_o = torch.full((Q,), N, dtype=torch.int32, device=dev); off = (torch.arange(Q + 1, device=dev) * N).to(torch.int32)
req_col = torch.randint(0, C, (K, Q), dtype=torch.int32, device=dev); req_tile = torch.arange(K * Q, dtype=torch.int32, device=dev).view(K, Q)
tiles = torch.empty(K * Q, 104, 104, device=dev)
w2s = L.pair_pack_w2_split(w2, HID2)
for _ in range(3):
    L.pair_ll_split(uv, HID1, pos, wg, w2s, b2, HID2, E, be, n_o, off, N, req_col, req_tile, None, tiles)
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_longlong * (8 * 8 * 64))()
lib.dfol_pair_trace_read.argtypes = [ctypes.c_void_p]
print("rc", lib.dfol_pair_trace_read(buf))
t = np.array(buf[:], dtype=np.int64).reshape(8, 8, 64)
for blk in range(2):
    base = t[blk, 0, 0]
    for w in (0, 4):
        x = t[blk, w]
        if x[0] == 0: continue
        for ti in range(3):
            b = 20 * ti
            if x[b] == 0: continue
            print("block %d wave %d task %d: start @%7d  setup %5d  build0 %5d  wait %5d  mult0 %5d  tail %5d" % (
                blk, w, ti, x[b] - base, (x[b + 17] - x[b]) if ti else 0, x[b + 18] - (x[b + 17] if ti else x[b]), x[b + 1] - x[b + 18], x[b + 2] - x[b + 1],
                (x[b + 19] - x[b + 2]) if ti else 0))
            prev = x[b + 19] if ti else x[b + 2]
            row = []
            for c in range(1, 8):
                row.append("c%d: build+wait %5d mult %5d" % (c, x[b + 1 + 2 * c] - prev, x[b + 2 + 2 * c] - x[b + 1 + 2 * c]))
                prev = x[b + 2 + 2 * c]
            print("      " + " | ".join(row))
