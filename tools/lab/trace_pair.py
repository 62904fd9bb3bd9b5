"""Reads the clock64 stamps of a -DDFOL_PAIR_TRACE build of the fp16x2 pair kernel (csrc/dfol_pair_h2.hip) at the bench shape.
usage: DFOL_LIB=build/lib_h2_trace.so python tools/lab/trace_pair.py"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
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
tiles = torch.full((K * Q, 104, 104), -30.0, device=dev)
w2s = L.pair_pack_w2_h2(w2, HID2)
for _ in range(3):
    L.pair_ll_h2(uv, HID1, pos, wg, w2s, b2, HID2, E, be, n_o, off, N, req_col, req_tile, None, tiles)
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_longlong * (8 * 8 * 64))()
lib.dfol_pair_h2_trace_read.argtypes = [ctypes.c_void_p]
print("rc", lib.dfol_pair_h2_trace_read(buf))
t = np.array(buf[:], dtype=np.int64).reshape(8, 8, 64)
for blk in range(3):
    base = t[blk, 0, 0]
    for w in ((0, 4) if blk else range(8)):
        x = t[blk, w]
        if x[0] == 0: continue
        print("block %d wave %d: start %d  prologue->loop %d  total %d  epilogue %d" % (blk, w, x[0] - base, x[2] - x[0], x[60] - x[0], x[60] - x[6 + 4 * 7]))
        for c in range(8):
            prev = x[2] if c == 0 else x[6 + 4 * (c - 1)]
            print("   chunk %d @%6d: build %5d (rows wait %5d, chunk request %4d, A pieces %5d)  barrier %5d  multiply %5d  barrier %5d" % (
                c, prev - base, x[3 + 4 * c] - prev, x[40 + c] - prev, x[50 + c] - x[40 + c], x[3 + 4 * c] - x[50 + c], x[4 + 4 * c] - x[3 + 4 * c],
                x[5 + 4 * c] - x[4 + 4 * c], x[6 + 4 * c] - x[5 + 4 * c]))
