"""Times the pair layer's first-stage kernels (csrc/dfol_pair_train.hip) at a train step's shape: Q images of N objects, HID1 wide.
usage: python tools/lab/time_hidden1.py [N] [Q] [HID1]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dfol_vqa_amd import _lib

N, Q, H1 = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 100), (2, 256), (3, 128)))
dev = torch.device("cuda:0")
n = np.full(Q, N, np.int64)
O, pairs = int(n.sum()), int((n * (n - 1)).sum())
obj_off = torch.tensor(np.concatenate([[0], np.cumsum(n)]).astype(np.int32), device=dev)
pair_off = torch.tensor(np.concatenate([[0], np.cumsum(n * (n - 1))]).astype(np.int64), device=dev)
n_obj = torch.tensor(n.astype(np.int32), device=dev)
g = torch.Generator(device=dev).manual_seed(1)
U, V = (torch.randn(O, H1, device=dev, generator=g) for _ in range(2))
pos = torch.rand(O, 4, device=dev, generator=g) * 0.8 + 0.05
Wg = torch.randn(H1, 4, device=dev, generator=g) * 0.5


def timed(f, reps=20):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps):
            f()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best


for store in (torch.float32, torch.bfloat16):
    z, geo = _lib.pair_hidden1_fwd(U, V, pos, Wg, obj_off, pair_off, n_obj, N, pairs, store)
    dz = torch.randn(pairs, H1, device=dev, generator=g).to(store)
    b = z.element_size()
    tf = timed(lambda: _lib.pair_hidden1_fwd(U, V, pos, Wg, obj_off, pair_off, n_obj, N, pairs, store))
    tb = timed(lambda: _lib.pair_hidden1_bwd(dz, z, geo, obj_off, pair_off, n_obj, N, O))
    if store == torch.float32:
        tr = timed(lambda: _lib.pair_hidden1_bwd(dz, None, geo, obj_off, pair_off, n_obj, N, O, uvw=(U, V, Wg)))
        print(f"  bwd rebuilding z: {tr * 1e3:7.1f} us ({pairs * (H1 * b + 16) / tr / 1e9:5.2f} TB/s read)")
    print(f"{str(store):16s} N={N} Q={Q} HID1={H1}: fwd {tf * 1e3:7.1f} us ({pairs * (H1 * b + 16) / tf / 1e9:5.2f} TB/s written)   "
          f"bwd {tb * 1e3:7.1f} us ({pairs * (2 * H1 * b + 16) / tb / 1e9:5.2f} TB/s read)")
