import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from dfol_vqa_amd import _lib
dev = lambda a: torch.as_tensor(a).cuda()
def run(n_list, hid1, hid2, K, seed):
    rng = np.random.RandomState(seed)
    Q, O, NS, C = len(n_list), sum(n_list), max(4, (max(n_list) + 7) // 8 * 8), 20
    off = np.concatenate([[0], np.cumsum(n_list)]).astype(np.int32)
    uv = rng.uniform(-1, 1, (O, 2 * hid1)).astype(np.float32)
    pos = rng.uniform(0.05, 0.9, (O, 4)).astype(np.float32)
    wg = rng.uniform(-0.5, 0.5, (hid1, 4)).astype(np.float32)
    w2 = np.zeros(((hid2 + 31) // 32 * 32, hid1), np.float32)
    w2[:hid2] = rng.normal(size=(hid2, hid1)).astype(np.float32) / np.sqrt(hid1)
    b2 = rng.normal(size=hid2).astype(np.float32)
    E = (rng.normal(size=(C, hid2)) / np.sqrt(hid2)).astype(np.float32)
    be = rng.normal(size=C).astype(np.float32)
    req_col = rng.randint(0, C, (K, Q)).astype(np.int32)
    req_tile = np.arange(K * Q, dtype=np.int32).reshape(K, Q)
    args = (dev(uv), hid1, dev(pos), dev(wg))
    tail = (dev(E), dev(be), dev(np.array(n_list, np.int32)), dev(off), max(n_list), dev(req_col), dev(req_tile), None)
    packed = _lib.pair_pack_w2_h2(dev(w2), hid2)
    out = {}
    for form in ("pingpong", "interleaved", "interleaved2"):
        os.environ["DFOL_PAIR_H2_FORM"] = form.rstrip("2")
        t = torch.full((K * Q, NS, NS), -30.0, device="cuda")
        out[form] = _lib.pair_ll_h2(*args, packed, dev(b2), hid2, *tail, t).clone().cpu().numpy()
        torch.cuda.synchronize()
    a, b, b2_ = out["pingpong"], out["interleaved"], out["interleaved2"]
    bad = np.argwhere(a != b)
    print(n_list, hid1, hid2, "mismatches", len(bad), "of", (a != -30).sum(), "| interleaved repeatable:", np.array_equal(b, b2_), "| max diff", np.abs(a - b).max())
    if len(bad):
        t_, s_, o_ = bad[:, 0], bad[:, 1], bad[:, 2]
        n = n_list[0]
        e = s_ * (n - 1) + o_ - (o_ > s_)
        print("  tiles", np.unique(t_)[:10], "pair index e range", e.min(), e.max(), "e%256 hist", np.bincount((e % 256) // 32, minlength=8), "first", bad[:5].tolist())
run([100], 256, 300, 1, 1)
run([64], 256, 300, 1, 2)
run([40], 256, 300, 1, 3)
run([40], 64, 300, 1, 4)
run([40], 32, 300, 1, 5)
run([17], 256, 300, 1, 6)
run([20, 20, 20, 20], 256, 300, 1, 7)
