"""Lab: the fused train-forward pair kernel (dfol_pair_train_fwd_h2_f32) alone at a train step's shape, HIP events; DFOL_LIB selects an A/B build."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from dfol_vqa_amd import _lib
    N, Q, hid1, hid2 = int(sys.argv[2]), 256, 256, 300
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    O = N * Q
    n = np.full(Q, N, np.int64)
    off = torch.tensor(np.concatenate([[0], np.cumsum(n)]).astype(np.int32), device=dev)
    pair_off = torch.tensor(np.concatenate([[0], np.cumsum(n * (n - 1))]).astype(np.int64)[:Q], device=dev)
    pairs = int((n * (n - 1)).sum())
    uv = torch.randn(O, 2 * hid1, device=dev, generator=g)
    pos = torch.rand(O, 4, device=dev, generator=g)
    wg = torch.randn(hid1, 4, device=dev, generator=g) * 0.1
    wp = torch.zeros(320, hid1, device=dev); wp[:hid2] = torch.randn(hid2, hid1, device=dev, generator=g) / 16
    img = _lib.pair_pack_w2_h2(wp, hid2)
    b2 = torch.randn(hid2, device=dev, generator=g)
    e_rows = torch.randn(Q, hid2, device=dev, generator=g) / 17
    req = torch.arange(Q, dtype=torch.int32, device=dev).view(1, Q)
    nobj = torch.tensor(n.astype(np.int32), device=dev)
    run = lambda: _lib.pair_train_fwd_h2(uv, hid1, pos, wg, img, b2, hid2, nobj, off, pair_off, N, pairs, e_rows, req)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-28s %.3f ms per launch  (%.2f TB/s of Z + pre2 written)" % (os.path.basename(os.environ.get("DFOL_LIB", "libdfolvqa.so")), ms, pairs * (hid1 + hid2) * 4 / ms / 1e9))
else:
    N = sys.argv[1] if len(sys.argv) > 1 else "100"
    libs = [None] + sorted(os.path.join(ROOT, "build", f) for f in os.listdir(os.path.join(ROOT, "build")) if f.startswith("lib_h2t_"))
    for lib in libs:
        env = dict(os.environ)
        if lib: env["DFOL_LIB"] = lib
        subprocess.call([sys.executable, os.path.abspath(__file__), "child", N], env=env)
