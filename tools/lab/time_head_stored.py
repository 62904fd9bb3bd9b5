"""Lab: the head's backward products (dz tall + weight gradient with the sums) at the train shape, default build against A/B builds under build/lib_hst_*.so."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from dfol_vqa_amd import _lib
    dev = torch.device("cuda:0")
    Q, n, H1, H2 = 256, 100, 256, 300
    per = n * (n - 1)
    M = Q * per
    g = torch.Generator(device=dev).manual_seed(1)
    p2 = torch.rand(M, H2, device=dev, generator=g) * 0.98 + 0.01
    z = torch.nn.functional.elu(torch.randn(M, H1, device=dev, generator=g))
    w2 = torch.randn(H2, H1, device=dev, generator=g) / 16
    E = torch.randn(Q, H2, device=dev, generator=g) * 0.1
    dx = torch.randn(M, device=dev, generator=g) * 1e-3
    pred_off = torch.arange(Q + 1, device=dev, dtype=torch.int64) * per
    rep = torch.arange(Q, device=dev, dtype=torch.int32).repeat_interleave(per)
    dz = torch.empty(M, H1, device=dev)
    names = ["dfol_pair_dz_fused_f32", "dfol_pair_dz_tall_f32", "dfol_pair_wgrad_fused_f32", "dfol_pair_wgrad_fused_sums_f32"]
    for sums in (False, True):
        for _ in range(2):
            _lib.pair_head_products(dx, p2, z, w2, E, pred_off, rep, True, True, dz_out=dz, sums=sums)
        _lib._timed = {k: [] for k in names}
        for _ in range(5):
            _lib.pair_head_products(dx, p2, z, w2, E, pred_off, rep, True, True, dz_out=dz, sums=sums)
        torch.cuda.synchronize()
        print("%-20s sums=%d " % (os.path.basename(os.environ.get("DFOL_LIB", "default")), sums),
              "  ".join("%s %.3f" % (k.replace("dfol_pair_", "").replace("_f32", ""), min(a.elapsed_time(b) for a, b in v)) for k, v in _lib._timed.items() if v))
        _lib._timed = None
else:
    libs = [None] + sorted(os.path.join(ROOT, "build", f) for f in os.listdir(os.path.join(ROOT, "build")) if f.startswith("lib_hst_"))
    for lib in libs:
        env = dict(os.environ)
        if lib: env["DFOL_LIB"] = lib
        subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=env)
