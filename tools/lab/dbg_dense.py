import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))); sys.path.insert(0, 'tests')
from dfol_vqa_amd import _lib
from oracle import dfol_oracle as orc
for (M, N, K, ldx_extra, act) in [(200, 333, 300, 0, 3), (257, 300, 256, 0, 1), (64, 49, 12, 4, 3)]:
    rng = np.random.RandomState(M + N + K)
    Xfull = (rng.uniform(-1, 1, (M, K + ldx_extra)) * np.exp(rng.uniform(-6, 2, (M, 1)))).astype(np.float32)
    W = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    W[N // 2] *= 1e-6
    W[N // 3] *= 1e4
    b = rng.normal(size=N).astype(np.float32)
    xt = torch.tensor(Xfull, device="cuda")[:, :K]
    z = Xfull[:, :K].astype(np.float64) @ W.astype(np.float64).T + b
    ref = [z, orc._sigmoid(z), orc._elu(z), orc._log_sigmoid(z)][act]
    for math in ("f16x2", "bf16x3"):
        with _lib.dense_math(math):
            got = _lib.linear_act_split(xt, torch.tensor(W, device="cuda"), torch.tensor(b, device="cuda"), act).cpu().numpy()
            pre = _lib.linear_act_split(xt, torch.tensor(W, device="cuda"), torch.tensor(b, device="cuda"), 0).cpu().numpy()
        bad = ~np.isclose(got, ref, rtol=2e-5, atol=2e-5)
        print(M, N, K, act, math, "bad", bad.sum(), "nan", np.isnan(got).sum(), "cols", sorted(set(np.nonzero(bad)[1].tolist()))[:10], "N//2", N // 2, "N//3", N // 3)
        if bad.any():
            i, j = np.argwhere(bad)[0]
            print("   at", i, j, "got", got[i, j], "ref", ref[i, j], "z", z[i, j], "pre", pre[i, j])
