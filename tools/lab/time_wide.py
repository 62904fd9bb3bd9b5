"""HIP-event timing of the wide forward products (featurizer 25600 x 2048(ld 2054) x 512 + Sigmoid; stacked first layer 25600 x 516 x 512): the
persistent wide kernel (csrc/dfol_dense_wide.hip) against the tiled kernel (DFOL_DENSE_WIDE=0 in a child process).
usage: python tools/lab/time_wide.py [objects]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    from dfol_vqa_amd import _lib as L
    M = int(sys.argv[2])
    torch.manual_seed(0)
    for (N, K, ld, act) in ((512, 2048, 2054, L.ACT_SIGMOID), (512, 516, 516, L.ACT_NONE)):
        X = torch.randn(M, ld, device="cuda")
        W = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        for _ in range(5):
            L.linear_act_split(X[:, :K], W, b, act, out)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            s.record()
            for _ in range(20):
                L.linear_act_split(X[:, :K], W, b, act, out)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 20 * 1e3)
        fl = 2.0 * M * N * K
        print("  M %d N %d K %d ld %d: %7.1f us (best of 5 x 20; %.0f TFLOP/s algorithmic, %.2f of the fp16 peak executed)" % (
            M, N, K, ld, min(ts), fl / min(ts) / 1e6, 3 * fl / min(ts) / 1e6 / 2500))
    sys.exit(0)
M = sys.argv[1] if len(sys.argv) > 1 else "25600"
for rep in range(2):
    for wide in ("1", "0"):
        env = dict(os.environ, DFOL_DENSE_WIDE=wide)
        print("DFOL_DENSE_WIDE=%s (%s)" % (wide, "persistent wide kernel" if wide == "1" else "tiled kernel"))
        sys.stdout.flush()
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", M], env=env)
