import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from dfol_vqa_amd import _lib as L
M, N, K = 25600, 512, 2048
x = torch.rand(M, K, device='cuda') - 0.5; w = torch.rand(N, K, device='cuda') - 0.5; b = torch.rand(N, device='cuda'); y = torch.empty(M, N, device='cuda')
for _ in range(3): L.linear_act_split(x, w, b, 2, y)
torch.cuda.synchronize()
lib = L.load(); buf = (ctypes.c_longlong * 256)(); lib.dfol_dense_trace_read.argtypes = [ctypes.c_void_p]; lib.dfol_dense_trace_read(buf)
t = np.array(buf[:], dtype=np.int64).reshape(4, 64)
for w_ in (0, 3):
    x_ = t[w_]
    for ks in range(1, 12):
        print("wave %d ks %2d: store_a %5d  wait+barrier %5d  issue+compute %5d  end barrier %5d" % (w_, ks, x_[4*ks+1]-x_[4*ks], x_[4*ks+2]-x_[4*ks+1], x_[4*ks+3]-x_[4*ks+2], x_[4*ks+4]-x_[4*ks+3]))
