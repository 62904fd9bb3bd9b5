"""Lab: the calibration LSTM cell and the attention-output kernel alone - a chain of dependent launches (each cell reads the last one's state), HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dfol_vqa_amd import _lib as L
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
wt_ih, wt_hh, b_ih, b_hh = (rnd(200, 318) * 0.05).t().contiguous(), (rnd(200, 50) * 0.1).t().contiguous(), rnd(200) * 0.1, rnd(200) * 0.1
x, h, c = rnd(rows, 318), rnd(rows, 50) * 0.5, rnd(rows, 50)
W, b = rnd(4, 100) * 0.1, rnd(4)
def chain(n):
    hh, cc = h, c
    for _ in range(n):
        hh, cc = L.lstm_cell(x, hh, cc, wt_ih, wt_hh, b_ih, b_hh)
    return hh
def mods(n):
    for _ in range(n):
        L.attention_modulations(h, c, W, b)
for name, fn in (("lstm_cell", chain), ("attention_modulations", mods)):
    fn(5); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(200); e1.record(); torch.cuda.synchronize()
    print("%s rows %d: %.2f us per launch (200 in a row)" % (name, rows, e0.elapsed_time(e1) * 1000 / 200))
