"""Where do the small torch launches of a train step come from?  One eager step of the bench's train configuration under torch.profiler with
Python stacks; prints, per torch operator that launched a kernel, the innermost frames inside this repository.
usage: python tools/lab/train_glue_stacks.py [bench args, e.g. --mlp-math bf16]"""
import collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from dfol_vqa_amd import parallel, training
args = bench.parse(["--mode", "train"] + sys.argv[1:])
dev = torch.device("cuda", 0)
model, ontology, paths, names = bench.build_model(args, dev, train=True)
_, pbs = bench.build_batch(args, 0, ontology, names, dev)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-4)
bucket = parallel.GradBucket(params)
step = lambda: training.train_batch(model, opt, pbs, 0.65, global_batch_size=args.batch, bucket=bucket, sync_loss=False)
for _ in range(3):
    step()
torch.cuda.synchronize()
import traceback
from torch.overrides import TorchFunctionMode
WATCH = ("zeros", "zeros_like", "new_zeros", "full", "ones_like", "to", "float", "long", "int", "contiguous", "clone", "where", "gt", "mul", "div",
         "add", "add_", "sum", "cat", "gather", "index_select", "__getitem__", "__and__", "abs", "amax", "exp", "zero_", "fill_", "copy_", "index_put_",
         "__setitem__", "__mul__", "__rmul__", "__truediv__", "__add__", "__sub__", "__gt__", "expand", "repeat", "stack", "type", "new_full", "masked_fill")
agg = collections.Counter()


class Trace(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", str(func))
        if name in WATCH:
            fr = [f for f in traceback.extract_stack()[:-1] if ROOT in f.filename and "tools/lab" not in f.filename][-2:]
            cuda = any(isinstance(a, torch.Tensor) and a.is_cuda for a in args) or "cuda" in str((kwargs or {}).get("device", ""))
            if fr and cuda:
                agg[(name, " <- ".join("%s:%d %s" % (f.filename.replace(ROOT + "/", ""), f.lineno, f.name) for f in reversed(fr)))] += 1
        return func(*args, **(kwargs or {}))


with torch.autograd.set_multithreading_enabled(False), Trace():
    step()
torch.cuda.synchronize()
print("torch calls on GPU tensors from repository code in one eager train step (call, count, innermost frames):")
for (name, where), n in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%3d  %-14s %s" % (n, name, where))
