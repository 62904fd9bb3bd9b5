# Scratch: tests/test_interpreter_gpu.py::test_graphed_forward_survives_cache_eviction step by step with synchronisation points
import gc, sys, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import test_interpreter_gpu as T
from dfol_vqa_amd import _lib, experiment, fol_types, host_util
from dfol_vqa_amd import synthetic as syn
from dfol_vqa_amd.interpreter import GraphedForward
DEV = T.DEV
def mark(s):
    torch.cuda.synchronize(); print("ok:", s, flush=True)
paths, names = syn.write_synthetic_ontology(tempfile.mkdtemp())
cfg = syn.reference_config(paths)
ont = experiment.build_ontology(cfg)
torch.manual_seed(4)
model = experiment.build_model(cfg, ont)
with torch.no_grad():
    model._oracle._embedding_network.linear.weight.normal_(0.0, 0.1)
    model._oracle._embedding_network.linear.bias.fill_(-2.0)
model = model.to(DEV).eval()
nm = (names["nouns"][:6], names["attributes"][:5], names["relations"][:4])
qs, scenes = T._neural_questions("choose_attr", 8, 12, 12, 2048, seed=21, names=nm)
pbs = [pb.to_cuda(DEV) for pb in T.TableCollater(2, ont, "X").collate([dict(q, scene=s) for q, s in zip(qs, scenes)])]
g = GraphedForward(model, pbs); mark("captured")
first = g(); mark("first replay")
lp0 = first["log_probability"].clone()
print("kept", len(g._keep))
host_util._upload_cache.clear(); _lib._SPLIT_W_CACHE.clear(); fol_types._geometry_cache.clear(); fol_types._pair_index_cache.clear()
ont.__dict__.get("_lower_cache", {}).clear(); model._oracle._split_cache = None; model._oracle._w2_cache = None
gc.collect(); mark("evicted")
if "--no-empty" not in sys.argv:
    torch.cuda.empty_cache(); mark("empty_cache")
qs2, scenes2 = T._neural_questions("exist", 6, 5, 9, 2048, seed=77, names=nm)
pbs2 = [pb.to_cuda(DEV) for pb in T.TableCollater(1, ont, "X").collate([dict(q, scene=s) for q, s in zip(qs2, scenes2)])]
with torch.no_grad():
    model(pbs2, False)
mark("eager forward on another batch")
junk = [torch.full((1 << k,), float("nan"), device=DEV) for k in range(4, 22)]; mark("junk")
again = g(); mark("second replay")
print("equal", torch.equal(again["log_probability"], lp0))
