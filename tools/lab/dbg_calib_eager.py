import os, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import dfol_vqa_amd as D
from dfol_vqa_amd import experiment, _lib, native_exec
from dfol_vqa_amd import synthetic as syn
device = torch.device("cuda", 0)
paths, names = syn.write_synthetic_ontology(tempfile.mkdtemp())
cfg = syn.reference_config(paths, activate_attention_transfer=True)
ontology = experiment.build_ontology(cfg)
model = experiment.build_model(cfg, ontology); bench.init_weights(model); model = model.to(device).eval()
voc = list(ontology._vocabulary["idx_to_arg"])
emb = torch.randn(len(voc), 300) * 0.1
class Collater(D.ProgramCollaterBase):
    def __init__(self): super(Collater, self).__init__("select", "relate", "filter", 1, ontology=ontology)
    def collate_object_features(self, qs):
        return torch.cat([torch.from_numpy(q["scene"]["X"]) for q in qs], 0), torch.cat([torch.full((q["scene"]["n"],), i, dtype=torch.int64) for i, q in enumerate(qs)])
    def collate_meta_data(self, qs): return {"index": {t: i for i, t in enumerate(voc)}, "embedding": emb}
nouns, attrs, rels = names["nouns"][:8], names["attributes"][:6], names["relations"][:5]
qs = []
for i in range(256):
    br, last = syn.three_hop_program(i, nouns, attrs, rels)
    qs.append(syn.question(i, br, last, "yes", syn.feature_scene(i, 100, 2048)))
pbs = [pb.to_cuda(device) for pb in Collater().collate(qs)]
import cProfile, pstats
with torch.no_grad():
    for k in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model(pbs, False)
        torch.cuda.synchronize()
        print(k, "ms %.3f" % ((time.perf_counter() - t0) * 1e3), dict(_lib.PATH_COUNTS), id(pbs[0]._native_plan))
    pr = cProfile.Profile(); pr.enable()
    for k in range(5):
        model(pbs, False)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
