"""cProfile of the host side of fresh-program batches (collate -> lower -> eager forward), top functions by cumulative / own time."""
import cProfile, json, pstats, sys, io, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import bench
from dfol_vqa_amd import synthetic as syn
import dfol_vqa_amd as D
args = bench.parse([])
dev = torch.device("cuda", 0)
model, ontology, paths, names = bench.build_model(args, dev)
cats = json.load(open(paths["attribute_file"]))
N, B = args.objects, args.batch
feats = torch.rand(B * N, 2054, device=dev)
feats[:, 2052], feats[:, 2053] = 640.0, 480.0
bindex = torch.arange(B, dtype=torch.int64).repeat_interleave(N)

class Collater(D.ProgramCollaterBase):
    def __init__(self):
        super(Collater, self).__init__("select", "relate", "filter", 1, ontology=ontology)
    def collate_object_features(self, questions):
        return feats, bindex
    def collate_meta_data(self, questions):
        return {"index": {}, "embedding": torch.zeros(1, 1)}

kinds = ["exist", "verify_rel", "choose_attr", "and", "query_attr", "verify_attrs", "or", "choose_rel"]
coll = Collater()
batches = [syn.full_size_questions(kinds[b % 8], B, N, N, names, cats, 9000 + b, with_scene=False) for b in range(26)]

def run(qs):
    pbs = coll.collate(qs)
    for pb in pbs:
        pb.create_sparse_tensors()
    pbs = [pb.to_cuda(dev) for pb in pbs]
    return model(pbs, False)

with torch.no_grad():
    for qs in batches[:2]:
        run(qs)
    pr = cProfile.Profile()
    pr.enable()
    for qs in batches[2:]:
        run(qs)
    pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[4:44]))
