"""Where does a fresh-programs batch spend its time?  Per-kernel device time and launch counts over the mixed-program batches of bench.py's
value_fresh_programs leg (eager), and the host time of the forward.  usage: python tools/lab/profile_fresh.py"""
import json, sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import bench
from dfol_vqa_amd import _lib as L
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
with torch.no_grad():
    for b, kind in enumerate(kinds * 2):
        qs = syn.full_size_questions(kind, B, N, N, names, cats, 9000 + b, with_scene=False)
        pbs = coll.collate(qs)
        for pb in pbs:
            pb.create_sparse_tensors()
        pbs = [pb.to_cuda(dev) for pb in pbs]
        if b < len(kinds):
            model(pbs, False)
            continue
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model(pbs, False)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        L.enable_kernel_timing(list(L.SIGNATURES))
        model(pbs, False)
        torch.cuda.synchronize()
        t = L.disable_kernel_timing()
        dev_ms = sum(s for n, s in t.values()) * 1e3
        top = sorted(((s * 1e3, n, k) for k, (n, s) in t.items() if n), reverse=True)[:5]
        print("%-13s wall %.2f ms, kernels %.2f ms in %d launches: %s" % (kind, wall * 1e3, dev_ms, sum(n for n, s in t.values()),
                                                                       ", ".join("%s %.2f (%d)" % (k[5:], ms, n) for ms, n, k in top)))
