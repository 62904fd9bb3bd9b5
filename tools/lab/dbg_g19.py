"""Lab: one g19 case through the HIP train step, log-probabilities and loss next to the reference's fp32 / fp64 (golden g19)."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from dfol_vqa_amd import experiment, training, synthetic as syn
from test_interpreter_gpu import TableCollater
DEV = torch.device("cuda:0")
paths, names = syn.write_synthetic_ontology(tempfile.mkdtemp())
cfg = syn.reference_config(paths, freeze_featurizer=False, freeze_attribute_network=False, freeze_relation_network=False, freeze_embedding_network=False, dropout=0.0)
ont = experiment.build_ontology(cfg)
model = experiment.build_model(cfg, ont)
a, meta = gu.load("g19_full_size_train_step")
syn.load_seeded_weights(model, meta["weight_seed"])
model = model.to(DEV).train()
np.set_printoptions(precision=5, linewidth=200, suppress=False)
for name in sys.argv[1:]:
    qs, cm, ref, grads = gu.g19_case(name, a, meta)
    pbs = [pb.to_cuda(DEV) for pb in TableCollater(1, ont, "X").collate([dict(q) for q in qs])]
    model.zero_grad(set_to_none=True)
    res = model(pbs, True)
    loss = training.compute_loss(pbs, res) / len(qs)
    lp = res["log_probability"].detach().cpu().numpy()
    print(name, "loss", float(loss), "ref32", ref["f32"][0], "ref64", ref["f64"][0])
    print(" ours ", lp)
    print(" ref32", ref["f32"][1])
    print(" ref64", ref["f64"][1])
    with torch.no_grad():
        model.eval()
        res_i = model(pbs, False)
        model.train()
    print(" ours (inference)", res_i["log_probability"].cpu().numpy())
    print(" answers", [q["answer"] for q in qs])
