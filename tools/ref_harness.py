"""Drives the imported reference (/root/reference/src) to produce golden vectors.

Runs ONLY in the build container (the reference never travels to the GPU box).
Used by tools/capture_goldens.py and tools/time_reference.py.  Nothing in the
product, the tests or bench.py imports this module.
"""

import os
import sys
import types

REF_SRC = "/root/reference/src"


def import_reference():
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    # h5py / cv2 are not installed; the in-memory path never touches them.
    for m in ("h5py", "cv2"):
        if m not in sys.modules:
            sys.modules[m] = types.ModuleType(m)
    import torch  # noqa: F401
    from nsvqa.nn.interpreter import util as ref_util
    from nsvqa.nn.interpreter import batch_base_ops, batch_base_types, batch_gqa_ops
    from nsvqa.nn.interpreter.batch_gqa_interpreter import BatchGQAInterpreter
    from nsvqa.nn.vision.classifier_oracle import ClassifierOracle
    from nsvqa.data import data_pipeline
    from nsvqa.data.batch_gqa_boxfeatures_pipeline import BatchGQABoxFeaturizer
    ns = types.SimpleNamespace(
        util=ref_util, base_ops=batch_base_ops, base_types=batch_base_types, gqa_ops=batch_gqa_ops,
        BatchGQAInterpreter=BatchGQAInterpreter, ClassifierOracle=ClassifierOracle,
        data_pipeline=data_pipeline, BatchGQABoxFeaturizer=BatchGQABoxFeaturizer)
    return ns


def build_ontology(ref, paths):
    return ref.gqa_ops.GQAOntology(paths["attribute_file"], paths["class_file"], paths["vocabulary_file"],
                                   paths["word_embedding_file"], relation_json_path=paths["relation_file"])


class TableFeaturizer(object):
    """Stands in for BatchGQABoxFeaturizer: the 'object features' ARE the attribute table and the
    relation table rides in meta_data['R'].  With ClassifierOracle(..., None, None, None, cached=True)
    the reference's compute_all_log_likelihood_2 passes both through untouched
    (classifier_oracle.py:145-156), so the logic path runs on controlled tables."""

    def __init__(self, ref):
        self._ref = ref

    def featurize_scene(self, device, objects_list, batch_index, meta_data):
        ind0, ind1, ind2 = self._ref.util.find_sparse_pair_indices(batch_index, batch_index, device,
                                                                   exclude_self_relations=True)
        return {"attribute_features": objects_list,
                "relation_features": {"features": meta_data["R"], "index": [ind0, ind1, ind2]},
                "object_num": objects_list.size()[0]}


def build_table_interpreter(ref, ontology, normalize=True):
    oracle = ref.ClassifierOracle(ontology, None, None, None, normalize=normalize, cached=True)
    model = ref.BatchGQAInterpreter("golden", oracle, ontology, TableFeaturizer(ref), cached=True)
    model.eval()
    return model


def make_collater(ref, split_num=1, mode="table", ontology=None):
    import torch

    class Collater(ref.data_pipeline.ProgramCollaterBase):
        def __init__(self):
            super(Collater, self).__init__("select", "relate", "filter", split_num)

        def collate_object_features(self, questions):
            feats = torch.cat([torch.as_tensor(q["scene"]["A" if mode == "table" else "X"]) for q in questions], 0)
            bi = torch.cat([torch.full((q["scene"]["n"],), i, dtype=torch.int64) for i, q in enumerate(questions)])
            return feats, bi

        def collate_meta_data(self, questions):
            md = {"index": {}, "embedding": torch.zeros(1, 1)}
            if ontology is not None:       # token embeddings for the calibration LSTMs (batch_gqa_boxfeatures_pipeline.py:88-92)
                names = list(ontology._vocabulary["idx_to_arg"])
                md = {"index": {t: i for i, t in enumerate(names)}, "embedding": torch.from_numpy(ontology.get_embeddings(names)).float()}
            if mode == "table":
                md["R"] = torch.cat([torch.as_tensor(q["scene"]["R"]) for q in questions], 0)
            return md

    return Collater()


_FP64 = {"on": False}


def set_fp64(ref, on):
    """The reference builds a few helper sparse maps with the legacy FloatTensor constructor, which is
    fp32-only (classifier_oracle.py:36-40).  Its fp64 runs are only an auxiliary conditioning yardstick
    for the goldens, so for those runs (and only those) the helper map is cast to fp64 after the
    reference built it.  fp32 goldens come from the unmodified reference."""
    import torch
    cls = ref.ClassifierOracle
    if not hasattr(cls, "_orig_build_map"):
        cls._orig_build_map = cls._build_map

        def _build_map(self, attribute_image_map):
            m = cls._orig_build_map(self, attribute_image_map)
            if m is not None and _FP64["on"]:
                m = m.to(torch.float64)
            return m

        cls._build_map = _build_map
    _FP64["on"] = bool(on)
