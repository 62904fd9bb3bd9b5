"""Visual oracle: concept log-likelihoods for the interpreter, in block layout.

Reference: src/nsvqa/nn/vision/base_oracle.py, classifier_oracle.py and the MLP definitions in
src/gqa_interpreter_experiments.py:18-77.  The neural stages run on the exact-fp32 MFMA GEMM with the
activation fused (csrc/dfol_dense.hip); the per-predicate likelihood blocks are gathered from the
cached tables by csrc/dfol_logic.hip.  Parameter names match the reference's state_dict
(`_network.1.weight`, ...) so its checkpoints load with strict=False.
"""

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .fol_types import TokenType
from .host_util import get_lowered, segments_of


# ---------------------------------------------------------------------------------------------------
# MLP holders (gqa_interpreter_experiments.py:18-77)
# ---------------------------------------------------------------------------------------------------
def _run_layers(seq, x):
    """Walk an nn.Sequential of (Dropout, Linear, activation) triples; each triple is one fused GEMM launch."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Dropout):
            if m.training and m.p > 0:
                raise L.DfolError("dropout > 0 in training mode is not built on the HIP path; use dropout: 0")
            i += 1
            continue
        assert isinstance(m, nn.Linear), "unexpected layer %r" % (m,)
        act, step = L.ACT_NONE, 1
        if i + 1 < len(mods):
            nxt = mods[i + 1]
            if isinstance(nxt, nn.ELU):
                act, step = L.ACT_ELU, 2
            elif isinstance(nxt, nn.Sigmoid):
                act, step = L.ACT_SIGMOID, 2
            elif isinstance(nxt, nn.LogSigmoid):
                act, step = L.ACT_LOGSIGMOID, 2
        x = L.linear_act(x, m.weight, m.bias, act)
        i += step
    return x


class RegularMLP(nn.Module):
    """[Dropout, Linear, ELU]* + [Dropout, Linear, Sigmoid]  (gqa_interpreter_experiments.py:18-36)."""

    def __init__(self, input_dim, output_dim, layers_config, dropout):
        super(RegularMLP, self).__init__()
        if layers_config is None:
            self._network = None
        else:
            layers, last = [], input_dim
            for width in layers_config:
                layers += [nn.Dropout(dropout), nn.Linear(last, width), nn.ELU()]
                last = width
            layers += [nn.Dropout(dropout), nn.Linear(last, output_dim), nn.Sigmoid()]
            self._network = nn.Sequential(*layers)

    def forward(self, input_tensor):
        return input_tensor if self._network is None else _run_layers(self._network, input_tensor)


class EmbeddingLayer(nn.Module):
    """Dropout + Linear(hidden -> concepts) + LogSigmoid  (gqa_interpreter_experiments.py:60-77)."""

    def __init__(self, input_dim, output_dim, dropout, weights=None, biases=None, freeze_bias=False, cluster_index=None):
        super(EmbeddingLayer, self).__init__()
        if cluster_index is not None:
            raise NotImplementedError("ClusteredLogSoftmax is constructed nowhere in the reference (SURVEY.md §2 row 1)")
        linear = nn.Linear(input_dim, output_dim, bias=not freeze_bias)
        if weights is not None:
            linear.weight = nn.Parameter(weights)
        if biases is not None and not freeze_bias:
            linear.bias = nn.Parameter(biases)
        self._network = nn.Sequential(nn.Dropout(dropout), linear, nn.LogSigmoid())

    def forward(self, input_tensor):
        return _run_layers(self._network, input_tensor)

    @property
    def linear(self):
        return self._network[1]


# ---------------------------------------------------------------------------------------------------
# oracle
# ---------------------------------------------------------------------------------------------------
class OracleBase(nn.Module):
    """base_oracle.py:11-55.  forward() keeps the reference signature; blocks come back with the
    reference's trailing feature dimension: [P, NS, 1] / [P, NS, NS, 1]."""

    def __init__(self, ontology, feature_dim=1):
        super(OracleBase, self).__init__()
        self._feature_dim = feature_dim
        self._ontology = ontology

    def forward(self, token_type, token_list, token_image_map, world, default_log_likelihood=-30, normalized_probability=True):
        if not isinstance(token_list, list):
            token_list = [token_list]
        low = get_lowered(token_list, self._ontology, token_type)
        if isinstance(token_image_map, torch.Tensor):
            host_map = token_image_map.cpu().numpy()
        else:
            host_map = np.asarray(token_image_map)
        res = self.block_likelihood(token_type, low, world.pred_q(token_image_map), host_map, world,
                                    default_log_likelihood, normalized_probability)
        return res.unsqueeze(-1)

    def block_likelihood(self, token_type, low, pred_q, pred_q_host, world, default_log_likelihood=-30,
                         normalized_probability=True, orientation=L.TILE_SUBJECT_ROWS):
        raise NotImplementedError

    def get_embedding(self, tokens, meta_data, device):       # base_oracle.py:45-55
        try:
            ind = [meta_data['index'][t] for t in tokens]
            return meta_data['embedding'][ind, :]
        except (KeyError, TypeError):
            return torch.from_numpy(self._ontology.get_embeddings(tokens)).float().to(device)


class ClassifierOracle(OracleBase):
    """classifier_oracle.py:11-156 with cached tables (the only mode the reference's experiments use:
    gqa_interpreter_experiments.py:209-210 builds it with cached=True)."""

    def __init__(self, ontology, attribute_network, relation_network, embedding_network, normalize=False, cached=False):
        super(ClassifierOracle, self).__init__(ontology, feature_dim=1)
        self._attribute_network = attribute_network
        self._relation_network = relation_network
        self._embedding_network = embedding_network
        self._normalize = normalize
        self._cached = cached
        self._rel_rows = None

    # ---- a3: the cached tables (classifier_oracle.py:145-156) ------------------------------------------
    def _relation_embedding(self):
        """Rows of the embedding layer that are relations ([:, relation_index] commutes with the GEMM)."""
        lin = self._embedding_network.linear
        idx = torch.as_tensor(self._ontology._relation_index, dtype=torch.int64, device=lin.weight.device)
        w = lin.weight.detach().index_select(0, idx).contiguous()
        b = None if lin.bias is None else lin.bias.detach().index_select(0, idx).contiguous()
        return w, b

    def compute_all_log_likelihood_2(self, object_features, pair_object_features):
        if self._embedding_network is None or self._attribute_network is None:
            attr_output = object_features
        else:
            attr_output = self._embedding_network(self._attribute_network(object_features))
        if self._embedding_network is None or self._relation_network is None or pair_object_features is None:
            rel_output = pair_object_features
        else:
            h = self._relation_network(pair_object_features)
            w, b = self._relation_embedding()
            rel_output = L.linear_act(h, w, b, L.ACT_LOGSIGMOID)      # only the 333 relation columns are computed
        return attr_output, rel_output

    # ---- a4 / a5: per-predicate blocks (classifier_oracle.py:44-137) -------------------------------------
    def block_likelihood(self, token_type, low, pred_q, pred_q_host, world, default_log_likelihood=-30,
                         normalized_probability=True, orientation=L.TILE_SUBJECT_ROWS):
        if not self._cached:
            raise NotImplementedError("only the cached-table oracle of the reference's experiments is built")
        dev = world._device
        cols, _, _ = low.on(dev)
        if token_type == TokenType.ATTRIBUTE:
            gather = lambda c, pq: L.attr_gather(world._attribute_features, world._obj_off, pq, c, world._NS,
                                                 float(default_log_likelihood))
        else:
            table = world._relation_features['features']
            if table is None:                                   # no image has two objects
                table = torch.zeros(1, 1, dtype=torch.float32, device=dev)
            gather = lambda c, pq: L.rel_gather(table, world._pair_off, world._n_obj, pq, c, world._NS, orientation,
                                                float(default_log_likelihood))
        if not (self._normalize and normalized_probability):
            return gather(cols, pred_q)
        valid = low.valid.astype(bool)
        seg = segments_of(np.asarray(pred_q_host)[valid])        # clusters of the compressed list (:23, :72, :124)
        if len(seg) - 1 == int(valid.sum()):                     # all singletons: cluster_map is None (:27-28)
            return gather(cols, pred_q)
        if low.all_valid:
            ll = gather(cols, pred_q)
            L.option_normalize_(ll, torch.as_tensor(seg).to(dev), pred_q, world._n_obj, world._NS)
            return ll
        # no-op tokens inside an option list: normalise the compressed list, then put default blocks back
        keep = torch.as_tensor(np.nonzero(valid)[0]).to(dev)
        pq_c = pred_q.index_select(0, keep).contiguous()
        ll_c = gather(cols.index_select(0, keep).contiguous(), pq_c)
        L.option_normalize_(ll_c, torch.as_tensor(seg).to(dev), pq_c, world._n_obj, world._NS)
        ll = torch.full((len(low.cols),) + tuple(ll_c.shape[1:]), float(default_log_likelihood), dtype=torch.float32, device=dev)
        ll[keep] = ll_c
        return ll
