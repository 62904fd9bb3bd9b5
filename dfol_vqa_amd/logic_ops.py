"""The differentiable first-order-logic operators in block layout.

Reference: src/nsvqa/nn/interpreter/batch_base_ops.py (BatchBayesianLogicCell :42-237, FilterBatch
:301-405, RelateBatch :471-596).  The arithmetic runs in csrc/dfol_logic.hip; these classes only
resolve tokens, pick priors/quantifiers and launch.
"""

import numpy as np
import torch
import torch.nn as nn

from . import ops as L
from .fol_types import BatchAttentionState, BatchVariableSet, Quantifier, TokenType
from .host_util import upload, detect_negations, get_lowered, is_valid_token


class BatchBayesianLogicCell(nn.Module):
    """batch_base_ops.py:42-237.  Block-layout shapes:
         log_prior      [Q, arity, NS]
         log_likelihood [P, NS(, NS)] (a trailing feature dim of 1 is accepted and dropped)
         quantifiers    [P, arity]
         batch_object_map = the BatchWorld (it carries n_obj); predicate_question_map = int32 [P] or None
       returns [P, arity, NS]."""

    def __init__(self, arity, trainable_module_type=None, feature_dim=1, trainable_gate=False):
        super(BatchBayesianLogicCell, self).__init__()
        if trainable_module_type is not None or trainable_gate:
            raise NotImplementedError("trainable_gate / operator_layers_config are off in every shipped config (SURVEY.md §2 row 3)")
        if arity not in (1, 2):
            raise NotImplementedError("Likelihood for arity > 2 is not implemented.")      # cf. :210
        self._arity = arity
        self._feature_dim = feature_dim

    def forward(self, log_prior, log_likelihood, quantifiers, dim_order, batch_object_map=None, predicate_question_map=None,
                is_negated=None, default_log_likelihood=-30, active=None, want=None, orientation=L.TILE_SUBJECT_ROWS,
                need=(True, True)):
        world = batch_object_map
        assert quantifiers.size()[1] == self._arity, "The number of quantifiers must match the arity of the operator."
        assert len(dim_order) == self._arity, "The number of dimension order elements must match the arity of the operator."
        assert log_prior.size()[1] == self._arity, "The second dimension of log-prior must be equal to the arity of the operator."
        assert list(dim_order) == list(range(self._arity)), "only the natural dim_order is built"
        ll = log_likelihood
        if ll.dim() == self._arity + 2:
            assert ll.size(-1) == 1, "feature_dim > 1 needs a trainable module, which is not built"
            ll = ll.squeeze(-1)                                  # mean over the single feature (:194)
        assert ll.dim() == self._arity + 1, "The number of dimensions of log-likelihood must be equal to the arity of the operator + 2."
        P, Q = ll.size(0), log_prior.size(0)
        assert P == Q or (predicate_question_map is not None and predicate_question_map.numel() == P), \
            "In case predicate_num != question_num, predicate_question_map of size (predicate_num) must be provided."
        pred_q = world._ident if predicate_question_map is None else predicate_question_map
        neg = None if is_negated is None else is_negated.to(torch.uint8)
        if self._arity == 1:
            out = L.filter_fwd(log_prior[:, 0, :].contiguous(), ll.contiguous(), pred_q, world._n_obj, neg, active)
            return out.unsqueeze(1)
        ps, po = L.relate_fwd(log_prior[:, 0, :].contiguous(), log_prior[:, 1, :].contiguous(), ll.contiguous(), pred_q, world._n_obj,
                              quantifiers[:, 0].contiguous(), quantifiers[:, 1].contiguous(), neg, active, want, orientation,
                              lone_forall_identity=(P == 1), need_s=need[0], need_o=need[1])
        if ps is None or po is None:
            return ps, po
        return torch.stack([ps, po], 1)


class BatchOperatorBase(nn.Module):
    """batch_base_ops.py:241-286."""

    def __init__(self, oracle, is_terminal, fan_in, fan_out, forward_attention_network=None, backward_attention_network=None,
                 attention_output_network=None):
        super(BatchOperatorBase, self).__init__()
        self._oracle = oracle
        self._is_terminal = is_terminal
        self._fan_in = fan_in
        self._fan_out = fan_out
        if forward_attention_network is not None and backward_attention_network is not None and attention_output_network is not None:
            self._forward_attention_network = forward_attention_network          # shared by every operator (:251-254)
            self._backward_attention_network = backward_attention_network
            self._attention_output_network = attention_output_network

    def is_terminal(self):
        return self._is_terminal

    def fan_in(self):
        return self._fan_in

    def fan_out(self):
        return self._fan_out

    def _get_features(self, world, token_list, op_features):                    # :265-273
        result = self._oracle.get_embedding(token_list, world._meta_data, world._device)
        if result.dim() < 2:
            result = result.unsqueeze(0)
        temp = op_features.repeat(len(token_list), 1) if len(token_list) > 1 else op_features.unsqueeze(0)
        return torch.cat([temp, result], dim=1)

    def _token_features(self, world, tokens, op_feature, type_flag):
        """LSTM input rows [op one-hot, type flag, token embedding]; zero rows for no-op tokens (:437-446, :628-637)."""
        # the forward and the backward calibration pass ask for the same rows: build them once per scene
        cache = world.__dict__.setdefault("_calib_features", {})
        key = (tuple(str(t) for t in tokens), float(type_flag), op_feature.data_ptr())
        hit = cache.get(key)
        if hit is not None:
            return hit
        feats = self._token_features_now(world, tokens, op_feature, type_flag)
        cache[key] = feats
        return feats

    def _token_features_now(self, world, tokens, op_feature, type_flag):
        ind = [is_valid_token(v) for v in tokens]
        kept = [t for t, k in zip(tokens, ind) if k]
        _, _, names = detect_negations(kept)
        host = getattr(op_feature, "_host", None)
        if host is not None:                                  # the constant columns (op one-hot, type flag) come from the host, memoised
            emb = self._oracle.get_embedding(names, world._meta_data, world._device)
            emb = emb.unsqueeze(0) if emb.dim() < 2 else emb
            const = np.concatenate([np.asarray(host, np.float32), np.asarray([type_flag], np.float32)])
            feats = torch.cat([upload(np.tile(const, (len(names), 1)), world._device), emb], dim=1)
        else:
            flag = torch.full((1,), float(type_flag), dtype=torch.float32, device=world._device)
            feats = self._get_features(world, names, torch.cat([op_feature, flag], dim=0))
        if all(ind):
            return feats
        full = torch.zeros(len(tokens), feats.shape[1], dtype=torch.float32, device=world._device)
        return full.index_copy_(0, upload(np.nonzero(ind)[0].astype(np.int64), world._device), feats)    # (a mask assignment synchronises)

    def _compute_attention_modulations(self, forward_state, backward_state):    # :275-286
        net = self._attention_output_network
        ref = backward_state[0] if forward_state is None else forward_state[0]
        plain = isinstance(net, nn.Sequential) and len(net) == 2 and isinstance(net[0], nn.Linear) and isinstance(net[1], nn.Sigmoid)
        needs_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in net.parameters()) or
                                                  any(t is not None and t[0].requires_grad for t in (forward_state, backward_state)))
        if plain and ref.is_cuda and ref.dtype == torch.float32 and not needs_grad:
            # inference: Linear(2 S -> 4) + Sigmoid on the two states in ONE launch, no concatenation (the kernel the native executor calls too)
            return L.attention_modulations(None if forward_state is None else forward_state[0], None if backward_state is None else backward_state[0],
                                           net[0].weight, net[0].bias)
        fs = torch.zeros_like(backward_state[0]) if forward_state is None else forward_state[0]
        bs = torch.zeros_like(forward_state[0]) if backward_state is None else backward_state[0]
        return net(torch.cat([fs, bs], dim=1))


class SelectBatch(BatchOperatorBase):
    """batch_base_ops.py:290-297."""

    def __init__(self, oracle, **kw):
        super(SelectBatch, self).__init__(oracle, is_terminal=False, fan_in=0, fan_out=1, **kw)

    def forward(self, id, world, name, quantifier=Quantifier.EXISTS):
        return world.variable_set(name, quantifier=quantifier)


def _expand_quantifier(quant, pred_q, identity):
    return quant if identity else L.gather_rows(quant.unsqueeze(1).contiguous(), pred_q).squeeze(1)


def _host_map(predicate_question_map, P):
    """predicate -> question list on the host (needed for the option clusters), None = identity."""
    m = predicate_question_map
    if m is None:
        return None
    if isinstance(m, torch.Tensor):
        host = getattr(m, "_host", None)
        if host is not None:
            return host
        if m.is_sparse:
            m = m.coalesce().indices()[1]
        return m.cpu().numpy().tolist()
    return list(m)


class FilterBatch(BatchOperatorBase):
    """batch_base_ops.py:301-405."""

    def __init__(self, oracle, trainable_module_type=None, feature_dim=1, trainable_gate=False, **kw):
        super(FilterBatch, self).__init__(oracle, is_terminal=False, fan_in=1, fan_out=1, **kw)
        self._blc = BatchBayesianLogicCell(arity=1, trainable_module_type=trainable_module_type, feature_dim=feature_dim,
                                           trainable_gate=trainable_gate)
        self._modulations = {}
        self._forward_state = {}

    def forward(self, op_id, world, variable_set, attribute_list, predicate_question_map=None, default_log_likelihood=-30,
                normalized_probability=True):
        if not isinstance(attribute_list, list):
            attribute_list = [attribute_list]
        low = get_lowered(attribute_list, self._oracle._ontology, TokenType.ATTRIBUTE)
        if not low.any_valid:                                    # :316-317
            return variable_set
        question_num = variable_set.batch_size()
        predicate_num = len(attribute_list)
        host_map = _host_map(predicate_question_map, predicate_num)
        assert question_num == predicate_num or (host_map is not None and len(host_map) == predicate_num), "Batch size mismatch."
        identity = host_map is None
        pred_q = world._ident if identity else world.pred_q(predicate_question_map)
        quantifier = _expand_quantifier(variable_set._quantifier, pred_q, identity)           # :341-343
        dev = variable_set.device
        _, neg_dev, valid_dev = low.on(dev)
        ll = self._oracle.block_likelihood(TokenType.ATTRIBUTE, low, pred_q, range(predicate_num) if identity else host_map,
                                           world, default_log_likelihood, normalized_probability)
        att = self._blc(variable_set._log_attention.unsqueeze(1), ll, quantifier.unsqueeze(1), [0], world, pred_q,
                        neg_dev if low.any_neg else None, default_log_likelihood,
                        active=None if low.all_valid else valid_dev)[:, 0, :]
        res = BatchVariableSet(variable_set._name, dev, variable_set.object_num(), predicate_num, quantifiers=quantifier,
                               log_attention=att, world=world, predicate_question_map=None if identity else pred_q,
                               base_cumulative_loss=variable_set.cumulative_loss(),
                               prev_variable_sets_num=variable_set._prev_variable_sets_num + 1)
        if op_id in self._modulations:                            # :401-403
            res = res.apply_modulations(self._modulations.pop(op_id), variable_set, predicate_question_map)
        return res


def _filter_transform_attention(self, op_id, is_forward, world, attention_state, attribute_list, op_feature, predicate_question_map=None):
    """FilterBatch.transform_attention, batch_base_ops.py:407-467."""
    if not isinstance(attribute_list, list):
        attribute_list = [attribute_list]
    if not any(is_valid_token(v) for v in attribute_list):
        return attention_state
    question_num, predicate_num = world.batch_size(), len(attribute_list)
    host_map = _host_map(predicate_question_map, predicate_num)
    assert question_num == predicate_num or (host_map is not None and len(host_map) == predicate_num), "Batch size mismatch."
    pred_q = None if host_map is None else world.pred_q(predicate_question_map)
    features = self._token_features(world, attribute_list, op_feature, 0.0)
    if is_forward:
        old = attention_state.expand(pred_q) if pred_q is not None else attention_state
        new_state_tuple = self._forward_attention_network(features, old._state)
        self._forward_state[op_id] = new_state_tuple
        return BatchAttentionState(attention_state._name, world._device, new_state_tuple)
    if op_id[-1] != 'n':
        self._modulations[op_id] = self._compute_attention_modulations(self._forward_state[op_id], attention_state._state)
    self._forward_state.pop(op_id, None)
    new_state = BatchAttentionState(attention_state._name, world._device, self._backward_attention_network(features, attention_state._state))
    return new_state.squeeze(pred_q, question_num, host=host_map) if pred_q is not None else new_state


FilterBatch.transform_attention = _filter_transform_attention


class RelateBatch(BatchOperatorBase):
    """batch_base_ops.py:471-596.  `need` = (subject, object) lets a caller that only consumes one
    posterior (GQARelateBatch keeps exactly one, batch_gqa_ops.py:371) skip the other direction."""

    def __init__(self, oracle, trainable_module_type=None, feature_dim=1, trainable_gate=False, **kw):
        super(RelateBatch, self).__init__(oracle, is_terminal=False, fan_in=2, fan_out=2, **kw)
        self._blc = BatchBayesianLogicCell(arity=2, trainable_module_type=trainable_module_type, feature_dim=feature_dim,
                                           trainable_gate=trainable_gate)
        self._subject_modulations = {}
        self._object_modulations = {}
        self._forward_subject_state = {}
        self._forward_object_state = {}

    def forward(self, op_id, world, subject_variable_set, object_variable_set, relation_list, predicate_question_map=None,
                default_log_likelihood=-30, normalized_probability=True, want=None):
        assert subject_variable_set.batch_size() == object_variable_set.batch_size(), \
            "The subject and object variable sets must have the same batch size."
        assert subject_variable_set.object_num() == object_variable_set.object_num(), \
            "The subject and object variable sets must have the same number of objects."
        if not isinstance(relation_list, list):
            relation_list = [relation_list]
        low = get_lowered(relation_list, self._oracle._ontology, TokenType.RELATION)
        if not low.any_valid:                                    # :491-492
            return subject_variable_set, object_variable_set
        question_num = world.batch_size()
        predicate_num = len(relation_list)
        host_map = _host_map(predicate_question_map, predicate_num)
        assert question_num == predicate_num or (host_map is not None and len(host_map) == predicate_num), "Batch size mismatch."
        identity = host_map is None
        pred_q = world._ident if identity else world.pred_q(predicate_question_map)
        dev = subject_variable_set.device
        q_s = _expand_quantifier(subject_variable_set._quantifier, pred_q, identity)          # :518-521
        q_o = _expand_quantifier(object_variable_set._quantifier, pred_q, identity)
        _, neg_dev, valid_dev = low.on(dev)
        tile = self._oracle.block_likelihood(TokenType.RELATION, low, pred_q, range(predicate_num) if identity else host_map,
                                             world, default_log_likelihood, normalized_probability)
        ps, po = L.relate_fwd(subject_variable_set._log_attention, object_variable_set._log_attention, tile, pred_q, world._n_obj,
                              q_s, q_o, neg_dev if low.any_neg else None, None if low.all_valid else valid_dev, want,
                              L.TILE_SUBJECT_ROWS, lone_forall_identity=(predicate_num == 1), diag_absent=True)
        n_prev = subject_variable_set._prev_variable_sets_num + object_variable_set._prev_variable_sets_num + 1
        pqm = None if identity else pred_q
        new_subject_set = BatchVariableSet(subject_variable_set._name, dev, subject_variable_set.object_num(), predicate_num,
                                           quantifiers=q_s, log_attention=ps, world=world, predicate_question_map=pqm,
                                           prev_variable_sets_num=n_prev)        # both take the subject's quantifier (:571-586)
        new_object_set = BatchVariableSet(object_variable_set._name, dev, object_variable_set.object_num(), predicate_num,
                                          quantifiers=q_s, log_attention=po, world=world, predicate_question_map=pqm,
                                          prev_variable_sets_num=n_prev)
        if op_id in self._subject_modulations:
            new_subject_set = new_subject_set.apply_modulations(self._subject_modulations.pop(op_id), subject_variable_set, predicate_question_map)
        if op_id in self._object_modulations:
            new_object_set = new_object_set.apply_modulations(self._object_modulations.pop(op_id), object_variable_set, predicate_question_map)
        return new_subject_set, new_object_set


def _relate_transform_attention(self, op_id, is_forward, world, subject_attention_state, object_attention_state, relation_list, op_feature,
                                predicate_question_map=None):
    """RelateBatch.transform_attention, batch_base_ops.py:598-684."""
    if not isinstance(relation_list, list):
        relation_list = [relation_list]
    if not any(is_valid_token(v) for v in relation_list):
        return subject_attention_state, object_attention_state
    question_num, predicate_num = world.batch_size(), len(relation_list)
    host_map = _host_map(predicate_question_map, predicate_num)
    assert question_num == predicate_num or (host_map is not None and len(host_map) == predicate_num), "Batch size mismatch."
    pred_q = None if host_map is None else world.pred_q(predicate_question_map)
    features = self._token_features(world, relation_list, op_feature, 1.0)
    dev = world._device
    if is_forward:
        s_old = subject_attention_state.expand(pred_q) if pred_q is not None else subject_attention_state
        o_old = object_attention_state.expand(pred_q) if pred_q is not None else object_attention_state
        new = self._forward_attention_network(features, (s_old._state[0] + o_old._state[0], s_old._state[1] + o_old._state[1]))
        copy = (new[0].clone(), new[1].clone())
        self._forward_subject_state[op_id] = new
        self._forward_object_state[op_id] = copy
        return BatchAttentionState(subject_attention_state._name, dev, new), BatchAttentionState(object_attention_state._name, dev, copy)
    if op_id[-1] != 'n':
        self._subject_modulations[op_id] = self._compute_attention_modulations(self._forward_subject_state[op_id], subject_attention_state._state)
        self._object_modulations[op_id] = self._compute_attention_modulations(self._forward_object_state[op_id], object_attention_state._state)
    self._forward_subject_state.pop(op_id, None)
    self._forward_object_state.pop(op_id, None)
    agg = (subject_attention_state._state[0] + object_attention_state._state[0], subject_attention_state._state[1] + object_attention_state._state[1])
    new = self._backward_attention_network(features, agg)
    new_s = BatchAttentionState(subject_attention_state._name, dev, new)
    new_o = BatchAttentionState(object_attention_state._name, dev, (new[0].clone(), new[1].clone()))
    if pred_q is not None:
        new_s, new_o = new_s.squeeze(pred_q, question_num, host=host_map), new_o.squeeze(pred_q, question_num, host=host_map)
    return new_s, new_o


RelateBatch.transform_attention = _relate_transform_attention
