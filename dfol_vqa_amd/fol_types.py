"""Basic types of the interpreter in BLOCK layout (reference: src/nsvqa/nn/interpreter/batch_base_types.py).

The reference keeps one flat attention row over every object of a ProgramBatch plus a sparse
[batch, total_obj] `batch_object_map`.  Here each predicate owns one block of NS floats (NS = padded
max objects per image) and the map is replaced by three small integer vectors that live on the GPU
for the whole batch: objects per image, first object of each image, first ordered pair of each image.
Class names, constructor arguments and methods follow the reference so that code written against it
keeps working; `flat_log_attention()` rebuilds the reference's [P, total_obj] view on request.
"""

from enum import IntEnum

import numpy as np
import torch

from . import ops as L


class Quantifier(IntEnum):          # batch_base_types.py:15-17
    FOR_ALL = 0
    EXISTS = 1


class QuestionType(IntEnum):        # batch_base_types.py:19-24
    BINARY = 0
    QUERY = 1
    STATEMENT = 2
    OBJECT_STATEMENT = 3
    SCENE_GRAPH = 4


class TokenType(IntEnum):           # batch_base_types.py:26-30
    ATTRIBUTE = 0
    RELATION = 1
    NAME = 2
    CATEGORY = 3


def _pad4(n):
    return max(4, (int(n) + 3) // 4 * 4)


_geometry_cache = L.LRUCache(256)
_pair_index_cache = L.LRUCache(8)          # 16 bytes per ordered pair: keep a handful of batch shapes


def _geometry_on_device(device, n_list):
    """Geometry tensors of a batch (object counts, object / pair offsets, identity map), kept on the device per distinct shape:
    a pageable host-to-device copy synchronises the stream, and a ProgramBatch that is run again (every training epoch, every
    benchmark step) would pay that several times per forward."""
    key = (str(device), n_list)
    hit = _geometry_cache.get(key)
    if hit is None:
        n = np.asarray(n_list, np.int64)
        hit = (torch.as_tensor(n.astype(np.int32)).to(device),
               torch.as_tensor(np.concatenate([[0], np.cumsum(n)]).astype(np.int32)).to(device),
               torch.as_tensor(np.concatenate([[0], np.cumsum(n * (n - 1))]).astype(np.int64)).to(device),
               torch.arange(len(n_list), dtype=torch.int32, device=device))
        hit[3]._dfol_sorted = True                               # the identity map (see _lib.require_sorted)
        _geometry_cache[key] = hit
    return L.keep_alive(hit)


class BatchWorld(object):
    """Scene of one ProgramBatch (batch_base_types.py:191-252): likelihood tables + the block geometry."""

    def __init__(self, device, object_num, attribute_features, relation_features, batch_index, meta_data=None,
                 attention_transfer_state_dim=0, object_nums=None, question_image=None):
        """`question_image` (None = the reference's layout, one scene per question): question k looks at scene question_image[k]; the
        object rows (`object_num`, `batch_index`, `object_nums`) then describe the DISTINCT scenes.  Two geometries result: the
        QUESTION-level one the logic kernels index (`_n_list`, `_n_obj`, `_ident`, `_NS`: attention rows are per question) and the
        IMAGE-level one the oracle kernels index (`_img_n_list`, `_img_n_obj`, `_obj_off`, `_pair_off`, `_pair_num`: object / pair rows
        are per scene), linked by `_q_img`.  Without sharing they coincide."""
        self._device = device
        self._lazy = None                 # set by the oracle's needed-columns mode: hidden activations instead of tables
        self._rel_tiles = {}              # relation tiles computed ahead of the execution loop, keyed by id(lowered tokens)
        self._attribute_features = attribute_features
        self._relation_features = relation_features
        self._object_num = int(object_num)
        self._object_image_map = batch_index
        self._meta_data = meta_data
        self._attention_transfer_state_dim = attention_transfer_state_dim

        if object_nums is None:       # the reference syncs here too (batch_base_types.py:202-205)
            bi = batch_index if isinstance(batch_index, torch.Tensor) else torch.as_tensor(batch_index)
            object_nums = torch.bincount(bi.to(torch.int64).cpu()).tolist()
        self._img_n_list = [int(n) for n in object_nums]
        assert sum(self._img_n_list) == self._object_num, "object counts do not add up to object_num"
        n = np.asarray(self._img_n_list, np.int64)
        self._img_n_obj, self._obj_off, self._pair_off, img_ident = _geometry_on_device(device, tuple(self._img_n_list))
        self._pair_num = int((n * (n - 1)).sum())
        self._shared = question_image is not None
        if self._shared:
            self._q_img = np.asarray(question_image, np.int64)
            assert self._q_img.ndim == 1 and (len(self._q_img) == 0 or (0 <= self._q_img.min() and self._q_img.max() < len(self._img_n_list))), \
                "question_image must index the scenes of this batch"
            self._n_list = [self._img_n_list[i] for i in self._q_img]
            self._n_obj, _, _, self._ident = _geometry_on_device(device, tuple(self._n_list))
            from .host_util import upload
            self._q_img_dev = upload(self._q_img.astype(np.int32), device)
        else:
            self._q_img = np.arange(len(self._img_n_list), dtype=np.int64)
            self._n_list, self._n_obj, self._ident, self._q_img_dev = self._img_n_list, self._img_n_obj, img_ident, img_ident
        self._batch_size = len(self._n_list)
        self._NS = _pad4(max(self._n_list))
        self._zeros = None

    def pred_img(self, pred_q):
        """Predicate -> scene map for the oracle kernels (int32 on the device) from a predicate -> question map."""
        if not self._shared:
            return pred_q
        if pred_q is self._ident:
            return self._q_img_dev
        return self._q_img_dev.index_select(0, pred_q)

    # The reference's cached tables ([O, 2335] and [pairs, 333]).  In needed-columns mode they are materialised
    # only if somebody actually reads them (API compatibility); the interpreter itself never does.
    @property
    def _attribute_features(self):
        if self._attr_table is None and self._lazy is not None:
            self._lazy.materialize_tables(self)
        return self._attr_table

    @_attribute_features.setter
    def _attribute_features(self, value):
        self._attr_table = value

    @property
    def _relation_features(self):
        if self._rel_table is None and self._lazy is not None:
            self._lazy.materialize_tables(self)
        return self._rel_table

    @_relation_features.setter
    def _relation_features(self, value):
        self._rel_table = value

    # -- reference API ---------------------------------------------------------------------------
    def to(self, dtype):
        if dtype != torch.float32:
            raise L.DfolError("the MI355X path computes in fp32 (got %s)" % dtype)
        return self

    @property
    def dtype(self):
        return torch.float32

    def batch_size(self):
        return self._batch_size

    def object_num(self):
        return self._object_num

    def word_embedding_dim(self):
        return self._meta_data['embedding'].size()[1]

    def variable_set(self, names, quantifier=Quantifier.EXISTS, log_attention=None):
        return BatchVariableSet(names, self._device, self._object_num, self._batch_size, quantifiers=quantifier,
                                log_attention=log_attention, world=self)

    def attention_state(self, name, state=None):               # batch_base_types.py:249-252
        if state is None:
            z = torch.zeros(self._batch_size, self._attention_transfer_state_dim, dtype=torch.float32, device=self._device)
            state = (z, torch.zeros_like(z))
        return BatchAttentionState(name, self._device, state)

    # -- block helpers ---------------------------------------------------------------------------
    def pair_index(self):
        """(subject row, object row) of every ordered same-image pair, in the reference's order (util.py:87-103)."""
        if getattr(self, "_pair_idx", None) is None:
            key = (str(self._device), tuple(self._img_n_list))
            hit = _pair_index_cache.get(key)
            if hit is None:
                s_all, o_all, first = [], [], 0
                for n in self._img_n_list:
                    s, o = np.nonzero(~np.eye(n, dtype=bool))
                    s_all.append(s + first)
                    o_all.append(o + first)
                    first += n
                hit = (torch.as_tensor(np.concatenate(s_all).astype(np.int64)).to(self._device),
                       torch.as_tensor(np.concatenate(o_all).astype(np.int64)).to(self._device))
                _pair_index_cache[key] = hit
            self._pair_idx = L.keep_alive(hit)
        return self._pair_idx

    def zeros_attention(self):
        if self._zeros is None:
            from .host_util import constant
            self._zeros = constant((self._batch_size, self._NS), 0.0, self._device)
        return self._zeros

    def pred_q(self, predicate_question_map):
        """Normalise a predicate->question map (None / python list / tensor / sparse [P,Q]) to int32 [P] on device.  Whether the map is
        non-decreasing (what the deterministic backward kernels need, _lib.require_sorted) is decided HERE, once per map: on the host for
        lists, with one device read for tensors - remembered per source tensor, so that a train step neither re-converts nor re-reads."""
        m = predicate_question_map
        if m is None:
            return self._ident
        if isinstance(m, torch.Tensor):
            from . import _lib
            memo = self.__dict__.setdefault("_pred_q_memo", {})
            key = (id(m), m._version)
            hit = memo.get(key)
            if hit is None or hit[0] is not m:
                idx = m.coalesce().indices()[1] if m.is_sparse else m
                out = idx.to(device=self._device, dtype=torch.int32)
                # (a capture cannot read the device - require_sorted then refuses an unchecked map; a forward without gradients never asks:
                # the read is a device synchronisation, 2.6 ms per option-list operator of a fresh ProgramBatch)
                if not _lib.capturing() and torch.is_grad_enabled():
                    out._dfol_sorted = bool(out.numel() < 2 or bool((out[1:] >= out[:-1]).all().item()))
                if len(memo) >= 64:
                    memo.clear()
                hit = memo[key] = (m, out)
            return _lib.keep_alive(hit)[1]
        from .host_util import upload                        # (host_util imports this module)
        host = np.asarray(m, np.int32)
        out = upload(host, self._device)                     # a python list (the operators' batch_index): content-keyed, the same tensor per content
        if getattr(out, "_dfol_sorted", None) is None:
            out._dfol_sorted = bool(host.size < 2 or bool((host[1:] >= host[:-1]).all()))
        return out

    def to_flat(self, block, pred_q=None, fill=0.0):
        """[P, NS] blocks -> the reference's flat [P, total_obj] rows (own image filled, the rest `fill`)."""
        pq = (self._ident if pred_q is None else pred_q).cpu().numpy()
        b = block.detach().cpu().numpy()
        off = np.concatenate([[0], np.cumsum(self._n_list)])
        out = np.full((len(pq), self._object_num), fill, np.float32)
        for p, q in enumerate(pq):
            out[p, off[q]:off[q + 1]] = b[p, :self._n_list[q]]
        return torch.from_numpy(out)

    def from_flat(self, flat, pred_q=None):
        """The reference's flat [P, total_obj] rows -> [P, NS] blocks (padding 0)."""
        pq = (self._ident if pred_q is None else pred_q).cpu().numpy()
        f = flat.detach().cpu().numpy()
        off = np.concatenate([[0], np.cumsum(self._n_list)])
        out = np.zeros((len(pq), self._NS), np.float32)
        for p, q in enumerate(pq):
            out[p, :self._n_list[q]] = f[p, off[q]:off[q + 1]]
        return torch.from_numpy(out).to(self._device)


class BatchVariableSet(object):
    """A batch of soft object sets (batch_base_types.py:34-187): log-attention [P, NS], quantifier [P]."""

    def __init__(self, names, device, object_num, batch_size=1, quantifiers=Quantifier.EXISTS, log_attention=None,
                 batch_object_map=None, predicate_question_map=None, base_cumulative_loss=0, prev_variable_sets_num=0,
                 world=None):
        world = world if world is not None else batch_object_map     # the world plays the batch_object_map's role
        assert isinstance(world, BatchWorld), "a BatchWorld must be provided (it carries the block geometry)."
        self._name = names
        self._device = device
        self._object_num = object_num
        self._batch_size = batch_size
        self._base_cumulative_loss = base_cumulative_loss
        self._prev_variable_sets_num = prev_variable_sets_num
        self._world = world
        self._batch_object_map = world

        # _quantifier_host: the same values on the host when they are known there (constructed from python values), so that operators whose
        # selection flags are host data too can pick quantifiers without a launch (GQARelateBatch); None for device-computed quantifiers
        from .host_util import constant, upload
        if isinstance(quantifiers, (int, float, Quantifier)):
            self._quantifier = constant((batch_size,), float(quantifiers), device)
            self._quantifier_host = np.full(batch_size, float(quantifiers), np.float32)
        elif isinstance(quantifiers, (list, tuple, np.ndarray)):
            self._quantifier_host = np.asarray(quantifiers, np.float32)
            self._quantifier = upload(self._quantifier_host, device)
        else:                                                # a tensor: operators hand their input's quantifier on, shadow included
            self._quantifier, self._quantifier_host = quantifiers, getattr(quantifiers, "_dfol_host", None)
        if self._quantifier_host is not None and getattr(self._quantifier, "_dfol_host", None) is None:
            try:
                self._quantifier._dfol_host = self._quantifier_host      # (cached tensors: the same content, so the same shadow, for every user)
            except Exception:
                pass

        if log_attention is None:
            self._log_attention = world.zeros_attention() if batch_size == world._batch_size else \
                torch.zeros(batch_size, world._NS, dtype=torch.float32, device=device)
        else:
            self._log_attention = log_attention
        assert self._log_attention.shape == (batch_size, world._NS), "log-attention must be [batch, NS] blocks"

        # predicate -> question index (int32 [P]); None means predicate p belongs to question p
        self._predicate_question_map = None if predicate_question_map is None else world.pred_q(predicate_question_map)

    def to(self, dtype):
        if dtype != torch.float32:
            raise L.DfolError("the MI355X path computes in fp32 (got %s)" % dtype)
        return self

    @property
    def dtype(self):
        return torch.float32

    @property
    def device(self):
        return self._device

    def object_num(self):
        return self._object_num

    def batch_size(self):
        return self._batch_size

    def pred_q(self):
        return self._world._ident if self._predicate_question_map is None else self._predicate_question_map

    def log_probability(self, hard_mode=False):
        """Quantifier aggregation, batch_base_types.py:103-125 (soft: sums; hard_mode: minimum)."""
        if hard_mode:
            return L.quantify_hard(self._log_attention, self._quantifier, self.pred_q(), self._world._n_obj, self._world.object_num())
        return L.quantify_fwd(self._log_attention, self._quantifier, self.pred_q(), self._world._n_obj)

    def cumulative_loss(self):          # batch_base_types.py:127-131
        return 0

    def mean_cumulative_loss(self):
        return self.cumulative_loss() / (self._prev_variable_sets_num + 1)

    def get_attention(self):
        return self._log_attention.exp()

    def flat_log_attention(self):
        """The reference's [P, total_obj] view (cross-image entries 0), for traces and visualisers."""
        return self._world.to_flat(self._log_attention, self.pred_q())

    def gate(self, variable_set, flag):
        """Per-question select (batch_base_types.py:149-168): rows with flag 1 come from self, the rest from `variable_set`."""
        if isinstance(flag, torch.Tensor):
            host_flag = getattr(flag, "_host", None)
            if host_flag is None:
                host_flag = flag.cpu().numpy().tolist()
        else:
            host_flag = [0 if f is None else f for f in flag]
        names = [x if f > 0 else y for x, y, f in zip(self._name, variable_set._name, host_flag)]
        if len(host_flag) == self._log_attention.shape[0] and all(f == 1 for f in host_flag):
            att, quant = self._log_attention, self._quantifier       # every row takes self: g x + (1 - g) y = x, no launch
        else:
            if isinstance(flag, torch.Tensor):
                g = flag.to(torch.float32)
            else:
                from .host_util import upload
                g = upload(np.asarray([float(f) for f in host_flag], np.float32), self._device)
            att, quant = L.gate(self._log_attention, variable_set._log_attention, self._quantifier, variable_set._quantifier, g)
        out = BatchVariableSet(names, self._device, self._object_num, self._batch_size, quantifiers=quant, log_attention=att,
                               world=self._world)
        out._predicate_question_map = self._predicate_question_map
        return out

    def apply_modulations(self, modulations, input_variable_set, predicate_question_map=None):
        """Attention calibration (batch_base_types.py:170-187) with the [P, 4] modulations of the attention-output network."""
        if modulations is not None:
            if modulations.size()[1] != 4:
                raise NotImplementedError("only the 4-column modulations the reference builds (output_dim = 4) are supported")
            self._log_attention = L.modulate(self._log_attention, modulations, self.pred_q(), self._world._n_obj)
        return self

    def __repr__(self):
        return "Object set of %d objects in %d blocks of %d" % (self._object_num, self._batch_size, self._world._NS)


class BatchAttentionState(object):
    """LSTM (h, c) state of the attention-calibration passes, one row per question (batch_base_types.py:256-310)."""

    def __init__(self, name, device, state, set_zeros=False):
        self._name = name
        self._device = device
        self._state = (torch.zeros_like(state[0]), torch.zeros_like(state[1])) if set_zeros else state

    def to(self, dtype):
        return self

    @property
    def dtype(self):
        return self._state[0].dtype

    @property
    def device(self):
        return self._device

    def state_size(self):
        return self._state[0].size()[1]

    def gate(self, attention_state, flag):                     # :279-298
        if isinstance(flag, torch.Tensor):
            g = flag.to(torch.float32)
            host = getattr(flag, "_host", None)
            if host is None:
                host = flag.cpu().numpy().tolist()
        else:
            host = [0 if f is None else f for f in flag]
            from .host_util import upload
            g = upload(np.asarray([float(f) for f in host], np.float32), self._device)
        names = [x if f > 0 else y for x, y, f in zip(self._name, attention_state._name, host)]
        binary = all(f in (0, 1, 0.0, 1.0, True, False) for f in host)
        if binary and all(f > 0 for f in host):                 # the flags are host data: a uniform gate costs no launch ...
            return BatchAttentionState(names, self._device, self._state)
        if binary and not any(f > 0 for f in host):
            return BatchAttentionState(names, self._device, attention_state._state)
        if binary:                                              # ... and a 0/1 gate is a select (g x + (1 - g) y exactly, for finite states)
            from .host_util import upload
            pick = upload(np.asarray([f > 0 for f in host], np.bool_), self._device).unsqueeze(1)
            return BatchAttentionState(names, self._device, (torch.where(pick, self._state[0], attention_state._state[0]),
                                                             torch.where(pick, self._state[1], attention_state._state[1])))
        g = g.unsqueeze(1)
        state0 = self._state[0] * g + attention_state._state[0] * (1.0 - g)
        state1 = self._state[1] * g + attention_state._state[1] * (1.0 - g)
        return BatchAttentionState(names, self._device, (state0, state1))

    def expand(self, predicate_question_map):                  # mm(pqm, state)  :300-304
        idx = predicate_question_map.to(torch.int64)
        return BatchAttentionState(self._name, self._device, (self._state[0][idx], self._state[1][idx]))

    def squeeze(self, predicate_question_map, question_num=None, host=None):   # mm(pqm^T, state)  :306-310
        idx = predicate_question_map.to(torch.int64)
        Q = int(question_num if question_num is not None else int(idx.max()) + 1)
        host = host if host is not None else getattr(predicate_question_map, "_host", None)
        host = None if host is None else list(host)
        s0 = self._state[0]
        if host is not None and s0.is_cuda and s0.dtype == torch.float32 and not (torch.is_grad_enabled() and (s0.requires_grad or self._state[1].requires_grad)) \
                and all(b >= a for a, b in zip(host, host[1:])):
            # the predicates of a question are consecutive (what flatten_list produces): a segmented row sum in a FIXED order - index_add_ on
            # the device is an atomic add, its rounding differs from run to run (and from the native executor's, which takes this kernel)
            from .host_util import upload
            counts = np.bincount(np.asarray(host, np.int64), minlength=Q)
            seg = upload(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), s0.device)
            return BatchAttentionState(self._name, self._device, (L.segment_sum_rows(s0.contiguous(), seg), L.segment_sum_rows(self._state[1].contiguous(), seg)))
        z = lambda s: torch.zeros(Q, s.shape[1], dtype=s.dtype, device=s.device).index_add_(0, idx, s)
        return BatchAttentionState(self._name, self._device, (z(self._state[0]), z(self._state[1])))
