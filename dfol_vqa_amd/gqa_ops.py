"""The GQA operator plugins (reference: src/nsvqa/nn/interpreter/batch_gqa_ops.py).

Same class names, constructor keywords and forward signatures as the reference's registry entries
(batch_gqa_interpreter.py:27-65).  Every forward is a handful of kernel launches on block-layout
state; the only host work is string bookkeeping (names, answers).
"""

import json
import math
import re
import threading

import numpy as np
import torch

from . import ops as L
from .fol_types import BatchVariableSet, Quantifier, QuestionType, TokenType
from .host_util import TokenList, find_max_ind, flatten_list, unflatten_list, upload
from .logic_ops import BatchOperatorBase, FilterBatch, RelateBatch

UNKNOWN = 'UNKNOWN'


class GQAOntology(object):
    """Vocabulary / attribute / class / relation metadata (batch_gqa_ops.py:25-148).  Attribute names follow the
    reference (including its spelling `_relation_reveresed_index`) because other components read them."""

    def __init__(self, attribute_json_path, class_json_path, vocab_json_file, embedding_file=None, relation_json_path=None,
                 frequency_json_path=None):
        with open(attribute_json_path) as f:
            self._attribute_dict = json.load(f)
        with open(class_json_path) as f:
            self._class_dict = json.load(f)
        self._nouns = list(set(sum(self._class_dict.values(), [])))
        self._adjectives = list(set(sum(self._attribute_dict.values(), [])))
        if frequency_json_path is not None:
            with open(frequency_json_path) as f:
                self._frequencies = json.load(f)
        self._inverted_class_dict = {}
        for parent, children in self._class_dict.items():
            for child in children:
                self._inverted_class_dict.setdefault(child, []).append(parent)

        self._embedding_file = embedding_file
        self._word_index = {}
        self._embedding_rows = None
        if embedding_file is not None:
            offsets = []
            with open(embedding_file, 'rb') as f:          # one pass: remember where each word's line starts
                pos = 0
                for i, line in enumerate(f):
                    self._word_index[line.split(b' ', 1)[0].decode('utf8')] = i
                    offsets.append(pos)
                    pos += len(line)
                    if i == 0:
                        self._embedding_dim = len(line.decode('utf8').rstrip('\n').split(' ')) - 1
            self._line_offsets = offsets

        with open(vocab_json_file) as f:
            self._vocabulary = json.load(f)
        a2i = self._vocabulary['arg_to_idx']
        self._noun_index = sorted(a2i[n] - 1 for n in self._nouns if n in a2i)

        if relation_json_path is not None:
            with open(relation_json_path) as f:
                self._relations = list(set(json.load(f)))
            self._relation_index = sorted(a2i[r] - 1 for r in self._relations if r in a2i)
            rel = set(self._relation_index)
            self._attribute_index = [i for i in range(len(a2i)) if i not in rel]
            self._attributes = [self._vocabulary['idx_to_arg'][i] for i in self._attribute_index]
            self._relation_reveresed_index = {i: j for j, i in enumerate(self._relation_index)}
            self._attribute_reveresed_index = {i: j for j, i in enumerate(self._attribute_index)}
            nouns = set(self._nouns)
            self._noun_subindex = sorted(j for j, i in enumerate(self._attribute_index) if self._vocabulary['idx_to_arg'][i] in nouns)
            noun_sub = set(self._noun_subindex)
            self._non_noun_subindex = [j for j in range(len(self._attribute_index)) if j not in noun_sub]

    def get_family_subindex(self, attribute):
        if attribute not in self._inverted_class_dict:
            return []
        children = set()
        for parent in self._inverted_class_dict[attribute]:
            children.update(self._class_dict[parent])
        return [j for j, a in enumerate(self._attributes) if a in children]

    def encode_token(self, token):
        t = str(token).lower().strip()
        negated = re.match(r"not\((\w|\s)+\)", t) is not None
        if negated:
            t = t[4:-1]
        return (-1 if negated else 1) * self._vocabulary['arg_to_idx'][t]

    def decode_token(self, idx):
        t = self._vocabulary['idx_to_arg'][abs(int(idx)) - 1]
        if t == 'true':
            return True
        if t == 'false':
            return False
        return t if idx >= 0 else 'not(' + t + ')'

    def encode_op(self, op):
        return self._vocabulary['op_to_idx'][op.lower().strip()]

    def decode_op(self, idx):
        return self._vocabulary['idx_to_op'][idx - 1]

    def encode_img_id(self, img_id):
        return self._vocabulary['img_to_idx'][img_id.lower().strip()]

    def decode_img_id(self, idx):
        return self._vocabulary['idx_to_img'][idx - 1]

    def query_attribute(self, attr_name):
        return self._attribute_dict.get(attr_name, UNKNOWN)

    def query_class(self, class_name):
        return self._class_dict.get(class_name, UNKNOWN)

    def query(self, name):                                    # batch_gqa_ops.py:114-124
        if name in self._attribute_dict:
            return self._attribute_dict[name]
        if name in self._class_dict:
            return self._class_dict[name]
        if name is None:
            return [None]
        if name == 'entity':
            return self._nouns
        return [name]

    def is_noun(self, name):
        return name in self._nouns

    def is_adjective(self, name):
        return name in self._adjectives

    def is_relation(self, name):
        return name in self._relations

    def get_embeddings(self, names):                          # batch_gqa_ops.py:135-148
        if self._embedding_file is None:
            return None
        res = np.zeros((len(names), self._embedding_dim), dtype=np.float32)
        with open(self._embedding_file, 'rb') as f:
            for i, name in enumerate(names):
                for t in name.split(' '):
                    if t in self._word_index:
                        f.seek(self._line_offsets[self._word_index[t]])
                        line = f.readline().decode('utf8').rstrip('\n')
                        res[i, :] += np.array([float(x) for x in line.split(' ')[1:]])
        return res


# ---------------------------------------------------------------------------------------------------
class GQABatchOperatorBase(BatchOperatorBase):

    def __init__(self, oracle, ontology, is_terminal, fan_in, fan_out):
        super(GQABatchOperatorBase, self).__init__(oracle, is_terminal, fan_in, fan_out)
        self._ontology = ontology


class _Deferred(threading.local):
    """While `queue` is a list, terminal operators do not read their log-probabilities back: they return empty answer lists and
    queue a closure that fills them.  The interpreter runs the closures after the last ProgramBatch has been enqueued (one
    device->host synchronisation per forward instead of one per ProgramBatch), or after replaying a captured graph."""
    queue = None
    outputs = None           # a list while a graph is captured: the device tensors the queued closures will read back
    snapshot = None          # {id(device tensor): numpy copy} while the answers of ONE replay of a pipelined graph are decoded


DEFERRED = _Deferred()


def _host(t):
    """The host copy a terminal operator decodes its answers from: the tensor read back now (one synchronisation), or - a pipelined graph replay
    (interpreter.GraphedForward.submit / collect) - the copy that replay left in pinned memory behind its own launches."""
    snap = DEFERRED.snapshot
    if snap is not None:
        hit = snap.get(id(t))
        if hit is not None:
            return hit
    return t.detach().cpu().numpy()


def _reads(*tensors):
    if DEFERRED.outputs is not None:
        DEFERRED.outputs.extend(t for t in tensors if t is not None)


def _answers(compute):
    """compute() -> (answer, answer_log_probability) from host copies; run now, or queued (see _Deferred)."""
    if DEFERRED.queue is None:
        return compute()
    answer, alp = [], []

    def fill():
        a, l = compute()
        answer[:] = a
        alp[:] = l
    DEFERRED.queue.append(fill)
    return answer, alp


def _binary_answer(log_probability, batch_size, give_answer):
    """yes/no from p > 0.5 plus the log-probability of the answer given (e.g. batch_gqa_ops.py:404-407)."""
    if not give_answer:
        return [], []

    def compute():
        # the one device->host sync of a binary op; safe_exp (util.py:19) on the host copy instead of one more launch
        probability = np.exp(_host(log_probability).astype(np.float32)).tolist()
        answer = [['yes'] if probability[i] > 0.5 else ['no'] for i in range(batch_size)]
        alp = [[math.log(probability[i])] if probability[i] > 0.5 else [math.log(1 - probability[i])] for i in range(batch_size)]
        return answer, alp
    _reads(log_probability)
    return _answers(compute)


def _result(answer, log_probability, options, variable_set, qtype, cumulative_loss, variable_sets_num, answer_log_probability):
    return {'answer': answer, 'log_probability': log_probability, 'options': options, 'variable_set': variable_set, 'type': qtype,
            'cumulative_loss': cumulative_loss, 'variable_sets_num': variable_sets_num, 'answer_log_probability': answer_log_probability}


def _kw(kw):
    return dict(trainable_module_type=kw.get('trainable_module_type'), feature_dim=kw.get('feature_dim', 1),
                trainable_gate=kw.get('trainable_gate', False), forward_attention_network=kw.get('forward_attention_network'),
                backward_attention_network=kw.get('backward_attention_network'), attention_output_network=kw.get('attention_output_network'))


def _select_names(attribute_list, batch_size):
    """Names and filter tokens of a select (batch_gqa_ops.py:171-180)."""
    if attribute_list is None:
        return ["entity"] * batch_size, None
    name = ["entity" if a is None or a.lower() in ("_", "scene") else a for a in attribute_list][:batch_size]
    att = [None if a is None or a.lower() in ("_", "scene") else a for a in attribute_list][:batch_size]
    return name, att


class GQASelectBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:160-203."""

    def __init__(self, oracle, ontology, **kw):
        super(GQASelectBatch, self).__init__(oracle, ontology, is_terminal=False, fan_in=0, fan_out=1)
        self._filter = FilterBatch(oracle, **_kw(kw))

    def forward(self, op_id, world, attribute_list=None, give_answer=True, predicate_question_map=None, likelihood_threshold=0,
                hard_mode=False):
        name, att = _select_names(attribute_list, world.batch_size())
        x = world.variable_set(name, quantifier=Quantifier.EXISTS)
        if att is None or all(a is None for a in att):
            return x
        tokens = attribute_list if (len(attribute_list) == len(att) and getattr(attribute_list, "lowered", None) is not None) else att
        return self._filter(op_id, world, x, tokens)


class GQAFilterBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:314-350."""

    def __init__(self, oracle, ontology, **kw):
        super(GQAFilterBatch, self).__init__(oracle, ontology, is_terminal=False, fan_in=1, fan_out=1)
        self._filter = FilterBatch(oracle, **_kw(kw))

    def forward(self, op_id, world, variable_set, attribute_list, give_answer=True, predicate_question_map=None,
                likelihood_threshold=0, hard_mode=False):
        return self._filter(op_id, world, variable_set, attribute_list)


def _subject_flags(is_subject, device):
    host = [0.0 if f is None else float(f) for f in is_subject]
    t = getattr(is_subject, "device_flags", None)
    if t is None or t.device != torch.device(device):
        t = upload(np.asarray(host, np.float32), device)
    t._host = host
    return t, host


class GQARelateBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:354-390: x = select(name); (subject, object) = (x, prev) or (prev, x) per question;
    relate; keep the posterior of whichever variable `prev` was not."""

    def __init__(self, oracle, ontology, **kw):
        super(GQARelateBatch, self).__init__(oracle, ontology, is_terminal=False, fan_in=1, fan_out=1)
        self._gqa_select = GQASelectBatch(oracle, ontology, **kw)
        self._relate = RelateBatch(oracle, **_kw(kw))

    def forward(self, op_id, world, variable_set, relation_list, is_subject, attribute_list=None, give_answer=True,
                predicate_question_map=None, likelihood_threshold=0, hard_mode=False):
        x = self._gqa_select(op_id, world, attribute_list, give_answer, predicate_question_map, likelihood_threshold)
        flag, host = _subject_flags(is_subject, world._device)
        fused = self._forward_fused(op_id, world, x, variable_set, relation_list, flag, host)
        if fused is not None:
            return fused
        subject_set = x.gate(variable_set, flag)
        object_set = variable_set.gate(x, flag)
        want = upload(np.asarray([L.WANT_SUBJECT if f > 0 else L.WANT_OBJECT for f in host], np.uint8), world._device)
        subject_set, object_set = self._relate(op_id, world, subject_set, object_set, relation_list, want=want)
        return subject_set.gate(object_set, flag)

    def _forward_fused(self, op_id, world, x, prev, relation_list, flag, host_flags=None):
        """Inference fast path: the three gates and the arity-2 cell in ONE launch (dfol_relate_one_fwd_f32), on tiles the
        oracle prefetched with the summed-out variable along rows.  Same result as the generic route below it."""
        oracle = self._oracle
        if torch.is_grad_enabled() and (x._log_attention.requires_grad or prev._log_attention.requires_grad):
            return None
        rel = self._relate
        calibrated = op_id in rel._subject_modulations
        if calibrated != (op_id in rel._object_modulations):
            return None                                        # one-sided calibration: RelateBatch.forward handles it
        if calibrated and (rel._subject_modulations[op_id] is None or rel._object_modulations[op_id] is None):
            return None
        if not hasattr(oracle, "oriented_tiles") or x.batch_size() != prev.batch_size() or prev._predicate_question_map is not None:
            return None
        low = getattr(relation_list, "lowered", None)
        tiles = None if low is None else oracle.oriented_tiles(world, low, relation_list)
        if tiles is None:
            return None
        _, neg_dev, valid_dev = low.on(world._device)
        kernel = L.relate_one_fwd_bf16 if tiles.dtype == torch.bfloat16 else L.relate_one_fwd
        post = kernel(x._log_attention, prev._log_attention, tiles, world._ident, world._n_obj, prev._quantifier,
                      neg_dev if low.any_neg else None, None if low.all_valid else valid_dev, lone_forall_identity=(x.batch_size() == 1))
        if calibrated:
            # RelateBatch.forward calibrates both posteriors (batch_base_ops.py:588-594) and GQARelateBatch keeps one per question:
            # calibrate the kept one with that side's modulations
            mods = torch.where(flag.unsqueeze(1) > 0, rel._subject_modulations.pop(op_id), rel._object_modulations.pop(op_id))
            post = L.modulate(post, mods, world._ident, world._n_obj)
        # both posteriors carry the subject's quantifier (:571-586); the flags are host data, so the selection costs at most one launch
        if host_flags is not None and all(f > 0 for f in host_flags):
            quant = x._quantifier
        elif host_flags is not None and not any(f > 0 for f in host_flags):
            quant = prev._quantifier
        elif host_flags is not None and x._quantifier_host is not None and prev._quantifier_host is not None:
            quant = np.where(np.asarray([f > 0 for f in host_flags]), x._quantifier_host, prev._quantifier_host)     # host values: no launch
        elif host_flags is not None:
            quant = torch.where(upload(np.asarray([f > 0 for f in host_flags], np.bool_), world._device), x._quantifier, prev._quantifier)
        else:
            quant = torch.where(flag > 0, x._quantifier, prev._quantifier)
        return BatchVariableSet(x._name, world._device, x.object_num(), x.batch_size(), quantifiers=quant, log_attention=post, world=world,
                                prev_variable_sets_num=x._prev_variable_sets_num + prev._prev_variable_sets_num + 1)


class GQAExistBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:394-413."""

    def __init__(self, oracle, ontology):
        super(GQAExistBatch, self).__init__(oracle, ontology, is_terminal=True, fan_in=1, fan_out=0)

    def forward(self, op_id, world, variable_set, give_answer=True, predicate_question_map=None, likelihood_threshold=0, hard_mode=False):
        log_probability = variable_set.log_probability(give_answer and hard_mode)
        answer, alp = _binary_answer(log_probability, variable_set.batch_size(), give_answer)
        return _result(answer, log_probability, ['no', 'yes'], variable_set, QuestionType.BINARY, variable_set.cumulative_loss(),
                       variable_set._prev_variable_sets_num + 1, alp)


class GQAEndBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:768-783."""

    def __init__(self, oracle, ontology):
        super(GQAEndBatch, self).__init__(oracle, ontology, is_terminal=True, fan_in=1, fan_out=0)

    def forward(self, op_id, world, variable_set, give_answer=True, predicate_question_map=None, likelihood_threshold=0, hard_mode=False):
        answer = [[name] for name in variable_set._name] if give_answer else []
        return _result(answer, variable_set.log_probability(give_answer and hard_mode), [], variable_set, QuestionType.STATEMENT,
                       variable_set.cumulative_loss(), variable_set._prev_variable_sets_num + 1, [])


def _seg_off(batch_index, question_num, device):
    counts = np.bincount(np.asarray(batch_index, np.int64), minlength=question_num)
    return upload(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), device)


class GQAVerifyAttrsBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:444-477: un-normalised Filter per attribute, posteriors summed per question, then EXISTS."""

    def __init__(self, oracle, ontology, **kw):
        super(GQAVerifyAttrsBatch, self).__init__(oracle, ontology, is_terminal=True, fan_in=1, fan_out=0)
        self._filter = FilterBatch(oracle, **_kw(kw))

    def forward(self, op_id, world, variable_set, attribute_list_list, give_answer=True, predicate_question_map=None,
                likelihood_threshold=0, hard_mode=False):
        attribute_list, batch_index = flatten_list(attribute_list_list)
        x = self._filter(op_id, world, variable_set, attribute_list, batch_index, normalized_probability=False)
        log_attention = L.segment_sum_rows(x._log_attention, _seg_off(batch_index, variable_set._batch_size, world._device))   # :457
        y = BatchVariableSet(variable_set._name, variable_set._device, variable_set._object_num, batch_size=variable_set._batch_size,
                             quantifiers=variable_set._quantifier, log_attention=log_attention, world=world,
                             base_cumulative_loss=x._base_cumulative_loss, prev_variable_sets_num=x._prev_variable_sets_num)
        log_probability = y.log_probability(give_answer and hard_mode)
        answer, alp = _binary_answer(log_probability, variable_set.batch_size(), give_answer)
        return _result(answer, log_probability, ['no', 'yes'], y, QuestionType.BINARY, y.cumulative_loss(), y._prev_variable_sets_num + 1, alp)


class GQAVerifyRelBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:481-504."""

    def __init__(self, oracle, ontology, **kw):
        super(GQAVerifyRelBatch, self).__init__(oracle, ontology, is_terminal=True, fan_in=1, fan_out=0)
        self._gqa_relate = GQARelateBatch(oracle, ontology, **kw)

    def forward(self, op_id, world, variable_set, relation_list, is_subject, attribute_list=None, give_answer=True,
                predicate_question_map=None, likelihood_threshold=0, hard_mode=False):
        x = self._gqa_relate(op_id, world, variable_set, relation_list, is_subject, attribute_list, give_answer, predicate_question_map,
                             likelihood_threshold)
        log_probability = x.log_probability(give_answer and hard_mode)
        answer, alp = _binary_answer(log_probability, variable_set.batch_size(), give_answer)
        return _result(answer, log_probability, ['no', 'yes'], x, QuestionType.BINARY, x.cumulative_loss(), x._prev_variable_sets_num + 1, alp)


def _choose_answer(log_probability, x, option_list, batch_index, question_num, likelihood_threshold, give_answer):
    if not give_answer:
        return [], []
    bi = np.asarray(batch_index)
    if log_probability.is_cuda and (len(bi) < 2 or bool(np.all(bi[1:] >= bi[:-1]))) and len(bi) > 0 and int(bi[-1]) < question_num:
        # util.find_max_ind on the device, queued with the operator's other launches (capturable); the host later reads the flags
        counts = np.bincount(bi, minlength=question_num)
        seg_off = upload(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), log_probability.device)
        dev_flags = L.find_max_ind(log_probability.detach().contiguous(), seg_off, float(likelihood_threshold))

        def compute():
            flags = _host(dev_flags).tolist()
            return unflatten_list(option_list, batch_index, flags), unflatten_list(_host(log_probability).tolist(), batch_index, flags)
        _reads(dev_flags, log_probability)
        return _answers(compute)

    def compute():
        flags = find_max_ind(log_probability, bi, question_num, likelihood_threshold).tolist()   # util.py:64-66
        return unflatten_list(option_list, batch_index, flags), unflatten_list(log_probability.detach().cpu().numpy().tolist(), batch_index, flags)
    return _answers(compute)


class GQAChooseAttrBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:207-232."""

    def __init__(self, oracle, ontology, **kw):
        super(GQAChooseAttrBatch, self).__init__(oracle, ontology, is_terminal=True, fan_in=1, fan_out=0)
        self._filter = FilterBatch(oracle, **_kw(kw))

    def forward(self, op_id, world, variable_set, attribute_list_list, give_answer=True, predicate_question_map=None,
                likelihood_threshold=0, hard_mode=False):
        attribute_list, batch_index = flatten_list(attribute_list_list)
        x = self._filter(op_id, world, variable_set, attribute_list, batch_index)
        log_probability = x.log_probability(give_answer and hard_mode)
        answer, alp = _choose_answer(log_probability, x, attribute_list, batch_index, variable_set.batch_size(), likelihood_threshold, give_answer)
        return _result(answer, log_probability, attribute_list_list, x, QuestionType.QUERY, x.cumulative_loss(), x._prev_variable_sets_num + 1, alp)


class GQAQueryAttrBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:296-310."""

    def __init__(self, oracle, ontology, **kw):
        super(GQAQueryAttrBatch, self).__init__(oracle, ontology, is_terminal=True, fan_in=1, fan_out=0)
        self._gqa_choose_attr = GQAChooseAttrBatch(oracle, ontology, **kw)

    def forward(self, op_id, world, variable_set, category_list, give_answer=True, predicate_question_map=None, likelihood_threshold=0,
                hard_mode=False):
        attribute_list_list = [self._ontology.query(category if category not in ['name', 'type'] else n)
                               for category, n in zip(category_list, variable_set._name)]
        return self._gqa_choose_attr(op_id, world, variable_set, attribute_list_list, give_answer, predicate_question_map, likelihood_threshold)


class GQAChooseRelBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:236-292."""

    def __init__(self, oracle, ontology, **kw):
        super(GQAChooseRelBatch, self).__init__(oracle, ontology, is_terminal=True, fan_in=1, fan_out=0)
        self._gqa_select = GQASelectBatch(oracle, ontology, **kw)
        self._relate = RelateBatch(oracle, **_kw(kw))

    def forward(self, op_id, world, variable_set, relation_list_list, is_subject, attribute_list=None, give_answer=True,
                predicate_question_map=None, likelihood_threshold=0, hard_mode=False):
        relation_list, batch_index = flatten_list(relation_list_list)
        pre = getattr(relation_list_list, "flat_lowered", None)        # lowered (and its tiles prefetched) by the oracle
        if pre is not None and len(pre.cols) == len(relation_list):
            relation_list = TokenList(relation_list)
            relation_list.lowered, relation_list.lowered_type = pre, TokenType.RELATION
        x = self._gqa_select(op_id, world, attribute_list, give_answer, predicate_question_map, likelihood_threshold)
        flag, host = _subject_flags(is_subject, world._device)
        subject_set = x.gate(variable_set, flag)
        object_set = variable_set.gate(x, flag)
        pred_host = [host[q] for q in batch_index]                                    # pqm @ is_subject  (:254-255)
        want = upload(np.asarray([L.WANT_SUBJECT if f > 0 else L.WANT_OBJECT for f in pred_host], np.uint8), world._device)
        subject_set, object_set = self._relate(op_id, world, subject_set, object_set, relation_list, batch_index, want=want)
        pflag = upload(np.asarray(pred_host, np.float32), world._device)     # (memoised by content, so `_host` below always matches)
        pflag._host = pred_host
        x = subject_set.gate(object_set, pflag)
        log_probability = x.log_probability(give_answer and hard_mode)
        answer, alp = _choose_answer(log_probability, x, relation_list, batch_index, variable_set.batch_size(), likelihood_threshold, give_answer)
        return _result(answer, log_probability, relation_list_list, x, QuestionType.QUERY, x.cumulative_loss(), x._prev_variable_sets_num + 1, alp)


def _lp_of(v, hard):
    return v.log_probability(hard) if isinstance(v, BatchVariableSet) else v['log_probability']


def _loss_num(v):
    if isinstance(v, BatchVariableSet):
        return v.cumulative_loss(), v._prev_variable_sets_num + 1
    return v['cumulative_loss'], v['variable_sets_num']


class _GQABinaryLogic(GQABatchOperatorBase):

    _op = None

    def __init__(self, oracle, ontology):
        super(_GQABinaryLogic, self).__init__(oracle, ontology, is_terminal=True, fan_in=2, fan_out=0)

    def forward(self, op_id, world, variable_set1, variable_set2, give_answer=True, predicate_question_map=None, likelihood_threshold=0,
                hard_mode=False):
        v1, v2 = _lp_of(variable_set1, give_answer and hard_mode), _lp_of(variable_set2, give_answer and hard_mode)
        log_probability = L.logic(self._op, v1, v2)
        answer, alp = _binary_answer(log_probability, v1.numel(), give_answer)
        (c1, n1), (c2, n2) = _loss_num(variable_set1), _loss_num(variable_set2)
        return _result(answer, log_probability, ['no', 'yes'], None, QuestionType.BINARY, c1 + c2, n1 + n2, alp)


class GQAAndBatch(_GQABinaryLogic):
    """batch_gqa_ops.py:508-537."""
    _op = L.LOGIC_AND


class GQAOrBatch(_GQABinaryLogic):
    """batch_gqa_ops.py:541-570."""
    _op = L.LOGIC_OR


def _category_options(ontology, category_list, names):
    lists = [ontology.query(category if category not in ['name', 'type'] else n) for category, n in zip(category_list, names)]
    return flatten_list(lists)


class GQAAllSameBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:574-613."""

    def __init__(self, oracle, ontology, **kw):
        super(GQAAllSameBatch, self).__init__(oracle, ontology, is_terminal=True, fan_in=1, fan_out=0)
        self._filter = FilterBatch(oracle, **_kw(kw))

    def forward(self, op_id, world, variable_set, category_list, give_answer=True, predicate_question_map=None, likelihood_threshold=0,
                hard_mode=False):
        attribute_list, batch_index = _category_options(self._ontology, category_list, variable_set._name)
        x = self._filter(op_id, world, variable_set, attribute_list, batch_index)
        pred_q = x.pred_q()
        log_posterior = L.implication(variable_set._log_attention, x._log_attention, pred_q, world._n_obj)          # :588-589
        temp = BatchVariableSet(x._name, variable_set.device, x.object_num(), batch_size=len(attribute_list), quantifiers=Quantifier.FOR_ALL,
                                log_attention=log_posterior, world=world, predicate_question_map=pred_q)
        log_probability = temp.log_probability(give_answer and hard_mode)
        log_probability = L.segment_or(log_probability, _seg_off(batch_index, variable_set.batch_size(), world._device),
                                       as_written=getattr(self, "_or_as_written", False))                                   # :597-598
        answer, alp = _binary_answer(log_probability, variable_set.batch_size(), give_answer)
        return _result(answer, log_probability, ['no', 'yes'], None, QuestionType.BINARY, x.cumulative_loss(), x._prev_variable_sets_num + 1, alp)


class GQAAllDifferentBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:617-642 (the logical NOT of all_same)."""

    def __init__(self, oracle, ontology, **kw):
        super(GQAAllDifferentBatch, self).__init__(oracle, ontology, is_terminal=True, fan_in=1, fan_out=0)
        self._gqa_all_same = GQAAllSameBatch(oracle, ontology, **kw)
        self._gqa_all_same._or_as_written = True              # its aggregate is negated next: the reference's fp32 formula (dfol_segment_or_ref_f32)

    def forward(self, op_id, world, variable_set, category_list, give_answer=True, predicate_question_map=None, likelihood_threshold=0,
                hard_mode=False):
        all_same = self._gqa_all_same(op_id, world, variable_set, category_list, give_answer, predicate_question_map, likelihood_threshold)
        log_probability = L.logic(L.LOGIC_NOT, all_same['log_probability'])
        answer, alp = _binary_answer(log_probability, variable_set.batch_size(), give_answer)
        return _result(answer, log_probability, ['no', 'yes'], all_same['variable_set'], QuestionType.BINARY, all_same['cumulative_loss'],
                       all_same['variable_sets_num'], alp)


class GQATwoSameBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:646-690."""

    def __init__(self, oracle, ontology, **kw):
        super(GQATwoSameBatch, self).__init__(oracle, ontology, is_terminal=True, fan_in=2, fan_out=0)
        self._filter = FilterBatch(oracle, **_kw(kw))

    def forward(self, op_id, world, variable_set1, variable_set2, category_list, give_answer=True, predicate_question_map=None,
                likelihood_threshold=0, hard_mode=False):
        attribute_list, batch_index = _category_options(self._ontology, category_list, variable_set1._name)
        x1 = self._filter(op_id + ':0', world, variable_set1, attribute_list, batch_index)
        x2 = self._filter(op_id + ':1', world, variable_set2, attribute_list, batch_index)
        hard = give_answer and hard_mode
        log_probability = L.logic(L.LOGIC_AND, x1.log_probability(hard), x2.log_probability(hard))
        log_probability = L.segment_or(log_probability, _seg_off(batch_index, variable_set1.batch_size(), world._device),
                                       as_written=getattr(self, "_or_as_written", False))                                   # :664-665
        answer, alp = _binary_answer(log_probability, variable_set1.batch_size(), give_answer)
        return _result(answer, log_probability, ['no', 'yes'], None, QuestionType.BINARY, x1.cumulative_loss() + x2.cumulative_loss(),
                       x1._prev_variable_sets_num + x2._prev_variable_sets_num + 2, alp)


class GQATwoDifferentBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:694-717."""

    def __init__(self, oracle, ontology, **kw):
        super(GQATwoDifferentBatch, self).__init__(oracle, ontology, is_terminal=True, fan_in=2, fan_out=0)
        self._gqa_two_same = GQATwoSameBatch(oracle, ontology, **kw)
        self._gqa_two_same._or_as_written = True              # negated next, as all_different's

    def forward(self, op_id, world, variable_set1, variable_set2, category_list, give_answer=True, predicate_question_map=None,
                likelihood_threshold=0, hard_mode=False):
        two_same = self._gqa_two_same(op_id, world, variable_set1, variable_set2, category_list, give_answer, predicate_question_map,
                                      likelihood_threshold)
        log_probability = L.logic(L.LOGIC_NOT, two_same['log_probability'])
        answer, alp = _binary_answer(log_probability, variable_set1.batch_size(), give_answer)
        return _result(answer, log_probability, ['no', 'yes'], None, QuestionType.BINARY, two_same['cumulative_loss'],
                       two_same['variable_sets_num'], alp)


class GQACompareBatch(GQABatchOperatorBase):
    """batch_gqa_ops.py:721-764."""

    def __init__(self, oracle, ontology, **kw):
        super(GQACompareBatch, self).__init__(oracle, ontology, is_terminal=True, fan_in=2, fan_out=0)
        self._filter = FilterBatch(oracle, **_kw(kw))

    def forward(self, op_id, world, variable_set1, variable_set2, attribute_list, is_less, give_answer=True, predicate_question_map=None,
                likelihood_threshold=0, hard_mode=False):
        x1 = self._filter(op_id + ':0', world, variable_set1, attribute_list)
        x2 = self._filter(op_id + ':1', world, variable_set2, attribute_list)
        hard = give_answer and hard_mode
        alpha = upload(np.asarray([float(bool(v)) for v in is_less], np.float32), variable_set1.device)
        log_probability = L.compare(x1.log_probability(hard), x2.log_probability(hard), alpha)       # [Q, 2]  (:734-738)
        options = list(zip(variable_set1._name, variable_set2._name))
        answer, alp = [], []
        if give_answer:
            _reads(log_probability)

            def compute():
                lp = _host(log_probability)
                ind = lp.argmax(1)
                n = variable_set1.batch_size()
                return [[options[i][ind[i]]] for i in range(n)], [[float(lp[i, ind[i]])] for i in range(n)]
            answer, alp = _answers(compute)
        return _result(answer, log_probability.view(-1), options, None, QuestionType.QUERY, x1.cumulative_loss() + x2.cumulative_loss(),
                       x1._prev_variable_sets_num + x2._prev_variable_sets_num + 2, alp)


# ---------------------------------------------------------------------------------------------------
# attention-calibration passes: transform_attention of every operator (the LSTM walks the program forward and
# backward before it is executed; reference batch_gqa_ops.py, the `transform_attention` method of each class)
# ---------------------------------------------------------------------------------------------------
def _select_ta(self, op_id, is_forward, world, attention_state, attribute_list, op_feature, predicate_question_map=None):      # :185-203
    name, att = _select_names(attribute_list, world.batch_size())
    plain = att is None or all(a is None for a in att)
    if is_forward:
        x = world.attention_state(name)
        return x if plain else self._filter.transform_attention(op_id, is_forward, world, x, att, op_feature)
    return attention_state if plain else self._filter.transform_attention(op_id, is_forward, world, attention_state, att, op_feature)


def _filter_ta(self, op_id, is_forward, world, attention_state, attribute_list, op_feature, predicate_question_map=None):       # :337-338
    return self._filter.transform_attention(op_id, is_forward, world, attention_state, attribute_list, op_feature)


def _zero_state(world, attention_state):
    return world.attention_state(attention_state._name, (torch.zeros_like(attention_state._state[0]), torch.zeros_like(attention_state._state[1])))


def _relate_ta(self, op_id, is_forward, world, attention_state, relation_list, is_subject, attribute_list, op_feature,
               predicate_question_map=None):                                                                                      # :373-390
    if is_forward:
        x = self._gqa_select.transform_attention(op_id, is_forward, world, None, attribute_list, op_feature)
        subject_set = x.gate(attention_state, is_subject)
        object_set = attention_state.gate(x, is_subject)
        subject_set, object_set = self._relate.transform_attention(op_id, is_forward, world, subject_set, object_set, relation_list, op_feature)
        return subject_set.gate(object_set, is_subject)
    x = _zero_state(world, attention_state)
    object_set = x.gate(attention_state, is_subject)
    subject_set = attention_state.gate(x, is_subject)
    subject_set, object_set = self._relate.transform_attention(op_id, is_forward, world, subject_set, object_set, relation_list, op_feature)
    self._gqa_select.transform_attention(op_id, is_forward, world, subject_set.gate(object_set, is_subject), attribute_list, op_feature)
    return object_set.gate(subject_set, is_subject)


def _identity_ta(self, op_id, is_forward, world, attention_state, op_feature, predicate_question_map=None):                      # :412-413, :782-783
    return attention_state


def _options_ta(self, op_id, is_forward, world, attention_state, attribute_list_list, op_feature, predicate_question_map=None):  # :230-232, :475-477
    attribute_list, batch_index = flatten_list(attribute_list_list)
    return self._filter.transform_attention(op_id, is_forward, world, attention_state, attribute_list, op_feature,
                                            batch_index if predicate_question_map is None else predicate_question_map)


def _query_ta(self, op_id, is_forward, world, attention_state, category_list, op_feature, predicate_question_map=None):          # :308-310
    lists = [self._ontology.query(c if c not in ['name', 'type'] else n) for c, n in zip(category_list, attention_state._name)]
    return self._gqa_choose_attr.transform_attention(op_id, is_forward, world, attention_state, lists, op_feature, predicate_question_map)


def _verify_rel_ta(self, op_id, is_forward, world, attention_state, relation_list, is_subject, attribute_list, op_feature,
                   predicate_question_map=None):                                                                                  # :503-504
    return self._gqa_relate.transform_attention(op_id, is_forward, world, attention_state, relation_list, is_subject, attribute_list, op_feature)


def _choose_rel_ta(self, op_id, is_forward, world, attention_state, relation_list_list, is_subject, attribute_list, op_feature,
                   predicate_question_map=None):                                                                                  # :269-292
    relation_list, batch_index = flatten_list(relation_list_list)
    pqm = batch_index if predicate_question_map is None else predicate_question_map
    host = [0.0 if f is None else float(f) for f in is_subject]
    pmap = batch_index if predicate_question_map is None else (getattr(predicate_question_map, "_host", None) or
                                                               predicate_question_map.to(torch.int64).cpu().tolist())
    pred_flags = [host[q] for q in pmap]                                         # mm(pqm, is_subject)
    if is_forward:
        x = self._gqa_select.transform_attention(op_id, is_forward, world, None, attribute_list, op_feature)
        subject_state = x.gate(attention_state, is_subject)
        object_state = attention_state.gate(x, is_subject)
        subject_state, object_state = self._relate.transform_attention(op_id, is_forward, world, subject_state, object_state, relation_list,
                                                                       op_feature, pqm)
        return subject_state.gate(object_state, pred_flags)
    x = _zero_state(world, attention_state)
    object_set = x.gate(attention_state, pred_flags)
    subject_set = attention_state.gate(x, pred_flags)
    subject_set, object_set = self._relate.transform_attention(op_id, is_forward, world, subject_set, object_set, relation_list, op_feature, pqm)
    self._gqa_select.transform_attention(op_id, is_forward, world, subject_set.gate(object_set, is_subject), attribute_list, op_feature)
    return object_set.gate(subject_set, is_subject)


def _pair_identity_ta(self, op_id, is_forward, world, attention_state1, attention_state2, op_feature, predicate_question_map=None):   # :536-537
    return attention_state1, attention_state2


def _category_ta(self, op_id, is_forward, world, attention_state, category_list, op_feature, predicate_question_map=None):       # :610-613
    attribute_list, batch_index = _category_options(self._ontology, category_list, attention_state._name)
    return self._filter.transform_attention(op_id, is_forward, world, attention_state, attribute_list, op_feature,
                                            batch_index if predicate_question_map is None else predicate_question_map)


def _all_different_ta(self, op_id, is_forward, world, attention_state, category_list, op_feature, predicate_question_map=None):  # :641-642
    return self._gqa_all_same.transform_attention(op_id, is_forward, world, attention_state, category_list, op_feature, predicate_question_map)


def _two_same_ta(self, op_id, is_forward, world, attention_state1, attention_state2, category_list, op_feature, predicate_question_map=None):   # :683-690
    attribute_list, batch_index = _category_options(self._ontology, category_list, attention_state1._name)
    pqm = batch_index if predicate_question_map is None else predicate_question_map
    x1 = self._filter.transform_attention(op_id + ':0', is_forward, world, attention_state1, attribute_list, op_feature, pqm)
    x2 = self._filter.transform_attention(op_id + ':1', is_forward, world, attention_state2, attribute_list, op_feature, pqm)
    return x1, x2


def _two_different_ta(self, op_id, is_forward, world, attention_state1, attention_state2, category_list, op_feature, predicate_question_map=None):   # :716-717
    return self._gqa_two_same.transform_attention(op_id, is_forward, world, attention_state1, attention_state2, category_list, op_feature,
                                                  predicate_question_map)


def _compare_ta(self, op_id, is_forward, world, attention_state1, attention_state2, attribute_list, is_less, op_feature,
                predicate_question_map=None):                                                                                     # :760-764
    x1 = self._filter.transform_attention(op_id + ':0', is_forward, world, attention_state1, attribute_list, op_feature)
    x2 = self._filter.transform_attention(op_id + ':1', is_forward, world, attention_state2, attribute_list, op_feature)
    return x1, x2


GQASelectBatch.transform_attention = _select_ta
GQAFilterBatch.transform_attention = _filter_ta
GQARelateBatch.transform_attention = _relate_ta
GQAExistBatch.transform_attention = _identity_ta
GQAEndBatch.transform_attention = _identity_ta
GQAVerifyAttrsBatch.transform_attention = _options_ta
GQAChooseAttrBatch.transform_attention = _options_ta
GQAQueryAttrBatch.transform_attention = _query_ta
GQAVerifyRelBatch.transform_attention = _verify_rel_ta
GQAChooseRelBatch.transform_attention = _choose_rel_ta
GQAAndBatch.transform_attention = _pair_identity_ta
GQAOrBatch.transform_attention = _pair_identity_ta
GQAAllSameBatch.transform_attention = _category_ta
GQAAllDifferentBatch.transform_attention = _all_different_ta
GQATwoSameBatch.transform_attention = _two_same_ta
GQATwoDifferentBatch.transform_attention = _two_different_ta
GQACompareBatch.transform_attention = _compare_ta
