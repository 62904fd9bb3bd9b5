"""Program batching: the operand format the interpreter consumes.

Reference: src/nsvqa/data/data_pipeline.py — OperatorBatch :31-143, ProgramBatch :147-290,
ProgramCollaterBase :626-783.  Same classes and fields; in addition every OperatorBatch can be
*lowered* once, at collate time (i.e. in a DataLoader worker), to integer arrays — table columns,
negation flags, validity, subject flags — so the interpreter's timed step does no string work.
"""

import math
import time

import numpy as np
import torch

from .fol_types import QuestionType, TokenType
from .host_util import TokenList, flatten_list, lower_tokens, upload

# which argument slot of an operator holds which kind of token (for lowering)
_ATTR_SLOT = {"select": 0, "filter": 0}
_REL_OPS = {"relate": (0, 1, 2), "verify_rel": (0, 1, 2)}          # (relation, is_subject, name) slots


def _has_token_lists(column):
    """True when a question holds more than one token in this argument slot (its predicates then outnumber the questions)."""
    for el in column:
        if el.__class__ is list and len(el) > 1:
            return True
    return False


_COPY = object()            # OperatorBatch(question_index=_COPY, process_args=False): a field-by-field copy, no token analysis


class OperatorBatch(object):
    """One operator applied across the questions of a batch (data_pipeline.py:31-143)."""

    def __init__(self, op_name, arguments, question_num, is_terminal, mask=None, question_index=None, process_args=True):
        self._op_name = op_name
        self._is_terminal = is_terminal
        self._op_id = None
        if process_args:
            arguments = list(arguments)
            if 0 < len(arguments) < question_num:
                arguments = arguments + [None] * (question_num - len(arguments))
            elif len(arguments) >= question_num:
                arguments = arguments[:question_num]
            else:
                arguments = []
            width = next((len(x) for x in arguments if isinstance(x, list)), 0)
            arguments = [[None] * width if x is None else x for x in arguments]
            self._arguments = [TokenList(col) for col in zip(*arguments)]        # one list per argument slot
        else:
            self._arguments = arguments
        self._question_num = question_num
        self._predicate_num = question_num
        self._predicate_question_map = None
        self._question_index = None
        if not process_args and question_index is _COPY:
            # (to_cuda's internal copy: the analysed fields - `_question_index`, `_predicate_num` - are handed over by the caller)
            pass
        elif len(self._arguments) > 0 and _has_token_lists(self._arguments[0]):       # (also for pre-transposed arguments: data_pipeline.py:55-62)
            flat, batch_index = flatten_list(self._arguments[0])
            self._predicate_num = len(flat)
            if question_index is not None:
                self._question_index = question_index
            elif self._predicate_num != self._question_num:
                self._question_index = torch.tensor(batch_index, dtype=torch.int64)
        if mask is None:
            self._mask = None
        else:
            self._mask = torch.from_numpy(mask).float() if isinstance(mask, np.ndarray) else mask
            self._mask._host = self._mask.tolist()

    # -- pickling (collate worker processes hand ProgramBatches to the process that launches): the small host tensors travel as numpy arrays
    # (a torch tensor goes through a shared-memory file and a descriptor hand-over, about a millisecond each, 20 per batch) and get their
    # python attributes back on arrival
    _TENSOR_FIELDS = ("_mask", "_question_index", "_predicate_question_map")

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_mask_host"] = getattr(self._mask, "_host", None) if self._mask is not None else None
        pqm = self._predicate_question_map
        d["_pqm_host"] = getattr(pqm, "_host", None) if pqm is not None else None
        for f in self._TENSOR_FIELDS:
            t = d.get(f)
            if isinstance(t, torch.Tensor) and not t.is_cuda:
                d[f] = ("__numpy__", t.numpy())
        return d

    def __setstate__(self, d):
        mask_host, pqm_host = d.pop("_mask_host", None), d.pop("_pqm_host", None)
        for f in self._TENSOR_FIELDS:
            t = d.get(f)
            if isinstance(t, tuple) and len(t) == 2 and t[0] == "__numpy__":
                d[f] = torch.from_numpy(t[1])
        self.__dict__.update(d)
        if self._mask is not None and mask_host is not None:
            self._mask._host = mask_host
        if self._predicate_question_map is not None and pqm_host is not None:
            self._predicate_question_map._host = pqm_host

    # -- lowering --------------------------------------------------------------------------------
    def lower(self, ontology):
        """Resolve the token arguments of select / filter / relate / verify_rel against the ontology."""
        name = self._op_name
        if name in _ATTR_SLOT and self._arguments:
            toks = self._arguments[_ATTR_SLOT[name]]
            if name == "select":        # '_' and 'scene' select everything (batch_gqa_ops.py:175-177)
                clean = [None if a is None or a.lower() in ("_", "scene") else a for a in toks]
            else:
                clean = toks
            toks.lowered, toks.lowered_type = lower_tokens(clean, ontology, TokenType.ATTRIBUTE), TokenType.ATTRIBUTE
        elif name in _REL_OPS and self._arguments:
            r, s, n = _REL_OPS[name]
            rel = self._arguments[r]
            rel.lowered, rel.lowered_type = lower_tokens(rel, ontology, TokenType.RELATION), TokenType.RELATION
            names = self._arguments[n]
            clean = [None if a is None or a.lower() in ("_", "scene") else a for a in names]
            names.lowered, names.lowered_type = lower_tokens(clean, ontology, TokenType.ATTRIBUTE), TokenType.ATTRIBUTE
        return self

    def create_sparse_map(self):
        """The reference builds a sparse [P, Q] map here (data_pipeline.py:76-83); the block layout only needs the
        predicate -> question index vector, which already exists as `_question_index`."""
        if self._question_index is not None:
            self._predicate_question_map = self._question_index.to(torch.int32)
            self._predicate_question_map._host = self._question_index.tolist()

    def to_cuda(self, device, non_blocking=True):
        res = OperatorBatch(self._op_name, self._arguments, self._question_num, self._is_terminal, mask=None, question_index=_COPY, process_args=False)
        res._question_index = self._question_index
        # the small per-operator tensors (masks, predicate -> question maps, subject flags) go through the content-keyed upload: masks recur
        # from batch to batch (all ones, the same ragged patterns), and a new one is staged in pinned memory instead of a pageable,
        # stream-synchronising copy each (18 of them per fresh 256-question batch: 1.2 ms of host time)
        on_gpu = torch.device(device).type == "cuda"
        if self._mask is not None:
            res._mask = upload(self._mask.numpy(), device) if on_gpu and not self._mask.is_cuda else self._mask.cuda(device, non_blocking=non_blocking)
            res._mask._host = self._mask._host
        if self._question_index is not None:
            qi32 = self._question_index.to(torch.int32)
            res._predicate_question_map = upload(qi32.numpy(), device) if on_gpu and not qi32.is_cuda else qi32.cuda(device, non_blocking=non_blocking)
            res._predicate_question_map._host = self._question_index.tolist()
            res._predicate_question_map._dfol_sorted = bool(all(a <= b for a, b in zip(res._predicate_question_map._host, res._predicate_question_map._host[1:])))
        res._predicate_num = self._predicate_num
        res._op_id = self._op_id
        for a in res._arguments:                       # pre-stage the lowered integer arrays and subject flags
            low = getattr(a, "lowered", None)
            if low is not None:
                low.on(device)
        if self._op_name in _REL_OPS and res._arguments:
            flags = res._arguments[_REL_OPS[self._op_name][1]]
            host_flags = np.asarray([0.0 if f is None else float(f) for f in flags], np.float32)
            flags.device_flags = upload(host_flags, device) if on_gpu else torch.from_numpy(host_flags).cuda(device)
        return res

    def to(self, dtype):
        if dtype != torch.float32:
            raise ValueError("the MI355X path computes in fp32")
        return self

    def pin_memory(self):
        if self._mask is not None:
            host = self._mask._host
            self._mask = self._mask.pin_memory()
            self._mask._host = host
        if self._question_index is not None:
            self._question_index = self._question_index.pin_memory()
        return self

    def __repr__(self):
        res = "Operation: %s\nTerminal: %s\n" % (self._op_name, self._is_terminal)
        if self._mask is not None:
            res += "Mask: %s\n" % self._mask._host
        if len(self._arguments) > 0:
            res += "Arguments: %s\n" % [list(a) for a in self._arguments]
        return res


class _LazyOps(list):
    """The operator batches of a ProgramBatch moved to the device on FIRST ACCESS (ProgramBatch.to_cuda of a batch that carries a native plan):
    an empty list until then, `host` the operator batches as collated."""

    def __init__(self, host_ops, device, non_blocking):
        super(_LazyOps, self).__init__()
        self.host, self._device, self._non_blocking, self._moved = host_ops, device, non_blocking, False

    def _move(self):
        if not self._moved:
            self._moved = True
            super(_LazyOps, self).extend(ob.to_cuda(self._device, self._non_blocking) for ob in self.host)

    def __len__(self):
        return len(self.host)

    def __getitem__(self, i):
        self._move()
        return super(_LazyOps, self).__getitem__(i)

    def __iter__(self):
        self._move()
        return super(_LazyOps, self).__iter__()

    def __reduce__(self):                                   # (pickled as the plain host list)
        return (list, (list(self.host),))


class _PickledOps(list):
    """The operator batches of a ProgramBatch that crossed a process boundary WITH a native plan: kept as pickled bytes until somebody
    reads them (the Python operator loop, a trace).  The launching process of a stream of unseen batches needs the plan, the answers and the
    terminal operator's name - rebuilding ~11 operator batches of 256 token lists each was 0.5 ms of its ~1 ms per batch."""

    def __init__(self, blob, count, last_name):
        super(_PickledOps, self).__init__()
        self._blob, self._count, self.last_name, self._loaded = blob, count, last_name, False

    def _load(self):
        if not self._loaded:
            import pickle
            self._loaded = True
            super(_PickledOps, self).extend(pickle.loads(self._blob))
            self._blob = None

    def __len__(self):
        return self._count

    def __getitem__(self, i):
        self._load()
        return super(_PickledOps, self).__getitem__(i)

    def __iter__(self):
        self._load()
        return super(_PickledOps, self).__iter__()

    def __reduce__(self):
        if self._loaded:
            return (list, (list(super(_PickledOps, self).__iter__()),))
        return (_PickledOps, (self._blob, self._count, self.last_name))


class ProgramBatch(object):
    """A batch of aligned programs plus its scenes (data_pipeline.py:147-290)."""

    def __getstate__(self):
        d = dict(self.__dict__)
        ops = d.get("_op_batch_list")
        if d.get("_native_plan") is not None and ops is not None and not isinstance(ops, _PickledOps):
            import pickle
            host = list(getattr(ops, "host", ops))
            if host:
                d["_op_batch_list"] = _PickledOps(pickle.dumps(host, protocol=pickle.HIGHEST_PROTOCOL), len(host), host[-1]._op_name)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)

    def __init__(self, device, op_batch_list, dependencies, answers, object_features, object_batch_index=None, original_dicts=None,
                 meta_data=None, object_nums=None, question_image=None):
        # `question_image` (an extension of this build; None = the reference's layout): question k looks at scene question_image[k] of
        # this batch - `object_features` / `object_batch_index` then hold every DISTINCT image once instead of one copy per question
        # (ProgramCollaterBase(share_scenes=True); GQA testdev-balanced asks ~30 questions per image)
        self._question_image = None if question_image is None else [int(i) for i in question_image]
        self._op_batch_list = op_batch_list
        self._object_features = object_features
        self._dependencies = dependencies
        self._answers = answers
        self._batch_size = op_batch_list[0]._question_num if op_batch_list else 0
        self._original_dicts = original_dicts
        self._meta_data = meta_data
        self._device = device
        if isinstance(object_batch_index, np.ndarray):
            object_batch_index = torch.from_numpy(object_batch_index)
        self._object_batch_index = object_batch_index
        if object_nums is None and object_batch_index is not None:          # kept on the host so that build_scene never syncs
            object_nums = torch.bincount(object_batch_index.cpu().to(torch.int64)).tolist()
        self._object_nums = object_nums
        batch_id = str(int(round(time.time() * 1000)))                       # :167-169
        for i, ob in enumerate(self._op_batch_list):
            if ob._op_id is None:
                ob._op_id = batch_id + ':' + str(i)
        self._question_type = QuestionType.QUERY if self._op_batch_list and self._op_batch_list[-1]._op_name in \
            ['query_attr', 'choose_attr', 'choose_rel'] else QuestionType.BINARY

    @property
    def device(self):
        return self._device

    def batch_size(self):
        return self._batch_size

    def lower(self, ontology):
        for ob in self._op_batch_list:
            ob.lower(ontology)
        return self

    def to_cuda(self, device, non_blocking=True):
        def move(obj):
            if isinstance(obj, torch.Tensor):
                return obj.cuda(device, non_blocking=non_blocking)
            if isinstance(obj, dict):
                return {k: move(v) for k, v in obj.items()}
            return obj
        plan = getattr(self, "_native_plan", None)
        if plan is not None and not isinstance(self._op_batch_list, _LazyOps):
            # Lowered for the native executor (native_plan.build_plan, at collate time): everything its launches read travels in the plan's
            # ONE blob, so the ~18 per-operator uploads (masks, maps, flags, lowered token arrays) happen only if somebody asks for the
            # operator batches after all - the Python operator loop (a trace, a training step, DFOL_NATIVE=0) - and then on first access.
            ops = _LazyOps(self._op_batch_list, device, non_blocking)
            pb = ProgramBatch.__new__(ProgramBatch)
            pb.__dict__.update(self.__dict__)
            pb._op_batch_list, pb._device = ops, device
            pb._object_features, pb._object_batch_index, pb._meta_data = move(self._object_features), move(self._object_batch_index), move(self._meta_data)
            return pb
        pb = ProgramBatch(device, [ob.to_cuda(device, non_blocking) for ob in self._op_batch_list], self._dependencies, self._answers,
                          move(self._object_features), move(self._object_batch_index), self._original_dicts, move(self._meta_data),
                          object_nums=self._object_nums, question_image=self._question_image)
        if hasattr(self, "_native_plan"):                      # (None: lowered, and found to have a shape the executor does not take)
            pb._native_plan = self._native_plan
        return pb

    def terminal_op_name(self):
        """Name of the last operator batch (the metrics' slot, trainer.py:477-485) - without uploading a lazily moved batch's operators."""
        ops = self._op_batch_list
        host = ops.host if isinstance(ops, _LazyOps) else ops
        return getattr(host, "last_name", None) or host[-1]._op_name

    def to(self, dtype):
        if dtype != torch.float32:
            raise ValueError("the MI355X path computes in fp32")
        return self

    def pin_memory(self):
        if isinstance(self._object_features, torch.Tensor):
            self._object_features = self._object_features.pin_memory()
        if isinstance(self._meta_data, dict):
            for k, v in self._meta_data.items():
                if isinstance(v, torch.Tensor):
                    self._meta_data[k] = v.pin_memory()
        if self._object_batch_index is not None:
            self._object_batch_index = self._object_batch_index.pin_memory()
        self._op_batch_list = [ob.pin_memory() for ob in self._op_batch_list]
        return self

    def create_sparse_tensors(self):                   # data_pipeline.py:242-244
        for ob in self._op_batch_list:
            ob.create_sparse_map()

    def retrieve_instance(self, index, trace=None):    # data_pipeline.py:273-290
        res = []
        if index >= self._batch_size:
            return res
        for i, ob in enumerate(self._op_batch_list):
            if ob._mask is not None and ob._mask._host[index] == 0:
                continue
            res.append((ob._op_name, [a[index] for a in ob._arguments]))
        return res


class ProgramCollaterBase(object):
    """Aligns ragged per-question programs into one canonical operator sequence (data_pipeline.py:626-783):
    per branch `starter, (filler*, separator)*`, then one terminal operator batch per terminal operator name."""

    def __init__(self, starter_op, sep_op, filler_op, split_num=1, ontology=None, share_scenes=False, native_spec=None):
        self._sep_op = sep_op
        self._filler_op = filler_op
        self._starter_op = starter_op
        self._split_num = split_num
        self._ontology = ontology
        # share_scenes: questions of a ProgramBatch that name the same `image_id` share ONE copy of its object features (and, in the
        # interpreter, one featurizer pass and one set of relation tiles per (image, concept)).  The reference collates one copy per
        # question (batch_gqa_boxfeatures_pipeline.py:37-73); results are identical either way (tests/test_interpreter_gpu.py).
        self._share_scenes = share_scenes
        # native_spec (native_exec.model_spec(model), needs `ontology`): every ProgramBatch is also lowered to the native executor's
        # instruction table here - i.e. in the DataLoader worker - so the process that launches does no per-operator work at all
        self._native_spec = native_spec

    def collate_programs(self, questions):
        B = len(questions)
        ops, deps, last_dep = [], [], []
        cursor = -1
        filler_op, sep_op = self._filler_op, self._sep_op
        programs = [q['program'] for q in questions]
        branches = [p['branches'] for p in programs]
        branch_num = max(map(len, branches))
        for b in range(branch_num):
            heads = [br[b][0] for br in branches]
            args = [h['arguments'] if h['operator'] == self._starter_op else ['_'] for h in heads]
            ops.append(OperatorBatch(self._starter_op, args, B, False, mask=np.ones(B, dtype=np.float32)))
            deps.append([])
            cursor += 1
            # slot[k] = fillers that run before the k-th separator, then the separator itself
            fillers, seps = [], []
            for k, br in enumerate(branches):
                f_i = s_i = 0
                for o in br[b][1:]:
                    name = o['operator']
                    if name == filler_op:
                        while len(fillers) <= s_i:
                            fillers.append([])
                            f_i = 0
                        if f_i >= len(fillers[s_i]):
                            fillers[s_i].append({'arguments': [None] * B, 'mask': np.zeros(B, dtype=np.float32)})
                        fillers[s_i][f_i]['mask'][k] = 1.0
                        fillers[s_i][f_i]['arguments'][k] = o['arguments']
                        f_i += 1
                    elif name == sep_op:
                        if s_i >= len(seps):
                            seps.append({'arguments': [None] * B, 'mask': np.zeros(B, dtype=np.float32)})
                        seps[s_i]['mask'][k] = 1.0
                        seps[s_i]['arguments'][k] = o['arguments']
                        s_i += 1
                        f_i = 0
            for n in range(max(len(seps), len(fillers))):
                for d in (fillers[n] if n < len(fillers) else []):
                    ops.append(OperatorBatch(self._filler_op, d['arguments'], B, False, mask=d['mask']))
                    deps.append([cursor])
                    cursor += 1
                if n < len(seps):
                    ops.append(OperatorBatch(self._sep_op, seps[n]['arguments'], B, False, mask=seps[n]['mask']))
                    deps.append([cursor])
                    cursor += 1
            last_dep.append(cursor)
        terminal = {}
        for k, p in enumerate(programs):
            o = p['last_op']
            slot = terminal.setdefault(o['operator'], {'arguments': [None] * B, 'mask': np.zeros(B, dtype=np.float32)})
            slot['arguments'][k] = o['arguments']
            slot['mask'][k] = 1.0
        for name, slot in terminal.items():
            ops.append(OperatorBatch(name, slot['arguments'], B, True, mask=slot['mask']))
            deps.append(last_dep)
        return ops, deps

    def collate_object_features(self, questions):
        return None, None

    def collate_meta_data(self, questions):
        return None

    def collate(self, questions):
        result = []
        n = len(questions)
        split_num = min(self._split_num, n)
        split_size = math.ceil(n / split_num)
        device = torch.device('cpu')
        for i in range(split_num):
            chunk = questions[i * split_size:min((i + 1) * split_size, n)]
            if not chunk:
                break
            ops, deps = self.collate_programs(chunk)
            question_image = None
            if self._share_scenes:
                first, question_image = {}, []
                for q in chunk:
                    question_image.append(first.setdefault(q['image_id'], len(first)))
                reps = [None] * len(first)
                for q, i in zip(chunk, question_image):
                    if reps[i] is None:
                        reps[i] = q
                object_features, object_batch_index = self.collate_object_features(reps)      # one representative question per image
            else:
                object_features, object_batch_index = self.collate_object_features(chunk)
            pb = ProgramBatch(device, ops, deps, [q['answer'] for q in chunk], object_features, object_batch_index,
                              [q.get('original_dict') for q in chunk], meta_data=self.collate_meta_data(chunk), question_image=question_image)
            for ob in pb._op_batch_list:
                ob._op_id = str(i) + ':' + ob._op_id            # :775-777
            if self._ontology is not None:
                pb.lower(self._ontology)
                if self._native_spec is not None:
                    from .native_plan import build_plan
                    pb._native_plan = build_plan(pb, self._ontology, self._native_spec)
            result.append(pb)
        return result
