"""Question and object-feature data path.

Reference: src/nsvqa/data/data_pipeline.py:294-622 (ProgramDataset: JSON lines or the HDF5 program bytecode),
src/gqa_preprocess.py:15-94 (GQAH5Encoder, the bytecode writer) and
src/nsvqa/data/batch_gqa_boxfeatures_pipeline.py:15-189 (BatchGQABoxFeaturesCollator).

The bytecode is six int32 arrays per question file — `answer[row]`, `image_id[row]`, `branch_ops[row, b, 10]`,
`branch_args[row, b, 10, 3]`, `last_op[row]`, `last_args[row, k]` — token codes are 1-based vocabulary indices,
negative = `not(token)`, 0 = empty.  The reference stores them in HDF5; this module reads and writes the same arrays
from `.npz` and from `.h5` (through h5py when installed, else through the HDF5 C library itself: h5lite.py).
"""

import collections
import json
import os
import random
import re

import numpy as np
import torch

from .program import ProgramCollaterBase

_NEG = re.compile(r"not\((\w|\s)+\)")
ARRAYS = ("answer", "image_id", "branch_ops", "branch_args", "last_op", "last_args")


def _open_arrays(path):
    """{name: array-like} for a bytecode / feature container: .npz, or .h5 (h5py when installed, else the HDF5 C library through
    h5lite - the files the reference writes and reads, gqa_preprocess.py:87-93 / data_pipeline.py:328-389)."""
    ext = os.path.splitext(path)[1]
    if ext == ".npz":
        return np.load(path)
    if ext == ".h5":
        from . import h5lite
        return h5lite.import_h5py().File(path, "r")
    raise ValueError("unsupported container %s" % path)


def write_arrays(path, arrays):
    """Write {name: array} as `.npz` or as the reference's `.h5` layout (one dataset per name, gqa_preprocess.py:87-93)."""
    ext = os.path.splitext(path)[1]
    if ext == ".npz":
        np.savez(path, **arrays)
    elif ext == ".h5":
        from . import h5lite
        f = h5lite.import_h5py().File(path, "w")
        try:
            for k, v in arrays.items():
                f.create_dataset(k, data=np.asarray(v))
        finally:
            f.close()
    else:
        raise ValueError("unsupported container %s" % path)


class ProgramCodec(object):
    """Program <-> bytecode (gqa_preprocess.py:51-94 / data_pipeline.py:342-367, 391-453)."""

    def __init__(self, ontology, max_branch_length=10):
        self._ontology = ontology
        self._max_branch_length = max_branch_length

    # ---- encode -------------------------------------------------------------------------------
    @staticmethod
    def _flat(arguments):
        return [item for sub in arguments for item in (sub if isinstance(sub, list) else [sub])]

    def sizes(self, first_question, row_n):            # gqa_preprocess.py:21-49: sized by the file's first terminal operator
        opn = first_question['program']['last_op']['operator']
        arg_n = 2 if opn in ('verify_attrs', 'choose_attr', 'compare') else 3 if opn == 'verify_rel' else 4 if opn == 'choose_rel' else 1
        branch_n = 2 if opn in ('and', 'or', 'two_same', 'two_different', 'compare') else 1
        return row_n, branch_n, arg_n

    def encode(self, questions):
        ont = self._ontology
        row_n, branch_n, arg_n = self.sizes(questions[0], len(questions))
        out = {"answer": np.zeros(row_n, np.int32), "image_id": np.zeros(row_n, np.int32),
               "branch_ops": np.zeros((row_n, branch_n, self._max_branch_length), np.int32),
               "branch_args": np.zeros((row_n, branch_n, self._max_branch_length, 3), np.int32),
               "last_op": np.zeros(row_n, np.int32), "last_args": np.zeros((row_n, arg_n), np.int32)}
        for i, q in enumerate(questions):
            out["image_id"][i] = ont.encode_img_id(q['imageId'])
            out["answer"][i] = ont.encode_token(q['answer'])
            for j, branch in enumerate(q['program']['branches']):
                for k, o in enumerate(branch):
                    out["branch_ops"][i, j, k] = ont.encode_op(o['operator'])
                    for t, arg in enumerate(self._flat(o['arguments'])):
                        out["branch_args"][i, j, k, t] = ont.encode_token(arg)
            last = q['program']['last_op']
            out["last_op"][i] = ont.encode_op(last['operator'])
            for t, arg in enumerate(self._flat(last['arguments'])):
                out["last_args"][i, t] = ont.encode_token(arg)
        return out

    # ---- decode -------------------------------------------------------------------------------
    def _args(self, op_name, codes):
        tok = self._ontology.decode_token
        if op_name in ('select', 'filter', 'query_attr', 'verify_attr', 'all_same', 'all_different', 'two_same', 'two_different'):
            return [tok(codes[0])]
        if op_name in ('relate', 'verify_rel'):
            return [tok(codes[i]) for i in range(3)]
        if op_name == 'choose_attr':
            return [[tok(codes[0]), tok(codes[1])]]
        if op_name == 'verify_attrs':
            return [[tok(codes[0])] + ([tok(codes[1])] if codes[1] != 0 else [])]
        if op_name == 'choose_rel':
            return [[tok(codes[0]), tok(codes[1])], tok(codes[2]), tok(codes[3])]
        if op_name == 'compare':
            return [tok(codes[0]), tok(codes[1])]
        if op_name in ('exist', 'and', 'or', 'end'):
            return []
        raise KeyError(op_name)

    def decode(self, arrays, idx):
        ont = self._ontology
        obj = {'imageId': ont.decode_img_id(int(arrays['image_id'][idx])), 'answer': ont.decode_token(int(arrays['answer'][idx]))}
        last_name = ont.decode_op(int(arrays['last_op'][idx]))
        last = {'operator': last_name, 'arguments': self._args(last_name, [int(c) for c in arrays['last_args'][idx]])}
        ops, args = np.asarray(arrays['branch_ops'][idx]), np.asarray(arrays['branch_args'][idx])
        branches = []
        for i in range(ops.shape[0]):
            branch = []
            for j in range(ops.shape[1]):
                if ops[i, j] == 0:
                    break
                name = ont.decode_op(int(ops[i, j]))
                branch.append({'operator': name, 'arguments': self._args(name, [int(c) for c in args[i, j]])})
            branches.append(branch)
        obj['program'] = {'branches': branches, 'last_op': last}
        return obj


def _strip_negation(tokens):                            # data_pipeline.py:455-471
    return [a.strip()[4:-1] if isinstance(a, str) and _NEG.match(a.strip()) else (a.strip() if isinstance(a, str) else a) for a in tokens]


def _entity(arg):
    return "entity" if arg is None or (isinstance(arg, str) and arg.lower() in ("_", "scene")) else arg


class ProgramDataset(torch.utils.data.Dataset):
    """data_pipeline.py:294-622.  `input_file`: a JSON-lines path, a bytecode container (.npz / .h5) or a list of dicts."""

    def __init__(self, input_file, ontology, in_memory, max_cache_size=100000, keep_original_dict=False, shuffle_options=True):
        super(ProgramDataset, self).__init__()
        self._input_file = input_file
        self._ontology = ontology
        self._keep_original_dict = keep_original_dict
        self._shuffle_options = shuffle_options         # the reference shuffles choose-options at load time (:596-597)
        self._codec = ProgramCodec(ontology)
        self._is_bytecode = isinstance(input_file, str) and os.path.splitext(input_file)[1] in (".npz", ".h5")
        self._is_h5 = self._is_bytecode
        self._in_memory = in_memory or isinstance(input_file, (list, tuple))
        self._cache = {} if self._in_memory else collections.OrderedDict()
        self._max_cache_size = max_cache_size
        self._arrays = None
        if self._is_bytecode:
            arrays = _open_arrays(input_file)
            self._row_num = arrays['image_id'].shape[0]
            if self._in_memory:
                self._arrays = {k: np.asarray(arrays[k]) for k in ARRAYS}
        elif isinstance(input_file, str):
            with open(input_file, 'r') as fh:
                lines = fh.readlines()
            self._row_num = len(lines)
            self._data = lines if self._in_memory else None
            if not self._in_memory:
                self._offsets = np.cumsum([0] + [len(l.encode('utf8')) for l in lines[:-1]]).tolist()
        else:
            self._data = input_file
            self._row_num = len(input_file)

    def __len__(self):
        return self._row_num

    def _raw(self, idx):
        if self._is_bytecode:
            if self._arrays is None:                    # opened lazily in each DataLoader worker
                self._arrays = _open_arrays(self._input_file)
            return self._codec.decode(self._arrays, idx)
        if self._data is not None:
            line = self._data[idx]
        else:
            with open(self._input_file, 'rb') as fh:
                fh.seek(self._offsets[idx])
                line = fh.readline().decode('utf8')
        return json.loads(line) if isinstance(line, str) else line

    def __getitem__(self, idx):
        if not self._in_memory and idx in self._cache:
            return self._cache[idx]
        result = self._transform_line(self._raw(idx))
        if not self._in_memory:
            if len(self._cache) >= self._max_cache_size:
                self._cache.popitem(last=False)
            self._cache[idx] = result
        return result

    # ---- tokens a question mentions (data_pipeline.py:473-569); the variable name threads through a branch ----
    def _category_options(self, category, name):
        # the reference appends to the ontology's own list here (:487-488), growing it on every call; a copy is used instead
        return list(self._ontology.query(category if category not in ['name', 'type'] else name)) + [category]

    def _extract(self, operator, arguments, name):
        if operator == 'select':
            args = _strip_negation(arguments)
            return args, _entity(args[0])
        if operator == 'filter':
            args = _strip_negation(arguments)
            return args, args[0] if self._ontology.is_noun(args[0]) else name
        if operator in ('relate', 'verify_rel'):
            a0, a2 = _strip_negation([arguments[0]]), _strip_negation([arguments[2]])
            return a0 + a2, _entity(a2[0])
        if operator == 'choose_rel':
            a0, a2 = _strip_negation(arguments[0]), _strip_negation([arguments[2]])
            return a0 + a2, _entity(a2[0])
        if operator in ('query_attr', 'all_same', 'all_different', 'two_same', 'two_different'):
            return _strip_negation(self._category_options(arguments[0], name)), name
        if operator in ('choose_attr', 'verify_attrs'):
            return _strip_negation(arguments[0]), name
        if operator == 'verify_attr':
            return _strip_negation(arguments), name
        if operator == 'compare':
            return _strip_negation([arguments[0]]), name
        if operator in ('exist', 'and', 'or', 'end'):
            return [], name
        raise KeyError(operator)

    def _collect_tokens(self, program):
        tokens, name = [], ''
        for branch in program['branches']:
            name = ''
            for o in branch:
                t, name = self._extract(o['operator'], o['arguments'], name)
                tokens += t
        t, name = self._extract(program['last_op']['operator'], program['last_op']['arguments'], name)
        return list(set(tokens + t))

    @staticmethod
    def _transform_answer(op_name, answer):             # data_pipeline.py:571-591
        if answer is None:
            return None
        if isinstance(answer, (list, tuple)):
            if len(answer) == 0:
                return []
            flat = [x for sub in answer for x in sub] if isinstance(answer[0], (list, tuple)) else list(answer)
            return [a.lower().strip() for a in flat]
        res = answer.lower().strip() if isinstance(answer, str) else answer
        if op_name == 'choose_rel':
            res = {'left': 'to the left of', 'right': 'to the right of'}.get(res, res)
        return res

    def _transform_line(self, q):                       # data_pipeline.py:593-622
        op_name = q['program']['last_op']['operator']
        if self._shuffle_options and op_name in ('choose_rel', 'choose_attr'):
            random.shuffle(q['program']['last_op']['arguments'][0])
        if 'answer' not in q:
            q['answer'] = ""
        text = not self._is_bytecode
        result = {'program': q['program'], 'image_id': q['imageId'], 'answer': self._transform_answer(op_name, q['answer']),
                  'tokens': self._collect_tokens(q['program']), 'original_dict': q if self._keep_original_dict else None,
                  'question': q['question'] if text and 'question' in q else None,
                  'question_id': q['question_id'] if text and 'question_id' in q else None}
        for key in ('object_pairs', 'attribute_dict', 'relation_list'):
            if key in q:
                result[key] = q[key]
        if 'weights' in q:
            w = q['weights']
            result['weights'] = [x for sub in w for x in sub] if len(w) > 0 and isinstance(w[0], (list, tuple)) else w
        return result


class BatchGQABoxFeaturesCollator(ProgramCollaterBase):
    """batch_gqa_boxfeatures_pipeline.py:15-189: object features of the questions' images from chunked containers.

    Chunk i is `<prefix>_<i>.npz` (or `.h5`) with `features [chunk, max_obj, F]` and `bboxes [chunk, max_obj, 4]` (x1, y1, x2, y2);
    `object_info_json_path` maps image id -> {objectsNum, width, height, idx, file}.  Produces the reference's
    `object_features [O, F + 6]` = [features, W, H, x, y, w, h] (:57-71) plus the host-side object counts."""

    def __init__(self, object_h5_path, file_prefix, chunk_num, object_info_json_path, ontology, split_num, lower=True, share_scenes=False):
        super(BatchGQABoxFeaturesCollator, self).__init__('select', 'relate', 'filter', split_num, ontology=ontology if lower else None,
                                                          share_scenes=share_scenes)
        self._object_h5_path = object_h5_path
        self._file_prefix = file_prefix
        self._chunk_num = chunk_num
        self._file_handles = None
        with open(object_info_json_path, 'r') as f:
            self._object_info = json.load(f)
        self._gqa_ontology = ontology
        first = self._chunk(0)
        self._chunck_size, self._max_object_per_image, self._feature_dim = first['features'].shape

    def _chunk_path(self, i):
        base = os.path.join(self._object_h5_path, "%s_%d" % (self._file_prefix, i))
        for ext in (".npz", ".h5"):
            if os.path.exists(base + ext):
                return base + ext
        raise FileNotFoundError(base + ".npz|.h5")

    def _chunk(self, i):
        return _open_arrays(self._chunk_path(i))

    def collate_object_features(self, questions):
        if self._file_handles is None:                  # opened lazily in each DataLoader worker (:38-39)
            self._file_handles = [self._chunk(i) for i in range(self._chunk_num)]
        info = [self._object_info[q['image_id']] for q in questions]
        feats, counts = [], []
        for inf in info:
            n = int(inf['objectsNum'])
            h = self._file_handles[inf['file']]
            f = np.asarray(h['features'][inf['idx']][:n], np.float32)
            b = np.array(h['bboxes'][inf['idx']][:n], np.float32)
            b[:, 2] -= b[:, 0]                          # (x1, y1, x2, y2) -> (x, y, w, h)  (:60-61)
            b[:, 3] -= b[:, 1]
            size = np.tile(np.asarray([[inf['width'], inf['height']]], np.float32), (n, 1))
            feats.append(np.concatenate([f, size, b], 1))
            counts.append(n)
        batch_ind = torch.from_numpy(np.repeat(np.arange(len(counts)), counts).astype(np.int64))
        return torch.from_numpy(np.concatenate(feats, 0)), batch_ind

    def collate_meta_data(self, questions):             # :75-92 (the direct-supervision extras are out of scope)
        tokens = sorted({t for q in questions for t in q['tokens']}, key=str)
        emb = self._gqa_ontology.get_embeddings([str(t) for t in tokens])
        return {'index': {t: i for i, t in enumerate(tokens)},
                'embedding': torch.zeros(len(tokens), 1) if emb is None else torch.from_numpy(emb).float(),
                'questions': [q.get('question') for q in questions], 'image_ids': [q['image_id'] for q in questions],
                'question_ids': [q.get('question_id') for q in questions]}


# ---------------------------------------------------------------------------------------------------------------------
# Program verifier (the part that guards the interpreter's input): nn/parser/parse_utils.py:24-240
# ---------------------------------------------------------------------------------------------------------------------
# ---- the epoch driver's batching: one batch = questions of ONE file = one terminal operator (data_pipeline.py:787-900) ---------------------------
class _FileBatches(object):
    """Batches of one dataset's indices: its own order or a random permutation (torch's BatchSampler over a Sequential / RandomSampler), or,
    distributed, this rank's strided share of them (DistributedSampler: padded by wrapping around to a multiple of the world size)."""

    def __init__(self, length, batch_size, drop_last, shuffle, replacement, distributed, rank, world, seed):
        self.length, self.batch_size, self.drop_last = int(length), int(batch_size), bool(drop_last)
        self.shuffle, self.replacement, self.distributed = bool(shuffle), bool(replacement), bool(distributed)
        self.rank, self.world, self.seed, self.epoch = int(rank), int(world), int(seed), 0
        self.samples = -(-self.length // self.world) if self.distributed else self.length

    def indices(self):
        g = torch.Generator()
        if self.distributed:
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.length, generator=g).tolist() if self.shuffle else list(range(self.length))
            total = self.samples * self.world
            while len(order) < total:                          # (pad by wrapping around, as DistributedSampler does)
                order += order[:total - len(order)]
            return order[self.rank:total:self.world]
        if not self.shuffle:
            return list(range(self.length))
        g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        if self.replacement:                                   # (torch's RandomSampler draws in chunks of 32, then the remainder)
            out = []
            for _ in range(self.length // 32):
                out += torch.randint(high=self.length, size=(32,), dtype=torch.int64, generator=g).tolist()
            return out + torch.randint(high=self.length, size=(self.length % 32,), dtype=torch.int64, generator=g).tolist()
        return torch.randperm(self.length, generator=g).tolist()

    def __iter__(self):
        idx = self.indices()
        for a in range(0, len(idx), self.batch_size):
            b = idx[a:a + self.batch_size]
            if len(b) == self.batch_size or not self.drop_last:
                yield b


class MultiSetSampler(torch.utils.data.Sampler):
    """data_pipeline.py:787-826: a batch sampler over SEVERAL datasets (one per program file, i.e. per terminal operator) that never mixes
    files inside a batch.  Every step draws the file with probability proportional to the questions it has left, takes that file's next
    batch of a random permutation and hands out indices into the concatenation of the datasets."""

    sequential = False

    def __init__(self, dataset_list, batch_size, drop_last, replacement=False, distributed=False, rank=None, world_size=None, seed=0):
        self._datasets = list(dataset_list)
        self._distributed = distributed
        if distributed and (rank is None or world_size is None):
            import torch.distributed as td
            rank, world_size = td.get_rank(), td.get_world_size()
        self._files = [_FileBatches(len(ds), batch_size, drop_last, not self.sequential, replacement, distributed, rank or 0, world_size or 1, seed)
                       for ds in self._datasets]
        self._lengths = [f.samples for f in self._files]
        self._cumulative_lengths = np.cumsum([len(ds) for ds in self._datasets]).tolist()
        self._num_samples = sum(self._lengths)

    def __len__(self):                                          # (the reference's length is in questions, not batches: :806-807)
        return self._num_samples

    def _offset(self, dataset_index, batch):
        return batch if dataset_index == 0 else [self._cumulative_lengths[dataset_index - 1] + i for i in batch]

    def __iter__(self):
        left = torch.tensor(self._lengths, dtype=torch.float32)
        iterators = [iter(f) for f in self._files]
        while float(left.sum()) > 0:
            k = int(torch.multinomial(left, 1)[0])
            try:
                batch = next(iterators[k])
            except StopIteration:                               # (drop_last: a file's remainder is never handed out)
                left[k] = 0
                continue
            left[k] = max(float(left[k]) - len(batch), 0.0)
            yield self._offset(k, batch)

    def set_epoch(self, epoch):
        if self._distributed:
            for f in self._files:
                f.epoch = int(epoch)


class MultiSetSequencialSampler(MultiSetSampler):
    """data_pipeline.py:830-871 (the reference's spelling): the files one after the other, each in its own order (test / predict)."""

    sequential = True

    def __init__(self, dataset_list, batch_size, drop_last, distributed=False, rank=None, world_size=None):
        super(MultiSetSequencialSampler, self).__init__(dataset_list, batch_size, drop_last, False, distributed, rank, world_size)

    def __iter__(self):
        for k, f in enumerate(self._files):
            for batch in f:
                yield self._offset(k, batch)


class GQADataManager(object):
    """data_pipeline.py:875-900: one ProgramDataset per question file of a directory (or one over a list of questions), and the DataLoader whose
    batches are single-file ProgramBatches collated in worker processes."""

    def __init__(self, data_path, ontology, in_memory, max_cache_size=100000, keep_original_dict=False):
        if isinstance(data_path, (list, tuple)):
            datasets = [ProgramDataset(data_path, ontology, in_memory, max_cache_size, keep_original_dict)]
        else:
            names = sorted(f for f in os.listdir(data_path) if os.path.isfile(os.path.join(data_path, f)) and f.endswith((".json", ".h5", ".npz")))
            datasets = [ProgramDataset(os.path.join(data_path, f), ontology, in_memory, max_cache_size, keep_original_dict) for f in names]
        self._dataset = torch.utils.data.ConcatDataset(datasets)

    def get_sampler(self, batch_size, drop_last=False, replacement=False, is_random=True, distributed=False, rank=None, world_size=None):
        if is_random:
            return MultiSetSampler(self._dataset.datasets, batch_size, drop_last, replacement, distributed=distributed, rank=rank, world_size=world_size)
        return MultiSetSequencialSampler(self._dataset.datasets, batch_size, drop_last, distributed=distributed, rank=rank, world_size=world_size)

    def get_loader(self, batch_size, collater, num_workers, use_cuda, drop_last=False, replacement=False, is_random=True, distributed=False):
        sampler = self.get_sampler(batch_size, drop_last, replacement, is_random, distributed)
        return torch.utils.data.DataLoader(dataset=self._dataset, batch_sampler=sampler, num_workers=num_workers, collate_fn=collater.collate,
                                           pin_memory=use_cuda)


class ParserError(Exception):
    pass


_NEGATED = re.compile(r"not\((\w|\s)+\)")
_BLANK = ("_", "scene")
# operator -> kinds of its arguments.  v: a vocabulary token; n: a noun slot (vocabulary token, "_" or "scene"); r: a relation;
# b: a bool; c: a category (attribute / class family, "name", "type"); V2 / V+ / R+: a list of exactly two / at least one such tokens
_SIGNATURES = {"select": ("n",), "filter": ("v",), "relate": ("r", "b", "n"), "verify_rel": ("r", "b", "n"), "choose_rel": ("R+", "b", "n"),
               "query_attr": ("c",), "all_same": ("c",), "all_different": ("c",), "two_same": ("c",), "two_different": ("c",),
               "choose_attr": ("V2",), "verify_attrs": ("V+",), "compare": ("v", "b"), "exist": (), "and": (), "or": ()}
_BRANCH_OPS = ("select", "filter", "relate")
_TWO_BRANCH = ("and", "or", "two_same", "two_different", "compare")


class GQAProgramVerifier(object):
    """Accepts exactly the programs the reference's verifier accepts (golden g13): operator names, argument counts and kinds,
    branch structure.  `verify` returns True or raises ParserError."""

    def __init__(self, attribute_json_path, class_json_path, vocab_json_path, relation_json_path):
        from .gqa_ops import GQAOntology
        self._ontology = GQAOntology(attribute_json_path, class_json_path, vocab_json_path, None, relation_json_path=relation_json_path)

    @staticmethod
    def _plain(tokens):
        """Strip `not(...)`; like the reference, whitespace is only stripped when some token of the list is negated (:31-46)."""
        negated = [_NEGATED.match(t.strip()) is not None for t in tokens]
        if not any(negated):
            return list(tokens)
        return [t.strip()[4:-1] if n else t.strip() for t, n in zip(tokens, negated)]

    def _check(self, name, kind, arg):
        ont = self._ontology
        vocab = ont._vocabulary['arg_to_idx']
        if kind == "b":
            if not isinstance(arg, bool):
                raise ParserError("'%s': a flag argument must be a boolean, got %s" % (name, type(arg)))
        elif kind == "c":
            if arg not in ont._class_dict and arg not in ont._attribute_dict and arg not in ('name', 'type'):
                raise ParserError("'%s' has an unknown category argument: %s" % (name, arg))
        elif kind in ("V2", "V+", "R+"):
            if kind == "V2" and len(arg) != 2:
                raise ParserError("'%s' must have 2 options, but has %d" % (name, len(arg)))
            if len(arg) == 0:
                raise ParserError("'%s' must have at least one option" % name)
            for t in self._plain(arg):
                if not (ont.is_relation(t.lower()) if kind == "R+" else t.lower() in vocab):
                    raise ParserError("'%s' option is not a %s: %s" % (name, "relation" if kind == "R+" else "vocabulary token", t))
        else:
            t = self._plain([arg])[0].lower()
            ok = ont.is_relation(t) if kind == "r" else (t in vocab or (kind == "n" and t in _BLANK))
            if not ok:
                raise ParserError("'%s' argument is not a %s: %s" % (name, {"r": "relation", "n": "noun", "v": "vocabulary token"}[kind], arg))

    def _verify_op(self, name, arguments):
        if name not in _SIGNATURES:
            raise ParserError("Invalid operator: %s" % name)
        kinds = _SIGNATURES[name]
        if len(arguments) != len(kinds):
            raise ParserError("'%s' must have %d argument(s), but has %d argument(s)." % (name, len(kinds), len(arguments)))
        for kind, arg in zip(kinds, arguments):
            self._check(name, kind, arg)

    def verify(self, program):
        if 'last_op' not in program or 'operator' not in program['last_op']:
            raise ParserError("The 'last_op' / 'operator' field is missing: " + str(program))
        last = program['last_op']['operator']
        if last in _BRANCH_OPS:
            raise ParserError("'%s' is not a terminal operator" % last)
        self._verify_op(last, program['last_op']['arguments'])
        if 'branches' not in program:
            raise ParserError("The 'branches' field is missing: " + str(program))
        want = 2 if last in _TWO_BRANCH else 1
        if len(program['branches']) != want:
            raise ParserError("'%s' must have exactly %d branch(es)." % (last, want))
        for branch in program['branches']:
            for i, op in enumerate(branch):
                if 'operator' not in op:
                    raise ParserError("The 'operator' field is missing: " + str(op))
                if (i == 0) != (op['operator'] == 'select') or op['operator'] not in _BRANCH_OPS:
                    raise ParserError("A branch is a 'select' followed by 'filter' / 'relate' operators: " + str(op['operator']))
                if 'arguments' not in op:
                    raise ParserError("The 'arguments' field is missing: " + str(op))
                self._verify_op(op['operator'], op['arguments'])
        return True
