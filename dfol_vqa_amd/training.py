"""Loss and train step (reference: src/nsvqa/train/trainer.py:181-262 `_compute_loss`, :429-442 `_train_batch`).

Only the step math is here; epoch loops, checkpoint cadence, metrics tables and the visualiser of VQATrainer are
out of scope.  The log-probabilities come from the HIP forward kernels and their gradients from the HIP backward
kernels (ops.py); the scalar loss on the [P]-vector is a few tensor ops.
"""

import torch
import torch.nn as nn

from .fol_types import QuestionType
from . import parallel

_YES = ('yes', 'yeah', 'yep', 'yup', 'aye', 'yea')


def safe_log(x):                                         # util.py:22-25
    return x.clamp(min=1e-20).log()


def _targets(values, device):
    """0/1 target vector on the device: through the content-keyed upload cache on a GPU (a fresh torch.tensor(..., device=) is a pageable,
    stream-synchronising copy per step, and not capturable in a HIP graph)."""
    if torch.device(device).type == "cuda":
        import numpy as np
        from .host_util import upload
        return upload(np.asarray(values, np.float32), device)
    return torch.tensor(values, dtype=torch.float32, device=device)


def compute_loss(program_batch_list, prediction, l1_lambda=0.0, parameters=None, l1_scale=1.0):
    """trainer.py:181-262 for STATEMENT / BINARY / QUERY predictions (sum over the batch, not yet divided by it).

    `l1_scale`: with R data-parallel ranks every rank adds the regulariser to its local loss and the gradients are then
    summed, so each rank passes 1/R and the summed gradient carries the term once, as the single-process step does."""
    lp = prediction['log_probability']
    device = lp.device
    qtype = prediction['type']
    if qtype == QuestionType.STATEMENT:                  # :182-183 returns before the L1 block
        return -lp.sum()
    if qtype == QuestionType.BINARY:                   # :185-194
        host = [float(a in _YES) for pb in program_batch_list for a in pb._answers]
        target = _targets(host, device)
        loss = nn.functional.binary_cross_entropy(lp.exp(), target, reduction='sum')
    elif qtype == QuestionType.QUERY:                    # :207-230
        answers = [a for pb in program_batch_list for a in pb._answers]
        target = [[a == o for o in op] for a, op in zip(answers, prediction['options'])]
        tflat = _targets([float(x) for t in target for x in t], device)
        if lp.is_cuda:                                   # per-question sums by the segment kernel (index_add is atomic: not repeatable)
            import numpy as np
            from . import ops
            from .host_util import upload
            seg_off = upload(np.concatenate([[0], np.cumsum([len(t) for t in target])]).astype(np.int32), device)
            denom = ops.segment_sum_rows(lp.exp().unsqueeze(1).contiguous(), seg_off).squeeze(1)
        else:
            seg = torch.tensor([i for i, t in enumerate(target) for _ in t], dtype=torch.int64, device=device)
            denom = torch.zeros(len(target), dtype=torch.float32, device=device).index_add(0, seg, lp.exp())
        loss = safe_log(denom).sum() - (tflat * lp).sum()
    else:
        raise NotImplementedError("direct-supervision losses (OBJECT_STATEMENT / SCENE_GRAPH) are out of scope")
    if l1_lambda and parameters is not None:             # :258-260
        allp = torch.cat([p.view(-1) for p in parameters if p.requires_grad])
        loss = loss + l1_scale * l1_lambda * torch.norm(allp, 1) / max(1, allp.numel())
    return loss


def _direct_grad(on):
    from .visual_oracle import direct_grad
    return direct_grad(on)


class FusedClipAdam(object):
    """nn.utils.clip_grad_norm_ + torch.optim.Adam.step() (trainer.py:439-441) over the flat gradient bucket as two launches of this library
    (three for a capturable optimizer: its step counters live on the device) instead of seventeen of torch's foreach forms
    (csrc/dfol_optim.hip).  The optimizer object stays the owner of the hyper-parameters and of the state - `exp_avg`, `exp_avg_sq`, `step`
    are the tensors torch.optim.Adam itself would have created, so `optimizer.state_dict()` / `load_state_dict()` and a later plain
    `optimizer.step()` keep working.  `FusedClipAdam.make(...)` returns None when the fused form does not apply (another optimizer class,
    amsgrad / maximize, several hyper-parameter groups, parameters that are not the bucket's, DFOL_FUSED_ADAM=0): callers then take torch's path."""

    @staticmethod
    def make(optimizer, bucket):
        import os
        if bucket is None or os.environ.get("DFOL_FUSED_ADAM", "1") == "0" or type(optimizer) is not torch.optim.Adam:
            return None
        groups = optimizer.param_groups
        g0 = groups[0]
        keys = ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize", "capturable")
        if any(any(g.get(k) != g0.get(k) for k in keys) for g in groups) or g0.get("amsgrad") or g0.get("maximize") or g0.get("differentiable"):
            return None
        if isinstance(g0["lr"], torch.Tensor):
            return None
        params = [p for g in groups for p in g["params"]]
        if len(params) != len(bucket.params) or any(a is not b for a, b in zip(params, bucket.params)):
            return None
        if not params or any((not p.is_cuda) or p.dtype != torch.float32 or not p.is_contiguous() for p in params) or not bucket.flat.is_cuda:
            return None
        return FusedClipAdam(optimizer, bucket, params)

    def __init__(self, optimizer, bucket, params):
        import numpy as np
        from . import _lib
        self._opt, self._bucket, self._params = optimizer, bucket, params
        g0 = optimizer.param_groups[0]
        self._capturable = bool(g0.get("capturable", False))
        dev = bucket.flat.device
        for p in params:                                     # the state torch.optim.Adam._init_group creates on its first step
            st = optimizer.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=dev) if self._capturable else torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        lib = _lib.load()
        chunk = lib.dfol_clip_adam_chunk()
        self._partials = torch.empty(lib.dfol_grad_sqnorm_parts(), dtype=torch.float32, device=dev)
        self.total_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        numel = [p.numel() for p in params]
        goff = np.concatenate([[0], np.cumsum(numel)])[:-1]
        ct, cs = [], []
        for t, n in enumerate(numel):
            for a in range(0, n, chunk):
                ct.append(t)
                cs.append(a)
        up = lambda a, dt: torch.from_numpy(np.asarray(a, dt)).to(dev)
        self._goff, self._numel = up(goff, np.int64), up(numel, np.int64)
        self._chunk_tensor, self._chunk_start, self._n_chunks = up(ct, np.int32), up(cs, np.int64), len(ct)
        self._tables_key, self._tables = None, None

    def _address_tables(self):
        """Device tables of the parameters' and the state's addresses, rebuilt when any of them moved (load_state_dict, .to())."""
        import numpy as np
        st = [self._opt.state[p] for p in self._params]
        key = tuple((p.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(), s["step"].data_ptr()) for p, s in zip(self._params, st))
        if key != self._tables_key:
            dev = self._bucket.flat.device
            cols = [np.asarray([k[i] for k in key], np.int64) for i in range(4)]
            self._tables = [torch.from_numpy(c).to(dev) for c in cols]
            self._tables_key = key
        return self._tables

    def step(self, clip_norm):
        from . import _lib
        g0 = self._opt.param_groups[0]
        bucket = self._bucket
        if any(p.grad is None or p.grad.data_ptr() != bucket.flat.data_ptr() + 4 * off for p, off in zip(self._params, self._goff_host())):
            raise _lib.DfolError("FusedClipAdam: a parameter's .grad is no longer its view of the gradient bucket (zero the bucket, not the optimizer)")
        param, m, v, step = self._address_tables()
        step_host = 0.0
        if not self._capturable:                             # host step counters: torch keeps them as 0-d CPU tensors
            steps = set(float(self._opt.state[p]["step"]) for p in self._params)
            if len(steps) > 1:
                # per-parameter counts differ (a partial load_state_dict): the kernel takes ONE host step for its bias corrections
                if clip_norm is not None:
                    nn.utils.clip_grad_norm_(self._params, clip_norm)
                self._opt.step()
                return None
            for p in self._params:
                self._opt.state[p]["step"] += 1
            step_host = float(self._opt.state[self._params[0]]["step"])
        _lib.call("dfol_grad_sqnorm_f32", bucket.flat.data_ptr(), bucket.flat.numel(), self._partials.data_ptr(), _lib._stream())
        _lib.call("dfol_clip_adam_f32", bucket.flat.data_ptr(), self._partials.data_ptr(), param.data_ptr(), m.data_ptr(), v.data_ptr(),
                  self._goff.data_ptr(), self._numel.data_ptr(), len(self._params), self._chunk_tensor.data_ptr(), self._chunk_start.data_ptr(),
                  self._n_chunks, step.data_ptr() if self._capturable else None, step_host, float(g0["lr"]), float(g0["betas"][0]),
                  float(g0["betas"][1]), float(g0["eps"]), float(g0["weight_decay"]), -1.0 if clip_norm is None else float(clip_norm),
                  self.total_norm.data_ptr(), _lib._stream())
        _lib.keep_alive((self._partials, param, m, v, step, self._goff, self._numel, self._chunk_tensor, self._chunk_start, self.total_norm))
        torch.autograd.graph.increment_version(self._params)  # (written behind autograd's back: version-keyed weight images must miss)
        return self.total_norm

    def _goff_host(self):
        if getattr(self, "_goff_h", None) is None:
            off, acc = [], 0
            for p in self._params:
                off.append(acc)
                acc += p.numel()
            self._goff_h = off
        return self._goff_h


def clip_and_step(model, optimizer, clip_norm, bucket=None, fused=None):
    """trainer.py:439-441.  `fused`: a FusedClipAdam (or None: torch's clip_grad_norm_ + optimizer.step())."""
    if fused is not None:
        fused.step(clip_norm)
    else:
        nn.utils.clip_grad_norm_(model.parameters(), clip_norm)
        optimizer.step()


_FUSED_OF = {}


def _fused_for(optimizer, bucket):
    """One FusedClipAdam per (optimizer, bucket) pair, made on first use."""
    key = (id(optimizer), id(bucket))
    hit = _FUSED_OF.get(key)
    if hit is None or hit[0]() is not optimizer or hit[1]() is not bucket:
        import weakref
        if len(_FUSED_OF) > 64:
            _FUSED_OF.clear()
        hit = _FUSED_OF[key] = (weakref.ref(optimizer), weakref.ref(bucket) if bucket is not None else (lambda: None), FusedClipAdam.make(optimizer, bucket))
    return hit[2]


def train_batch(model, optimizer, data, clip_norm, global_batch_size=None, group=None, l1_lambda=0.0, bucket=None, sync_loss=True):
    """trainer.py:429-442: zero_grad -> forward -> loss / B -> backward -> clip_grad_norm_ -> step.

    With `group` set (one process per GPU), `data` is this rank's shard: the loss is divided by the GLOBAL batch size, the
    gradients of all ranks are summed with ONE all-reduce of a flat bucket (`bucket`: a persistent parallel.GradBucket, no
    gather/scatter copies), and every rank applies the same clip + step, so parameters that start equal
    (parallel.broadcast_parameters) stay equal.  The L1 term is scaled by 1/world so the summed gradient carries it once.
    Returns (this rank's share of the summed loss, result): all-reduce the scalar if the global value is wanted.  sync_loss=False
    returns the loss as a 0-d GPU tensor instead of a float: the reference reads it back every step (trainer.py:438), which makes the
    host wait for the GPU before it can prepare the next batch (~2 ms of a 16 ms step); accumulate the tensor and read it per epoch."""
    if bucket is not None:
        bucket.zero_()
    else:
        optimizer.zero_grad()
    result = model(data, True)
    world = 1
    if group is not None:
        import torch.distributed as dist
        world = dist.get_world_size(group)
    loss = compute_loss(data, result, l1_lambda, list(model.parameters()), l1_scale=1.0 / world)
    local_b = sum(d.batch_size() for d in data)
    b = global_batch_size if global_batch_size is not None else local_b
    loss = loss / b
    with _direct_grad(bucket is not None):
        loss.backward()
    if group is not None:
        if bucket is not None:
            bucket.allreduce(group)
        else:
            parallel.allreduce_gradients(model.parameters(), group)
    clip_and_step(model, optimizer, clip_norm, bucket, _fused_for(optimizer, bucket) if bucket is not None else None)
    return (float(loss.detach()) * b if sync_loss else loss.detach() * b), result


class GraphedTrainStep(object):
    """trainer.py:429-442 for ONE fixed list of ProgramBatches as captured HIP graphs: zero the gradients -> forward -> loss / B -> backward
    -> [all-reduce] -> clip_grad_norm_ -> optimizer step, replayed with one host call.  A train step is 250 - 700 launches (the calibrator
    phases are host-bound: 4.6 ms of GPU work in a 6.9 ms step), and its launch sequence is static for a given batch shape, exactly like
    the inference forward's (interpreter.GraphedForward).  The optimizer must be capturable (torch.optim.Adam(..., capturable=True): its
    step counter lives on the device); new scenes of the same shapes are served by copying into `program_batch._object_features`; a batch
    with other programs or shapes needs its own capture.  `loss` is a 0-d device tensor that every replay overwrites (this rank's share of
    the summed loss).

    Data parallelism (`group`, one process per GPU; `data` is this rank's shard, the loss is divided by the GLOBAL batch size, the bucket
    is required: one flat buffer, one collective).  The default is the ROBUST form: TWO graphs - A = zero + forward + loss + backward,
    B = clip + optimizer step - with the bucket's all-reduce issued EAGERLY between the two replays: no collective is ever captured, so
    nothing of the process group's machinery (its watchdog thread polls the events of issued collectives) can meet a stream capture; one
    extra launch boundary is ~10 us of a 3 - 12 ms step, and it works with every backend (gloo included).  `graph_collective=True` is
    the single-graph form with RCCL's all-reduce as a node of the graph (RCCL only): it aborted the process from the watchdog thread on
    one box of round 3 (SIGABRT, not catchable), so it is opt-in and a caller that wants to survive it must run it in a child process.
    Every rank must construct and replay the same number of times.

    Replays rewrite the parameters on the device without touching their version counters, and every packed-weight cache of the library is
    keyed on (data_ptr, _version): after each replay the versions of the trainable parameters are bumped, so an eager / GraphedForward
    evaluation between replays repacks the CURRENT weights (the trainer's train-epoch-then-validate loop).

    Same lifetime rule as GraphedForward: tensors that came out of caches are referenced from `self._keep`."""

    def __init__(self, model, optimizer, data, clip_norm, l1_lambda=0.0, bucket=None, warmup=2, group=None, global_batch_size=None,
                 graph_collective=False):
        from ._lib import keeping
        if warmup < 1:
            # the first eager step allocates the optimizer state, fills the host-side caches and packs the weight images: without it the
            # capture would miss the pack kernels of a warm cache and replay stale weights
            raise ValueError("GraphedTrainStep needs warmup >= 1")
        self._model, self._opt, self._data, self._bucket = model, optimizer, data, bucket
        self._clip, self._l1 = clip_norm, l1_lambda
        self._group, self._world, self._graph_collective = group, 1, bool(graph_collective)
        if group is not None:
            import torch.distributed as dist
            if bucket is None:
                raise ValueError("GraphedTrainStep with a process group needs the persistent gradient bucket (parallel.GradBucket)")
            if getattr(bucket, "_segments", None) is not None:
                raise ValueError("GraphedTrainStep issues ONE all-reduce after the backward: use a bucket without enable_overlap()")
            self._world = dist.get_world_size(group)
            if self._graph_collective and dist.get_backend(group) != "nccl":
                raise ValueError("graph_collective=True captures the all-reduce: RCCL (backend 'nccl') only; host-side backends cannot be captured")
        self._local_batch = sum(d.batch_size() for d in data)
        self._batch = global_batch_size if global_batch_size is not None else self._local_batch
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("GraphedTrainStep needs a capturable optimizer (torch.optim.Adam(params, lr=..., capturable=True))")
        self._params = [p for g in optimizer.param_groups for p in g["params"]]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                        # warm-up on a side stream (fills every host-side cache, allocates optimizer state)
            for _ in range(warmup):
                self._front()
                if group is not None:
                    bucket.allreduce(group)
                self._back()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if group is not None:
            self._quiesce()
        torch.cuda.empty_cache()                             # the warm-up's activations go back before the graph's private pool is sized
        self._keep = []
        from . import _lib as _L
        _L.CAPTURE_RANGE_HOST = self._range_host = _L.new_range_host()      # (pinned memory for the captured copy of the fp16-range status word)
        _L.CAPTURE_RANGE_WORD = self._range_word = torch.zeros(1, dtype=torch.int32, device=next(model.parameters()).device)   # (this graph's own word)
        _L.CAPTURE_RANGE_CHECKS[:] = []
        # (thread-local capture mode: the check for capture-unsafe calls is restricted to the capturing thread; a process group's watchdog
        # thread may query an event of an earlier collective at any time, which under the default "global" mode invalidates the capture)
        mode = {} if group is None else {"capture_error_mode": "thread_local"}
        self._graph = torch.cuda.CUDAGraph()
        self._graph_b = None
        if group is None or self._graph_collective:
            with keeping(self._keep), torch.cuda.graph(self._graph, **mode):
                self.loss, self.result = self._front()
                if group is not None:
                    bucket.allreduce(group)
                self._back()
        else:
            with keeping(self._keep), torch.cuda.graph(self._graph, **mode):
                self.loss, self.result = self._front()
            self._graph_b = torch.cuda.CUDAGraph()
            with keeping(self._keep), torch.cuda.graph(self._graph_b, pool=self._graph.pool(), **mode):
                self._back()
        self._range_checks, _L.CAPTURE_RANGE_CHECKS[:] = list(_L.CAPTURE_RANGE_CHECKS), []
        _L.CAPTURE_RANGE_HOST = None
        _L.CAPTURE_RANGE_WORD = None

    @staticmethod
    def _quiesce():
        """No collective may be in flight, nor still on the watchdog's list, when a capture starts: the device is idle (the caller
        synchronised), and the watchdog drops completed work within one poll interval (100 ms)."""
        import time
        time.sleep(0.25)

    def _front(self):
        if self._bucket is not None:
            self._bucket.zero_()
        else:
            self._opt.zero_grad(set_to_none=False)
        result = self._model(self._data, True)
        loss = compute_loss(self._data, result, self._l1, list(self._model.parameters()), l1_scale=1.0 / self._world) / self._batch
        with _direct_grad(self._bucket is not None):
            loss.backward()
        return loss.detach() * self._batch, result

    def _back(self):
        clip_and_step(self._model, self._opt, self._clip, self._bucket, _fused_for(self._opt, self._bucket) if self._bucket is not None else None)

    def __call__(self):
        self._graph.replay()
        if self._graph_b is not None:
            self._bucket.allreduce(self._group)              # eager, between the two replays, on the replays' stream
            self._graph_b.replay()
        torch.autograd.graph.increment_version(self._params)   # the replay rewrote them: version-keyed weight caches must miss
        for check in self._range_checks:                       # fp16 range flags of the replay BEFORE this one (no host wait per step)
            check(sync=False)
        return self.loss, self.result


# ---- evaluation metrics (trainer.py:64-86, 264-318, 477-485) ---------------------------------------------
OP_INDEX = {'query_attr': 1, 'choose_attr': 2, 'verify_attrs': 3, 'choose_rel': 4, 'verify_rel': 5, 'exist': 6, 'and': 7, 'or': 8,
            'all_same': 9, 'all_different': 10, 'two_same': 11, 'two_different': 12, 'compare': 13, 'object_attr': 14, 'object_rel': 15,
            'scene': 16}
ERROR_DIM = len(OP_INDEX) + 1


def compute_evaluation_metrics(program_batch_list, prediction, first_answer=False):
    """Error rate of a batch (trainer.py:264-318): a QUERY answer set scores 1/|set| when it contains the gold answer."""
    import numpy as np
    answers = [a for pb in program_batch_list for a in pb._answers]
    if first_answer:
        match = [a in op[0] if len(op) > 0 else False for a, op in zip(answers, prediction['answer'])]
    elif prediction['type'] == QuestionType.QUERY:
        match = [float(any(a in o for o in op)) / float(len(op)) if len(op) > 0 else 0 for a, op in zip(answers, prediction['answer'])]
    else:
        match = [any(a in o for o in op) if len(op) > 0 else False for a, op in zip(answers, prediction['answer'])]
    return 1.0 - float(np.mean(np.array(match, dtype=np.float32)))


def _terminal_name(pb):
    """The last operator batch's name (without moving a lazily uploaded ProgramBatch's operators to the device)."""
    name = getattr(pb, "terminal_op_name", None)
    return name() if name is not None else pb._op_batch_list[-1]._op_name


def accumulate_test_batch(error, total_example_num, data, prediction, first_answer=False):
    """trainer.py:477-485: overall error in slot 0, per-terminal-operator error in the operator's slot."""
    b = sum(d.batch_size() for d in data)
    err = b * compute_evaluation_metrics(data, prediction, first_answer)
    slot = OP_INDEX[_terminal_name(data[0])]
    error[0] += err
    error[slot] += err
    total_example_num[0] += b
    total_example_num[slot] += b
    return error, total_example_num


def test_epoch(model, loader, device, first_answer=False):
    """The validation / test loop of VQATrainer._test_epoch (trainer.py:444-475): per-operator error rates over a stream of collated
    batches (lists of ProgramBatches on the host), one thread, software-pipelined - a batch's launches are enqueued with
    `forward_async`, the NEXT batch is pulled from `loader` (its collate), given its sparse maps, uploaded AND launched while the device
    runs, and only then are the first batch's answers read back and scored (two batches in flight: the device never waits for the host).  -> error / total_example_num, ERROR_DIM floats (slot 0 overall)."""
    import numpy as np
    import torch
    error = np.zeros(ERROR_DIM, dtype=np.float32)
    total = np.zeros(ERROR_DIM, dtype=np.float32)

    def prepared(it):
        for data in it:
            if len(data) > 0:
                for d in data:
                    d.create_sparse_tensors()
                yield [d.to_cuda(device) for d in data]

    model.eval()
    with torch.no_grad():
        stream = prepared(iter(loader))
        data = next(stream, None)
        pending = model.forward_async(data, False) if data is not None else None
        while data is not None:
            nxt = next(stream, None)                          # host work of the next batch under this batch's kernels ...
            nxt_pending = model.forward_async(nxt, False) if nxt is not None else None     # ... and its launches queued behind them
            accumulate_test_batch(error, total, data, pending.result(), first_answer)
            data, pending = nxt, nxt_pending
    with np.errstate(invalid="ignore", divide="ignore"):
        return error / total                                  # (NaN in the slot of an operator no batch ended with, as the reference's)


def collect_predictions(program_batch_list, prediction, is_submission=False):
    """The prediction records VQATrainer._print_predictions appends per test batch (trainer.py:320-337): question id, predicted
    answer (the first of the arg-max set for binary questions and submissions, the whole set for QUERY questions), question type
    ('open' for query_attr programs, 'binary' otherwise) and, for QUERY questions, the option list."""
    question_ids = [qid for pb in program_batch_list for qid in pb._meta_data['question_ids']]
    if is_submission:
        return [{'questionId': qid, 'prediction': p[0]} for qid, p in zip(question_ids, prediction['answer'])]
    query = prediction['type'] == QuestionType.QUERY
    answers = [p if query else p[0] for p in prediction['answer']]
    types = ['open' if _terminal_name(pb) == 'query_attr' else 'binary' for pb in program_batch_list for _ in range(pb.batch_size())]
    if query:
        return [{'questionId': qid, 'prediction': a, 'type': t, 'options': opt} for qid, a, t, opt in zip(question_ids, answers, types, prediction['options'])]
    return [{'questionId': qid, 'prediction': a, 'type': t} for qid, a, t in zip(question_ids, answers, types)]


def metric_dict(error):                                  # trainer.py:85-86
    return dict(zip(['over_all'] + list(OP_INDEX.keys()), [float(x) for x in error]))
