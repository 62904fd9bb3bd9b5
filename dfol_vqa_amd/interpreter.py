"""The program interpreter (reference: src/nsvqa/nn/interpreter/batch_base_interpreter.py,
batch_gqa_interpreter.py, data_parallel.py:15-50 and the featurizer of
src/nsvqa/data/batch_gqa_boxfeatures_pipeline.py:193-281)."""

import inspect
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops as L
from . import gqa_ops as gqa
from .fol_types import BatchAttentionState, BatchVariableSet, BatchWorld, QuestionType
from .host_util import keeping, reverse_dependencies, upload


def gather_results(outputs, target_device=None, is_cuda=True):
    """Concatenate per-ProgramBatch results (data_parallel.py:15-50)."""
    if outputs[0]['type'] == QuestionType.SCENE_GRAPH:
        raise NotImplementedError("scene-graph (direct supervision) results are out of scope (SURVEY.md §2 row 4)")
    log_probability = torch.cat([o['log_probability'].reshape(-1) for o in outputs]) if len(outputs) > 1 else \
        outputs[0]['log_probability'].reshape(-1)
    answer = [a for o in outputs for a in o['answer']]
    answer_log_probability = [a for o in outputs for a in o['answer_log_probability']]
    if outputs[0]['type'] == QuestionType.QUERY:
        options = [x for o in outputs for x in o['options']]
    else:
        options = outputs[0]['options']
    return {'answer': answer, 'log_probability': log_probability, 'options': options, 'variable_set': None, 'type': outputs[0]['type'],
            'cumulative_loss': sum(o['cumulative_loss'] for o in outputs), 'variable_sets_num': sum(o['variable_sets_num'] for o in outputs),
            'answer_log_probability': answer_log_probability}


class BatchGQABoxFeaturizer(nn.Module):
    """Object and pair features of a scene (batch_gqa_boxfeatures_pipeline.py:193-281).

    featurize_scene keeps the reference's return dict.  The pair matrix [pairs, 2(D+4)+4] is the reference's
    layout; it is materialised only because compute_all_log_likelihood_2 asks for it (full cached tables)."""

    def __init__(self, featurizer_network=None):
        super(BatchGQABoxFeaturizer, self).__init__()
        self._featurizer_network = featurizer_network

    def featurize_scene(self, device, objects_list, batch_index, meta_data, world_geometry=None):
        object_num = objects_list.size()[0]
        raw_cols = objects_list.size()[1]
        feat = objects_list[:, :raw_cols - 6]                    # a strided view: the GEMM reads it in place
        net = self._featurizer_network
        direct = net is not None and getattr(net, "_network", None) is not None and hasattr(net, "output_width") and feat.is_cuda and \
            not (torch.is_grad_enabled() and (feat.requires_grad or any(p.requires_grad for p in net.parameters())))
        if direct:
            # inference: the featurizer's last layer writes straight into the object matrix (row stride D), no [O, D - 4] copy
            D = net.output_width() + 4
            obj = torch.empty(object_num, D, dtype=torch.float32, device=device)
            net(feat, out=obj[:, :D - 4])
        else:
            f = net(feat) if net is not None and getattr(net, "_network", None) is not None else feat
            D = f.size(1) + 4
            obj = torch.empty(object_num, D, dtype=torch.float32, device=device)
            obj[:, :D - 4] = f
        L.box_positions(objects_list, obj, D - 4)                # :208-211
        geo = world_geometry
        pair = None
        if geo is not None and geo._pair_num > 0:
            pair = L.pair_features(obj, D, geo._obj_off, geo._pair_off, geo._batch_size, max(geo._n_list), geo._pair_num,
                                   pair_index=geo.pair_index)                                                                # :252-279
        return {'attribute_features': obj, 'relation_features': {'features': pair, 'index': None}, 'object_num': object_num}


class _LazyGather(object):
    """Per-ProgramBatch results whose answers are still queued (inside a graph capture); gather() after the queue has run."""

    def __init__(self, results, device):
        self.results, self.device = results, device

    def gather(self):
        return gather_results(self.results, self.device, True)


class PendingForward(object):
    """A forward whose launches are enqueued but whose answers have not been read back (`BatchInterpreterBase.forward_async`): the host
    is free - to collate the next ProgramBatch, say - while the device runs this one; `result()` waits for the device, reads the terminal
    operators' log-probabilities back and returns what `forward` would have."""

    def __init__(self, lazy, queue):
        self._lazy, self._queue, self._result = lazy, queue, None

    def result(self):
        if self._result is None:
            for fill in self._queue:
                fill()
            self._result = self._lazy.gather()
            self._lazy = self._queue = None
        return self._result


class GraphedForward(object):
    """The inference forward of one fixed list of ProgramBatches as a captured HIP graph (the launch sequence of a program batch is
    static: 16 launches for a 3-hop program).  Replaying it removes the per-launch host work, which is 7-10 % of a step at 36
    objects per scene.  The object features are read from the ProgramBatches' own tensors, so new scenes of the same shapes are
    served by copying into `program_batch._object_features` before `__call__`.  Answers are decoded after the replay.

    Lifetime: the graph holds raw device addresses.  Tensors created during the capture live in the graph's own memory pool; tensors
    that came out of a cache (uploaded index arrays, batch geometry, packed / split weight images, transposed LSTM weights) are
    referenced from `self._keep`, so cache evictions cannot free them under the graph.  The graph reads the weight images of the
    weight VERSION it was captured with: rebuild it after an optimizer step or load_state_dict."""

    def __init__(self, model, program_batch_list, warmup=2):
        from . import gqa_ops, native_exec
        self._model, self._pbs = model, program_batch_list
        with torch.no_grad():
            with native_exec.suspended():                        # (the capture records the Python operator loop: warm ITS caches)
                for _ in range(warmup):                          # fills the host-side caches (geometry, lowered tokens, packed weights)
                    model(program_batch_list, False)
            torch.cuda.synchronize()
            self._queue = []
            self._keep = []                                      # everything the caches handed to the captured launches (host_util.keeping)
            self._graph = torch.cuda.CUDAGraph()
            gqa_ops.DEFERRED.queue = self._queue
            gqa_ops.DEFERRED.outputs = self._outputs = []        # the device tensors the answers are decoded from
            from . import _lib
            _lib.CAPTURE_RANGE_HOST = self._range_host = _lib.new_range_host()     # (the replay copies the fp16-range status word into it)
            dev = program_batch_list[0].device
            _lib.CAPTURE_RANGE_DEV = self._range_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.CAPTURE_RANGE_WORD = self._range_word = torch.zeros(1, dtype=torch.int32, device=dev)   # (this graph's own: see ReplayLanes)
            try:
                with keeping(self._keep), torch.cuda.graph(self._graph):
                    self._lazy = model(program_batch_list, False)
            finally:
                gqa_ops.DEFERRED.queue = None
                gqa_ops.DEFERRED.outputs = None
                _lib.CAPTURE_RANGE_HOST = None
                _lib.CAPTURE_RANGE_DEV = None
                _lib.CAPTURE_RANGE_WORD = None
        self._outputs.extend(r['log_probability'] for r in self._lazy.results)
        seen, uniq = set(), []
        for t in self._outputs + [self._range_dev]:
            if id(t) not in seen:
                seen.add(id(t))
                uniq.append(t)
        self._outputs, self._slots, self._next = uniq, [], 0

    def __call__(self):
        self._graph.replay()
        for fill in self._queue:
            fill()
        return self._lazy.gather()

    # ---- pipelined replays: the host decodes replay i's answers while the device runs replay i + 1 ------------------------------------------------
    def submit(self, depth=2):
        """Replay, and queue behind it the copies of everything the answers are decoded from (log-probabilities, arg-max flags, the fp16-range
        status word) into one of `depth` pinned slots - no host wait.  Returns the ticket `collect` takes.  At most `depth` tickets may be
        outstanding; new features for the next replay may be copied into the ProgramBatches' tensors right after this call (stream order)."""
        while len(self._slots) < max(1, depth):
            self._slots.append({"host": [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in self._outputs], "event": torch.cuda.Event(), "busy": False})
        slot = self._slots[self._next % len(self._slots)]
        if slot["busy"]:
            raise RuntimeError("GraphedForward.submit: %d replays outstanding; collect() the oldest first" % len(self._slots))
        self._next += 1
        self._graph.replay()
        for t, h in zip(self._outputs, slot["host"]):
            h.copy_(t, non_blocking=True)
        slot["event"].record()
        slot["busy"] = True
        return slot

    def collect(self, ticket):
        """Wait for that replay (only), decode its answers from its own copies and return what `forward` would have - with the log-probabilities
        as a host tensor (the device buffers belong to the replays behind it)."""
        from . import gqa_ops
        ticket["event"].synchronize()
        snap = {id(t): h.numpy() for t, h in zip(self._outputs, ticket["host"])}
        gqa_ops.DEFERRED.snapshot = snap
        try:
            for fill in self._queue:
                if getattr(fill, "range_check", False):
                    fill(False, int(snap[id(self._range_dev)][0]))
                else:
                    fill()
            res = self._lazy.gather()
        finally:
            gqa_ops.DEFERRED.snapshot = None
            ticket["busy"] = False
        res = dict(res)
        res['answer'] = [list(a) for a in res['answer']]         # (the queued closures refill the same lists at every replay)
        res['answer_log_probability'] = [list(a) for a in res['answer_log_probability']]
        res['log_probability'] = torch.cat([torch.from_numpy(snap[id(r['log_probability'])].reshape(-1).copy()) for r in self._lazy.results])
        return res


class ReplayLanes(object):
    """Several captured forwards of one batch shape - each over its OWN ProgramBatch tensors, intermediates and fp16-range status word - replayed
    round-robin on as many HIP streams, so that consecutive batches OVERLAP on the device: a forward ends in a dozen small logic launches (a few
    workgroups each, the rest of the 256 CUs idle, ~10 us apart) and begins with the featurizer and the pair kernel, which fill the chip - side by
    side the next batch's head runs in the previous one's tail.  256 questions x 100 objects: 1.52 ms per batch one replay at a time, 1.45 with
    two replays of one graph in flight on one stream (the host's read-back hidden), 1.35 on two lanes (tools/lab/time_step_forms.py).
    submit(fill) -> ticket: `fill(program_batch_list)` (optional) runs on the lane's stream before its replay - copy the batch's new features into
    the lane's tensors there; collect(ticket) -> the result dict of that batch (GraphedForward.collect).  At most one outstanding ticket per lane."""

    def __init__(self, model, program_batch_lists):
        assert len(program_batch_lists) >= 1
        self._graphs = [GraphedForward(model, pbs) for pbs in program_batch_lists]
        dev = program_batch_lists[0][0].device
        self._streams = [torch.cuda.Stream(device=dev) for _ in self._graphs]
        self._next = 0
        cur = torch.cuda.current_stream(dev)
        for s in self._streams:
            s.wait_stream(cur)                                   # (whatever uploaded the batches comes first)

    def __len__(self):
        return len(self._graphs)

    def batches(self, lane):
        return self._graphs[lane]._pbs

    def submit(self, fill=None):
        lane = self._next % len(self._graphs)
        self._next += 1
        g = self._graphs[lane]
        with torch.cuda.stream(self._streams[lane]):
            if fill is not None:
                fill(g._pbs)
            ticket = g.submit(depth=1)
        return (g, ticket)

    def collect(self, ticket):
        return ticket[0].collect(ticket[1])


class BatchInterpreterBase(nn.Module):
    """batch_base_interpreter.py:14-183."""

    def __init__(self, name, oracle, featurizer=None, attention_transfer_state_dim=0, apply_modulation_everywhere=True, cached=False,
                 visual_rule_learner=None, calibrator=None):
        super(BatchInterpreterBase, self).__init__()
        self._featurizer = featurizer
        self._oracle = oracle
        self._name = name
        self._global_step = nn.Parameter(torch.tensor([0], dtype=torch.float), requires_grad=False)
        self._has_modulator = False
        self._attention_transfer_state_dim = attention_transfer_state_dim
        self._apply_modulation_everywhere = apply_modulation_everywhere
        self._cached = cached
        if visual_rule_learner is not None or calibrator is not None:
            raise NotImplementedError("visual_rule_learner / calibrator are None in every reference configuration")

    def _execute(self, op_id, world, operator_batch, input_tuple, is_terminal, is_training):
        raise NotImplementedError

    def parameter_count(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def save(self, export_path_base):                          # :39-40
        torch.save(self.state_dict(), os.path.join(export_path_base, self._name))

    def load(self, import_path_base):                          # :42-43
        self.load_state_dict(torch.load(os.path.join(import_path_base, self._name)), strict=False)

    def build_scene(self, device, object_features, batch_index, meta_data, object_nums=None, question_image=None):
        """batch_base_interpreter.py:45-70: featurizer + oracle MLPs -> cached likelihood tables -> BatchWorld.
        `question_image` (ProgramBatch._question_image; None = one scene per question, the reference's layout): the object rows hold
        every distinct scene once and question k looks at scene question_image[k]."""
        if self._featurizer is None:
            raise NotImplementedError("a featurizer is required (the reference's featurizer-less branch :62-67 is dead code)")
        # needed-columns mode: no pair matrix, no full tables; the oracle keeps hidden activations and evaluates the concept
        # columns a program names.  The fused kernels are forward-only: they run whenever no gradient has to reach the oracle's
        # or the featurizer's weights (inference, and the calibrator-only phases cur6-7 where both are frozen); when those
        # weights train, the same columns come from differentiable tensor ops (visual_oracle._*_autograd).
        oracle_trains = torch.is_grad_enabled() and any(p.requires_grad for m in (self._oracle, self._featurizer)
                                                        if isinstance(m, nn.Module) for p in m.parameters())
        needed = self._cached and isinstance(self._featurizer, BatchGQABoxFeaturizer) and \
            getattr(self._oracle, "supports_needed_columns", lambda: False)()
        if question_image is not None and (not needed or oracle_trains or (self._has_modulator and torch.is_grad_enabled())):
            # shared scenes are served by the needed-columns inference dataflow; the other dataflows (full cached tables, training)
            # take the reference's layout: one copy of its scene per question
            if object_nums is None:
                bi = batch_index if isinstance(batch_index, torch.Tensor) else torch.as_tensor(batch_index)
                object_nums = torch.bincount(bi.to(torch.int64).cpu()).tolist()
            off = np.concatenate([[0], np.cumsum(object_nums)])
            rows = np.concatenate([np.arange(off[i], off[i + 1]) for i in question_image]) if len(question_image) else np.zeros(0, np.int64)
            object_features = object_features.index_select(0, torch.as_tensor(rows, dtype=torch.int64).to(object_features.device))
            object_nums = [int(object_nums[i]) for i in question_image]
            batch_index = torch.as_tensor(np.repeat(np.arange(len(object_nums)), object_nums).astype(np.int64)).to(object_features.device)
            question_image = None
        geometry = BatchWorld(device, object_features.size(0), None, None, batch_index, meta_data,
                              attention_transfer_state_dim=self._attention_transfer_state_dim, object_nums=object_nums,
                              question_image=question_image)
        if needed:
            features = self._featurizer.featurize_scene(device, object_features, batch_index, meta_data, world_geometry=None)
            self._oracle.prepare_scene(geometry, features['attribute_features'], train=oracle_trains)
            return geometry
        if 'world_geometry' in inspect.signature(self._featurizer.featurize_scene).parameters:
            features = self._featurizer.featurize_scene(device, object_features, batch_index, meta_data, world_geometry=geometry)
        else:                   # a featurizer written against the reference's 4-argument signature
            features = self._featurizer.featurize_scene(device, object_features, batch_index, meta_data)
        attribute_features = features['attribute_features']
        relation_features = features['relation_features']
        if self._cached:
            attribute_features, relation_features['features'] = self._oracle.compute_all_log_likelihood_2(
                attribute_features, relation_features['features'])
        geometry._attribute_features = attribute_features
        geometry._relation_features = relation_features
        return geometry

    def forward(self, program_batch_list, is_training, return_trace=False, modulator_switch=True):
        """batch_base_interpreter.py:72-183.  Terminal operators defer reading their log-probabilities back until every
        ProgramBatch has been enqueued: one device->host synchronisation per forward."""
        from . import gqa_ops
        outer = gqa_ops.DEFERRED.queue
        queue = [] if outer is None else outer                 # (a graph capture installs its own queue, see GraphedForward)
        gqa_ops.DEFERRED.queue = queue
        watch = None
        try:
            # `_mlp_math` (config key `mlp_math`, experiment.build_interpreter): "bf16" runs the large dense products of the featurizer and
            # the oracle MLPs - forward, and through the autograd functions' recorded mode their backward - on bf16-rounded operands with
            # fp32 accumulation (BASELINE configs[3]: "bf16 fwd / fp32 logic"); the logic kernels and everything small stay fp32
            from . import _lib
            dev0 = program_batch_list[0].device if program_batch_list else None
            watch = _lib.RangeWatch(dev0) if dev0 is not None and torch.device(dev0).type == "cuda" else None
            with _lib.dense_math(getattr(self, "_mlp_math", None)):
                all_results, all_traces, device = self._run_batches(program_batch_list, is_training, modulator_switch, return_trace)
            if watch is not None:
                check = watch.finish()                           # (runs after the answers' read-backs: raises if a kernel left fp16's range)
                if outer is None and _lib.capturing():
                    _lib.CAPTURE_RANGE_CHECKS.append(check)      # a captured train step: its owner checks after replays
                else:
                    queue.append(check)
        finally:
            gqa_ops.DEFERRED.queue = outer
            if watch is not None:
                _lib.load().dfol_set_range_status(None)          # (also when the forward raised before watch.finish())
        if outer is None:
            for fill in queue:
                fill()
        result = gather_results(all_results, device, True) if outer is None else _LazyGather(all_results, device)
        return (result, all_traces) if return_trace else result

    def forward_async(self, program_batch_list, is_training=False, modulator_switch=True):
        """`forward` without its device->host synchronisation: enqueues every launch and returns a PendingForward.  The reference's test()
        loop (trainer.py:685-720) reads the answers of a batch before it collates the next; with this the two overlap on one thread."""
        from . import gqa_ops
        if gqa_ops.DEFERRED.queue is not None:
            raise RuntimeError("forward_async inside a graph capture or another pending forward's launch")
        queue = []
        gqa_ops.DEFERRED.queue = queue
        try:
            lazy = self.forward(program_batch_list, is_training, modulator_switch=modulator_switch)
        finally:
            gqa_ops.DEFERRED.queue = None
        return PendingForward(lazy, queue)

    def _native_spec(self, is_training, modulator_switch, return_trace):
        """The lowering spec when this forward may run on the native executor (native_exec / native_plan: one C call per ProgramBatch instead
        of one Python dispatch per operator), else None: inference without gradients reaching the oracle or the calibrator, soft quantifiers, no
        trace, and not inside a graph capture (a capture records the Python loop's launches, as before).  Since round 6 the calibrated forward
        (activate_attention_transfer, the reference's default), shared scenes and bf16 relation tiles stay on the executor."""
        from . import _lib, native_exec
        if not native_exec.enabled() or is_training or return_trace or _lib.capturing() or getattr(self, "_hard_mode", False):
            return None
        calibrate = bool(self._has_modulator and modulator_switch)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return None                                      # gradients reach the oracle or the calibrator: the differentiable Python operators
        if not hasattr(self, "_ontology"):
            return None
        return native_exec.model_spec(self, calibrate)

    def _run_batches(self, program_batch_list, is_training, modulator_switch, return_trace=False):
        from . import _lib
        all_traces, all_results = [], []
        device = program_batch_list[0].device
        spec = self._native_spec(is_training, modulator_switch, return_trace)
        for program_batch in program_batch_list:
            if spec is not None and isinstance(program_batch._object_features, torch.Tensor) and program_batch._object_features.is_cuda:
                from . import gqa_ops, native_exec
                plan = native_exec.plan_for(self, program_batch, spec)
                if plan is not None:
                    all_results.append(native_exec.run(self, program_batch, plan, gqa_ops.DEFERRED.queue, give_answer=not is_training))
                    all_traces.append([])
                    continue
            _lib.note("python_program")                         # (route counter: a ProgramBatch that ran the Python operator loop)
            world = self.build_scene(program_batch.device, program_batch._object_features, program_batch._object_batch_index,
                                     program_batch._meta_data, object_nums=getattr(program_batch, "_object_nums", None),
                                     question_image=getattr(program_batch, "_question_image", None))
            if self._has_modulator and modulator_switch:
                self._calibration_passes(world, program_batch, device, is_training)
            if world._lazy is not None:
                # with a training calibrator the attentions carry gradients and the relates run on the generic cell
                self._oracle.prefetch_relations(world, program_batch, fused=not (self._has_modulator and modulator_switch and torch.is_grad_enabled()))
                if hasattr(self._oracle, "prefetch_attributes"):
                    self._oracle.prefetch_attributes(world, program_batch)
            ops = program_batch._op_batch_list
            trace = []
            for i, op_batch in enumerate(ops):                   # execution loop :145-172
                deps = program_batch._dependencies[i]
                input_tuple = tuple(trace[d] for d in deps)
                x, terminate = self._execute(op_batch._op_id, world, op_batch, input_tuple, i == len(ops) - 1, is_training)
                if isinstance(x, BatchVariableSet) and len(input_tuple) > 0 and op_batch._mask is not None:
                    x = x.gate(input_tuple[0], op_batch._mask)   # questions lacking this op keep their attention (:166-167)
                trace.append(x)
                if terminate:
                    break
            all_results.append(trace[-1] if trace else None)
            all_traces.append(trace)
        return all_results, all_traces, device


def _calibration_passes(self, world, program_batch, device, is_training):
    """The LSTM walks the aligned program forward (batch_base_interpreter.py:92-111) and backward (:113-140); each operator
    leaves its [P, 4] modulations in its own dictionary, keyed by op id, for the execution loop to apply."""
    ops, deps_all = program_batch._op_batch_list, program_batch._dependencies
    last = len(ops) - 1
    trace = []
    for i, op_batch in enumerate(ops):
        deps = deps_all[i]
        input_tuple = tuple(trace[d] for d in deps) if deps else (None,)
        x, _ = self._transform_attention(op_batch._op_id, True, world, op_batch, input_tuple, i == last, is_training)
        if i < last and input_tuple[0] is not None and op_batch._mask is not None:
            x = x.gate(input_tuple[0], op_batch._mask)
        trace.append(x)
    reversed_dependencies = reverse_dependencies(deps_all)
    final = trace[-1]
    if isinstance(final, (tuple, list)):
        first_state = tuple(BatchAttentionState(a._name, device, a._state, set_zeros=True) for a in final)
    else:
        first_state = (BatchAttentionState(final._name, device, final._state, set_zeros=True),)
    trace = [None] * len(ops)
    for i in reversed(range(len(ops))):
        op_batch = ops[i]
        if len(reversed_dependencies[i]) == 1:
            temp = trace[reversed_dependencies[i][0]]
            if isinstance(temp, (tuple, list)):
                input_tuple = (temp[1],) if i == len(ops) - 2 else (temp[0],)
            else:
                input_tuple = (temp,)
        else:
            input_tuple = first_state
        x, _ = self._transform_attention(op_batch._op_id, False, world, op_batch, input_tuple, i == 0, is_training)
        if len(deps_all[i]) > 0 and op_batch._mask is not None and isinstance(x, BatchAttentionState) and i != last:
            x = x.gate(input_tuple[0], op_batch._mask)
        trace[i] = x


BatchInterpreterBase._calibration_passes = _calibration_passes


class BatchGQAInterpreter(BatchInterpreterBase):
    """batch_gqa_interpreter.py:13-86: the operator registry and its dispatch."""

    def __init__(self, name, oracle, ontology, featurizer=None, trainable_module_type=None, feature_dim=1, trainable_gate=False,
                 likelihood_threshold=0, hard_mode=False, attention_transfer_state_dim=0, forward_attention_network=None,
                 backward_attention_network=None, attention_output_network=None, apply_modulation_everywhere=True, cached=False,
                 visual_rule_learner=None, calibrator=None):
        super(BatchGQAInterpreter, self).__init__(name, oracle, featurizer, attention_transfer_state_dim=attention_transfer_state_dim,
                                                  apply_modulation_everywhere=apply_modulation_everywhere, cached=cached,
                                                  visual_rule_learner=visual_rule_learner, calibrator=calibrator)
        self._ontology = ontology
        self._likelihood_threshold = likelihood_threshold
        self._hard_mode = hard_mode
        self._has_modulator = forward_attention_network is not None and backward_attention_network is not None and \
            attention_output_network is not None
        kw = dict(trainable_module_type=trainable_module_type, feature_dim=feature_dim, trainable_gate=trainable_gate,
                  forward_attention_network=forward_attention_network, backward_attention_network=backward_attention_network,
                  attention_output_network=attention_output_network)
        o, t = self._oracle, self._ontology
        self._ops = nn.ModuleDict({
            'select': gqa.GQASelectBatch(o, t, **kw), 'filter': gqa.GQAFilterBatch(o, t, **kw), 'relate': gqa.GQARelateBatch(o, t, **kw),
            'query_attr': gqa.GQAQueryAttrBatch(o, t, **kw), 'choose_attr': gqa.GQAChooseAttrBatch(o, t, **kw),
            'verify_attrs': gqa.GQAVerifyAttrsBatch(o, t, **kw), 'choose_rel': gqa.GQAChooseRelBatch(o, t, **kw),
            'verify_rel': gqa.GQAVerifyRelBatch(o, t, **kw), 'exist': gqa.GQAExistBatch(o, t), 'and': gqa.GQAAndBatch(o, t),
            'or': gqa.GQAOrBatch(o, t), 'all_same': gqa.GQAAllSameBatch(o, t, **kw), 'all_different': gqa.GQAAllDifferentBatch(o, t, **kw),
            'two_same': gqa.GQATwoSameBatch(o, t, **kw), 'two_different': gqa.GQATwoDifferentBatch(o, t, **kw),
            'compare': gqa.GQACompareBatch(o, t, **kw), 'end': gqa.GQAEndBatch(o, t),
        })

    # one-hot position of every operator in the LSTM input (batch_gqa_interpreter.py:67-70)
    _OPS_INDEX = {'all_different': 0, 'all_same': 1, 'and': 2, 'choose_attr': 3, 'choose_rel': 4, 'compare': 5, 'end': 6, 'exist': 7,
                  'filter': 8, 'or': 9, 'query_attr': 10, 'relate': 11, 'select': 12, 'two_different': 13, 'two_same': 14,
                  'verify_attrs': 15, 'verify_rel': 16}

    def _transform_attention(self, op_id, is_forward, world, operator_batch, input_tuple, is_terminal, is_training):   # :80-86
        onehot = np.zeros(len(self._OPS_INDEX), np.float32)
        onehot[self._OPS_INDEX[operator_batch._op_name]] = 1.0
        temp = upload(onehot, world._device)                   # (memoised; an indexed assignment on the device is not graph-capturable)
        temp._host = onehot                                    # lets the operators build their constant feature columns on the host
        x = self._ops[operator_batch._op_name].transform_attention(*((op_id, is_forward, world) + input_tuple + tuple(operator_batch._arguments) +
                                                                     (temp, operator_batch._predicate_question_map)))
        return x, is_terminal

    def _execute(self, op_id, world, operator_batch, input_tuple, is_terminal, is_training):
        op = self._ops[operator_batch._op_name]
        x = op(*((op_id, world) + input_tuple + tuple(operator_batch._arguments) +
                 (not is_training, operator_batch._predicate_question_map, self._likelihood_threshold, self._hard_mode)))
        if is_terminal and not operator_batch._is_terminal:      # append `end` to read out a log-likelihood (:75-76)
            return self._ops['end'](op_id, world, x, not is_training, operator_batch._predicate_question_map), is_terminal
        return x, is_terminal
