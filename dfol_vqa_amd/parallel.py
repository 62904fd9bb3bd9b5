"""Data parallelism: one process per GPU, questions sharded across ranks.

Reference: src/nsvqa/nn/interpreter/data_parallel.py (ProgramDataParallel: one Python thread per GPU, parameters
broadcast each forward, gradients reduced to device 0, log-probabilities gathered).  Questions and their scenes are
independent, so inference needs no exchange at all except gathering the answers; training needs exactly one
all-reduce (sum) of the flat fp32 gradient bucket per step (RCCL over xGMI on MI355X: backend "nccl"; "gloo" on CPU).
"""

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(costs, world_size):
    """Contiguous shards of a question list, balanced by cost (use sum of N^2 per question for ragged scenes).
    Contiguity keeps the reference's question order when results are concatenated (data_parallel.py:59-80)."""
    costs = np.asarray(costs, np.float64)
    n = len(costs)
    if n == 0:
        return [(0, 0)] * world_size
    cum = np.concatenate([[0.0], np.cumsum(costs)])
    total = cum[-1]
    bounds, start = [], 0
    for r in range(world_size):
        if r == world_size - 1:
            end = n
        else:
            target = total * (r + 1) / world_size
            end = int(np.searchsorted(cum, target, side="left"))          # first prefix whose cost reaches the target
            if end > start + 1 and abs(cum[end - 1] - target) < abs(cum[min(end, n)] - target):
                end -= 1                                                   # the shorter prefix is closer
            end = min(max(end, start), n)
        bounds.append((start, end))
        start = end
    return bounds


def shard_questions(questions, rank, world_size, costs=None):
    if costs is None:
        costs = [float(q["scene"]["n"]) ** 2 if isinstance(q, dict) and "scene" in q else 1.0 for q in questions]
    s, e = shard_bounds(costs, world_size)[rank]
    return questions[s:e]


def gather_results(result, group=None):
    """All ranks receive the concatenation (in rank order) of every rank's result dict
    (the distributed counterpart of data_parallel.gather_results :15-50)."""
    world = dist.get_world_size(group)
    payload = {"answer": result["answer"], "options": result["options"], "answer_log_probability": result["answer_log_probability"],
               "log_probability": result["log_probability"].detach().cpu().numpy(), "type": int(result["type"]),
               "variable_sets_num": result["variable_sets_num"], "cumulative_loss": result["cumulative_loss"]}
    parts = [None] * world
    dist.all_gather_object(parts, payload, group=group)
    query = parts[0]["type"] == 1
    return {"answer": [a for p in parts for a in p["answer"]],
            "log_probability": torch.from_numpy(np.concatenate([p["log_probability"] for p in parts])),
            "options": [o for p in parts for o in p["options"]] if query else parts[0]["options"],
            "variable_set": None, "type": result["type"],
            "cumulative_loss": sum(p["cumulative_loss"] for p in parts), "variable_sets_num": sum(p["variable_sets_num"] for p in parts),
            "answer_log_probability": [a for p in parts for a in p["answer_log_probability"]]}


def broadcast_parameters(model, src=0, group=None):
    """Every rank takes rank `src`'s parameters and buffers (ONE broadcast of a flat fp32 bucket).  The reference's
    ProgramDataParallel replicates device 0's module on every forward (data_parallel.py:54-83); with one process per GPU the
    replicas must be made equal once - after build_model / load_state_dict - and the identical all-reduced gradient plus the
    identical clip + Adam step keep them equal from then on.  Returns the number of bytes exchanged."""
    tensors = [p.data for p in model.parameters()] + [b.data for b in model.buffers()]
    tensors = [t for t in tensors if t.numel() > 0]
    if not tensors:
        return 0
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    offset = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[offset:offset + n].view_as(t))      # copy_ bumps the version counter: packed weight images are rebuilt
            offset += n
    return flat.numel() * 4


def parameters_digest(model):
    """A float64 checksum pair of all parameters (sum, sum of squares) for cross-rank equality checks."""
    ps = [p.detach().reshape(-1).to(torch.float64) for p in model.parameters() if p.numel() > 0]
    flat = torch.cat(ps) if ps else torch.zeros(1, dtype=torch.float64)
    return torch.stack([flat.sum(), (flat * flat).sum()])


class GradBucket(object):
    """The flat fp32 gradient bucket of SURVEY.md 8(e), persistent: every trainable parameter's `.grad` is a view into ONE
    contiguous buffer, so the step's all-reduce needs no gather / scatter copies (autograd accumulates into the views in place).
    Use `bucket.zero_()` instead of `optimizer.zero_grad()` (which would drop the views)."""

    def __init__(self, parameters):
        self.params = [p for p in parameters if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._segments = None            # overlap mode: [(first element, end element, first parameter, end parameter)]
        self._attach()

    def _attach(self):
        offset = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[offset:offset + n].view_as(p)
            offset += n

    def zero_(self):
        if any(p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or
               p.grad.data_ptr() >= self.flat.data_ptr() + 4 * max(1, self.flat.numel()) for p in self.params):
            self._attach()                                   # someone called zero_grad(set_to_none=True)
        self.flat.zero_()
        if self._segments is not None:
            self._pending = [b - a for _, _, a, b in self._segments]
            self._works = [None] * len(self._segments)
            self._next = len(self._segments) - 1
            self._finished = False

    def nbytes(self):
        return self.flat.numel() * 4

    def enable_overlap(self, group=None, segments=3):
        """Overlap the collective with the backward pass: the bucket is cut into `segments` contiguous ranges at parameter boundaries and
        the all-reduce of a range starts (async, from a side stream that waits for the backward's stream) the moment autograd has accumulated the last gradient of
        the range - post-accumulate-grad hooks - while the rest of the backward still runs.  Autograd reaches the parameters roughly in
        reverse order, so the ranges are issued last to first - strictly in that order on every rank (a rank whose batch finished an earlier
        range first holds it back), because collectives must match across ranks.  `allreduce()` then starts whatever has not started
        (parameters that got no gradient this step) and waits for all of it.  The summed values per element are those of the single
        all-reduce."""
        if not self.params or segments <= 1:
            return self
        total, bounds, acc = self.flat.numel(), [0], 0
        for i, p in enumerate(self.params):
            acc += p.numel()
            if len(bounds) < segments and acc >= total * len(bounds) / segments and i + 1 < len(self.params):
                bounds.append(i + 1)
        bounds.append(len(self.params))
        starts = [0]
        for p in self.params:
            starts.append(starts[-1] + p.numel())
        self._segments = [(starts[a], starts[b], a, b) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
        self._group = group
        self._pending = [b - a for _, _, a, b in self._segments]
        self._works = [None] * len(self._segments)
        self._next = len(self._segments) - 1
        self._finished = False
        # Streams, explicitly: a range's all-reduce is issued from a dedicated side stream that first waits for the stream the hook runs
        # on (the backward's: autograd runs a node - and its hooks - on the stream of the node's forward), so the collective starts after
        # the gradients of the range are complete, whatever stream the backward used; allreduce() makes the caller's stream (the
        # optimizer's) wait for the side stream after the works have completed on it.
        self._side = torch.cuda.Stream(device=self.flat.device) if self.flat.is_cuda else None
        seg_of = {}
        for k, (_, _, a, b) in enumerate(self._segments):
            for i in range(a, b):
                seg_of[i] = k
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(lambda _p, k=seg_of[i]: self._ready(k))
        return self

    # Collectives must be issued in the same order on every rank, whatever order the ranks' (data-dependent) graphs finish their ranges
    # in: ranges are launched strictly last to first, a finished range waiting for the ones after it.
    def _launch_ready(self, force=False):
        while self._next >= 0 and (force or self._pending[self._next] <= 0):
            a, b, _, _ = self._segments[self._next]
            if b > a:
                if self._side is not None:
                    self._side.wait_stream(torch.cuda.current_stream(self.flat.device))
                    with torch.cuda.stream(self._side):
                        self._works[self._next] = dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self._group, async_op=True)
                else:
                    self._works[self._next] = dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self._group, async_op=True)
            self._next -= 1

    def _ready(self, k):
        # ONE backward per zero_(): a second backward (gradient accumulation, retain_graph) would add local gradients into a range that
        # has already been summed across ranks, and the replicas would drift apart silently
        if getattr(self, "_finished", False) or self._pending[k] <= 0 or self._next < k:
            raise RuntimeError("GradBucket overlap mode: a gradient arrived for a range whose all-reduce has already been issued - "
                               "call bucket.zero_() before every backward (one backward per step; no gradient accumulation in overlap mode)")
        self._pending[k] -= 1
        if self._pending[k] == 0:
            self._launch_ready()

    def allreduce(self, group=None):
        """All-reduce (sum) of the whole bucket - RCCL over xGMI with backend "nccl", gloo on CPU: ONE collective, or, after
        enable_overlap(), the completion of the per-range collectives the backward already started."""
        if self._segments is not None:
            self._launch_ready(force=True)
            if self._side is not None:
                with torch.cuda.stream(self._side):
                    for w in self._works:
                        if w is not None:
                            w.wait()
                torch.cuda.current_stream(self.flat.device).wait_stream(self._side)
            else:
                for w in self._works:
                    if w is not None:
                        w.wait()
            self._finished = True
            return self.nbytes()
        if self.flat.numel():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        return self.nbytes()


def allreduce_gradients(parameters, group=None):
    """One all-reduce (sum) of ONE flat fp32 bucket holding every trainable gradient (9.2 MB in the oracle phase,
    0.59 MB in the calibrator phase).  With the loss divided by the GLOBAL batch size on every rank
    (trainer.py:433-436) the summed gradient equals the single-process gradient; every rank then applies the same
    clip + Adam step and parameters stay identical without any broadcast."""
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return 0
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1).to(torch.float32) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    offset = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[offset:offset + n].view_as(p.grad))
        offset += n
    return flat.numel() * 4
