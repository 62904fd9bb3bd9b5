"""Host-side helpers of the interpreter (reference: src/nsvqa/nn/interpreter/util.py) and token lowering.

Lowering turns the per-question token strings of an operator batch into three small integer arrays
(table column, negation flag, validity) once, at collate time, so that the timed step launches
kernels instead of doing dictionary look-ups and regular-expression matches per question.
"""

import re

import threading

import numpy as np
import torch

from ._lib import LRUCache, keep_alive, keeping  # noqa: F401
from .fol_types import TokenType

_NEG = re.compile(r"not\((\w|\s)+\)")


def detect_negations(a_list, device=None):          # util.py:68-85
    is_negated = [_NEG.match(a.strip()) is not None for a in a_list]
    any_negated = any(is_negated)
    if any_negated:
        b_list = [a.strip()[4:-1] if n else a.strip() for a, n in zip(a_list, is_negated)]
    else:
        b_list = a_list
    return any_negated, is_negated, b_list


def flatten_list(a_list_list):                      # util.py:52-57
    a_list = [a if a is not None else [None] for a in a_list_list]
    batch_index = [i for i, sublist in enumerate(a_list) for _ in sublist]
    return [item for sublist in a_list for item in sublist], batch_index


def unflatten_list(a_list, batch_index, flags):     # util.py:59-62
    d = {i: [] for i in set(batch_index)}
    for x, y, z in zip(batch_index, a_list, flags):
        if z > 0:
            d.setdefault(x, []).append(y)
    return list(d.values())


def find_max_ind(log_likelihood, pred_q, question_num, likelihood_threshold=0):
    """util.py:64-66 on the host: flag, per predicate, whether it attains its question's maximum probability."""
    lp = log_likelihood.detach().cpu().numpy() if isinstance(log_likelihood, torch.Tensor) else np.asarray(log_likelihood)
    pq = pred_q.cpu().numpy() if isinstance(pred_q, torch.Tensor) else np.asarray(pred_q)
    p = np.exp(lp)
    mx = np.zeros(question_num, p.dtype)
    np.maximum.at(mx, pq, p)
    return ((p == mx[pq]) & (p > likelihood_threshold)).astype(np.int64)


def reverse_dependencies(pred):                     # util.py:153-161
    length = len(pred)
    return [sorted(i for i in range(length) if j in pred[i]) for j in range(length)]


def is_valid_token(val):                            # batch_base_ops.py:315
    return val is not None and val.strip() not in ('', '_')


class Lowered(object):
    """Integer form of a token list for one table: column (-1 = no-op token), negation, validity."""

    __slots__ = ("cols", "neg", "valid", "any_neg", "any_valid", "all_valid", "_dev")

    def __init__(self, cols, neg, valid):
        self.cols = np.asarray(cols, np.int32)
        self.neg = np.asarray(neg, np.uint8)
        self.valid = np.asarray(valid, np.uint8)
        self.any_neg = bool(self.neg.any())
        self.any_valid = bool(self.valid.any())
        self.all_valid = bool(self.valid.all())
        self._dev = {}

    def on(self, device):
        key = str(device)
        if key not in self._dev:
            # ONE upload for the three arrays (columns | negations | validity in one byte buffer, three views of its device copy): a fresh
            # 256-question batch lowers ~13 token lists, and every separate small copy is ~15 us of host time
            n = len(self.cols)
            if n == 0 or torch.device(device).type != "cuda":
                self._dev[key] = (upload(self.cols, device), upload(self.neg, device), upload(self.valid, device))
            else:
                buf = upload(np.concatenate([self.cols.view(np.uint8), self.neg, self.valid]), device)
                self._dev[key] = (buf[:4 * n].view(torch.int32), buf[4 * n:5 * n], buf[5 * n:6 * n], buf)
        return keep_alive(self._dev[key])[:3]    # (a hit must reach a capturing graph's keep-alive list too: the Lowered object may be evicted)


_upload_cache = LRUCache(1024)


class _PinnedRing(object):
    """Host staging for small uploads: slices of ONE pinned buffer per device, copied with non_blocking=True.  A pageable host-to-device copy
    makes the host wait for the stream (~20 us each, ~60 per fresh 256-question batch); from pinned memory the copy is queued behind the
    stream's work and the host moves on.  A slice is reused only after the ring has wrapped, and then waits for the events of the copies that
    read it (one event per copy, recorded on the stream the copy was queued on).  `copy_to` stages and queues the copy under one
    lock: collate / prefetch threads upload too."""

    def __init__(self, device=None, nbytes=8 << 20):
        self.buf = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.view = self.buf.numpy()
        self.pos = 0
        self.device = device
        self.lock = threading.Lock()
        self.pending = []                                    # [start, end, event or None, stream] of the copies queued from the ring, oldest first
        self.unmarked = 0                                    # bytes queued since the last recorded event

    def _mark(self):
        """One event per stream covers every copy queued on that stream before it (an event per 100-byte upload would cost more host time
        than the copy): the entries still without an event get one recorded NOW on their stream."""
        events = {}
        for e in self.pending:
            if e[2] is None:
                ev = events.get(e[3])
                if ev is None:
                    ev = events[e[3]] = torch.cuda.Event()
                    ev.record(e[3])
                e[2] = ev
        self.unmarked = 0

    def stage(self, a):
        n = a.nbytes
        start = (self.pos + 15) & ~15
        if start + n > self.buf.numel():
            start = 0
        # a slice is rewritten only after the copies that read it have run: wait for THEIR events (a device-wide synchronize from a collate
        # thread would invalidate a graph capture running on the main thread, and waits for every other stream's work as well)
        hit = [i for i, e in enumerate(self.pending) if e[0] < start + n and start < e[1]]
        if hit:
            self._mark()
            for i in hit:
                self.pending[i][2].synchronize()
            gone = set(hit)
            self.pending = [e for i, e in enumerate(self.pending) if i not in gone]
        self.view[start:start + n] = a.reshape(-1).view(np.uint8)
        self.pos = start + n
        return self.buf[start:start + n], start

    def copy_to(self, dst_bytes, a):
        """dst_bytes (a flat uint8 device tensor of a.nbytes) <- a, through the ring, asynchronously."""
        with self.lock:
            src, start = self.stage(a)
            dst_bytes.copy_(src, non_blocking=True)
            st = torch.cuda.current_stream(dst_bytes.device)
            last = self.pending[-1] if self.pending else None
            if last is not None and last[2] is None and last[3] == st and last[1] <= start:
                last[1] = start + a.nbytes                   # (a run of copies on one stream is one entry: the list stays a few dozen long)
            else:
                self.pending.append([start, start + a.nbytes, None, st])
            self.unmarked += a.nbytes
            if self.unmarked >= (256 << 10):
                self._mark()


_rings = {}
_rings_lock = threading.Lock()


def ring_for(device):
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ring = _rings.get(key)
    if ring is None:
        with _rings_lock:
            ring = _rings.get(key)
            if ring is None:
                ring = _rings[key] = _PinnedRing(key)
    return ring


_TORCH_DTYPE = {np.dtype(k).str: v for k, v in ((np.float32, torch.float32), (np.int32, torch.int32), (np.int64, torch.int64), (np.uint8, torch.uint8),
                                                (np.float64, torch.float64), (np.int16, torch.int16), (np.bool_, torch.bool))}


def upload(array, device):
    """Small host index arrays (segment offsets, keep lists, predicate -> question maps) as device tensors, uploaded once per
    content: the same arrays recur on every forward of a batch.  New content goes through a pinned staging ring (no host wait per copy)."""
    a = np.ascontiguousarray(array)
    key = (str(device), a.dtype.str, a.shape, a.tobytes())
    hit = _upload_cache.get(key)
    if hit is None:
        dev = torch.device(device)
        if dev.type == "cuda" and a.ndim >= 1 and 0 < a.nbytes <= (1 << 20) and not torch.cuda.is_current_stream_capturing():
            hit = torch.empty(a.shape, dtype=_TORCH_DTYPE.get(a.dtype.str) or torch.as_tensor(a[:0]).dtype, device=dev)
            ring_for(dev).copy_to(hit.view(torch.uint8).reshape(-1), a)
        else:
            hit = torch.as_tensor(a).to(device)
        _upload_cache[key] = hit
    return keep_alive(hit)


_const_cache = LRUCache(64)


def constant(shape, value, device):
    """A READ-ONLY fp32 tensor of one value (the quantifier of a fresh variable set, the all-ones attention of a select), shared by every
    caller: torch.full / torch.zeros would launch a fill kernel - and add a graph node - per use.  Nothing on the path writes into its
    inputs in place (outputs are always new tensors), and tests that replay a batch would see a corrupted constant."""
    key = (str(device), tuple(shape), float(value))
    hit = _const_cache.get(key)
    if hit is None:
        hit = _const_cache[key] = torch.full(tuple(shape), float(value), dtype=torch.float32, device=device)
    return keep_alive(hit)


def lower_tokens(tokens, ontology, token_type):
    """Resolve tokens against the ontology exactly as the reference's oracle does
    (classifier_oracle.py:49-56 for attributes: column = arg_to_idx-1 of the full table;
    :89-96 for relations: column in the 333-wide table via _relation_reveresed_index).
    Unknown tokens raise KeyError like the reference's itemgetter."""
    # memoised by content: the per-question category expansions of query / same / different operators are thousands of tokens
    # (resolving 6656 strings costs 7 ms of host time per call, more than the whole GPU step), and a recurring list maps to the
    # same Lowered object, whose device copies are then uploaded once
    key = None
    if len(tokens) >= 1:
        cache = ontology.__dict__.setdefault("_lower_cache", LRUCache(256))       # lives and dies with the ontology it was resolved against
        key = (int(token_type), tuple(tokens))
        hit = cache.get(key)
        if hit is not None:
            return hit
    # per-token memo (token string -> column, negation, validity packed into one integer): a fresh batch names the same few hundred concepts
    # as the one before it, and resolving a token from scratch is a strip, a regex and two dictionary reads (4.5 k tokens per 256-question
    # batch: 3.6 ms).  A list whose tokens have all been seen is one pass of dict.get inside numpy.fromiter (a Python loop with three
    # appends per token was 0.9 ms of a fresh batch's 1.7 ms of collate).
    memo = ontology.__dict__.setdefault("_token_memo", {}).setdefault(int(token_type), {})
    try:
        codes = np.fromiter(map(memo.get, tokens), np.int64, len(tokens))
    except TypeError:                                        # a token not met before (dict.get -> None)
        arg_to_idx = ontology._vocabulary['arg_to_idx']
        if len(memo) > 65536:
            memo.clear()
        for t in tokens:
            if t in memo:
                continue
            if not is_valid_token(t):
                memo[t] = 0                                  # column -1, no negation, not valid
                continue
            s = t.strip()
            n = _NEG.match(s) is not None
            if n:
                s = s[4:-1]
            idx = arg_to_idx[s.strip()] - 1                  # (an unknown token raises KeyError, as the reference's itemgetter does: not memoised)
            if token_type == TokenType.RELATION:
                idx = ontology._relation_reveresed_index[idx]
            memo[t] = ((int(idx) + 1) << 2) | (int(n) << 1) | 1
        codes = np.fromiter(map(memo.get, tokens), np.int64, len(tokens))
    cols, neg, valid = ((codes >> 2) - 1).astype(np.int32), ((codes >> 1) & 1).astype(np.uint8), (codes & 1).astype(np.uint8)
    low = Lowered(cols, neg, valid)
    if key is not None:
        cache[key] = low
    return low


class TokenList(list):
    """A plain list of tokens that remembers its lowered form (filled by OperatorBatch.lower)."""

    lowered = None
    lowered_type = None


def get_lowered(tokens, ontology, token_type):
    low = getattr(tokens, "lowered", None)
    if low is not None and getattr(tokens, "lowered_type", None) == token_type:
        return low
    return lower_tokens(tokens, ontology, token_type)


def segments_of(image_map):
    """Runs of equal consecutive values (torch.unique_consecutive, classifier_oracle.py:23): -> offsets [S+1]."""
    m = np.asarray(image_map)
    if len(m) == 0:
        return np.zeros(1, np.int32)
    change = np.nonzero(m[1:] != m[:-1])[0] + 1
    return np.concatenate([[0], change, [len(m)]]).astype(np.int32)
