"""Host side of the native executor (include/dfol_vqa.h: dfol_run_program): the model's device pointers as the C structs, one upload of a
plan's side arrays, one C call per ProgramBatch, one read-back of the results.

The interpreter (interpreter.BatchInterpreterBase._run_batches) takes this route for every ProgramBatch that has a plan
(native_plan.build_plan: inference, needed-columns oracle, no calibration); everything else runs the Python operator loop.  The result
dict has the reference's keys (batch_gqa_ops.py: answer, log_probability, options, variable_set, type, cumulative_loss,
variable_sets_num, answer_log_probability), with `variable_set` None (what data_parallel.gather_results keeps of it anyway).
"""

import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import native_plan as NP

_p, _i32, _i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
DENSE_F32, DENSE_F16X2, DENSE_BF16X3, DENSE_BF16 = 0, 1, 2, 3
PAIR_PLAIN, PAIR_PACKED, PAIR_BF16X3, PAIR_F16X2 = 0, 1, 2, 3


class DenseLayer(ctypes.Structure):
    _fields_ = [("kind", _i32), ("act", _i32), ("N", _i32), ("K", _i32), ("weight", _p), ("ldw", _i64), ("packed", _p), ("bias", _p)]


class ProgramModel(ctypes.Structure):
    _fields_ = [("n_featurizer", _i32), ("n_attribute", _i32), ("featurizer", ctypes.POINTER(DenseLayer)), ("attribute", ctypes.POINTER(DenseLayer)),
                ("uv", DenseLayer), ("pair_kind", _i32), ("hid1", _i32), ("hid2", _i32), ("w2_rows", _i32), ("wg", _p), ("w2", _p), ("ld_w2", _i64),
                ("b2", _p), ("emb_w", _p), ("ld_e", _i64), ("emb_b", _p), ("emb_in", _i32), ("D", _i32),
                ("lstm_wih_t", _p * 2), ("lstm_whh_t", _p * 2), ("lstm_ld_wih", _i64 * 2), ("lstm_ld_whh", _i64 * 2), ("lstm_bih", _p * 2), ("lstm_bhh", _p * 2),
                ("lstm_kx", _i32), ("lstm_h", _i32), ("att_out_w", _p), ("ld_att_out", _i64), ("att_out_b", _p), ("att_out_n", _i32)]


class ProgramScene(ctypes.Structure):
    _fields_ = [("features", _p), ("ld_features", _i64), ("raw_cols", _i32), ("O", _i32), ("NS", _i32), ("max_n", _i32), ("n_obj", _i64),
                ("img_n_obj", _i64), ("obj_off", _i64)]


_SUSPENDED = [0]


def enabled():
    # (per-entry-point HIP-event timing, _lib.enable_kernel_timing, brackets the individual calls of the Python loop: same kernels)
    return os.environ.get("DFOL_NATIVE", "1") != "0" and not _SUSPENDED[0] and _lib._timed is None


class suspended(object):
    """with suspended(): forwards inside run the Python operator loop (the warm-up of a graph capture: the capture records that loop's
    launches, so its host-side caches - uploaded index arrays, lowered tokens - must be filled by the same loop)."""

    def __enter__(self):
        _SUSPENDED[0] += 1

    def __exit__(self, *exc):
        _SUSPENDED[0] -= 1
        return False


def _layers(seq):
    """(Linear, activation code) pairs of an nn.Sequential of (Dropout, Linear, activation) triples, as visual_oracle._run_layers walks it."""
    mods, out, i = list(seq), [], 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Dropout):
            if m.training and m.p > 0:               # a model left in train(): the Python loop raises there (visual_oracle._run_layers), the
                return None                          # reference would drop activations - the executor must not silently run without it
            i += 1
            continue
        if not isinstance(m, nn.Linear):
            return None
        act, step = _lib.ACT_NONE, 1
        if i + 1 < len(mods):
            nxt = mods[i + 1]
            if isinstance(nxt, nn.ELU):
                act, step = _lib.ACT_ELU, 2
            elif isinstance(nxt, nn.Sigmoid):
                act, step = _lib.ACT_SIGMOID, 2
            elif isinstance(nxt, nn.LogSigmoid):
                act, step = _lib.ACT_LOGSIGMOID, 2
        out.append((m, act))
        i += step
    return out


def calibrator(model):
    """(forward LSTM cell, backward LSTM cell, output Linear) of an interpreter whose attention calibrator the executor can run - the modules
    every operator shares (batch_base_ops.py:251-254) - or None: no calibrator, or shapes the one-launch kernels do not take."""
    from .visual_oracle import CalibrationLSTMCell
    if not getattr(model, "_has_modulator", False):
        return None
    flt = model._ops['filter']._filter
    fwd, bwd, out = (getattr(flt, n, None) for n in ("_forward_attention_network", "_backward_attention_network", "_attention_output_network"))
    if not (isinstance(fwd, CalibrationLSTMCell) and isinstance(bwd, CalibrationLSTMCell) and isinstance(out, nn.Sequential) and len(out) == 2
            and isinstance(out[0], nn.Linear) and isinstance(out[1], nn.Sigmoid)):
        return None
    S, KX = fwd.hidden_size, fwd.input_size
    if (bwd.hidden_size, bwd.input_size) != (S, KX) or out[0].in_features != 2 * S or out[0].out_features != 4 or 4 * (KX + 5 * S) * 4 > 65536:
        return None
    if not all(p.is_cuda and p.dtype == torch.float32 for m in (fwd, bwd, out) for p in m.parameters()):
        return None
    return fwd, bwd, out[0]


def model_spec(model, calibrate=False):
    """native_plan.ModelSpec of an interpreter, or None when its modules are not the shapes the executor drives.  calibrate: the forward runs
    the attention-calibration passes (activate_attention_transfer, modulator_switch on)."""
    from .interpreter import BatchGQABoxFeaturizer
    feat, oracle = model._featurizer, model._oracle
    if not isinstance(feat, BatchGQABoxFeaturizer) or getattr(feat._featurizer_network, "_network", None) is None:
        return None
    if not (model._cached and getattr(oracle, "supports_needed_columns", lambda: False)()):
        return None
    fl, al = _layers(feat._featurizer_network._network), _layers(oracle._attribute_network._network)
    if not fl or not al:
        return None
    lin1, lin2 = [m for m in oracle._relation_network._network if isinstance(m, nn.Linear)][:2]
    # bf16 tile storage (config key relation_tile_dtype: bf16): the pair kernels that write bf16 tiles are the packed ones over a second layer of
    # more than 256 rows (visual_oracle.prefetch_relations' rule; the weight's shape decides, not its values)
    tile_bf16 = getattr(oracle, "_tile_dtype", torch.float32) == torch.bfloat16 and 256 < lin2.out_features <= 320 and lin1.out_features <= 256 and \
        lin1.out_features % 16 == 0 and os.environ.get("DFOL_PAIR_PACKED", "1") != "0"
    calib = None
    if calibrate:
        nets = calibrator(model)
        if nets is None:
            return None
        calib = dict(state_dim=int(nets[0].hidden_size), lstm_in=int(nets[0].input_size), ops_index=dict(model._OPS_INDEX))
    return NP.ModelSpec([l.out_features for l, _ in fl], [l.out_features for l, _ in al], lin1.out_features, fl[-1][0].out_features + 4,
                        oracle._normalize, model._likelihood_threshold, oracle._ontology._relation_index, tile_bf16=tile_bf16, calib=calib)


class NativeModel(object):
    """The C view of an interpreter's neural modules for ONE weight version and dense-math mode.  Holds every tensor whose address it
    hands out (raw weights, packed images) for as long as it lives."""

    def __init__(self, model):
        feat, oracle = model._featurizer, model._oracle
        self._hold = []
        math = _lib._dense_math()
        kind, pieces = {"f16x2": (DENSE_F16X2, 2), "bf16x3": (DENSE_BF16X3, 3), "bf16": (DENSE_BF16, 1), "f32": (DENSE_F32, 0)}[math]

        def dense(weight, bias, act):
            w = weight.detach()
            N, K = w.shape
            d = DenseLayer()
            d.kind, d.act, d.N, d.K = DENSE_F32, int(act), int(N), int(K)
            d.weight, d.ldw = w.data_ptr(), w.stride(0)
            d.packed = None
            d.bias = None if bias is None else bias.detach().data_ptr()
            self._hold += [w, bias]
            if kind != DENSE_F32 and N * K >= _lib.SPLIT_MIN_WEIGHT and w.stride(1) == 1:
                img = _lib.linear_pack_w_split(weight, False, pieces)
                d.kind, d.packed = kind, img.data_ptr()
                self._hold.append(img)
            return d

        fl, al = _layers(feat._featurizer_network._network), _layers(oracle._attribute_network._network)
        self._fl = (DenseLayer * len(fl))(*[dense(l.weight, l.bias, a) for l, a in fl])
        self._al = (DenseLayer * len(al))(*[dense(l.weight, l.bias, a) for l, a in al])
        wuv, buv, wg, hid1, D = oracle._split_first_layer()
        w2p, b2, hid2, packed = oracle._padded_second_layer()
        emb = oracle._embedding_network.linear
        m = ProgramModel()
        m.n_featurizer, m.n_attribute = len(fl), len(al)
        m.featurizer, m.attribute = self._fl, self._al
        m.uv = dense(wuv, buv, _lib.ACT_NONE)
        if isinstance(packed, tuple):
            m.pair_kind, w2 = (PAIR_F16X2 if packed[0] == "f16x2" else PAIR_BF16X3), packed[1]
        elif packed is not None:
            m.pair_kind, w2 = PAIR_PACKED, packed
        else:
            m.pair_kind, w2 = PAIR_PLAIN, w2p
        m.hid1, m.hid2, m.w2_rows = int(hid1), int(hid2), int(w2p.shape[0])
        m.wg, m.w2, m.ld_w2, m.b2 = wg.data_ptr(), w2.data_ptr(), w2p.stride(0), b2.data_ptr()
        ew = emb.weight.detach()
        m.emb_w, m.ld_e = ew.data_ptr(), ew.stride(0)
        m.emb_b = None if emb.bias is None else emb.bias.detach().data_ptr()
        m.emb_in, m.D = int(ew.shape[1]), int(D)
        self._hold += [wuv, buv, wg, w2p, b2, w2, ew, emb.bias]
        nets = calibrator(model)
        if nets is not None:
            for k, cell in enumerate(nets[:2]):
                _, wih_t, whh_t = cell._transposed()
                m.lstm_wih_t[k], m.lstm_whh_t[k] = wih_t.data_ptr(), whh_t.data_ptr()
                m.lstm_ld_wih[k], m.lstm_ld_whh[k] = wih_t.stride(0), whh_t.stride(0)
                m.lstm_bih[k] = None if cell.bias_ih is None else cell.bias_ih.detach().data_ptr()
                m.lstm_bhh[k] = None if cell.bias_hh is None else cell.bias_hh.detach().data_ptr()
                self._hold += [wih_t, whh_t, cell.bias_ih, cell.bias_hh]
            m.lstm_kx, m.lstm_h = int(nets[0].input_size), int(nets[0].hidden_size)
            ow = nets[2].weight.detach()
            ow = ow if ow.stride(1) == 1 else ow.contiguous()
            m.att_out_w, m.ld_att_out, m.att_out_n = ow.data_ptr(), ow.stride(0), int(ow.shape[0])
            m.att_out_b = None if nets[2].bias is None else nets[2].bias.detach().data_ptr()
            self._hold += [ow, nets[2].bias]
        self.struct = m
        self.D = int(D)

    @staticmethod
    def version_key(model):
        params = list(model._featurizer.parameters()) + list(model._oracle.parameters())
        nets = calibrator(model)
        if nets is not None:
            params += [p for m in nets for p in m.parameters()]
        return (tuple((p.data_ptr(), p._version) for p in params), _lib._dense_math(), _lib.pair_math(), str(params[0].device) if params else "")


def native_model(model):
    key = NativeModel.version_key(model)
    hit = model.__dict__.get("_native_model")
    if hit is None or hit[0] != key:
        hit = model.__dict__["_native_model"] = (key, NativeModel(model))
    return hit[1]


# ---- pinned read-back buffers: a small free list (pinned allocations cost ~100 us; a forward needs one per ProgramBatch) -----------------------
_PINNED = {}


def _pinned(nbytes):
    size = 256
    while size < nbytes:
        size *= 2
    free = _PINNED.setdefault(size, [])
    return (free.pop() if free else torch.empty(size, dtype=torch.uint8).pin_memory()), size


def _release(buf, size):
    free = _PINNED.setdefault(size, [])
    if len(free) < 64:
        free.append(buf)


def plan_for(model, program_batch, spec):
    """The ProgramBatch's plan (built at collate time by a collater that was given the spec, or here on first use), or None."""
    plan = getattr(program_batch, "_native_plan", False)
    if plan is False or (plan is not None and plan.key != spec.key()):
        plan = NP.build_plan(program_batch, model._ontology, spec)
        program_batch._native_plan = plan
    return plan


def blob_on(plan, device):
    """The plan's side arrays on the device: one copy through the pinned staging ring, kept with the plan (a batch that runs again pays nothing)."""
    cache = plan.__dict__.setdefault("_blob_dev", {})
    key = str(device)
    hit = cache.get(key)
    if hit is None:
        from . import host_util
        hit = torch.empty(plan.blob.nbytes, dtype=torch.uint8, device=device)
        if plan.blob.nbytes <= (1 << 20) and not torch.cuda.is_current_stream_capturing():
            host_util.ring_for(device).copy_to(hit, plan.blob)
        else:
            hit.copy_(torch.from_numpy(plan.blob))
        cache[key] = hit
    return hit


def run(model, program_batch, plan, queue, give_answer=True):
    """Enqueue the batch; -> the result dict, its answers filled by a closure appended to `queue` (the interpreter's deferred read-back)."""
    feats = program_batch._object_features
    device = feats.device
    if feats.dtype != torch.float32 or feats.stride(1) != 1 or feats.shape[0] != plan.scene["O"]:
        raise _lib.DfolError("native executor: object features must be fp32 rows, one per object of the batch")
    nm = native_model(model)
    if feats.shape[1] - 6 != nm.struct.featurizer[0].K:
        raise _lib.DfolError("native executor: feature width %d does not match the featurizer (%d + 6)" % (feats.shape[1], nm.struct.featurizer[0].K))
    blob = blob_on(plan, device)
    # (sizes in coarse buckets: every fresh batch's plan has its own arena size, and a request that fits no cached block of the caching
    # allocator costs a hipMalloc - ~0.4 ms on the launching thread, more than the C call itself)
    gran = (1 << 25) if plan.ws_bytes >= (1 << 25) else (1 << 20)
    ws = torch.empty((plan.ws_bytes + gran - 1) // gran * gran, dtype=torch.uint8, device=device)
    sc = ProgramScene()
    sc.features, sc.ld_features, sc.raw_cols, sc.O = feats.data_ptr(), feats.stride(0), feats.shape[1], plan.scene["O"]
    sc.NS, sc.max_n = plan.scene["NS"], plan.scene["max_n"]
    sc.n_obj, sc.img_n_obj, sc.obj_off = plan.scene["n_obj"], plan.scene["img_n_obj"], plan.scene["obj_off"]
    instrs = plan.instrs
    _lib.call("dfol_run_program", ctypes.byref(nm.struct), ctypes.byref(sc), instrs.ctypes.data, instrs.shape[0], blob.data_ptr(), ws.data_ptr(),
              _lib._stream())
    r = plan.result
    lp = ws[r["lp"]:r["lp"] + 4 * r["count"]].view(torch.float32)
    answer, alp = [], []
    if give_answer:
        host, size = _pinned(plan.out_bytes)
        host[:plan.out_bytes].copy_(ws[:plan.out_bytes], non_blocking=True)
        done = torch.cuda.Event()
        done.record()

        def fill():
            done.synchronize()
            a, l = NP.decode(plan, host.numpy()[:plan.out_bytes], True)
            answer[:] = a
            alp[:] = l
            _release(host, size)
        queue.append(fill)
    _lib.note("native_program")
    return {'answer': answer, 'log_probability': lp, 'options': r["options"], 'variable_set': None, 'type': r["type"], 'cumulative_loss': 0,
            'variable_sets_num': r["num"], 'answer_log_probability': alp, '_workspace': ws}
