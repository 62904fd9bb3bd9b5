"""Lowering of a collated ProgramBatch to the instruction table of the native executor (include/dfol_vqa.h: dfol_run_program).

The reference dispatches one Python call per operator of a batch (batch_base_interpreter.py:145-172 -> batch_gqa_interpreter.py:72-78 ->
batch_gqa_ops.py); this module performs that walk ONCE, symbolically, on the host - in the collate worker, where the reference already
builds its ProgramBatches (data_pipeline.py:893-898) - and writes down, per launch the operator classes of gqa_ops.py / logic_ops.py /
visual_oracle.py would issue, one fixed-width instruction: opcode + operands (byte offsets into a blob of small side arrays and into a
workspace arena).  Everything a launch needs that the Python operators work out per call - concept columns, negation / validity flags,
predicate -> question maps, option clusters, quantifiers (always host-known: they start as EXISTS constants and are only ever gated by
host flags), gate flags, pair-kernel requests, names, `variable_sets_num` - is decided here.  The executor then issues the same
entry points with the same arguments, so its results equal the Python loop's bit for bit (tests/test_native_gpu.py).

Scope: inference (is_training False, no gradients), the needed-columns oracle with fp32 tiles, no attention calibration, soft
quantifiers (hard_mode False), no trace.  Anything else - and the rare shapes listed at `Unsupported` - returns None and the interpreter
runs its Python loop.
"""

import math

import numpy as np

from .fol_types import Quantifier, QuestionType, TokenType
from .host_util import flatten_list, get_lowered, lower_tokens, segments_of, unflatten_list

# opcodes of include/dfol_vqa.h
(OP_DENSE, OP_BOX_POSITIONS, OP_FILL, OP_PAIR_LL, OP_ATTR_LL, OP_OPTION_NORMALIZE, OP_FILTER, OP_RELATE_ONE, OP_RELATE, OP_QUANTIFY, OP_GATE,
 OP_LOGIC, OP_SEGMENT_SUM_ROWS, OP_SEGMENT_OR, OP_IMPLICATION, OP_COMPARE, OP_FIND_MAX_IND, OP_GATHER_TILES, OP_CALIB_FEATURES, OP_LSTM_CELL,
 OP_SELECT_ROWS, OP_ATT_MODULATIONS, OP_MODULATE, OP_CALIB_WALK) = range(24)
WALK_FILL, WALK_SELECT, WALK_ADD, WALK_LSTM, WALK_ATT_MODULATIONS = range(5)      # steps of an OP_CALIB_WALK table
INSTR_WIDTH = 16
LOGIC_AND, LOGIC_OR, LOGIC_NOT = 0, 1, 2
TILE_SUBJECT_ROWS, TILE_OBJECT_ROWS = 0, 1
WANT_SUBJECT, WANT_OBJECT = 1, 2
RELATE_LONE_FORALL_IDENTITY, RELATE_DIAG_ABSENT = 1, 2
_NEG30_BITS = int(np.float32(-30.0).view(np.int32))
_NEG30_BF16X2 = int(np.array([0xC1F0C1F0], np.uint32).view(np.int32)[0])             # two bf16 -30.0 per 32-bit word (bf16 relation tiles)
TILE_F32, TILE_BF16 = 0, 1
_ALIGN = 256


class Unsupported(Exception):
    """A shape the executor does not take (the Python loop does): raised inside the lowering, turned into `None` by build_plan."""


def shared_requests(q_img, n_img, items):
    """Shared scenes: the distinct (scene, relation column, orientation) triples among `items` = [(full columns, predicate -> question,
    orientation)], as pair-kernel request arrays [K', scenes] over the IMAGE-level geometry, plus, per item, the index of every
    predicate's tile among the distinct ones (`U` = the extra all-absent tile for no-op tokens).  One definition for the Python operators
    (visual_oracle._prefetch_relations_shared) and the lowering below."""
    uniq, per_item = {}, []
    for full, pq, orient in items:
        idx = np.empty(len(pq), np.int64)
        for p in range(len(pq)):
            idx[p] = -1 if full[p] < 0 else uniq.setdefault((int(q_img[pq[p]]), int(full[p]), int(orient[p])), len(uniq))
        per_item.append(idx)
    U = len(uniq)
    slot_of, keys = np.zeros(n_img, np.int64), sorted(uniq, key=uniq.get)
    slots = np.empty(U, np.int64)
    for u, (img, _, _) in enumerate(keys):
        slots[u] = slot_of[img]
        slot_of[img] += 1
    K = int(slot_of.max()) if U else 1
    col, til, ori = np.full((K, n_img), -1, np.int32), np.zeros((K, n_img), np.int32), np.zeros((K, n_img), np.uint8)
    for u, (img, c, o) in enumerate(keys):
        col[slots[u], img], til[slots[u], img], ori[slots[u], img] = c, u, o
    return U, col, til, ori, [np.where(i < 0, U, i) for i in per_item]


class ModelSpec(object):
    """What the lowering must know about the model: widths (they size the workspace), the oracle's option normalisation and the
    interpreter's likelihood threshold.  Picklable (collate workers build plans)."""

    def __init__(self, featurizer_widths, attribute_widths, hid1, D, normalize, likelihood_threshold, relation_index, tile_bf16=False, calib=None):
        self.featurizer_widths = [int(w) for w in featurizer_widths]      # output width of every featurizer layer (the last = D - 4)
        self.attribute_widths = [int(w) for w in attribute_widths]        # output width of every attribute-network layer
        self.hid1, self.D = int(hid1), int(D)
        self.normalize = bool(normalize)
        self.likelihood_threshold = float(likelihood_threshold)
        self.relation_index = np.asarray(relation_index, np.int32)        # 333-column index -> column of the full concept table
        # relation tiles stored as bf16 where every consumer reads them directly (relation_tile_dtype: bf16 with a packed second layer of > 256 rows:
        # visual_oracle.prefetch_relations' rule); decided per batch below (NS % 8 == 0, no choose_rel)
        self.tile_bf16 = bool(tile_bf16)
        # attention calibration (activate_attention_transfer with the modulator switched on): None, or {"state_dim": S, "lstm_in": 18 + token
        # embedding width, "ops_index": operator -> one-hot position (batch_gqa_interpreter.py:67-70)}
        self.calib = None if calib is None else dict(state_dim=int(calib["state_dim"]), lstm_in=int(calib["lstm_in"]), ops_index=dict(calib["ops_index"]))

    def key(self):
        return (tuple(self.featurizer_widths), tuple(self.attribute_widths), self.hid1, self.D, self.normalize, self.likelihood_threshold,
                self.relation_index.tobytes(), self.tile_bf16,
                None if self.calib is None else (self.calib["state_dim"], self.calib["lstm_in"], tuple(sorted(self.calib["ops_index"].items()))))


class _W(object):
    """A workspace reference: region 'o' (results, at the start of the arena) or 't' (intermediates, behind the results)."""
    __slots__ = ("region", "off")

    def __init__(self, region, off):
        self.region, self.off = region, off

    def at(self, nbytes):
        return _W(self.region, self.off + int(nbytes))


class _VS(object):
    """Symbolic BatchVariableSet: names, attention block [rows, NS] in the workspace, quantifiers on the host, predicate -> question map."""
    __slots__ = ("names", "att", "rows", "quant", "pq", "prev_num")

    def __init__(self, names, att, rows, quant, pq=None, prev_num=0):
        self.names, self.att, self.rows, self.quant, self.pq, self.prev_num = names, att, rows, np.asarray(quant, np.float32), pq, prev_num


class _AS(object):
    """Symbolic BatchAttentionState (fol_types.py; batch_base_types.py:256-310): names and the LSTM state (h, c) [rows, S] in the workspace - c directly
    behind h wherever `_Calibration` allocates it, so that a gate / gather / sum / add of a state is ONE launch over [2 rows, S].  `zero`: the all-zero
    state; `gate_of` = (x, y, flags) when the state is BatchAttentionState.gate(x, y, flags) (what lets `_Calibration.add` see x + y in
    gate(x, y, f) + gate(y, x, f))."""
    __slots__ = ("names", "h", "c", "rows", "zero", "gate_of")

    def __init__(self, names, h, c, rows, zero=False, gate_of=None):
        self.names, self.h, self.c, self.rows, self.zero, self.gate_of = names, h, c, rows, zero, gate_of


class NativePlan(object):
    """instrs [n, 16] int64 (host), blob (uint8, one upload), workspace size, the leading `out_bytes` of which the host reads back, the scene
    header and what the host needs to turn the read-back into the reference's result dict."""

    def __init__(self):
        self.instrs = None
        self.blob = None
        self.ws_bytes = 0
        self.out_bytes = 0
        self.scene = None             # dict: O, NS, max_n, n_obj / img_n_obj / obj_off blob offsets, raw_cols
        self.result = None            # dict: kind, type, lp (offset, count), ...
        self.key = None
        self.launches = 0


class _Builder(object):

    def __init__(self, pb, ontology, spec):
        self.pb, self.ont, self.spec = pb, ontology, spec
        self.instrs = []
        self.calibration = None
        self._blob_parts, self._blob_size, self._blob_memo = [], 0, {}
        self._size = {"o": 0, "t": 0}
        if getattr(pb, "_object_nums", None) is None:
            raise Unsupported("a ProgramBatch without per-image object counts")
        # two geometries, as fol_types.BatchWorld keeps them: per IMAGE (object rows, pair kernel, attribute columns) and per QUESTION (attention
        # rows, logic kernels); they coincide unless the collater shared scenes (`_question_image`: question -> image)
        img_n = [int(n) for n in pb._object_nums]
        qi = getattr(pb, "_question_image", None)
        self.shared = qi is not None
        self.q_img = np.arange(len(img_n), dtype=np.int64) if qi is None else np.asarray(qi, np.int64)
        if not img_n or min(img_n) < 1 or len(self.q_img) == 0 or self.q_img.min() < 0 or self.q_img.max() >= len(img_n):
            raise Unsupported("an empty batch or an image without objects")
        n_list = [img_n[i] for i in self.q_img]
        self.n_list, self.Q = n_list, len(n_list)
        self.img_n, self.n_img = img_n, len(img_n)
        self.O = int(sum(img_n))
        self.NS = max(4, (max(n_list) + 3) // 4 * 4)
        self.max_n = max(n_list)
        if max(img_n) > self.NS:
            raise Unsupported("a scene no question looks at is larger than the batch's blocks")
        n = np.asarray(img_n, np.int64)
        self.pair_num = int((n * (n - 1)).sum())
        self.b_img_n_obj = self.arr(n.astype(np.int32))
        self.b_n_obj = self.arr(np.asarray(n_list, np.int32))
        self.b_obj_off = self.arr(np.concatenate([[0], np.cumsum(n)]).astype(np.int32))
        self.ident = np.arange(self.Q, dtype=np.int32)
        self.b_ident = self.arr(self.ident)
        self._zeros = None

    # ---- blob / arena -----------------------------------------------------------------------------------------------------------------
    def arr(self, a):
        a = np.ascontiguousarray(a)
        key = (a.dtype.str, a.tobytes())
        hit = self._blob_memo.get(key)
        if hit is None:
            pad = (-self._blob_size) % 16
            if pad:
                self._blob_parts.append(np.zeros(pad, np.uint8))
                self._blob_size += pad
            hit = self._blob_memo[key] = self._blob_size
            self._blob_parts.append(a.reshape(-1).view(np.uint8))
            self._blob_size += a.nbytes
        return hit

    def alloc(self, nbytes, region="t"):
        off = self._size[region]
        self._size[region] = off + (int(nbytes) + _ALIGN - 1) // _ALIGN * _ALIGN
        return _W(region, off)

    def block(self, rows, region="t"):
        return self.alloc(rows * self.NS * 4, region)

    def emit(self, *ops):
        assert len(ops) <= INSTR_WIDTH
        self.instrs.append(list(ops) + [0] * (INSTR_WIDTH - len(ops)))

    def zeros(self):
        if self._zeros is None:
            self._zeros = self.block(self.Q)
            self.emit(OP_FILL, self._zeros, self.Q * self.NS, 0)
        return self._zeros

    # ---- scene stage (interpreter.build_scene / visual_oracle.prepare_scene) --------------------------------------------------------------
    def scene_stage(self):
        sp, O = self.spec, self.O
        D = sp.D
        self.obj = self.alloc(O * D * 4)
        x, ldx, src = 0, 0, 0                                      # the raw features: pointer and row stride come from the scene header
        for i, w in enumerate(sp.featurizer_widths):
            last = i == len(sp.featurizer_widths) - 1
            y, ldy = (self.obj, D) if last else (self.alloc(O * w * 4), w)
            self.emit(OP_DENSE, 0, i, src, x, ldx, y, ldy, O)
            x, ldx, src = y, ldy, 1
        if sp.featurizer_widths[-1] != D - 4:
            raise Unsupported("featurizer width")
        self.emit(OP_BOX_POSITIONS, self.obj, D, D - 4)
        x, ldx = self.obj, D
        for i, w in enumerate(sp.attribute_widths):
            y = self.alloc(O * w * 4)
            self.emit(OP_DENSE, 1, i, 1, x, ldx, y, w, O)
            x, ldx = y, w
        self.hidden, self.H = x, ldx
        self.uv = self.alloc(O * 2 * sp.hid1 * 4)
        self.emit(OP_DENSE, 2, 0, 1, self.obj, D, self.uv, 2 * sp.hid1, O)

    # ---- relation tiles: ONE pair-kernel launch for every relation operator (visual_oracle.prefetch_relations) ----------------------------
    def relation_stage(self, ops):
        Q, NS = self.Q, self.NS
        entries = []                                               # (op index, lowered tokens, predicate -> question, orientation)
        for i, ob in enumerate(ops):
            if not ob._arguments:
                continue
            if ob._op_name in ("relate", "verify_rel"):
                low = get_lowered(ob._arguments[0], self.ont, TokenType.RELATION)
                if low.any_valid and len(low.cols) == Q:
                    orient = np.asarray([TILE_OBJECT_ROWS if f else TILE_SUBJECT_ROWS for f in ob._arguments[1]], np.uint8)
                    entries.append((i, low, np.arange(Q), orient))
            elif ob._op_name == "choose_rel":
                flat, batch_index = flatten_list(ob._arguments[0])
                low = lower_tokens(flat, self.ont, TokenType.RELATION)
                if low.any_valid:
                    entries.append((i, low, np.asarray(batch_index, np.int64), np.zeros(len(flat), np.uint8)))
        self.tiles = {}
        self.tile_dtype = TILE_F32
        if not entries:
            return
        # (images of one object have no pairs: the tiles keep their absent fill and the pair kernel, which returns at once for max_n < 2, is not
        # even requested - the Python operators' route does the same through dfol_pair_ll_*'s early return)
        bf16 = self.spec.tile_bf16 and NS % 8 == 0 and all(ob._op_name != "choose_rel" for ob in ops)
        self.tile_dtype = TILE_BF16 if bf16 else TILE_F32
        esz = 2 if bf16 else 4
        fill_words = lambda count: (count * NS * NS * esz // 4, _NEG30_BF16X2 if bf16 else _NEG30_BITS)
        D = self.spec.D
        if self.shared:
            # one tile per distinct (scene, concept, orientation) from the pair kernel, then every operator's per-predicate tiles are row gathers
            # of those (visual_oracle._prefetch_relations_shared)
            items = []
            for i, low, pq, orient in entries:
                full = np.where(low.cols >= 0, self.spec.relation_index[np.maximum(low.cols, 0)], -1).astype(np.int32)
                items.append((full, np.asarray(pq, np.int64), orient))
            U, col, til, ori, maps = shared_requests(self.q_img, self.n_img, items)
            distinct = self.alloc((U + 1) * NS * NS * esz)                # tile U: all absent (no-op tokens)
            self.emit(OP_FILL, distinct, *fill_words(U + 1))
            if U and self.pair_num > 0:
                self.emit(OP_PAIR_LL, self.uv, 2 * self.spec.hid1, self.obj.at((D - 4) * 4), D, self.arr(col), self.arr(til), self.arr(ori), col.shape[0], distinct,
                          self.n_img, self.tile_dtype)
            for (i, low, pq, orient), m in zip(entries, maps):
                P = len(pq)
                mine = self.alloc(P * NS * NS * esz)
                self.emit(OP_GATHER_TILES, distinct, self.arr(m.astype(np.int32)), P, mine, NS * NS * esz // 4)
                self.tiles[i] = (mine, low)
            return
        total = sum(len(e[1].cols) for e in entries)
        tiles = self.alloc(total * NS * NS * esz)
        self.emit(OP_FILL, tiles, *fill_words(total))
        rows_col, rows_tile, rows_orient, base = [], [], [], 0
        for i, low, pq, orient in entries:
            P = len(pq)
            pq = np.asarray(pq, np.int64)
            slot = np.zeros(P, np.int64)                           # j-th predicate of its question, in predicate order
            if P > 1 and not (P == Q and pq[0] == 0 and pq[-1] == Q - 1 and (np.diff(pq) == 1).all()):
                order = np.argsort(pq, kind="stable")
                sq = pq[order]
                start = np.flatnonzero(np.concatenate([[True], sq[1:] != sq[:-1]]))
                slot[order] = np.arange(P) - np.repeat(start, np.diff(np.concatenate([start, [P]])))
            K = int(slot.max()) + 1 if P else 1
            col = np.full((K, Q), -1, np.int32)
            til = np.zeros((K, Q), np.int32)
            ori = np.zeros((K, Q), np.uint8)
            full = np.where(low.cols >= 0, self.spec.relation_index[np.maximum(low.cols, 0)], -1).astype(np.int32)
            col[slot, pq] = full
            til[slot, pq] = base + np.arange(P, dtype=np.int32)
            ori[slot, pq] = orient
            rows_col.append(col), rows_tile.append(til), rows_orient.append(ori)
            self.tiles[i] = (tiles.at(base * NS * NS * esz), low)
            base += P
        col, til, ori = np.concatenate(rows_col), np.concatenate(rows_tile), np.concatenate(rows_orient)
        if self.pair_num > 0:
            self.emit(OP_PAIR_LL, self.uv, 2 * self.spec.hid1, self.obj.at((D - 4) * 4), D, self.arr(col), self.arr(til), self.arr(ori), col.shape[0], tiles, Q,
                      self.tile_dtype)

    # ---- attribute blocks: ONE launch for every attribute token list of the batch ---------------------------------------------------------
    def attribute_stage(self, requests):
        """requests: [(key, lowered, predicate -> question int32)] -> self.attr[key] = block [P, NS]."""
        self.attr = {}
        todo = [(k, low, pq) for k, low, pq in requests if low.any_valid]
        if not todo:
            return
        total = sum(len(low.cols) for _, low, _ in todo)
        ll = self.block(total)
        cols = np.concatenate([low.cols for _, low, _ in todo]).astype(np.int32)
        pimg = np.concatenate([self.q_img[np.asarray(pq, np.int64)].astype(np.int32) for _, _, pq in todo])        # predicate -> scene
        self.emit(OP_ATTR_LL, self.hidden, self.H, self.arr(pimg), self.arr(cols), total, ll)
        base = 0
        for k, low, pq in todo:
            self.attr[k] = ll.at(base * self.NS * 4)
            base += len(low.cols)

    # ---- operators --------------------------------------------------------------------------------------------------------------------
    def _normalize(self, ll, low, pq, rank, normalized, esz=4):
        """The option normalisation of visual_oracle._block_likelihood_needed (classifier_oracle.py:72-75, 124-127), in place."""
        if not (self.spec.normalize and normalized):
            return
        valid = low.valid.astype(bool)
        seg = segments_of(np.asarray(pq)[valid])
        if len(seg) - 1 == int(valid.sum()):
            return
        pq = np.asarray(pq, np.int32)
        if low.all_valid:
            self.emit(OP_OPTION_NORMALIZE, ll, self.arr(seg.astype(np.int32)), len(seg) - 1, self.arr(pq), rank)
            return
        # no-op tokens inside an option list: the compressed list is normalised and the default blocks put back (classifier_oracle.py:56-60,
        # 72-75; visual_oracle._block_likelihood_needed) - here: gather the valid rows, normalise them, gather back through a map whose no-op
        # rows point at one extra all-default row
        if rank == 2 and esz != 4:
            raise Unsupported("option lists over bf16 tiles")
        keep = np.flatnonzero(valid).astype(np.int32)
        width = self.NS if rank == 1 else self.NS * self.NS            # 32-bit words per row
        compact = self.alloc((len(keep) + 1) * width * 4)
        self.emit(OP_GATHER_TILES, ll, self.arr(keep), len(keep), compact, width)
        self.emit(OP_FILL, compact.at(len(keep) * width * 4), width, _NEG30_BITS)
        self.emit(OP_OPTION_NORMALIZE, compact, self.arr(seg.astype(np.int32)), len(seg) - 1, self.arr(pq[keep]), rank)
        back = np.full(len(pq), len(keep), np.int32)
        back[keep] = np.arange(len(keep), dtype=np.int32)
        self.emit(OP_GATHER_TILES, compact, self.arr(back), len(pq), ll, width)

    def modulate(self, att, mkey, pq_arr, P):
        """BatchVariableSet.apply_modulations (batch_base_types.py:170-187) when the calibration passes left modulations for this operator."""
        mods = None if self.calibration is None else self.calibration.mods.pop(mkey, None)
        if mods is None:
            return att
        out = self.block(P)
        self.emit(OP_MODULATE, att, mods, self.arr(np.asarray(pq_arr, np.int32)), P, out)
        return out

    def filter(self, vs, tokens, key, pq=None, normalized=True, mkey=None):
        """FilterBatch.forward (logic_ops.py; batch_base_ops.py:311-405).  mkey: the (operator index, role) its modulations were left under."""
        low = get_lowered(tokens, self.ont, TokenType.ATTRIBUTE)
        if not low.any_valid:
            return vs
        P = len(tokens)
        if pq is None:
            if P != vs.rows:
                raise Unsupported("batch size mismatch")
            pq_arr, quant = self.ident if vs.rows == self.Q else np.arange(P, dtype=np.int32), vs.quant
        else:
            pq_arr = np.asarray(pq, np.int32)
            if len(pq_arr) != P or vs.rows != self.Q:
                raise Unsupported("batch size mismatch")
            quant = vs.quant[pq_arr]
        if vs.pq is not None:
            raise Unsupported("a filter over an expanded variable set")
        ll = self.attr[key]
        self._normalize(ll, low, pq_arr, 1, normalized)
        out = self.block(P)
        self.emit(OP_FILTER, vs.att, ll, self.arr(pq_arr), self.arr(low.neg) if low.any_neg else -1, -1 if low.all_valid else self.arr(low.valid), P, out)
        if mkey is not None:
            out = self.modulate(out, mkey, pq_arr, P)                                  # :401-403
        return _VS(vs.names, out, P, quant, None if pq is None else pq_arr, vs.prev_num + 1)

    def select(self, tokens, key, mkey=None):
        """GQASelectBatch.forward (batch_gqa_ops.py:168-183)."""
        Q = self.Q
        if tokens is None:
            names, att = ["entity"] * Q, None
        else:
            names = ["entity" if a is None or a.lower() in ("_", "scene") else a for a in tokens][:Q]
            att = [None if a is None or a.lower() in ("_", "scene") else a for a in tokens][:Q]
        x = _VS(names, self.zeros(), Q, np.full(Q, float(Quantifier.EXISTS), np.float32))
        if att is None or all(a is None for a in att):
            return x
        return self.filter(x, att, key, mkey=mkey)

    @staticmethod
    def select_tokens(tokens, Q):
        """The token list GQASelectBatch hands its filter (names `_` / `scene` are no-ops)."""
        if tokens is None:
            return None
        return [None if a is None or a.lower() in ("_", "scene") else a for a in tokens][:Q]

    def gate_names(self, x_names, y_names, flags):
        return [a if f > 0 else b for a, b, f in zip(x_names, y_names, flags)]

    def gate(self, x, y, flags):
        """BatchVariableSet.gate (fol_types.py; batch_base_types.py:149-168): rows with flag 1 come from x."""
        flags = [0 if f is None else f for f in flags]
        names = self.gate_names(x.names, y.names, flags)
        g = np.asarray([float(f) for f in flags], np.float32)
        if len(flags) == x.rows and all(f == 1 for f in flags):
            return _VS(names, x.att, x.rows, x.quant, x.pq, 0)
        out, outq = self.block(x.rows), self.alloc(x.rows * 4)
        self.emit(OP_GATE, x.att, y.att, self.arr(x.quant), self.arr(y.quant), self.arr(g), x.rows, out, outq)
        return _VS(names, out, x.rows, g * x.quant + (1.0 - g) * y.quant, x.pq, 0)

    def mask_gate(self, x, prev, mask, valid):
        """The interpreter's pass-through for questions lacking an operator (batch_base_interpreter.py:166-167).  Where every masked-out
        question carries a no-op token (what collate produces), filter_fwd / relate_one_fwd already left the incoming row in place
        (active == 0 copies the prior row), so no launch is needed - only names and quantifiers follow the mask; otherwise a real gate."""
        if mask is None or x is prev:
            return x
        g = np.asarray(mask, np.float32)
        # (a calibrated operator modulated EVERY row of its result, the pass-through rows included - apply_modulations knows no mask - so the
        # incoming rows must really be put back)
        if self.calibration is not None and bool((g == 0).any()):
            return self.gate(x, prev, mask)
        if len(mask) != x.rows or x.rows != prev.rows or bool(((g == 0) & np.asarray(valid, bool)[:len(mask)]).any()) or not bool(((g == 0) | (g == 1)).all()):
            return self.gate(x, prev, mask)
        names = self.gate_names(x.names, prev.names, mask)
        return _VS(names, x.att, x.rows, g * x.quant + (1.0 - g) * prev.quant, x.pq, 0)

    def relate(self, i, prev, relation_list, is_subject, names_tokens, key):
        """GQARelateBatch.forward on the fused single-posterior kernel (gqa_ops.GQARelateBatch._forward_fused; batch_gqa_ops.py:364-371)."""
        x = self.select(names_tokens, key, mkey=(i, "sel"))
        host = [0.0 if f is None else float(f) for f in is_subject]
        hit = self.tiles.get(i)
        if hit is None or x.rows != prev.rows or prev.pq is not None or x.rows != self.Q:
            raise Unsupported("a relate without prefetched tiles")
        tiles, low = hit
        out = self.block(self.Q)
        self.emit(OP_RELATE_ONE, x.att, prev.att, tiles, self.b_ident, self.arr(prev.quant), self.arr(low.neg) if low.any_neg else -1,
                  -1 if low.all_valid else self.arr(low.valid), self.Q, 1 if self.Q == 1 else 0, out, self.tile_dtype)
        if self.calibration is not None and (i, "rel_s") in self.calibration.mods:
            # RelateBatch.forward calibrates both posteriors (batch_base_ops.py:588-594) and GQARelateBatch keeps one per question: the kept one
            # with that side's modulations (gqa_ops.GQARelateBatch._forward_fused)
            ms, mo = self.calibration.mods.pop((i, "rel_s")), self.calibration.mods.pop((i, "rel_o"))
            if all(f > 0 for f in host):
                mods = ms
            elif not any(f > 0 for f in host):
                mods = mo
            else:
                mods = self.alloc(self.Q * 4 * 4)
                self.emit(OP_SELECT_ROWS, ms, mo, self.arr(np.asarray([1 if f > 0 else 0 for f in host], np.uint8)), self.Q, 4, mods)
            out2 = self.block(self.Q)
            self.emit(OP_MODULATE, out, mods, self.b_ident, self.Q, out2)
            out = out2
        quant = np.where(np.asarray([f > 0 for f in host]), x.quant, prev.quant).astype(np.float32)
        return _VS(x.names, out, self.Q, quant, None, x.prev_num + prev.prev_num + 1), low

    def quantify(self, vs, quant=None, region="t"):
        lp = self.alloc(vs.rows * 4, region)
        pq = self.ident if vs.pq is None and vs.rows == self.Q else (np.arange(vs.rows, dtype=np.int32) if vs.pq is None else vs.pq)
        if vs.pq is None and vs.rows != self.Q:
            raise Unsupported("an unmapped variable set of another size")
        self.emit(OP_QUANTIFY, vs.att, self.arr(vs.quant if quant is None else quant), self.arr(pq), vs.rows, lp)
        return lp

    def seg_off(self, batch_index):
        counts = np.bincount(np.asarray(batch_index, np.int64), minlength=self.Q)
        return self.arr(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32))

    def find_max(self, lp, batch_index, P):
        bi = np.asarray(batch_index)
        if not (len(bi) > 0 and (len(bi) < 2 or bool(np.all(bi[1:] >= bi[:-1]))) and int(bi[-1]) < self.Q):
            raise Unsupported("an unsorted option list")
        flags = self.alloc(P, "o")
        thr = int(np.float32(self.spec.likelihood_threshold).view(np.int32))
        self.emit(OP_FIND_MAX_IND, lp, self.seg_off(batch_index), self.Q, thr, flags)
        return flags

    def category_options(self, category_list, names):
        lists = [self.ont.query(c if c not in ["name", "type"] else n) for c, n in zip(category_list, names)]
        return lists, flatten_list(lists)


# ---- attention calibration: the LSTM walks the aligned program forward and backward before it is executed (batch_base_interpreter.py:87-140) ------
class _Calibration(object):
    """The symbolic walk of interpreter._calibration_passes and of every operator's transform_attention (gqa_ops.py / logic_ops.py;
    batch_gqa_ops.py and batch_base_ops.py:407-467, 598-684): emits the LSTM cells, state gates and attention-output products of both passes and
    leaves, per (operator index, role), the workspace reference of the [P, 4] modulations the execution pass applies.  Roles: "sel" (the filter of
    an operator's select), "flt" / "flt:0" / "flt:1" (its own filter[s]), "rel_s" / "rel_o" (its relate's two posteriors)."""

    def __init__(self, b, pb):
        self.b, self.c = b, b.spec.calib
        self.S = self.c["state_dim"]
        self.mods, self.fwd = {}, {}
        self._zero, self._feat_memo, self._table, self._table_rows = {}, {}, {}, []
        self._walk = []                                           # (index in b.instrs, None or (rows, [WALK_* step])) per launch of the passes
        md = pb._meta_data if isinstance(getattr(pb, "_meta_data", None), dict) else {}
        self.index, self.embedding = md.get("index"), md.get("embedding")
        self.E, self._emb_host = None, None

    # -- token embeddings: a plan-local table in the blob (the rows get_embedding would hand the LSTM: base_oracle.py:45-55) -------------------------
    def _embedding_rows(self, names):
        try:
            ind = [self.index[t] for t in names]
            if self._emb_host is None:                            # (one host copy per plan: a device tensor when the plan is built after to_cuda)
                emb = self.embedding
                self._emb_host = np.asarray(emb if not hasattr(emb, "detach") else emb.detach().cpu().numpy())
            rows = self._emb_host[ind, :]
            return np.asarray(rows, np.float32).reshape(len(names), -1), "i"
        except (KeyError, TypeError, IndexError):
            rows = self.b.ont.get_embeddings(names)
            if rows is None:
                raise Unsupported("tokens without an embedding (no index entry, no embedding file)")
            return np.asarray(rows, np.float32).reshape(len(names), -1), "o"

    def features(self, tokens, op_name, type_flag):
        """-> the description of [P, lstm_in] rows [operator one-hot, type flag, token embedding], zero rows for no-op tokens: (head, n_head, table, E, idx)."""
        from .host_util import detect_negations, is_valid_token
        key = (tuple(str(t) for t in tokens), float(type_flag), op_name)
        hit = self._feat_memo.get(key)
        if hit is not None:
            return hit
        ind = [is_valid_token(v) for v in tokens]
        kept = [t for t, k in zip(tokens, ind) if k]
        _, _, names = detect_negations(kept)
        rows, src = self._embedding_rows(list(names))
        if self.E is None:
            self.E = rows.shape[1]
        head = np.zeros(len(self.c["ops_index"]) + 1, np.float32)
        head[self.c["ops_index"][op_name]] = 1.0
        head[-1] = type_flag
        if rows.shape[1] != self.E or len(head) + self.E != self.c["lstm_in"]:
            raise Unsupported("token embeddings of another width than the calibration LSTM's input")
        idx, j = np.full(len(tokens), -1, np.int32), 0
        for p, k in enumerate(ind):
            if k:
                tk = (src, names[j])
                if tk not in self._table:
                    self._table[tk] = len(self._table_rows)
                    self._table_rows.append(rows[j])
                idx[p] = self._table[tk]
                j += 1
        # (the rows are built inside the LSTM launch that reads them - dfol_lstm_cell_tokens_f32 - from this description; the table's blob offset is
        # patched in by finish(): rows keep arriving while the passes are walked)
        out = (self.b.arr(head), len(head), ("table", self), self.E, self.b.arr(idx))
        self._feat_memo[key] = out
        return out

    def _emit(self, ops, rows=None, step=None):
        """One launch of the passes; step: the same work as a WALK_* entry over states of `rows` rows (None: not row-wise, e.g. an option-list gather)."""
        self.b.emit(*ops)
        self._walk.append((len(self.b.instrs) - 1, None if step is None else (rows, list(step))))

    def finish(self):
        off = self.b.arr(np.stack(self._table_rows)) if self._table_rows else -1
        for row in self.b.instrs + [w[1] for _, w in self._walk if w is not None]:
            for k, v in enumerate(row):
                if isinstance(v, tuple) and v[0] == "table" and v[1] is self:
                    row[k] = off

    def fold_runs(self, resolve):
        """DFOL_CALIB_WALK=1 (opt-in): runs of row-wise launches over states of one size -> one OP_CALIB_WALK each (the workgroup that owns 16 rows walks
        the run: csrc/dfol_program.hip calib_walk_kernel; same device code, bit-identical results).  Measured SLOWER than the launches it replaces
        (256 questions: 1.89 against 1.70 ms per batch): a workgroup then pulls all 294 KB of a cell's weights through one CU's L1, which
        the stand-alone cell spreads over 112 CUs - so the default keeps the separate launches."""
        import os
        if os.environ.get("DFOL_CALIB_WALK", "0") != "1" or not self._walk:
            return
        instrs, out, k = self.b.instrs, [], 0
        first = self._walk[0][0]
        assert [i for i, _ in self._walk] == list(range(first, first + len(self._walk)))      # (the passes are emitted in one piece)
        out.extend(instrs[:first])
        while k < len(self._walk):
            w = self._walk[k][1]
            e = k + 1
            if w is not None:
                while e < len(self._walk) and self._walk[e][1] is not None and self._walk[e][1][0] == w[0]:
                    e += 1
            if w is None or e - k < 2:
                out.append(instrs[first + k])
                k += 1
                continue
            table = np.zeros((e - k, INSTR_WIDTH), np.int64)
            for t, (_, (_, step)) in enumerate(self._walk[k:e]):
                table[t, :len(step)] = [resolve(v) for v in step]
            row = [OP_CALIB_WALK, self.b.arr(table), e - k, w[0]]
            out.append(row + [0] * (INSTR_WIDTH - len(row)))
            k = e
        out.extend(instrs[first + len(self._walk):])
        self.b.instrs = out

    # -- attention states ---------------------------------------------------------------------------------------------------------------------
    def _state(self, rows):
        """Workspace for one state: h [rows, S], then c [rows, S]."""
        h = self.b.alloc(2 * rows * self.S * 4)
        return h, h.at(rows * self.S * 4)

    def _paired(self, x):
        return x.c.region == x.h.region and x.c.off == x.h.off + x.rows * self.S * 4

    def zero_state(self, names, rows):
        z = self._zero.get(rows)
        if z is None:
            z = self._zero[rows] = self._state(rows)
            self._emit((OP_FILL, z[0], 2 * rows * self.S, 0), rows, (WALK_FILL, z[0], 2, self.S, 0))
        return _AS(list(names), z[0], z[1], rows, zero=True)

    def gate(self, x, y, flags):
        """BatchAttentionState.gate (:279-298) with host flags: rows whose flag is > 0 come from x."""
        host = [0 if f is None else f for f in flags]
        names = [a if f > 0 else c for a, c, f in zip(x.names, y.names, host)]
        if not all(f in (0, 1, 0.0, 1.0, True, False) for f in host):
            raise Unsupported("a fractional state gate")
        of = (x, y, tuple(1 if f > 0 else 0 for f in host))
        if all(f > 0 for f in host):
            return _AS(names, x.h, x.c, x.rows, zero=x.zero, gate_of=of)
        if not any(f > 0 for f in host):
            return _AS(names, y.h, y.c, y.rows, zero=y.zero, gate_of=of)
        if x.rows != y.rows or len(host) != x.rows:
            raise Unsupported("a state gate over differing batch sizes")
        same = lambda u, v: u.region == v.region and u.off == v.off
        if same(x.h, y.h) and same(x.c, y.c):                    # both sides are one state (two zero states, a relate's twin posteriors): no launch
            return _AS(names, x.h, x.c, x.rows, zero=x.zero and y.zero, gate_of=of)
        pick = np.asarray(of[2], np.uint8)
        h, c = self._state(x.rows)
        if self._paired(x) and self._paired(y):
            pick2 = self.b.arr(np.concatenate([pick, pick]))
            self._emit((OP_SELECT_ROWS, x.h, y.h, pick2, 2 * x.rows, self.S, h), x.rows, (WALK_SELECT, x.h, y.h, pick2, 2, self.S, h))
        else:
            pick = self.b.arr(pick)
            self._emit((OP_SELECT_ROWS, x.h, y.h, pick, x.rows, self.S, h), x.rows, (WALK_SELECT, x.h, y.h, pick, 1, self.S, h))
            self._emit((OP_SELECT_ROWS, x.c, y.c, pick, x.rows, self.S, c), x.rows, (WALK_SELECT, x.c, y.c, pick, 1, self.S, c))
        return _AS(names, h, c, x.rows, gate_of=of)

    def expand(self, x, pq):
        P = len(pq)
        pq = np.asarray(pq, np.int32)
        if x.zero:                                                # (rows of the zero state)
            return self.zero_state(x.names, P)
        h, c = self._state(P)
        if self._paired(x):
            self._emit((OP_GATHER_TILES, x.h, self.b.arr(np.concatenate([pq, pq + np.int32(x.rows)])), 2 * P, h, self.S))
        else:
            idx = self.b.arr(pq)
            self._emit((OP_GATHER_TILES, x.h, idx, P, h, self.S))
            self._emit((OP_GATHER_TILES, x.c, idx, P, c, self.S))
        return _AS(x.names, h, c, P)

    def squeeze(self, x, pq):
        pq = np.asarray(pq, np.int64)
        if len(pq) > 1 and not bool(np.all(pq[1:] >= pq[:-1])):
            raise Unsupported("an unsorted option list")
        Q = self.b.Q
        h, c = self._state(Q)
        if self._paired(x) and len(pq) == x.rows:
            counts = np.bincount(pq, minlength=Q)
            seg2 = self.b.arr(np.concatenate([[0], np.cumsum(np.concatenate([counts, counts]))]).astype(np.int32))
            self._emit((OP_SEGMENT_SUM_ROWS, x.h, seg2, 2 * Q, self.S, h))
        else:
            seg = self.b.seg_off(pq)
            self._emit((OP_SEGMENT_SUM_ROWS, x.h, seg, Q, self.S, h))
            self._emit((OP_SEGMENT_SUM_ROWS, x.c, seg, Q, self.S, c))
        return _AS(x.names, h, c, Q)

    def add(self, x, y):
        """x + y (BatchAttentionState.__add__).  A relate adds the two gates of one pair of states, gate(a, b, f) + gate(b, a, f): every row of
        that sum is a + b (b + a: the same float), so the gates are not needed for it - and with the zero state on one side (a select without a
        token) the sum is the other state itself (bit for bit up to the sign of a zero)."""
        if x.rows != y.rows:
            raise Unsupported("states of differing batch sizes")
        gx, gy = x.gate_of, y.gate_of
        if gx is not None and gy is not None and gx[2] == gy[2] and gx[0] is gy[1] and gx[1] is gy[0] and gx[0].rows == gx[1].rows == x.rows:
            names, x, y = x.names, gx[0], gx[1]
            x = _AS(names, x.h, x.c, x.rows, zero=x.zero)
        if y.zero:
            return _AS(x.names, x.h, x.c, x.rows, zero=x.zero)
        if x.zero:
            return _AS(x.names, y.h, y.c, y.rows)
        h, c = self._state(x.rows)
        if self._paired(x) and self._paired(y):
            self._emit((OP_LOGIC, LOGIC_AND, x.h, y.h, 2 * x.rows * self.S, h), x.rows, (WALK_ADD, x.h, y.h, 0, 2, self.S, h))
        else:
            self._emit((OP_LOGIC, LOGIC_AND, x.h, y.h, x.rows * self.S, h), x.rows, (WALK_ADD, x.h, y.h, 0, 1, self.S, h))
            self._emit((OP_LOGIC, LOGIC_AND, x.c, y.c, x.rows * self.S, c), x.rows, (WALK_ADD, x.c, y.c, 0, 1, self.S, c))
        return _AS(x.names, h, c, x.rows)

    def lstm(self, which, feats, state, rows):
        if state.rows != rows:
            raise Unsupported("an LSTM state of another batch size than its tokens")
        h, c = self._state(rows)
        self._emit((OP_LSTM_CELL, which, -1, state.h, state.c, rows, h, c) + tuple(feats), rows, (WALK_LSTM, which, state.h, state.c, h, c) + tuple(feats))
        return _AS(state.names, h, c, rows)

    def modulations(self, fwd, bwd, rows):
        if fwd.rows != rows or bwd.rows != rows:
            raise Unsupported("modulations over differing batch sizes")
        out = self.b.alloc(rows * 4 * 4)
        self._emit((OP_ATT_MODULATIONS, fwd.h, bwd.h, rows, out), rows, (WALK_ATT_MODULATIONS, fwd.h, bwd.h, out))
        return out

    # -- FilterBatch / RelateBatch.transform_attention (logic_ops.py; batch_base_ops.py:407-467, 598-684) ----------------------------------------
    def filter_ta(self, key, is_forward, state, tokens, op_name, pq=None):
        from .host_util import is_valid_token
        tokens = tokens if isinstance(tokens, (list, tuple)) else [tokens]
        if not any(is_valid_token(v) for v in tokens):
            return state
        P = len(tokens)
        if pq is None and P != self.b.Q:
            raise Unsupported("batch size mismatch")
        feats = self.features(tokens, op_name, 0.0)
        if is_forward:
            old = self.expand(state, pq) if pq is not None else state
            new = self.lstm(0, feats, old, P)
            self.fwd[key] = new
            return _AS(state.names, new.h, new.c, P)
        if key not in self.fwd:
            raise Unsupported("a backward step without its forward state")
        self.mods[key] = self.modulations(self.fwd.pop(key), state, P)
        new = self.lstm(1, feats, state, P)
        new = _AS(state.names, new.h, new.c, P)
        return self.squeeze(new, pq) if pq is not None else new

    def relate_ta(self, i, is_forward, s_state, o_state, tokens, op_name, pq=None):
        from .host_util import is_valid_token
        tokens = tokens if isinstance(tokens, (list, tuple)) else [tokens]
        if not any(is_valid_token(v) for v in tokens):
            return s_state, o_state
        P = len(tokens)
        if pq is None and P != self.b.Q:
            raise Unsupported("batch size mismatch")
        feats = self.features(tokens, op_name, 1.0)
        if is_forward:
            both = self.add(s_state, o_state)                     # (the rows of a sum = the sum of the rows: one gather for an option list)
            new = self.lstm(0, feats, self.expand(both, pq) if pq is not None else both, P)
            self.fwd[(i, "rel_s")] = self.fwd[(i, "rel_o")] = new
            return _AS(s_state.names, new.h, new.c, P), _AS(o_state.names, new.h, new.c, P)
        if (i, "rel_s") not in self.fwd:
            raise Unsupported("a backward step without its forward state")
        self.mods[(i, "rel_s")] = self.modulations(self.fwd.pop((i, "rel_s")), s_state, P)
        self.mods[(i, "rel_o")] = self.modulations(self.fwd.pop((i, "rel_o")), o_state, P)
        new = self.lstm(1, feats, self.add(s_state, o_state), P)
        new_s, new_o = _AS(s_state.names, new.h, new.c, P), _AS(o_state.names, new.h, new.c, P)
        if pq is not None:
            sq = self.squeeze(new_s, pq)
            new_s, new_o = sq, _AS(o_state.names, sq.h, sq.c, sq.rows)
        return new_s, new_o

    # -- the operators (gqa_ops.py `_*_ta`; batch_gqa_ops.py) ---------------------------------------------------------------------------------------
    def select_ta(self, i, is_forward, state, tokens, op_name):
        Q = self.b.Q
        if tokens is None:
            names, att = ["entity"] * Q, None
        else:
            names = ["entity" if a is None or a.lower() in ("_", "scene") else a for a in tokens][:Q]
            att = [None if a is None or a.lower() in ("_", "scene") else a for a in tokens][:Q]
        plain = att is None or all(a is None for a in att)
        if is_forward:
            x = self.zero_state(names, Q)
            return x if plain else self.filter_ta((i, "sel"), True, x, att, op_name)
        return state if plain else self.filter_ta((i, "sel"), False, state, att, op_name)

    def gqa_relate_ta(self, i, is_forward, state, relation_list, is_subject, attribute_list, op_name):
        if is_forward:
            x = self.select_ta(i, True, None, attribute_list, op_name)
            s, o = self.relate_ta(i, True, self.gate(x, state, is_subject), self.gate(state, x, is_subject), relation_list, op_name)
            return self.gate(s, o, is_subject)
        x = self.zero_state(state.names, state.rows)
        o_set = self.gate(x, state, is_subject)
        s_set = self.gate(state, x, is_subject)
        s_set, o_set = self.relate_ta(i, False, s_set, o_set, relation_list, op_name)
        self.select_ta(i, False, self.gate(s_set, o_set, is_subject), attribute_list, op_name)
        return self.gate(o_set, s_set, is_subject)

    def choose_rel_ta(self, i, is_forward, state, lists, is_subject, attribute_list, op_name):
        flat, bi = flatten_list(lists)
        host = [0.0 if f is None else float(f) for f in is_subject]
        pred_flags = [host[q] for q in bi]
        if is_forward:
            x = self.select_ta(i, True, None, attribute_list, op_name)
            s, o = self.relate_ta(i, True, self.gate(x, state, is_subject), self.gate(state, x, is_subject), flat, op_name, bi)
            return self.gate(s, o, pred_flags)
        x = self.zero_state(state.names, state.rows)
        o_set = self.gate(x, state, pred_flags)
        s_set = self.gate(state, x, pred_flags)
        s_set, o_set = self.relate_ta(i, False, s_set, o_set, flat, op_name, bi)
        self.select_ta(i, False, self.gate(s_set, o_set, is_subject), attribute_list, op_name)
        return self.gate(o_set, s_set, is_subject)

    def transform(self, i, ob, is_forward, ins):
        """BatchGQAInterpreter._transform_attention (interpreter.py; batch_gqa_interpreter.py:80-86) for operator batch i."""
        name, args = ob._op_name, ob._arguments
        if name == "select":
            return self.select_ta(i, is_forward, ins[0], args[0] if args else None, name)
        if name == "filter":
            return self.filter_ta((i, "flt"), is_forward, ins[0], args[0], name)
        if name in ("relate", "verify_rel"):
            return self.gqa_relate_ta(i, is_forward, ins[0], args[0], args[1], args[2] if len(args) > 2 else None, name)
        if name in ("exist", "end"):
            return ins[0]
        if name in ("verify_attrs", "choose_attr"):
            flat, bi = flatten_list(args[0])
            return self.filter_ta((i, "flt"), is_forward, ins[0], flat, name, bi)
        if name in ("query_attr", "all_same", "all_different"):
            _, (flat, bi) = self.b.category_options(args[0], ins[0].names)
            return self.filter_ta((i, "flt"), is_forward, ins[0], flat, name, bi)
        if name == "choose_rel":
            return self.choose_rel_ta(i, is_forward, ins[0], args[0], args[1], args[2] if len(args) > 2 else None, name)
        if name in ("and", "or"):
            return (ins[0], ins[1])
        if name in ("two_same", "two_different"):
            _, (flat, bi) = self.b.category_options(args[0], ins[0].names)
            return (self.filter_ta((i, "flt:0"), is_forward, ins[0], flat, name, bi), self.filter_ta((i, "flt:1"), is_forward, ins[1], flat, name, bi))
        if name == "compare":
            return (self.filter_ta((i, "flt:0"), is_forward, ins[0], args[0], name), self.filter_ta((i, "flt:1"), is_forward, ins[1], args[0], name))
        raise Unsupported("operator %r" % name)

    def run(self, ops, deps_all):
        """interpreter._calibration_passes (batch_base_interpreter.py:92-140)."""
        from .host_util import reverse_dependencies
        last = len(ops) - 1
        trace = []
        for i, ob in enumerate(ops):
            deps = deps_all[i]
            ins = tuple(trace[d] for d in deps) if deps else (None,)
            if any(isinstance(v, tuple) for v in ins):
                raise Unsupported("an operator that reads a terminal operator's states")
            mask = None if ob._mask is None else ob._mask._host
            x = self.transform(i, ob, True, ins)
            if i < last and ins[0] is not None and mask is not None:
                if isinstance(x, tuple):
                    raise Unsupported("a masked two-branch operator")
                x = self.gate(x, ins[0], mask)
            trace.append(x)
        rev = reverse_dependencies(deps_all)
        final = trace[-1]
        first = tuple(self.zero_state(a.names, a.rows) for a in final) if isinstance(final, tuple) else (self.zero_state(final.names, final.rows),)
        trace = [None] * len(ops)
        for i in reversed(range(len(ops))):
            ob = ops[i]
            if len(rev[i]) == 1:
                temp = trace[rev[i][0]]
                ins = ((temp[1],) if i == len(ops) - 2 else (temp[0],)) if isinstance(temp, tuple) else (temp,)
            else:
                ins = first
            mask = None if ob._mask is None else ob._mask._host
            if ins[0] is None:
                raise Unsupported("a backward step without an incoming state")
            x = self.transform(i, ob, False, ins)
            if len(deps_all[i]) > 0 and mask is not None and isinstance(x, _AS) and i != last:
                x = self.gate(x, ins[0], mask)
            trace[i] = x
        self.finish()


def _attribute_requests(b, ops, deps):
    """Every attribute token list the operators will filter with, keyed (op index, slot): decided before the operators are lowered so that
    one launch evaluates all of them.  Category options depend on variable NAMES, which flow through selects and gates: a names-only
    pre-pass of the program mirrors the walk below."""
    Q, ont = b.Q, b.ont
    reqs, names_of = [], []
    ident = b.ident

    def sel_names(tokens):
        if tokens is None:
            return ["entity"] * Q
        return ["entity" if a is None or a.lower() in ("_", "scene") else a for a in tokens][:Q]

    for i, ob in enumerate(ops):
        name, args, d = ob._op_name, ob._arguments, deps[i]
        mask = None if ob._mask is None else ob._mask._host
        prev = names_of[d[0]] if d else None
        out = prev
        if name == "select":
            toks = args[0] if args else None
            out = sel_names(toks)
            st = b.select_tokens(toks, Q)
            if st is not None and any(a is not None for a in st):
                reqs.append(((i, 0), get_lowered(st, ont, TokenType.ATTRIBUTE), ident))
        elif name == "filter":
            reqs.append(((i, 0), get_lowered(args[0], ont, TokenType.ATTRIBUTE), ident))
        elif name in ("relate", "verify_rel", "choose_rel"):
            toks = args[2] if len(args) > 2 else None
            out = sel_names(toks)
            st = b.select_tokens(toks, Q)
            if st is not None and any(a is not None for a in st):
                reqs.append(((i, 0), get_lowered(st, ont, TokenType.ATTRIBUTE), ident))
        elif name in ("verify_attrs", "choose_attr"):
            flat, bi = flatten_list(args[0])
            reqs.append(((i, 0), lower_tokens(flat, ont, TokenType.ATTRIBUTE), np.asarray(bi, np.int32)))
        elif name in ("query_attr", "all_same", "all_different", "two_same", "two_different"):
            _, (flat, bi) = b.category_options(args[0], prev)
            reqs.append(((i, 0), lower_tokens(flat, ont, TokenType.ATTRIBUTE), np.asarray(bi, np.int32)))
        elif name == "compare":
            reqs.append(((i, 0), get_lowered(args[0], ont, TokenType.ATTRIBUTE), ident))
        if name in ("select", "filter", "relate") and d and mask is not None and prev is not None:
            out = [a if f > 0 else p for a, p, f in zip(out, prev, mask)]
        names_of.append(out)
    return reqs


def _lower(pb, ontology, spec):
    b = _Builder(pb, ontology, spec)
    ops, deps = getattr(pb._op_batch_list, "host", pb._op_batch_list), pb._dependencies      # (a lazily moved batch: the collated operators)
    if not ops:
        raise Unsupported("an empty program")
    b.scene_stage()
    b.relation_stage(ops)
    b.attribute_stage(_attribute_requests(b, ops, deps))
    if spec.calib is not None:
        b.calibration = _Calibration(b, pb)
        b.calibration.run(ops, deps)
    Q = b.Q
    trace, result = [], None
    last = len(ops) - 1
    for i, ob in enumerate(ops):
        name, args = ob._op_name, ob._arguments
        ins = [trace[d] for d in deps[i]]
        mask = None if ob._mask is None else ob._mask._host
        if ob._is_terminal and i != last:
            trace.append(None)                                     # an earlier terminal's result is dropped by the reference's loop too
            continue
        if any(v is None for v in ins):
            raise Unsupported("an operator that reads a terminal operator's result")
        valid = None
        if name == "select":
            x = b.select(args[0] if args else None, (i, 0), mkey=(i, "sel"))
        elif name == "filter":
            x = b.filter(ins[0], args[0], (i, 0), mkey=(i, "flt"))
            valid = get_lowered(args[0], ontology, TokenType.ATTRIBUTE).valid
        elif name == "relate":
            x, low = b.relate(i, ins[0], args[0], args[1], args[2] if len(args) > 2 else None, (i, 0))
            valid = low.valid
        elif name == "exist":
            lp = b.quantify(ins[0], region="o")
            result = dict(kind="binary", type=QuestionType.BINARY, lp=lp, count=Q, options=["no", "yes"], num=ins[0].prev_num + 1)
        elif name == "end":
            lp = b.quantify(ins[0], region="o")
            result = dict(kind="end", type=QuestionType.STATEMENT, lp=lp, count=ins[0].rows, options=[], num=ins[0].prev_num + 1, names=list(ins[0].names))
        elif name == "verify_rel":
            x, _ = b.relate(i, ins[0], args[0], args[1], args[2] if len(args) > 2 else None, (i, 0))
            lp = b.quantify(x, region="o")
            result = dict(kind="binary", type=QuestionType.BINARY, lp=lp, count=Q, options=["no", "yes"], num=x.prev_num + 1)
        elif name == "verify_attrs":
            flat, bi = flatten_list(args[0])
            x = b.filter(ins[0], flat, (i, 0), bi, normalized=False, mkey=(i, "flt"))
            if x is ins[0]:
                raise Unsupported("verify_attrs without attributes")
            summed = b.block(Q)
            b.emit(OP_SEGMENT_SUM_ROWS, x.att, b.seg_off(bi), Q, b.NS, summed)
            y = _VS(ins[0].names, summed, Q, ins[0].quant, None, x.prev_num)
            lp = b.quantify(y, region="o")
            result = dict(kind="binary", type=QuestionType.BINARY, lp=lp, count=Q, options=["no", "yes"], num=y.prev_num + 1)
        elif name in ("choose_attr", "query_attr"):
            if name == "query_attr":
                lists, (flat, bi) = b.category_options(args[0], ins[0].names)
            else:
                lists, (flat, bi) = args[0], flatten_list(args[0])
            x = b.filter(ins[0], flat, (i, 0), bi, mkey=(i, "flt"))
            if x is ins[0]:
                raise Unsupported("an option list without options")
            lp = b.quantify(x, region="o")
            flags = b.find_max(lp, bi, len(flat))
            result = dict(kind="choose", type=QuestionType.QUERY, lp=lp, count=len(flat), flags=flags, flat=flat, batch_index=bi, options=lists, num=x.prev_num + 1)
        elif name == "choose_rel":
            result = _choose_rel(b, i, ins[0], args)
        elif name in ("and", "or"):
            lp1, lp2 = b.quantify(ins[0]), b.quantify(ins[1])
            lp = b.alloc(Q * 4, "o")
            b.emit(OP_LOGIC, LOGIC_AND if name == "and" else LOGIC_OR, lp1, lp2, Q, lp)
            result = dict(kind="binary", type=QuestionType.BINARY, lp=lp, count=Q, options=["no", "yes"], num=ins[0].prev_num + ins[1].prev_num + 2)
        elif name in ("all_same", "all_different"):
            _, (flat, bi) = b.category_options(args[0], ins[0].names)
            x = b.filter(ins[0], flat, (i, 0), bi, mkey=(i, "flt"))
            if x is ins[0]:
                raise Unsupported("a category without options")
            post = b.block(x.rows)
            b.emit(OP_IMPLICATION, ins[0].att, x.att, b.arr(x.pq), x.rows, post)
            temp = _VS(x.names, post, x.rows, np.zeros(x.rows, np.float32), x.pq)
            lp_p = b.quantify(temp)
            same = name == "all_same"
            lp = b.alloc(Q * 4, "o" if same else "t")
            b.emit(OP_SEGMENT_OR, lp_p, b.seg_off(bi), Q, lp, 0 if same else 1)      # (negated next: the reference's fp32 formula)
            if not same:
                lp2 = b.alloc(Q * 4, "o")
                b.emit(OP_LOGIC, LOGIC_NOT, lp, -1, Q, lp2)
                lp = lp2
            result = dict(kind="binary", type=QuestionType.BINARY, lp=lp, count=Q, options=["no", "yes"], num=x.prev_num + 1)
        elif name in ("two_same", "two_different"):
            _, (flat, bi) = b.category_options(args[0], ins[0].names)
            # (the Python operators evaluate and normalise the same blocks twice; once is the same values)
            x1 = b.filter(ins[0], flat, (i, 0), bi, mkey=(i, "flt:0"))
            if x1 is ins[0]:
                raise Unsupported("a category without options")
            spec_norm, b.spec.normalize = b.spec.normalize, False          # already normalised in place by the first filter
            try:
                x2 = b.filter(ins[1], flat, (i, 0), bi, mkey=(i, "flt:1"))
            finally:
                b.spec.normalize = spec_norm
            lp1, lp2 = b.quantify(x1), b.quantify(x2)
            both = b.alloc(x1.rows * 4)
            b.emit(OP_LOGIC, LOGIC_AND, lp1, lp2, x1.rows, both)
            same = name == "two_same"
            lp = b.alloc(Q * 4, "o" if same else "t")
            b.emit(OP_SEGMENT_OR, both, b.seg_off(bi), Q, lp, 0 if same else 1)
            if not same:
                lp2_ = b.alloc(Q * 4, "o")
                b.emit(OP_LOGIC, LOGIC_NOT, lp, -1, Q, lp2_)
                lp = lp2_
            result = dict(kind="binary", type=QuestionType.BINARY, lp=lp, count=Q, options=["no", "yes"], num=x1.prev_num + x2.prev_num + 2)
        elif name == "compare":
            x1 = b.filter(ins[0], args[0], (i, 0), mkey=(i, "flt:0"))
            x2 = b.filter(ins[1], args[0], (i, 0), mkey=(i, "flt:1"))
            lp1, lp2 = b.quantify(x1), b.quantify(x2)
            lp = b.alloc(Q * 2 * 4, "o")
            b.emit(OP_COMPARE, lp1, lp2, b.arr(np.asarray([float(bool(v)) for v in args[1]], np.float32)), Q, lp)
            result = dict(kind="compare", type=QuestionType.QUERY, lp=lp, count=2 * Q, options=list(zip(ins[0].names, ins[1].names)),
                          num=x1.prev_num + x2.prev_num + 2)
        else:
            raise Unsupported("operator %r" % name)
        if result is not None:
            break
        if i == last:                  # a program that ends without a terminal operator: `end` reads the operator's own (un-gated) result
            lp = b.quantify(x, region="o")                         # (batch_gqa_interpreter.py:75-76)
            result = dict(kind="end", type=QuestionType.STATEMENT, lp=lp, count=x.rows, options=[], num=x.prev_num + 1, names=list(x.names))
            break
        if ins and mask is not None:
            x = b.mask_gate(x, ins[0], mask, valid)
        trace.append(x)
    if result is None:
        raise Unsupported("a program without a result")
    return b, result


def _choose_rel(b, i, prev, args):
    """GQAChooseRelBatch.forward (gqa_ops.py; batch_gqa_ops.py:246-267): the generic two-posterior cell over the flattened option list."""
    Q = b.Q
    lists = args[0]
    flat, bi = flatten_list(lists)
    x = b.select(args[2] if len(args) > 2 else None, (i, 0), mkey=(i, "sel"))
    host = [0.0 if f is None else float(f) for f in args[1]]
    subject_set = b.gate(x, prev, host)
    object_set = b.gate(prev, x, host)
    hit = b.tiles.get(i)
    if hit is None or prev.pq is not None or prev.rows != Q:
        raise Unsupported("choose_rel without prefetched tiles")
    tiles, low = hit
    P = len(flat)
    pq = np.asarray(bi, np.int32)
    q_s, q_o = subject_set.quant[pq], object_set.quant[pq]
    pred_host = [host[q] for q in bi]
    want = np.asarray([WANT_SUBJECT if f > 0 else WANT_OBJECT for f in pred_host], np.uint8)
    b._normalize(tiles, low, pq, 2, True)
    ps, po = b.block(P), b.block(P)
    flags = (RELATE_LONE_FORALL_IDENTITY if P == 1 else 0) | RELATE_DIAG_ABSENT
    b.emit(OP_RELATE, subject_set.att, object_set.att, tiles, b.arr(pq), b.arr(q_s), b.arr(q_o), b.arr(low.neg) if low.any_neg else -1,
           -1 if low.all_valid else b.arr(low.valid), b.arr(want), P,
           TILE_SUBJECT_ROWS, flags, ps, po)
    ps = b.modulate(ps, (i, "rel_s"), pq, P)                      # RelateBatch.forward :588-594
    po = b.modulate(po, (i, "rel_o"), pq, P)
    n_prev = subject_set.prev_num + object_set.prev_num + 1
    s_set = _VS(subject_set.names, ps, P, q_s, pq, n_prev)
    o_set = _VS(object_set.names, po, P, q_s, pq, n_prev)
    xx = b.gate(s_set, o_set, pred_host)
    lp = b.quantify(xx, region="o")
    fl = b.find_max(lp, bi, P)
    return dict(kind="choose", type=QuestionType.QUERY, lp=lp, count=P, flags=fl, flat=flat, batch_index=bi, options=lists, num=xx.prev_num + 1)


def build_plan(program_batch, ontology, spec):
    """-> NativePlan, or None when the batch has a shape the executor does not take."""
    try:
        b, result = _lower(program_batch, ontology, spec)
    except Unsupported:
        return None
    plan = NativePlan()
    out_bytes = b._size["o"]
    base = {"o": 0, "t": out_bytes}

    def resolve(v):
        return base[v.region] + v.off if isinstance(v, _W) else int(v)

    if b.calibration is not None:
        b.calibration.fold_runs(resolve)
    plan.instrs = np.asarray([[resolve(v) for v in row] for row in b.instrs], np.int64).reshape(-1, INSTR_WIDTH)
    plan.blob = np.concatenate(b._blob_parts) if b._blob_parts else np.zeros(16, np.uint8)
    plan.out_bytes = out_bytes
    plan.ws_bytes = out_bytes + b._size["t"]
    plan.scene = dict(O=b.O, Q=b.Q, NS=b.NS, max_n=b.max_n, n_obj=b.b_n_obj, img_n_obj=b.b_img_n_obj, obj_off=b.b_obj_off)
    for k in ("lp", "flags"):
        if k in result:
            result[k] = resolve(result[k])
    plan.result = result
    plan.key = spec.key()
    plan.launches = len(b.instrs)
    return plan


# ---- answers from the read-back (the host halves of the terminal operators, gqa_ops.py) ----------------------------------------------------
def decode(plan, out_host, give_answer=True):
    """out_host: uint8 numpy view of the first plan.out_bytes of the workspace -> (answer, answer_log_probability)."""
    r = plan.result
    if not give_answer:
        return [], []
    lp = out_host[r["lp"]:r["lp"] + 4 * r["count"]].view(np.float32)
    kind = r["kind"]
    if kind == "binary":                                           # gqa_ops._binary_answer (e.g. batch_gqa_ops.py:404-407)
        probability = np.exp(lp.astype(np.float32)).tolist()
        answer = [['yes'] if p > 0.5 else ['no'] for p in probability]
        alp = [[math.log(p)] if p > 0.5 else [math.log(1 - p)] for p in probability]
        return answer, alp
    if kind == "end":                                              # batch_gqa_ops.py:768-783
        return [[n] for n in r["names"]], []
    if kind == "choose":                                           # gqa_ops._choose_answer (util.py:59-66)
        flags = out_host[r["flags"]:r["flags"] + r["count"]].tolist()
        return unflatten_list(r["flat"], r["batch_index"], flags), unflatten_list(lp.tolist(), r["batch_index"], flags)
    if kind == "compare":                                          # batch_gqa_ops.py:730-758
        v = lp.reshape(-1, 2)
        ind = v.argmax(1)
        opts = r["options"]
        return [[opts[i][ind[i]]] for i in range(len(opts))], [[float(v[i, ind[i]])] for i in range(len(opts))]
    raise ValueError(kind)
