// The native executor of a lowered ProgramBatch: ONE C-ABI call enqueues every launch of a batch's forward (scene stage + operators).
//
// The reference runs one Python dispatch per operator (batch_base_interpreter.py:145-172 -> batch_gqa_interpreter.py:72-78 -> the operator
// classes of batch_gqa_ops.py), each of which issues a few tensor ops; at MI355X speed the ~35 launches of a fresh 256-question batch
// cost ~55 us of Python each while their kernels take 1.7 - 2.1 ms together (profiles/r04: the device idles for half of the step).
// Here the host lowers a collated ProgramBatch ONCE, in the collate worker, to a table of kernel-level instructions
// (dfol_vqa_amd/native_plan.py) and this function walks the table, calling the very same entry points of this library the Python
// operators call - same kernels, same arguments, so the results are bit-identical to the Python loop's (tests/test_native_gpu.py).
//
// Memory: `blob` holds every small side array of the batch (geometry, columns, negation / validity flags, predicate -> question maps,
// segment offsets, quantifiers, gate flags, pair-kernel requests), uploaded by the caller with ONE copy; `workspace` is one caller-owned
// arena in which the plan has assigned every intermediate (object matrix, hidden activations, likelihood blocks, relation tiles,
// attentions) and, at its start, the results the host reads back (log-probabilities, arg-max flags).  No allocation, no synchronisation,
// no global state: the call only enqueues work on `stream`.
#include <string.h>

#include "dfol_calib.h"
#include "dfol_common.h"

namespace {

// ---- a run of the attention-calibration passes in ONE launch (DFOL_OP_CALIB_WALK) -----------------------------------------------------------------
// The passes (batch_base_interpreter.py:87-140) are a chain of small row-wise steps over [rows, 50] LSTM states - a cell, a gate between two states, a
// sum, the attention-output product - each of which reads what the step before wrote: eight cells and about as many small steps per forward, ~9 us per
// cell launch and ~5 - 7 us per small one with the device otherwise idle.  Every step treats the rows independently, so a workgroup that owns
// LC_ROWS rows can run the whole chain on them with workgroup barriers between the steps (a workgroup's own global writes are visible to it after a
// barrier: its wavefronts share the CU's L1).  The steps come as a table of WALK_* entries in the blob; they call the same device functions as the
// stand-alone kernels (dfol_calib.h), so the results are bit-identical to the launches they replace.
__global__ __launch_bounds__(LC_THREADS) void calib_walk_kernel(const int64_t* __restrict__ table, int n_steps, int rows, char* __restrict__ ws,
                                                                const char* __restrict__ blob, DfolProgramModel m) {
    extern __shared__ __attribute__((aligned(16))) float cw_s[];
    const int r0 = blockIdx.x * LC_ROWS, tid = threadIdx.x;
    auto W = [&](int64_t off) { return off < 0 ? nullptr : reinterpret_cast<float*>(ws + off); };
    for (int i = 0; i < n_steps; ++i) {
        const int64_t* a = table + (int64_t)i * DFOL_INSTR_WIDTH;
        switch (a[0]) {
            case DFOL_WALK_FILL: {       // dst, planes, width, bits
                uint32_t* dst = reinterpret_cast<uint32_t*>(W(a[1]));
                const int planes = (int)a[2], width = (int)a[3];
                for (int e = tid; e < planes * LC_ROWS * width; e += LC_THREADS) {
                    const int pl = e / (LC_ROWS * width), rem = e - pl * LC_ROWS * width, row = r0 + rem / width;
                    if (row < rows) dst[((int64_t)pl * rows + row) * width + rem % width] = (uint32_t)a[4];
                }
                break;
            }
            case DFOL_WALK_SELECT:       // x, y, flags (blob), planes, width, out
            case DFOL_WALK_ADD: {        // x, y, -, planes, width, out
                const float* x = W(a[1]);
                const float* y = W(a[2]);
                const uint8_t* flags = a[0] == DFOL_WALK_SELECT ? reinterpret_cast<const uint8_t*>(blob + a[3]) : nullptr;
                const int planes = (int)a[4], width = (int)a[5];
                float* out = W(a[6]);
                for (int e = tid; e < planes * LC_ROWS * width; e += LC_THREADS) {
                    const int pl = e / (LC_ROWS * width), rem = e - pl * LC_ROWS * width, row = r0 + rem / width;
                    if (row < rows) {
                        const int64_t pr = (int64_t)pl * rows + row, at_ = pr * width + rem % width;
                        out[at_] = flags ? (flags[pr] ? x : y)[at_] : x[at_] + y[at_];
                    }
                }
                break;
            }
            case DFOL_WALK_LSTM: {       // which, h, c, h_out, c_out, head (blob), n_head, table (blob), E, idx (blob)
                const int w = a[1] ? 1 : 0;
                LcCell p{nullptr, 0, m.lstm_kx, W(a[2]), m.lstm_h, W(a[3]), m.lstm_wih_t[w], m.lstm_ld_wih[w], m.lstm_whh_t[w], m.lstm_ld_whh[w], m.lstm_bih[w],
                         m.lstm_bhh[w], rows, m.lstm_h, W(a[4]), W(a[5]), nullptr,
                         LcTokens{reinterpret_cast<const float*>(blob + a[6]), (int)a[7], reinterpret_cast<const float*>(blob + a[8]), (int)a[9],
                                  reinterpret_cast<const int32_t*>(blob + a[10])}};
                lc_stage(p, r0, cw_s);
                __syncthreads();
                lc_wide(p, r0, cw_s, cw_s + LC_ROWS * (m.lstm_kx + m.lstm_h));
                break;
            }
            case DFOL_WALK_ATT_MODULATIONS:  // forward state h, backward state h, out
                if (tid < 256) am_rows(W(a[1]), W(a[2]), m.att_out_w, m.ld_att_out, m.att_out_b, rows, m.lstm_h, m.att_out_n, W(a[3]), r0, tid);
                break;
            default: break;
        }
        __syncthreads();
    }
}

int run_calib_walk(const DfolProgramModel* model, const int64_t* table_dev, int32_t n_steps, int32_t rows, void* workspace, const void* blob, void* stream) {
    DFOL_REQUIRE(n_steps >= 0 && rows >= 0, "calib_walk: bad sizes");
    if (n_steps == 0 || rows == 0) return 0;
    DFOL_REQUIRE(model->lstm_wih_t[0] && model->lstm_whh_t[0] && model->lstm_wih_t[1] && model->lstm_whh_t[1] && model->lstm_h > 0 && model->lstm_kx > 0 &&
                     model->att_out_w && model->att_out_n > 0,
                 "calib_walk: the model has no calibration networks");
    const size_t lds = sizeof(float) * lc_wide_lds_floats(model->lstm_kx, model->lstm_h);
    DFOL_REQUIRE(lds <= 64 * 1024, "calib_walk: input width %d + hidden %d too large for the staging buffer", model->lstm_kx, model->lstm_h);
    hipLaunchKernelGGL(calib_walk_kernel, dim3(dfol_cdiv(rows, LC_ROWS)), dim3(LC_THREADS), lds, (hipStream_t)stream, table_dev, n_steps, rows,
                       static_cast<char*>(workspace), static_cast<const char*>(blob), *model);
    DFOL_LAUNCH_CHECK("calib_walk");
    return 0;
}

inline const void* at(const void* base, int64_t off) { return off < 0 ? nullptr : static_cast<const char*>(base) + off; }
inline void* at(void* base, int64_t off) { return off < 0 ? nullptr : static_cast<char*>(base) + off; }

int run_dense(const DfolDenseLayer& L, const float* X, int64_t ldx, float* Y, int64_t ldy, int32_t M, void* stream) {
    // the dispatch of _lib.linear_act: the split-operand kernels for weights of >= 65536 elements when the input view allows it
    const bool split_ok = L.kind != DFOL_DENSE_F32 && L.packed != nullptr && L.K % 4 == 0 && ldx % 2 == 0 && (reinterpret_cast<uintptr_t>(X) % 8 == 0);
    if (!split_ok) return dfol_linear_act_f32(X, ldx, L.weight, L.ldw, L.bias, Y, ldy, M, L.N, L.K, L.act, stream);
    switch (L.kind) {
        case DFOL_DENSE_F16X2: return dfol_linear_act_h2_f32(X, ldx, L.packed, L.bias, Y, ldy, M, L.N, L.K, L.act, stream);
        case DFOL_DENSE_BF16X3: return dfol_linear_act_split_f32(X, ldx, L.packed, L.bias, Y, ldy, M, L.N, L.K, L.act, stream);
        case DFOL_DENSE_BF16: return dfol_linear_act_bf16_f32(X, ldx, L.packed, L.bias, Y, ldy, M, L.N, L.K, L.act, stream);
    }
    dfol_set_error("run_program: unknown dense kind %d", L.kind);
    return 1;
}

}  // namespace

extern "C" int dfol_run_program(const DfolProgramModel* model, const DfolProgramScene* scene, const int64_t* instr_host, int32_t n_instr,
                                const void* blob, void* workspace, void* stream) {
    DFOL_REQUIRE(model && scene && instr_host && blob && workspace, "run_program: null pointer");
    DFOL_REQUIRE(n_instr >= 0 && scene->NS > 0 && scene->NS % 4 == 0, "run_program: bad sizes (n_instr %d, NS %d)", n_instr, scene->NS);
    const int32_t NS = scene->NS;
    const int32_t* n_obj = static_cast<const int32_t*>(at(blob, scene->n_obj));            // per QUESTION
    const int32_t* img_n_obj = static_cast<const int32_t*>(at(blob, scene->img_n_obj));    // per SCENE (= n_obj without shared scenes)
    const int32_t* obj_off = static_cast<const int32_t*>(at(blob, scene->obj_off));
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int32_t i = 0; i < n_instr; ++i) {
        const int64_t* a = instr_host + static_cast<int64_t>(i) * DFOL_INSTR_WIDTH;
        int rc = 0;
        switch (a[0]) {
            case DFOL_OP_DENSE: {        // set, layer, x source (0 = the raw features, 1 = workspace), x, ldx, y, ldy, M
                const DfolDenseLayer* layers = a[1] == 0 ? model->featurizer : (a[1] == 1 ? model->attribute : &model->uv);
                const int32_t count = a[1] == 0 ? model->n_featurizer : (a[1] == 1 ? model->n_attribute : 1);
                DFOL_REQUIRE(layers && a[2] >= 0 && a[2] < count, "run_program[%d]: dense layer %lld of set %lld does not exist", i, (long long)a[2], (long long)a[1]);
                const float* X = a[3] == 0 ? scene->features + a[4] / 4 : static_cast<const float*>(at(workspace, a[4]));
                rc = run_dense(layers[a[2]], X, a[3] == 0 ? scene->ld_features : a[5], static_cast<float*>(at(workspace, a[6])), a[7], static_cast<int32_t>(a[8]), stream);
                break;
            }
            case DFOL_OP_BOX_POSITIONS:  // obj, ld_obj, pos_col
                rc = dfol_box_positions_f32(scene->features, scene->ld_features, scene->raw_cols, scene->O, static_cast<float*>(at(workspace, a[1])), a[2],
                                            static_cast<int32_t>(a[3]), stream);
                break;
            case DFOL_OP_FILL: {         // dst, count (32-bit words), bit pattern
                if (a[2] > 0) {
                    DFOL_REQUIRE(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(at(workspace, a[1])), static_cast<int>(a[3]), static_cast<size_t>(a[2]), s) == hipSuccess,
                                 "run_program[%d]: hipMemsetD32Async failed", i);
                }
                break;
            }
            case DFOL_OP_PAIR_LL: {      // uv, ld_uv, pos, ld_pos, req_col, req_tile, req_orient, K, tiles, scenes, tile dtype (DFOL_TILE_*)
                const float* uv = static_cast<const float*>(at(workspace, a[1]));
                const float* pos = static_cast<const float*>(at(workspace, a[3]));
                const int32_t* rc_ = static_cast<const int32_t*>(at(blob, a[5]));
                const int32_t* rt = static_cast<const int32_t*>(at(blob, a[6]));
                const uint8_t* ro = static_cast<const uint8_t*>(at(blob, a[7]));
                void* tiles = at(workspace, a[9]);
                const int32_t K = static_cast<int32_t>(a[8]), Qimg = static_cast<int32_t>(a[10]), tdt = static_cast<int32_t>(a[11]);
                DFOL_REQUIRE(tdt == DFOL_TILE_F32 || model->pair_kind != DFOL_PAIR_PLAIN, "run_program[%d]: bf16 tiles need a packed second layer", i);
                switch (model->pair_kind) {
                    case DFOL_PAIR_F16X2:
                        rc = dfol_pair_ll_h2_f32(uv, a[2], model->hid1, pos, a[4], model->wg, model->w2, model->b2, model->hid2, model->emb_w, model->ld_e, model->emb_b,
                                                 img_n_obj, obj_off, Qimg, scene->max_n, rc_, rt, ro, K, NS, -30.0f, tdt, tiles, stream);
                        break;
                    case DFOL_PAIR_BF16X3:
                        rc = dfol_pair_ll_split_f32(uv, a[2], model->hid1, pos, a[4], model->wg, model->w2, model->b2, model->hid2, model->emb_w, model->ld_e, model->emb_b,
                                                    img_n_obj, obj_off, Qimg, scene->max_n, rc_, rt, ro, K, NS, -30.0f, tdt, tiles, stream);
                        break;
                    case DFOL_PAIR_PACKED:
                        rc = dfol_pair_ll_packed_f32(uv, a[2], model->hid1, pos, a[4], model->wg, static_cast<const float*>(model->w2), model->b2, model->hid2, model->emb_w,
                                                     model->ld_e, model->emb_b, img_n_obj, obj_off, Qimg, scene->max_n, rc_, rt, ro, K, NS, -30.0f, tdt, tiles, stream);
                        break;
                    case DFOL_PAIR_PLAIN:
                        rc = dfol_pair_ll_f32(uv, a[2], model->hid1, pos, a[4], model->wg, static_cast<const float*>(model->w2), model->ld_w2, model->w2_rows, model->b2,
                                              model->hid2, model->emb_w, model->ld_e, model->emb_b, img_n_obj, obj_off, Qimg, scene->max_n, rc_, rt, ro, K, NS, -30.0f,
                                              static_cast<float*>(tiles), stream);
                        break;
                    default:
                        dfol_set_error("run_program[%d]: unknown pair kernel kind %d", i, model->pair_kind);
                        return 1;
                }
                break;
            }
            case DFOL_OP_ATTR_LL:        // hidden, ld_hidden, pred_img, cols, P, ll
                rc = dfol_attr_ll_f32(static_cast<const float*>(at(workspace, a[1])), a[2], model->emb_in, model->emb_w, model->ld_e, model->emb_b, obj_off,
                                      static_cast<const int32_t*>(at(blob, a[3])), static_cast<const int32_t*>(at(blob, a[4])), static_cast<int32_t>(a[5]), NS, -30.0f,
                                      static_cast<float*>(at(workspace, a[6])), stream);
                break;
            case DFOL_OP_OPTION_NORMALIZE:   // ll, seg_off, segments, pred_q, rank
                rc = dfol_option_normalize_f32(static_cast<float*>(at(workspace, a[1])), static_cast<const int32_t*>(at(blob, a[2])), static_cast<int32_t>(a[3]),
                                               static_cast<const int32_t*>(at(blob, a[4])), n_obj, NS, static_cast<int32_t>(a[5]), stream);
                break;
            case DFOL_OP_FILTER:         // att_in, ll, pred_q, neg, active, P, out
                rc = dfol_filter_fwd_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const float*>(at(workspace, a[2])),
                                         static_cast<const int32_t*>(at(blob, a[3])), n_obj, static_cast<const uint8_t*>(at(blob, a[4])), a[4] >= 0 ? 1 : 0,
                                         static_cast<const uint8_t*>(at(blob, a[5])), static_cast<int32_t>(a[6]), NS, static_cast<float*>(at(workspace, a[7])), stream);
                break;
            case DFOL_OP_RELATE_ONE:     // x, prev, tile, pred_q, quant_prev, neg, active, P, lone_forall_identity, out, tile dtype (DFOL_TILE_*)
                if (a[11] == DFOL_TILE_BF16) {
                    rc = dfol_relate_one_fwd_bf16(static_cast<const float*>(at(workspace, a[1])), static_cast<const float*>(at(workspace, a[2])),
                                                  static_cast<const uint16_t*>(at(workspace, a[3])), static_cast<const int32_t*>(at(blob, a[4])), n_obj,
                                                  static_cast<const float*>(at(blob, a[5])), static_cast<const uint8_t*>(at(blob, a[6])), a[6] >= 0 ? 1 : 0,
                                                  static_cast<const uint8_t*>(at(blob, a[7])), static_cast<int32_t>(a[8]), NS, static_cast<int32_t>(a[9]),
                                                  static_cast<float*>(at(workspace, a[10])), stream);
                    break;
                }
                rc = dfol_relate_one_fwd_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const float*>(at(workspace, a[2])),
                                             static_cast<const float*>(at(workspace, a[3])), static_cast<const int32_t*>(at(blob, a[4])), n_obj,
                                             static_cast<const float*>(at(blob, a[5])), static_cast<const uint8_t*>(at(blob, a[6])), a[6] >= 0 ? 1 : 0,
                                             static_cast<const uint8_t*>(at(blob, a[7])), static_cast<int32_t>(a[8]), NS, static_cast<int32_t>(a[9]),
                                             static_cast<float*>(at(workspace, a[10])), stream);
                break;
            case DFOL_OP_RELATE:         // prior_s, prior_o, tile, pred_q, quant_s, quant_o, neg, active, want, P, orientation, flags, post_s, post_o
                rc = dfol_relate_fwd_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const float*>(at(workspace, a[2])),
                                         static_cast<const float*>(at(workspace, a[3])), static_cast<const int32_t*>(at(blob, a[4])), n_obj,
                                         static_cast<const float*>(at(blob, a[5])), static_cast<const float*>(at(blob, a[6])), static_cast<const uint8_t*>(at(blob, a[7])),
                                         a[7] >= 0 ? 1 : 0, static_cast<const uint8_t*>(at(blob, a[8])), static_cast<const uint8_t*>(at(blob, a[9])), static_cast<int32_t>(a[10]), NS,
                                         static_cast<int32_t>(a[11]), static_cast<int32_t>(a[12]), static_cast<float*>(at(workspace, a[13])), static_cast<float*>(at(workspace, a[14])),
                                         stream);
                break;
            case DFOL_OP_QUANTIFY:       // att, quant, pred_q, P, lp
                rc = dfol_quantify_fwd_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const float*>(at(blob, a[2])),
                                           static_cast<const int32_t*>(at(blob, a[3])), n_obj, static_cast<int32_t>(a[4]), NS, static_cast<float*>(at(workspace, a[5])), stream);
                break;
            case DFOL_OP_GATE:           // x, y, x_quant, y_quant, g, P, out, out_quant
                rc = dfol_gate_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const float*>(at(workspace, a[2])), static_cast<const float*>(at(blob, a[3])),
                                   static_cast<const float*>(at(blob, a[4])), static_cast<const float*>(at(blob, a[5])), static_cast<int32_t>(a[6]), NS,
                                   static_cast<float*>(at(workspace, a[7])), static_cast<float*>(at(workspace, a[8])), stream);
                break;
            case DFOL_OP_LOGIC:          // op, a, b, n, out
                rc = dfol_logic_f32(static_cast<int32_t>(a[1]), static_cast<const float*>(at(workspace, a[2])), static_cast<const float*>(at(workspace, a[3])), a[4],
                                    static_cast<float*>(at(workspace, a[5])), stream);
                break;
            case DFOL_OP_SEGMENT_SUM_ROWS:   // src, seg_off, Q, width, out
                rc = dfol_segment_sum_rows_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const int32_t*>(at(blob, a[2])), static_cast<int32_t>(a[3]),
                                               static_cast<int32_t>(a[4]), static_cast<float*>(at(workspace, a[5])), stream);
                break;
            case DFOL_OP_SEGMENT_OR:     // lp, seg_off, Q, out, as the reference writes it (1) or in the complement form (0)
                rc = (a[5] ? dfol_segment_or_ref_f32 : dfol_segment_or_f32)(static_cast<const float*>(at(workspace, a[1])), static_cast<const int32_t*>(at(blob, a[2])),
                                                                            static_cast<int32_t>(a[3]), static_cast<float*>(at(workspace, a[4])), stream);
                break;
            case DFOL_OP_IMPLICATION:    // prior, x, pred_q, P, out
                rc = dfol_implication_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const float*>(at(workspace, a[2])),
                                          static_cast<const int32_t*>(at(blob, a[3])), n_obj, static_cast<int32_t>(a[4]), NS, static_cast<float*>(at(workspace, a[5])), stream);
                break;
            case DFOL_OP_COMPARE:        // lp1, lp2, is_less, Q, out
                rc = dfol_compare_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const float*>(at(workspace, a[2])), static_cast<const float*>(at(blob, a[3])),
                                      static_cast<int32_t>(a[4]), static_cast<float*>(at(workspace, a[5])), stream);
                break;
            case DFOL_OP_FIND_MAX_IND: { // lp, seg_off, Q, threshold (float bits), flags
                float thr;
                const int32_t bits = static_cast<int32_t>(a[4]);
                memcpy(&thr, &bits, 4);
                rc = dfol_find_max_ind_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const int32_t*>(at(blob, a[2])), static_cast<int32_t>(a[3]), thr,
                                           static_cast<uint8_t*>(at(workspace, a[5])), stream);
                break;
            }
            case DFOL_OP_GATHER_TILES:   // src rows, index, count, dst rows, width in 32-bit words (0: NS * NS): dst[p] = src[index[p]] - shared scenes (one tile per distinct
                                         // (scene, concept, orientation); bf16 tiles are NS * NS / 2 words), compacted option lists, LSTM states by question
                rc = dfol_gather_rows_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const int32_t*>(at(blob, a[2])), static_cast<int32_t>(a[3]),
                                          a[5] > 0 ? static_cast<int32_t>(a[5]) : NS * NS, static_cast<float*>(at(workspace, a[4])), stream);
                break;
            case DFOL_OP_CALIB_FEATURES: // head, n_head, table, E, idx, P, out
                rc = dfol_calib_features_f32(static_cast<const float*>(at(blob, a[1])), static_cast<int32_t>(a[2]), static_cast<const float*>(at(blob, a[3])),
                                             static_cast<int32_t>(a[4]), static_cast<const int32_t*>(at(blob, a[5])), static_cast<int32_t>(a[6]),
                                             static_cast<float*>(at(workspace, a[7])), stream);
                break;
            case DFOL_OP_LSTM_CELL: {    // which, x (-1: built from tokens), h, c, rows, h_out, c_out; then head, n_head, table, E, idx
                const int w = a[1] ? 1 : 0;
                DFOL_REQUIRE(model->lstm_wih_t[w] && model->lstm_whh_t[w] && model->lstm_h > 0 && model->lstm_kx > 0, "run_program[%d]: the model has no calibration LSTM", i);
                if (a[2] < 0) {
                    DFOL_REQUIRE(a[9] + a[11] == model->lstm_kx, "run_program[%d]: token rows of %lld floats for an LSTM of input width %d", i,
                                 (long long)(a[9] + a[11]), model->lstm_kx);
                    rc = dfol_lstm_cell_tokens_f32(static_cast<const float*>(at(blob, a[8])), static_cast<int32_t>(a[9]), static_cast<const float*>(at(blob, a[10])),
                                                   static_cast<int32_t>(a[11]), static_cast<const int32_t*>(at(blob, a[12])),
                                                   static_cast<const float*>(at(workspace, a[3])), model->lstm_h, static_cast<const float*>(at(workspace, a[4])),
                                                   model->lstm_wih_t[w], model->lstm_ld_wih[w], model->lstm_whh_t[w], model->lstm_ld_whh[w], model->lstm_bih[w],
                                                   model->lstm_bhh[w], static_cast<int32_t>(a[5]), model->lstm_h, static_cast<float*>(at(workspace, a[6])),
                                                   static_cast<float*>(at(workspace, a[7])), stream);
                    break;
                }
                rc = dfol_lstm_cell_f32(static_cast<const float*>(at(workspace, a[2])), model->lstm_kx, model->lstm_kx, static_cast<const float*>(at(workspace, a[3])),
                                        model->lstm_h, static_cast<const float*>(at(workspace, a[4])), model->lstm_wih_t[w], model->lstm_ld_wih[w], model->lstm_whh_t[w],
                                        model->lstm_ld_whh[w], model->lstm_bih[w], model->lstm_bhh[w], static_cast<int32_t>(a[5]), model->lstm_h,
                                        static_cast<float*>(at(workspace, a[6])), static_cast<float*>(at(workspace, a[7])), stream);
                break;
            }
            case DFOL_OP_SELECT_ROWS:    // x, y, flags, P, width, out
                rc = dfol_select_rows_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const float*>(at(workspace, a[2])),
                                          static_cast<const uint8_t*>(at(blob, a[3])), static_cast<int32_t>(a[4]), static_cast<int32_t>(a[5]),
                                          static_cast<float*>(at(workspace, a[6])), stream);
                break;
            case DFOL_OP_ATT_MODULATIONS: // fs, bs, P, out
                DFOL_REQUIRE(model->att_out_w && model->att_out_n > 0, "run_program[%d]: the model has no attention-output network", i);
                rc = dfol_attention_modulations_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const float*>(at(workspace, a[2])), model->att_out_w,
                                                    model->ld_att_out, model->att_out_b, static_cast<int32_t>(a[3]), model->lstm_h, model->att_out_n,
                                                    static_cast<float*>(at(workspace, a[4])), stream);
                break;
            case DFOL_OP_CALIB_WALK:     // table of DFOL_WALK_* steps (blob), steps, rows
                rc = run_calib_walk(model, static_cast<const int64_t*>(at(blob, a[1])), static_cast<int32_t>(a[2]), static_cast<int32_t>(a[3]), workspace, blob, stream);
                break;
            case DFOL_OP_MODULATE:       // att, mods, pred_q, P, out
                DFOL_REQUIRE(model->att_out_n == 4, "run_program[%d]: dfol_modulate_f32 takes 4-column modulations", i);
                rc = dfol_modulate_f32(static_cast<const float*>(at(workspace, a[1])), static_cast<const float*>(at(workspace, a[2])),
                                       static_cast<const int32_t*>(at(blob, a[3])), n_obj, static_cast<int32_t>(a[4]), NS, static_cast<float*>(at(workspace, a[5])), stream);
                break;
            default:
                dfol_set_error("run_program[%d]: unknown opcode %lld", i, (long long)a[0]);
                return 1;
        }
        if (rc != 0) {
            char msg[480];
            snprintf(msg, sizeof(msg), "%s", dfol_last_error());
            dfol_set_error("run_program: instruction %d (opcode %lld): %s", i, (long long)a[0], msg);
            return rc;
        }
    }
    return 0;
}
