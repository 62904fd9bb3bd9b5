/*
 * dfol_vqa.h — C ABI of the MI355X-native ∇-FOL (DFOL-VQA) interpreter hot path.
 *
 * The reference (microsoft/DFOL-VQA) is pure Python/PyTorch: it has no FFI of its own for this
 * path.  Each entry point below therefore names the reference *Python* interface it replaces
 * (file:line under /root/reference/src/nsvqa), and INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add at that line to call it.
 *
 * Conventions
 *  - All pointers are DEVICE pointers into memory the caller owns, unless the name ends in `_host`.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls only enqueue work.
 *  - Return value: 0 on success; nonzero on error, with a message in dfol_last_error()
 *    (thread-local).  Nothing is written on an argument error.
 *  - BLOCK LAYOUT.  The reference keeps one flat attention row over all objects of a ProgramBatch
 *    ([P, total_obj], and [P, total_obj, total_obj] relation likelihoods with every cross-image
 *    entry at the default -30).  Here every predicate p owns one block:
 *        attention / attribute likelihood : float [P, NS]        (object o of image q at column o)
 *        relation likelihood tile         : float [P, NS, NS]    (tile[p][r][c])
 *    with NS >= max objects per image the row stride (NS % 4 == 0), n_obj[q] the true object count
 *    of image q and pred_q[p] the question/image that predicate p belongs to.  Columns >= n_obj
 *    are padding: never read, written as 0 (log 1) by the kernels that produce attention.
 *    The arity-2 forward kernels (dfol_relate_fwd_f32, dfol_relate_one_fwd_f32) keep a tile row in one wavefront's registers up to
 *    NS = 256 (64 lanes x 4 columns) and fall back to a plain one-workgroup-per-predicate kernel above that (same formulas, no fast
 *    paths); dfol_relate_bwd_f32 takes NS <= 2048 (7 NS floats of LDS).  LIMIT: dfol_relate_one_fwd_bf16 takes NS <= 256 only and
 *    rejects larger NS with an error.  GQA scenes have at most 100 objects (BASELINE.json configs[2]); the largest configuration,
 *    configs[4], has 256.
 *  - Log-space constants are the reference's: absent likelihood -30 (batch_base_ops.py:154),
 *    floor log(1e-20) (util.py:22-25), quantifier 1 = EXISTS / 0 = FOR_ALL (batch_base_types.py:15-17).
 *  - Precision: fp32 arithmetic throughout ("f32" suffix).
 */
#ifndef DFOL_VQA_H
#define DFOL_VQA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 6): dfol_pair_ll_h2_f32 takes UV in units of ln 2 and dfol_pair_pack_w2_f16x2 folds ln 2 into W2 (the round-5 change of contract that
 * kept version 2: a caller written against 2 passes unscaled UV and must be refused, not answered wrongly); dfol_pair_ll_h2_f32 reports saturated
 * ELU outputs through the dfol_set_range_status word (DFOL_RANGE_PAIR_SATURATED); dfol_run_program knows the calibration instructions. */
#define DFOL_ABI_VERSION 3

/* ---- library ---------------------------------------------------------------------------------- */
int dfol_abi_version(void);
const char* dfol_last_error(void);

/* tile orientation for relation likelihoods */
#define DFOL_TILE_SUBJECT_ROWS 0 /* tile[p][s][o]  (the reference's [P, O(subject), O(object)]) */
#define DFOL_TILE_OBJECT_ROWS 1  /* tile[p][o][s]  (transposed) */

/* which posteriors dfol_relate_fwd_f32 must produce, per predicate (bit mask) */
#define DFOL_RELATE_LONE_FORALL_IDENTITY 1 /* flags of dfol_relate_fwd_f32 */
#define DFOL_RELATE_DIAG_ABSENT 2
#define DFOL_TILE_F32 0  /* element type of relation tiles written by dfol_pair_ll_packed_f32 */
#define DFOL_TILE_BF16 1
#define DFOL_WANT_SUBJECT 1
#define DFOL_WANT_OBJECT 2

/* activations for dfol_linear_act_f32 */
#define DFOL_ACT_NONE 0
#define DFOL_ACT_SIGMOID 1    /* nn.Sigmoid     */
#define DFOL_ACT_ELU 2        /* nn.ELU(alpha=1) */
#define DFOL_ACT_LOGSIGMOID 3 /* nn.LogSigmoid  */

/* binary / unary log-space ops for dfol_logic_*_f32 */
#define DFOL_LOGIC_AND 0 /* util.py:29-30 */
#define DFOL_LOGIC_OR 1  /* util.py:32-33 */
#define DFOL_LOGIC_NOT 2 /* util.py:35-36 */

/* ---- oracle gathers (cached tables) -------------------------------------------------------------
 * Replaces ClassifierOracle._compute_attribute_log_likelihood, classifier_oracle.py:44-82
 * (cached=True branch :62-63 and the dense reshape :78-80), without the option normalisation.
 *   table    [O, ld_table]   cached attribute log-likelihood table (world._attribute_features)
 *   obj_off  [Q+1]           first object row of every image (prefix sum of n_obj)
 *   pred_q   [P]             image of predicate p            (attribute_image_map)
 *   pred_col [P]             column = arg_to_idx[token]-1 ; < 0 marks a no-op token (None / '_'),
 *                            whose block is filled with default_ll (batch_base_ops.py:364)
 *   ll       [P, NS]   out   ll[p][o] = table[obj_off[q]+o][col] for o < n_obj[q]; padding = default_ll
 */
int dfol_attr_gather_f32(const float* table, int64_t ld_table, const int32_t* obj_off, const int32_t* pred_q,
                         const int32_t* pred_col, int32_t P, int32_t NS, float default_ll, float* ll, void* stream);

/* Replaces ClassifierOracle._compute_relation_log_likelihood, classifier_oracle.py:84-137
 * (cached=True, the non-`relation_pairobject_map` branch :113-135), without the option normalisation.
 *   table    [pairs, ld_table]  cached relation table; pairs of image q start at pair_off[q] and are
 *                               the ordered (s,o), s != o pairs, row-major in s (util.py:87-103)
 *   pair_off [Q+1]              int64 prefix sum of n(n-1)
 *   pred_col [P]                column in the 333-wide relation table (_relation_reveresed_index); <0 = no-op
 *   tile     [P, NS, NS]  out   tile[p][s][o] (or [o][s] if orientation == DFOL_TILE_OBJECT_ROWS);
 *                               diagonal, padding and no-op blocks = default_ll
 */
int dfol_rel_gather_f32(const float* table, int64_t ld_table, const int64_t* pair_off, const int32_t* n_obj,
                        const int32_t* pred_q, const int32_t* pred_col, int32_t P, int32_t NS, int32_t orientation,
                        float default_ll, float* tile, void* stream);

/* Option normalisation: classifier_oracle.py:72-75 and :124-127 with _build_map :22-42.
 * A segment is a run of consecutive predicates of one question (torch.unique_consecutive of the image map).
 *   ll[p][e] -= log(max(sum_{p' in seg(p)} exp(ll[p'][e]), 1e-20))   for every real entry e of the block
 *   seg_off [S+1]  predicate range of every segment;   rank: 1 = [P,NS] blocks, 2 = [P,NS,NS] tiles
 * For rank 2 only real pairs (r != c, both < n_obj) are touched (the reference normalises over pairs,
 * classifier_oracle.py:121-127, and fills the diagonal afterwards).  In place.
 */
int dfol_option_normalize_f32(float* ll, const int32_t* seg_off, int32_t S, const int32_t* pred_q, const int32_t* n_obj,
                              int32_t NS, int32_t rank, void* stream);

/* ---- the logic cell -----------------------------------------------------------------------------
 * Arity 1: replaces BatchBayesianLogicCell.forward/_forward_core for FilterBatch,
 * batch_base_ops.py:153-215, 62-151 (ll clamp :194, negation :212-213, log-AND with the prior :138)
 * and the no-op row restore of FilterBatch.forward :385.
 *   att_in   [Q, NS]  prior attention, row pred_q[p] is read for predicate p
 *   ll       [P, NS]  raw likelihood block
 *   neg      [P]      1 = predicate is negated ("not(x)"); may be NULL when any_neg == 0
 *   any_neg           the reference applies log_parametric_not(ll, is_negated, 1) to EVERY predicate
 *                     of an op batch as soon as one is negated (:212-213, :376-380)
 *   active   [P]      0 = no-op token: the output row is the prior row (:385); may be NULL (all active)
 *   att_out  [P, NS]  out; padding columns (>= n_obj) are written as 0
 */
int dfol_filter_fwd_f32(const float* att_in, const float* ll, const int32_t* pred_q, const int32_t* n_obj,
                        const uint8_t* neg, int32_t any_neg, const uint8_t* active, int32_t P, int32_t NS,
                        float* att_out, void* stream);

/* Arity 2: replaces BatchBayesianLogicCell.forward/_forward_core for RelateBatch,
 * batch_base_ops.py:153-215, 62-151, and the no-op row restore of RelateBatch.forward :563-564.
 * Per predicate, with l' = negation(min(tile,0)), E the off-diagonal:
 *   post_s[s] = prior_s[s] + F_o( sum_{o != s} F_o(l'[s,o] + prior_o[o]) )
 *   post_o[o] = prior_o[o] + F_s( sum_{s != o} F_s(l'[s,o] + prior_s[s]) )
 *   F_x(v) = log(max(q_x + (1-2 q_x) e^v, 1e-20)),  q_x the quantifier of the variable summed out.
 *   prior_s, prior_o [Q, NS]   row pred_q[p] is read
 *   tile             [P, NS, NS] in `orientation`
 *   quant_s, quant_o [P]       quantifiers (float 0/1) of the subject / object variable
 *   want             [P]       DFOL_WANT_* bits; an unwanted posterior row is written as zeros.  NULL = both.
 *   post_s, post_o   [P, NS]   out (either may be NULL if no predicate wants it)
 *   flags                      DFOL_RELATE_LONE_FORALL_IDENTITY reproduces the reference's single-predicate literal branch
 *                              (:104-108,:129-133: P == 1 and FOR_ALL leaves the value untouched);
 *                              DFOL_RELATE_DIAG_ABSENT promises that every tile's diagonal holds the absent likelihood
 *                              (<= -30, what dfol_rel_gather_f32 / dfol_pair_ll_f32 write and option normalisation leaves
 *                              alone); since round 3 it is a hint only: every path drops self-relations explicitly
 *                              (batch_base_ops.py:112).  Results are the same either way.
 *                              EXISTS aggregations are evaluated in the complement form q <- q + y - q y (csrc/dfol_common.h,
 *                              dfol_or): the float64 value of the reference's log(1 - prod(1 - y)) to ~1e-6, without the
 *                              cancellation noise the formula has when evaluated as written in fp32.
 */
int dfol_relate_fwd_f32(const float* prior_s, const float* prior_o, const float* tile, const int32_t* pred_q,
                        const int32_t* n_obj, const float* quant_s, const float* quant_o, const uint8_t* neg,
                        int32_t any_neg, const uint8_t* active, const uint8_t* want, int32_t P, int32_t NS,
                        int32_t orientation, int32_t flags, float* post_s, float* post_o, void* stream);

/* Relate with ONE posterior: what GQARelateBatch / verify_rel / choose_rel keep (batch_gqa_ops.py:364-371: gate x/prev into
 * subject/object, RelateBatch, gate the wanted posterior back).  Fuses the three gates and the arity-2 cell:
 *     post[p][c] = x_att[p][c] + F( sum_{r != c} F( l'[r][c] + prev_att[pred_q[p]][r] ) ),  F by quant_prev[p]
 *   x_att    [P, NS]  attention of the freshly selected variable (select(name) of the relate), one row per predicate
 *   prev_att [Q, NS]  incoming attention of the other variable
 *   tile     [P, NS, NS] with the SUMMED-OUT variable (prev's) along rows: DFOL_TILE_OBJECT_ROWS for is_subject
 *            predicates, DFOL_TILE_SUBJECT_ROWS otherwise (dfol_pair_ll_f32 / dfol_rel_gather_f32 write either)
 *            Self-relations (the diagonal) are dropped explicitly, whatever the tile holds there (batch_base_ops.py:112).
 *   active[p] == 0: post[p] = prev_att row (the interpreter's pass-through for questions lacking the operator,
 *            batch_base_interpreter.py:166-167)
 */
int dfol_relate_one_fwd_f32(const float* x_att, const float* prev_att, const float* tile, const int32_t* pred_q,
                            const int32_t* n_obj, const float* quant_prev, const uint8_t* neg, int32_t any_neg,
                            const uint8_t* active, int32_t P, int32_t NS, int32_t lone_forall_identity, float* post, void* stream);

/* The same operator on bf16 tiles (BASELINE configs[4]: 256-object scenes; the stored likelihoods are rounded to bf16, which halves
 * the HBM bytes of the tile stream; arithmetic stays fp32).  tile [P, NS, NS] of bf16 bit patterns, NS a multiple of 8, written by
 * dfol_pair_ll_packed_f32 with tile_dtype = DFOL_TILE_BF16.  An opt-in storage mode, not the reference's numerics: results differ
 * from the fp32 path by the rounding of the likelihoods (relative 2^-9).
 */
int dfol_relate_one_fwd_bf16(const float* x_att, const float* prev_att, const uint16_t* tile, const int32_t* pred_q,
                             const int32_t* n_obj, const float* quant_prev, const uint8_t* neg, int32_t any_neg,
                             const uint8_t* active, int32_t P, int32_t NS, int32_t lone_forall_identity, float* post, void* stream);

/* Soft quantifier aggregation: replaces BatchVariableSet.log_probability (soft mode),
 * batch_base_types.py:113-123:   lp[p] = F_q( sum_{o < n} F_q(att[p][o]) ),  q = quant[p].
 */
int dfol_quantify_fwd_f32(const float* att, const float* quant, const int32_t* pred_q, const int32_t* n_obj,
                          int32_t P, int32_t NS, float* lp, void* stream);

/* util.find_max_ind (util.py:64-66): flag[p] = 1 iff predicate p attains the maximum probability exp(lp) among the predicates of
 * its question (contiguous: seg_off [Q+1]) and that probability exceeds likelihood_threshold.  Answer decoding of the choose / query
 * operators (batch_gqa_ops.py:215-228, 246-267, 304-306) without a host round trip per operator. */
int dfol_find_max_ind_f32(const float* lp, const int32_t* seg_off, int32_t Q, float likelihood_threshold, uint8_t* flag, void* stream);

/* hard_mode aggregation: replaces BatchVariableSet.log_probability(hard_mode=True), batch_base_types.py:104-112 (a test-time
 * option of BatchGQAInterpreter, :23,:73):   lp[p] = F_q( min( min_{o < n} F_q(att[p][o]), 0 if total_obj > n ) ).
 * The 0 is the reference's product with the dense batch-object mask: objects of the batch's other images take part in the
 * minimum as 0.  total_obj = number of objects of the whole ProgramBatch.  Forward only (the reference uses it with give_answer).
 */
int dfol_quantify_hard_f32(const float* att, const float* quant, const int32_t* pred_q, const int32_t* n_obj, int32_t P,
                           int32_t NS, int32_t total_obj, float* lp, void* stream);

/* Per-question select between two variable sets: replaces BatchVariableSet.gate,
 * batch_base_types.py:149-168.   out = g*x + (1-g)*y  on attention rows and quantifiers, g in {0,1}.
 *   x_att, y_att [P, NS]; x_quant, y_quant [P]; g [P] (float, as the reference's mask)
 */
int dfol_gate_f32(const float* x_att, const float* y_att, const float* x_quant, const float* y_quant, const float* g,
                  int32_t P, int32_t NS, float* out_att, float* out_quant, void* stream);

/* Row gather: out[p][:] = src[idx[p]][:]  (the reference's mm(predicate_question_map, X), e.g.
 * batch_base_ops.py:75,343 and batch_gqa_ops.py:588) for rows of `width` floats. */
int dfol_gather_rows_f32(const float* src, const int32_t* idx, int32_t P, int32_t width, float* out, void* stream);

/* Segmented row sum: out[q][:] = sum_{p in seg q} src[p][:]   (mm(pqm^T, X), batch_gqa_ops.py:457). */
int dfol_segment_sum_rows_f32(const float* src, const int32_t* seg_off, int32_t Q, int32_t width, float* out,
                              void* stream);
/* Per-predicate gradient rows of the embedding layer combined per concept (visual_oracle._combine_concept_rows: gather_rows +
 * segment_sum_rows + index_copy in one launch): out[ucols[u]][c] (accumulate: +)= sum over the slots k in [seg_off[u], seg_off[u + 1]) of
 * rows[order[k]][c], added in slot order (predicates naming the same concept in predicate order: repeatable bit for bit, no atomics);
 * ucols unique.  With accumulate = 1 and `out` the weight's gradient itself, the dense zero-filled intermediate never exists. */
int dfol_concept_rows_f32(const float* rows, int64_t ld_rows, const int32_t* order, const int32_t* seg_off, const int64_t* ucols, int32_t U,
                          int32_t width, float* out, int64_t ld_out, int32_t accumulate, void* stream);

/* Elementwise log-space logic on vectors: util.py:29-36.  b is ignored for DFOL_LOGIC_NOT. */
int dfol_logic_f32(int32_t op, const float* a, const float* b, int64_t n, float* out, void* stream);

/* log_parametric_not(x, alpha, 1) with a per-row alpha (util.py:46-47): x [rows, width], alpha [rows]. */
int dfol_parametric_not_f32(const float* x, const float* alpha, int32_t rows, int32_t width, float* out, void* stream);

/* Segmented log-OR: out[q] = log_not( sum_{p in seg q} log_not(lp[p]) ), batch_gqa_ops.py:597-598, :664-665. */
int dfol_segment_or_f32(const float* lp, const int32_t* seg_off, int32_t Q, float* out, void* stream);
/* The same aggregate evaluated as the reference writes it - log_not(sum log_not(lp_p)) in fp32, not the complement form - for the two
 * operators that negate it next (all_different batch_gqa_ops.py:631, two_different :706): where the aggregate approaches log 1 the
 * reference's arithmetic saturates to exactly 0 (and its negation to log 1e-20 with zero gradient); see csrc/dfol_logic.hip. */
int dfol_segment_or_ref_f32(const float* lp, const int32_t* seg_off, int32_t Q, float* out, void* stream);

/* all_same implication, batch_gqa_ops.py:588-589:  out = log_not(prior[pred_q[p]] + log_not(x))  on [P, NS]. */
int dfol_implication_f32(const float* prior, const float* x, const int32_t* pred_q, const int32_t* n_obj, int32_t P,
                         int32_t NS, float* out, void* stream);

/* compare, batch_gqa_ops.py:734-738: out[q][0..1] = log_parametric_not(log_softmax([lp1[q], lp2[q]]), is_less[q], 1) */
int dfol_compare_f32(const float* lp1, const float* lp2, const float* is_less, int32_t Q, float* out, void* stream);

/* ---- dense contractions (MFMA) ------------------------------------------------------------------
 * Y = act(X W^T + b): replaces the nn.Linear + activation stages of RegularMLP / EmbeddingLayer,
 * gqa_interpreter_experiments.py:26-33, 73-74.  Exact fp32 (v_mfma_f32_32x32x2_f32).
 *   X [M, ldx] (K used), W [N, ldw] (torch Linear layout), bias [N] or NULL, Y [M, ldy]
 */
int dfol_linear_act_f32(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y,
                        int64_t ldy, int32_t M, int32_t N, int32_t K, int32_t act, void* stream);

/* The same contraction with fp32 results from the bf16 matrix pipes (dfol_vqa_amd/csrc/dfol_dense_split.hip): each fp32 operand is
 * cut exactly into three bf16 pieces and six of the nine piece products are accumulated in fp32 (v_mfma_f32_16x16x32_bf16); the
 * dropped products are below 2^-23 |x w|, so results agree with dfol_linear_act_f32 to fp32 rounding.  W_split is produced once
 * per weight version by dfol_linear_pack_w_bf16x3: ceil(N/128) * ceil(K/32) * 24576 bytes, 16-byte aligned.
 * Limits: K % 4 == 0, ldx % 2 == 0, X 8-byte aligned (16-byte aligned rows take the wider loads).
 */
int dfol_linear_pack_w_bf16x3(const float* W, int64_t ldw, int32_t N, int32_t K, void* W_split, void* stream);
int dfol_linear_act_split_f32(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy, int32_t M,
                              int32_t N, int32_t K, int32_t act, void* stream);

/* The same contraction with fp32 results from the fp16 matrix pipe and THREE products per fp32 product (round 4; the default of the
 * FORWARD products): each operand is cut into two fp16 pieces x = h + l (22-23 significand bits), xl wh + xh wl + xh wh accumulated
 * in fp32 (v_mfma_f32_16x16x32_f16).  Rows of W are scaled by powers of two chosen by the pack kernel and un-scaled in the epilogue
 * (exact); X is split UNSCALED: an element's error is max(2^-22 |x|, 2^-25), i.e. fp32-class for operands of order one (features,
 * Sigmoid / ELU outputs) - use dfol_linear_act_split_f32 for operands of arbitrary magnitude (gradients); |x| > 65504 gives NaN.
 * W_split: dfol_linear_w_f16x2_bytes(N, K) bytes from dfol_linear_pack_w_f16x2, 16-byte aligned.  Limits as dfol_linear_act_split_f32.
 * Replaces the nn.Linear + activation lines of gqa_interpreter_experiments.py:18-36, 60-77 and batch_gqa_boxfeatures_pipeline.py:199-213. */
int64_t dfol_linear_w_f16x2_bytes(int32_t N, int32_t K);
int dfol_linear_pack_w_f16x2(const float* W, int64_t ldw, int32_t N, int32_t K, void* W_split, void* stream);
int dfol_linear_act_h2_f32(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy, int32_t M,
                           int32_t N, int32_t K, int32_t act, void* stream);
/* The same product for WIDE outputs (256 < N <= 512: the featurizer's 2048 -> 512, the pair MLP's stacked first layer 516 -> 512) as ONE
 * persistent workgroup per CU that owns 128 rows x all columns (csrc/dfol_dense_wide.hip): X is fetched and split into its pieces once
 * instead of once per 128-column block.  Same image, same limits on X (8-byte aligned rows, ldx % 2 == 0, K % 4 == 0; K >= 128), results
 * bit for bit those of dfol_linear_act_h2_f32 - which forwards here by itself when dfol_linear_wide_supported says the shape pays (four
 * column blocks: N > 384, and 128-row blocks that fill at least three quarters of the CUs in every round; DFOL_DENSE_WIDE=0 never, =2
 * whenever the shape is taken). */
int dfol_linear_wide_supported(int64_t M, int32_t N, int32_t K);
int dfol_linear_wide_h2_f32(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy, int32_t M,
                            int32_t N, int32_t K, int32_t act, void* stream);

/* The bf16 mode of the same kernel (BASELINE configs[3] "bf16 fwd / fp32 logic"; config key `mlp_math: bf16`): both operands rounded to
 * bf16 (nearest even), ONE product per operand pair, fp32 accumulation, fp32 output.  NOT the reference's numerics (relative error
 * ~2^-8 per product); a sixth of the matrix-pipe time.  W_bf16: ceil(N/128) * ceil(K/32) * 8192 bytes from dfol_linear_pack_w_bf16. */
int dfol_linear_pack_w_bf16(const float* W, int64_t ldw, int32_t N, int32_t K, void* W_bf16, void* stream);
int dfol_linear_act_bf16_f32(const float* X, int64_t ldx, const void* W_bf16, const float* bias, float* Y, int64_t ldy, int32_t M,
                             int32_t N, int32_t K, int32_t act, void* stream);
/* The bf16 mode with bf16 STORAGE on both sides: X [M, K] and Y [M, N] are rows of bfloat16 (ldx, ldy in elements; K % 4 == 0, ldx % 4 == 0,
 * X 8-byte aligned), fp32 accumulation and bias, the result rounded to nearest even.  What torch.autocast(bfloat16) makes of the same
 * nn.Linear line (gqa_interpreter_experiments.py:26-33, 73-74): used for the per-pair activations of a train step in the bf16 mode
 * (Z -> pre2 and dpre2 -> dZ), which are 14 of the 17 GB an fp32-storage step moves. */
int dfol_linear_act_bf16_bf16(const void* X_bf16, int64_t ldx, const void* W_bf16, const float* bias, void* Y_bf16, int64_t ldy, int32_t M,
                              int32_t N, int32_t K, int32_t act, void* stream);


/* Box positional features: replaces batch_gqa_boxfeatures_pipeline.py:208-211.
 *   raw [O, ld_raw]: the last 6 columns (ending at column `raw_cols`) are (W, H, x, y, w, h);
 *   writes pos = (x, y, w, h) / max((W, H, W, H), 1) into obj[:, pos_col .. pos_col+3].
 */
int dfol_box_positions_f32(const float* raw, int64_t ld_raw, int32_t raw_cols, int32_t O, float* obj, int64_t ld_obj,
                           int32_t pos_col, void* stream);

/* Pair features: replaces batch_gqa_boxfeatures_pipeline.py:252-279 (same-image ordered pairs, s != o).
 *   obj [O, ld_obj] with D = feat_dim + 4 used columns whose last 4 are the positional features;
 *   pair [pairs, 2D + 4] out = [obj_s, obj_o, distance, asin(dy / max(distance, 1e-10)), sign(x_o-x_s), sign(y_o-y_s)]
 */
int dfol_pair_features_f32(const float* obj, int64_t ld_obj, int32_t D, const int32_t* obj_off, const int64_t* pair_off,
                           int32_t Q, int32_t max_n, float* pair, int64_t ld_pair, void* stream);

/* Attention calibration: replaces BatchVariableSet.apply_modulations, batch_base_types.py:170-179, for the 4-column
 * modulations the reference's attention_output_network emits (gqa_interpreter_experiments.py:119-132):
 *   alpha = 10 m0, beta = 10 m1, c = 10 m2, d = m3;  t = alpha a + slog(c) + slog(d);
 *   out = t - slog(exp(beta * log_not(a) + slog(1 - d)) + exp(t))          att, out [P, NS]; mods [P, 4]
 */
int dfol_modulate_f32(const float* att, const float* mods, const int32_t* pred_q, const int32_t* n_obj, int32_t P, int32_t NS,
                      float* out, void* stream);

/* LSTM cell of the attention-calibration passes (batch_base_interpreter.py:87-140; nn.LSTMCell(318 -> 50) built at
 * gqa_interpreter_experiments.py:115-138), pointwise stage: igates = x W_ih^T + b_ih and hgates = h W_hh^T + b_hh ([rows, 4H], gate
 * order i, f, g, o; two dfol_linear_act_f32 launches) ->  c' = sigmoid(f) c + sigmoid(i) tanh(g),  h' = sigmoid(o) tanh(c').
 */
int dfol_lstm_pointwise_f32(const float* igates, const float* hgates, const float* c, int32_t rows, int32_t H, float* h_out,
                            float* c_out, void* stream);
/* The whole LSTM cell in one launch (gate products, biases, pointwise stage): x [rows, ld_x] (KX used), h [rows, ld_h], c [rows, H],
 * Wih = W_ih^T [KX, ld_wih >= 4H], Whh = W_hh^T [H, ld_whh >= 4H] (TRANSPOSED weights: coalesced across the gate threads),
 * b_ih / b_hh [4H] or NULL -> h_out, c_out [rows, H].  What the calibration passes run.
 */
int dfol_lstm_cell_f32(const float* x, int64_t ld_x, int32_t KX, const float* h, int64_t ld_h, const float* c, const float* Wih,
                       int64_t ld_wih, const float* Whh, int64_t ld_whh, const float* bih, const float* bhh, int32_t rows, int32_t H,
                       float* h_out, float* c_out, void* stream);
/* The same cell on an operator's TOKENS: row p of x is built while it is staged - [head (n_head floats: operator one-hot, token-type flag) |
 * table[idx[p]] (E floats: token embedding)], all zeros where idx[p] < 0 - what dfol_calib_features_f32 would write (batch_base_ops.py:265-273,
 * 437-446, 628-637), without the round trip through memory; KX = n_head + E.  Bit-identical to dfol_calib_features_f32 + dfol_lstm_cell_f32.
 */
int dfol_lstm_cell_tokens_f32(const float* head, int32_t n_head, const float* table, int32_t E, const int32_t* idx, const float* h, int64_t ld_h,
                              const float* c, const float* Wih, int64_t ld_wih, const float* Whh, int64_t ld_whh, const float* bih, const float* bhh,
                              int32_t rows, int32_t H, float* h_out, float* c_out, void* stream);

/* Small pieces of the calibration passes, shared by the Python operators and the native executor (round 6):
 * select_rows: out[p] = flags[p] ? x[p] : y[p] over rows of `width` floats - BatchAttentionState.gate with 0 / 1 flags (batch_base_types.py:279-298).
 * calib_features: the LSTM input rows of an operator's tokens (batch_base_ops.py:265-273, 437-446, 628-637): out[p] = [head (n_head floats: operator
 *   one-hot, token-type flag) | table[idx[p]] (E floats: token embedding)], all zeros where idx[p] < 0 (no-op token); out [P, n_head + E].
 * attention_modulations: BatchOperatorBase._compute_attention_modulations (batch_base_ops.py:275-286) with the Linear(2 S -> N) + Sigmoid output
 *   network (gqa_interpreter_experiments.py:119-132): out[p][j] = Sigmoid(b[j] + W[j][:S] . fs[p] + W[j][S:] . bs[p]); fs / bs [P, S] or NULL (= zeros).
 */
int dfol_select_rows_f32(const float* x, const float* y, const uint8_t* flags, int32_t P, int32_t width, float* out, void* stream);
int dfol_calib_features_f32(const float* head, int32_t n_head, const float* table, int32_t E, const int32_t* idx, int32_t P, float* out, void* stream);
int dfol_attention_modulations_f32(const float* fs, const float* bs, const float* W, int64_t ld_w, const float* b, int32_t P, int32_t S, int32_t N,
                                   float* out, void* stream);

/* Training of the attention calibrator (the curriculum's cur6-7 phases: oracle frozen, the two LSTM cells and the attention output
 * layer train; trainer.py:429-442 over batch_base_interpreter.py:87-140).  dfol_lstm_cell_train_f32 is dfol_lstm_cell_f32 that also
 * stores the ACTIVATED gates [rows, 4H] (sigmoid(i), sigmoid(f), tanh(g), sigmoid(o)); dfol_lstm_cell_bwd_f32 is the backward of the
 * pointwise stage (torch's lstm_cell_backward): -> d_gates [rows, 4H] w.r.t. the PRE-activation gates and d_c_prev [rows, H]; d_hy / d_cy
 * may be NULL (no gradient through that output).  The weight / input products of the backward are dfol_linear_act_f32 and
 * dfol_linear_wgrad_bias_f32 calls on d_gates.
 */
int dfol_lstm_cell_train_f32(const float* x, int64_t ld_x, int32_t KX, const float* h, int64_t ld_h, const float* c, const float* Wih,
                             int64_t ld_wih, const float* Whh, int64_t ld_whh, const float* bih, const float* bhh, int32_t rows, int32_t H,
                             float* h_out, float* c_out, float* gates, void* stream);
int dfol_lstm_cell_bwd_f32(const float* gates, const float* c_prev, const float* c_new, const float* d_hy, const float* d_cy, int32_t rows,
                           int32_t H, float* d_gates, float* d_c_prev, void* stream);

/* Backward of dfol_modulate_f32 (apply_modulations, batch_base_types.py:170-179): g_out [P, NS] -> g_att [P, NS] and g_mods [P, 4]
 * (the sums over a predicate's objects taken by one wavefront in a fixed order: deterministic).
 */
int dfol_modulate_bwd_f32(const float* g_out, const float* att, const float* mods, const int32_t* pred_q, const int32_t* n_obj, int32_t P,
                          int32_t NS, float* g_att, float* g_mods, void* stream);

/* ---- needed-columns oracle (MI355X-first: nothing the program does not ask for is computed) --------
 * The reference evaluates the embedding layer for all 2335 concepts on every object and every ordered
 * object pair (classifier_oracle.py:145-156; 64 % of its CPU time) and then gathers a
 * handful of columns (:63, :116).  These two entry points produce the same per-predicate blocks directly
 * from the hidden activations, for the requested (image, concept) pairs only.
 *
 * Attribute blocks:  ll[p][o] = LogSigmoid(hidden[obj_off[q]+o] . E[col] + be[col])
 *   hidden [O, ld_hidden] (H used) = attribute_network output (Sigmoid layer), E [C, ld_e] / be [C] the
 *   embedding layer (gqa_interpreter_experiments.py:60-77); pred_col indexes the FULL concept table
 *   (arg_to_idx - 1, classifier_oracle.py:49-56); < 0 = no-op token -> default block.
 */
int dfol_attr_ll_f32(const float* hidden, int64_t ld_hidden, int32_t H, const float* E, int64_t ld_e, const float* be,
                     const int32_t* obj_off, const int32_t* pred_q, const int32_t* pred_col, int32_t P, int32_t NS,
                     float default_ll, float* ll, void* stream);

/* Relation tiles, fused pair MLP (replaces batch_gqa_boxfeatures_pipeline.py:252-279 + the relation branch of
 * classifier_oracle.py:151-154 + the gather :113-135, for the requested columns only).
 * With the pair MLP  h = Sigmoid(W2 ELU(W1 [obj_s, obj_o, geo] + b1) + b2)  (gqa_interpreter_experiments.py:164-167):
 *   UV   [O, ld_uv]   per-OBJECT partial products of the first layer: UV[:, :HID1] = W1[:, :D] obj + b1,
 *                     UV[:, HID1:2 HID1] = W1[:, D:2D] obj         (one dfol_linear_act_f32 launch, DFOL_ACT_NONE)
 *   pos  [O, ld_pos]  the 4 positional features of every object (x, y, w, h normalised)
 *   Wg   [HID1, 4]    W1[:, 2D:2D+4] (distance, angle, h_side, v_side), contiguous
 *   W2 [HID2, ld_w2], b2 [HID2];   E [C, ld_e], be [C] (may be NULL) the embedding layer
 *   w2_rows_alloc     rows of W2 that may be read (>= HID2).  When W2 is allocated zero-padded to a multiple of 32 rows
 *                     and HID1 is a multiple of 32, the main loop runs without bounds checks.
 *   requests: K rows over the Q images.  req_col[k*Q+q] = column of the FULL concept table wanted for image q
 *             (< 0: none); req_tile[k*Q+q] = index of the destination tile; req_orient[k*Q+q] = DFOL_TILE_* (NULL = 0)
 *   tiles [T, NS, NS] out: tile[s][o] = LogSigmoid(h(s,o) . E[col] + be[col]); the diagonal is written as default_ll;
 *             rows/columns >= n_obj are NOT written (pre-fill ragged batches with default_ll).
 * Limits: HID1 <= 256 (multiple of 4), HID2 <= 320, max_n <= NS.
 */
int dfol_pair_ll_f32(const float* UV, int64_t ld_uv, int32_t HID1, const float* pos, int64_t ld_pos, const float* Wg,
                     const float* W2, int64_t ld_w2, int32_t w2_rows_alloc, const float* b2, int32_t HID2, const float* E,
                     int64_t ld_e, const float* be, const int32_t* n_obj, const int32_t* obj_off, int32_t Q, int32_t max_n,
                     const int32_t* req_col, const int32_t* req_tile, const uint8_t* req_orient, int32_t K, int32_t NS,
                     float default_ll, float* tiles, void* stream);

/* The same relation tiles from a PACKED second layer (the geometry the full-size oracle uses: two independent 4-wavefront
 * workgroups per CU, K chunks of 16, see dfol_vqa_amd/csrc/dfol_pair.hip).  W2_packed is produced once per weight update by
 * dfol_pair_pack_w2_f32 ((HID1/16)*320*16 floats, 16-byte aligned): chunk-major [HID1/16][320][16], rows >= HID2 zero,
 * k-groups swizzled for conflict-free LDS reads.  tile_dtype: DFOL_TILE_F32, or DFOL_TILE_BF16 (tiles are then [T, NS, NS] bf16
 * bit patterns for dfol_relate_one_fwd_bf16; needs HID2 > 256).  All other arguments as dfol_pair_ll_f32.
 * Limits: HID1 <= 256 and a multiple of 16, HID2 <= 320.
 */
int dfol_pair_pack_w2_f32(const float* W2, int64_t ld_w2, int32_t HID2, int32_t HID1, float* W2_packed, void* stream);
int dfol_pair_ll_packed_f32(const float* UV, int64_t ld_uv, int32_t HID1, const float* pos, int64_t ld_pos, const float* Wg,
                            const float* W2_packed, const float* b2, int32_t HID2, const float* E, int64_t ld_e, const float* be,
                            const int32_t* n_obj, const int32_t* obj_off, int32_t Q, int32_t max_n, const int32_t* req_col,
                            const int32_t* req_tile, const uint8_t* req_orient, int32_t K, int32_t NS, float default_ll,
                            int32_t tile_dtype, void* tiles, void* stream);

/* The same relation tiles with the second layer on the bf16 matrix pipes at fp32 accuracy (dfol_vqa_amd/csrc/dfol_pair_split.hip):
 * each fp32 operand is cut exactly into three bf16 pieces and six of the nine piece products are accumulated in fp32 by
 * v_mfma_f32_16x16x32_bf16; the dropped products are below 2^-23 of |a w|, the rounding error of one fp32 FMA, so the results
 * agree with dfol_pair_ll_packed_f32 to fp32 rounding (they are NOT a reduced-precision mode).  W2_split is produced once per
 * weight update by dfol_pair_pack_w2_bf16x3: (HID1/32) * 61440 bytes, 16-byte aligned; per 32 k two regions (column tiles 0-7
 * and 8-19) of [3 pieces][rows][32] bf16, rows >= HID2 zero, k-groups swizzled for conflict-free LDS reads.
 * All other arguments as dfol_pair_ll_packed_f32, with ONE difference: this kernel enumerates the ordered pairs s != o only and never
 * touches a tile's diagonal - pre-fill the tiles with default_ll (the diagonal and the padding keep that fill).
 * Limits: HID1 <= 256 and a multiple of 32, 256 < HID2 <= 320.
 */
int dfol_pair_pack_w2_bf16x3(const float* W2, int64_t ld_w2, int32_t HID2, int32_t HID1, void* W2_split, void* stream);
int dfol_pair_ll_split_f32(const float* UV, int64_t ld_uv, int32_t HID1, const float* pos, int64_t ld_pos, const float* Wg,
                           const void* W2_split, const float* b2, int32_t HID2, const float* E, int64_t ld_e, const float* be,
                           const int32_t* n_obj, const int32_t* obj_off, int32_t Q, int32_t max_n, const int32_t* req_col,
                           const int32_t* req_tile, const uint8_t* req_orient, int32_t K, int32_t NS, float default_ll,
                           int32_t tile_dtype, void* tiles, void* stream);

/* The same relation tiles with the second layer on the fp16 matrix pipe at fp32 accuracy with THREE products per fp32 product
 * (dfol_vqa_amd/csrc/dfol_pair_h2.hip; the default of the full-size oracle since round 4): each fp32 operand is cut into two fp16 pieces
 * x = h + l (22-23 significand bits) and al*wh + ah*wl + ah*wh are accumulated in fp32 by v_mfma_f32_16x16x32_f16.  Every row of W2
 * is scaled by a power of two chosen by the pack kernel (largest magnitude into [2^13, 2^14): the low pieces stay normal fp16 numbers,
 * no finite weight overflows) and un-scaled in the epilogue; activations (ELU outputs) are split unscaled and SATURATE at 6e4 - fp16's
 * range; the first layer is fed by Sigmoid outputs and box geometry, so this needs first-layer weights of magnitude ~58.  Measured
 * against float64 the results are as accurate as the fp32-pipe kernel's (tests/test_kernels_gpu.py; profiles/r04_split_accuracy_lab.txt).
 * W2_split is produced once per weight update by dfol_pair_pack_w2_f16x2: dfol_pair_w2_f16x2_bytes(HID1) bytes, 16-byte aligned
 * ((HID1/32) chunks of [2 pieces][320 rows][32] fp16, k-groups swizzled, then 320 floats -log2(e) 2^-e_r and the 320 int32 exponents).
 * UV IS TAKEN IN UNITS OF ln 2 (round 5): UV[o] = log2(e) x (the per-object halves of the first layer, bias included) - multiply the stacked
 * weight and bias by log2(e) once per weight update (dfol_vqa_amd/visual_oracle.py:_split_first_layer); Wg is passed as it is (the kernel
 * scales its 1 K geometry weights itself) and dfol_pair_pack_w2_f16x2 folds the factor ln 2 into W2.  The ELU then costs two instructions per
 * element - v_exp_f32 with the clamp modifier (2^min(z', 0)) and one fma - instead of four; a build tick of this kernel is paced by its
 * instruction count (csrc/dfol_pair_h2.hip).  Activations saturate at 6e4 in those units, i.e. at ELU outputs of 4.16e4; with a status
 * word set (dfol_set_range_status) a saturating batch is flagged DFOL_RANGE_PAIR_SATURATED.
 * All other arguments, the ordered-pairs-only enumeration (pre-fill the tiles with default_ll: the diagonal and the padding keep that
 * fill) and the limits as dfol_pair_ll_split_f32: HID1 <= 256 and a multiple of 32, 256 < HID2 <= 320.
 * Replaces classifier_oracle.py:145-156 for the relation columns a program names (gqa_interpreter_experiments.py:18-36, 60-77). */
int64_t dfol_pair_w2_f16x2_bytes(int32_t HID1);
int dfol_pair_pack_w2_f16x2(const float* W2, int64_t ld_w2, int32_t HID2, int32_t HID1, void* W2_split, void* stream);
int dfol_pair_ll_h2_f32(const float* UV, int64_t ld_uv, int32_t HID1, const float* pos, int64_t ld_pos, const float* Wg,
                        const void* W2_split, const float* b2, int32_t HID2, const float* E, int64_t ld_e, const float* be,
                        const int32_t* n_obj, const int32_t* obj_off, int32_t Q, int32_t max_n, const int32_t* req_col,
                        const int32_t* req_tile, const uint8_t* req_orient, int32_t K, int32_t NS, float default_ll,
                        int32_t tile_dtype, void* tiles, void* stream);

/* The FORWARD of a train step's pair MLP in one launch (round 6; csrc/dfol_pair_h2.hip, the kernel of dfol_pair_ll_h2_f32 with its operands'
 * layout: UV in units of ln 2, W2_split from dfol_pair_pack_w2_f16x2): per ordered pair row (image-major, subject-major: util.py:87-103;
 * pair_off[q] = first row of image q)
 *   Z    [pairs, HID1]    ELU(U[s] + V[o] + Wg geo)
 *   pre2 [pairs, ld_pre2] W2 ELU(..) + b2 (the hidden layer before its Sigmoid)          geo [pairs, 4] the pair geometry
 *   x    [K, ld_x]        x[k][row] = Sigmoid(pre2[row]) . E[req_row[k][image(row)]] for reader slot k (req_row [K, Q] int32, < 0: none; no bias)
 * Replaces dfol_pair_hidden1_fwd_f32 + the tall product over Z + the first reader's dfol_pair_logit_fwd_f32 (classifier_oracle.py:145-156 under
 * trainer.py:429-442).  HID1 <= 256 and a multiple of 32, 256 < HID2 <= 320. */
int dfol_pair_train_fwd_h2_f32(const float* UV, int64_t ld_uv, int32_t HID1, const float* pos, int64_t ld_pos, const float* Wg,
                               const void* W2_split, const float* b2, int32_t HID2, const float* E, int64_t ld_e, const int32_t* n_obj,
                               const int32_t* obj_off, const int64_t* pair_off, int32_t Q, int32_t max_n, const int32_t* req_row, int32_t K,
                               float* Z, float* pre2, int64_t ld_pre2, float* geo, float* x, int64_t ld_x, void* stream);

/* ---- training path of the pair MLP: the stages around its two tall GEMMs, fused (dfol_vqa_amd/csrc/dfol_pair_train.hip) ----------
 * Rows are the reference's ordered pairs (util.py:87-103): image-major, subject-major, the diagonal left out; pair_off[q] = first row of
 * image q ([Q] int64), obj_off[q] = its first object ([Q] int32), n_obj [Q].
 *   hidden1_fwd: Z[r, :] = ELU(U[s(r), :] + V[o(r), :] + Wg geo(r)) and geo[r, 0..3] (batch_gqa_boxfeatures_pipeline.py:263-279;
 *                gqa_interpreter_experiments.py:26-33 with the first layer split per object as in dfol_pair_ll_f32)
 *   hidden1_bwd: from dZ and Z: dU [O, ld_du], dV [O, ld_dv] (every object row written; no atomics) and one partial of the geometry-weight
 *                gradient per image, dWg_partial [Q, HID1, 4] (sum over Q on the caller's side).  max_n <= 16 * 4096 / HID1.
 *   logit_fwd:   x[r] = sum_j Sigmoid(P2[r, j]) E[p(r), j] + be[p(r)]; predicate p owns the rows pred_off[p] .. pred_off[p+1]
 *                ([P+1] int64, ascending, pred_off[P] = rows; max_rows = the largest range; P < 65536); E [P, ld_e] are the predicates'
 *                embedding rows, be [P] or NULL
 *   logit_bwd:   dP2[r, j] = dx[r] E[p, j] h (1 - h), dE[p, j] = sum_r dx[r] h[r, j], dbe[p] = sum_r dx[r] (NULL to skip); h recomputed
 * HID1 in {16, 32, ..., 1024}; HID2 <= 512.
 */
int dfol_pair_hidden1_fwd_f32(const float* U, int64_t ld_u, const float* V, int64_t ld_v, const float* pos, int64_t ld_pos, const float* Wg,
                              const int32_t* obj_off, const int64_t* pair_off, const int32_t* n_obj, int32_t Q, int32_t max_n, int32_t HID1,
                              float* Z, float* geo, void* stream);
int dfol_pair_hidden1_bwd_f32(const float* dZ, const float* Z, const float* geo, const int32_t* obj_off, const int64_t* pair_off,
                              const int32_t* n_obj, int32_t Q, int32_t max_n, int32_t HID1, float* dU, int64_t ld_du, float* dV,
                              int64_t ld_dv, float* dWg_partial, void* stream);

/* The same three sums without reading Z (fp32 storage): z is rebuilt inside the kernel from U [O, HID1], V [O, HID1] (rows 16-byte
 * aligned; the image's V rows are staged in LDS), Wg [HID1, 4] and the geometry, with dfol_pair_hidden1_fwd_f32's own expression - the
 * pass reads dZ alone.  dfol_pair_hidden1_bwd_recompute_supported(max_n, HID1): images of up to 16 * (512 / (HID1 / 4)) objects.
 * Replaces the same autograd nodes as dfol_pair_hidden1_bwd_f32 (oracle.py:304-316 backward). */
int dfol_pair_hidden1_bwd_recompute_supported(int32_t max_n, int32_t H1);
int dfol_pair_hidden1_bwd_recompute_f32(const float* dZ, const float* U, int64_t ld_u, const float* V, int64_t ld_v, const float* Wg,
                                        const float* geo, const int32_t* obj_off, const int64_t* pair_off, const int32_t* n_obj,
                                        int32_t Q, int32_t max_n, int32_t H1, float* dU, int64_t ld_du, float* dV, int64_t ld_dv,
                                        float* dWg_partial, void* stream);
int dfol_pair_logit_fwd_f32(const float* P2, int64_t ld_p2, int32_t HID2, const float* E, int64_t ld_e, const float* be,
                            const int64_t* pred_off, int32_t P, int64_t rows, int64_t max_rows, float* x, void* stream);
int dfol_pair_logit_bwd_f32(const float* dx, const float* P2, int64_t ld_p2, int32_t HID2, const float* E, int64_t ld_e,
                            const int64_t* pred_off, int32_t P, float* dP2, int64_t ld_dp2, float* dE, int64_t ld_de, float* dbe, void* stream);
/* Round 4: the backward of the pair MLP's head WITHOUT the [pairs, HID2] gradient dpre2 in memory (what autograd materialises between
 * the logit layer and the second Linear of gqa_interpreter_experiments.py:26-33 under trainer.py:436): the three consumers of dpre2
 * rebuild it from pre2, dx and the rows' embedding rows - dpre2[r, j] = dx[r] E[row_pred[r], j] h (1 - h), h = Sigmoid(pre2[r, j]).
 *   logit_bwd_sums: dE, dbe as dfol_pair_logit_bwd_f32 and dB2 [P, HID2] = every predicate's column sums of dpre2 (the second layer's
 *                   bias gradient is the sum of its P rows); dfol_pair_logit_bwd_f32 itself now takes dP2 = NULL (dE and dbe only)
 *   dz_fused:       dZ [M, HID1] (+)= dpre2 W2, W2t_split = dfol_linear_pack_w_f16x2 of W2^T [HID1, HID2]; row_pred[r] = the row of E
 *                   (or -1: no gradient), emax[p] = max |E[p, :]|; the row's dpre2 is scaled by a power of two from the bound
 *                   |dx[r]| emax / 4 before its fp16 split (csrc/dfol_dense_split.hip, LsProducer)
 *   wgrad_fused:    dW2 [HID2, HID1] = dpre2^T Z in one pass over pre2 and Z (csrc/dfol_dense_wgrad.hip); row_pred non-decreasing and
 *                   valid, pred_off [P + 1] the predicates' first rows, scale -> {S, 1 / S} on the device, S a power of two with
 *                   S max_r |dx[r]| emax[row_pred[r]] / 4 <= 2^14; HID2 <= 320, HID1 <= 256, multiples of 4
 * HID2 % 4 == 0, rows of pre2, Z and E 16-byte aligned.
 * linear_logit_h2 (the forward counterpart): Y = X W^T + b as dfol_linear_act_h2_f32 and, from the epilogue of the same pass, the logit
 * layer's forward in partial sums: x_part[s][r] = sum over the s-th 64-column half block of Sigmoid(Y[r, j]) E[row_pred[r], j],
 * s < 2 ceil(N / 128) (row_pred < 0: 0) - dfol_pair_logit_fwd_f32 without its pass over Y (the caller adds a row's slots and the bias). */
int dfol_linear_logit_h2_f32(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy, int32_t M, int32_t N,
                             int32_t K, const int32_t* row_pred, const float* E, int64_t ld_e, float* x_part, int64_t ld_xp, void* stream);
/* The same two tall products (Y = X W^T + b with the optional logit partial sums; dZ (+)= dpre2 W2) as ONE persistent workgroup per CU that
 * walks 128-row blocks over all columns (csrc/dfol_dense_tall.hip): M >= 16384, N <= 320 (dfol_linear_tall_supported), results bit for bit
 * those of dfol_linear_act_h2_f32 / dfol_pair_dz_fused_f32.  linear_tall_h2: x_part (or NULL) has FOUR slots [4, ld_xp >= M], one per
 * quarter of the padded columns.  pair_dz_tall: workspace of 2 M + 4 floats (the rows' scaled dx and un-scaling factors, then the
 * launch's largest bound and, at workspace + 2 M + 1, the {S, 1 / S} dfol_pair_wgrad_fused_f32 takes as `scale`). */
int dfol_linear_tall_supported(int64_t M, int32_t N, int32_t K);
int dfol_linear_tall_h2_f32(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy, int32_t M, int32_t N,
                            int32_t K, const int32_t* row_pred, const float* E, int64_t ld_e, float* x_part, int64_t ld_xp, void* stream);
int dfol_pair_dz_tall_f32(const float* pre2, int64_t ld_p2, const float* dx, const int32_t* row_pred, const float* E, int64_t ld_e,
                          const float* emax, const void* W2t_split, float* dZ, int64_t ld_dz, int32_t M, int32_t HID1, int32_t HID2,
                          int32_t accumulate, float* workspace, void* stream);
/* Several readers of one trunk (the relate hops of a program, choose_rel's option slots) in ONE such pass:
 * dZ (+)= dpre2 W2 with dpre2[r][j] = h (1 - h) sum_k dx_k[r] E_k[row_pred[r]][j], k < nr <= 4 - on its own every reader pays a pass over pre2 and,
 * from the second on, a read-modify-write of dZ.  The readers share row_pred (every pair row of the batch under one predicate per reader, in
 * order); dx [nr][dx_stride >= M], E [nr][P][ld_e], emax [nr][P] (max_j |E_k[p][j]|); workspace: (nr + 1) M floats. */
int dfol_pair_dz_tall_multi_f32(const float* pre2, int64_t ld_p2, const float* dx, int64_t dx_stride, int32_t nr, const int32_t* row_pred, const float* E,
                                int64_t ld_e, int32_t P, const float* emax, const void* W2t_split, float* dZ, int64_t ld_dz, int32_t M, int32_t HID1,
                                int32_t HID2, int32_t accumulate, float* workspace, void* stream);
/* The bf16 mode's forms of the same (bf16-STORED activations: `void*` rows of bfloat16, strides in elements, multiples of 4; one bf16 piece
 * per operand, fp32 accumulation, bfloat16 results rounded to nearest even): bit for bit dfol_linear_act_bf16_bf16 resp.
 * dfol_pair_logit_bwd_bf16 followed by it; wgrad_fused_sums_bf16 always yields the sums (every predicate >= 64 rows or none), needs no
 * scale (bf16 has fp32's exponent range) and only HID2 % 4 == 0; workspace as dfol_pair_wgrad_fused_sums_workspace. */
int dfol_linear_tall_bf16_bf16(const void* X_bf16, int64_t ldx, const void* W_bf16, const float* bias, void* Y_bf16, int64_t ldy, int32_t M, int32_t N,
                               int32_t K, const int32_t* row_pred, const float* E, int64_t ld_e, float* x_part, int64_t ld_xp, void* stream);
int dfol_pair_dz_tall_bf16(const void* pre2_bf16, int64_t ld_p2, const float* dx, const int32_t* row_pred, const float* E, int64_t ld_e,
                           const void* W2t_bf16, void* dZ_bf16, int64_t ld_dz, int32_t M, int32_t HID1, int32_t HID2, int32_t accumulate, void* stream);
int dfol_pair_wgrad_fused_sums_bf16(const void* pre2_bf16, int64_t ld_p2, const float* dx, const int32_t* row_pred, const int64_t* pred_off, int32_t P,
                                    const float* E, int64_t ld_e, const void* Z_bf16, int64_t ld_z, int64_t M, int32_t HID2, int32_t HID1,
                                    float* workspace, float* dW, float* dE, int64_t ld_de, float* dbe, float* db2, void* stream);
int dfol_pair_logit_bwd_sums_f32(const float* dx, const float* P2, int64_t ld_p2, int32_t HID2, const float* E, int64_t ld_e,
                                 const int64_t* pred_off, int32_t P, float* dE, int64_t ld_de, float* dbe, float* dB2, int64_t ld_db2, void* stream);
int dfol_pair_dz_fused_f32(const float* pre2, int64_t ld_p2, const float* dx, const int32_t* row_pred, const float* E, int64_t ld_e,
                           const float* emax, const void* W2t_split, float* dZ, int64_t ld_dz, int32_t M, int32_t HID1, int32_t HID2,
                           int32_t accumulate, void* stream);
int64_t dfol_pair_wgrad_fused_workspace(int64_t M, int32_t HID2, int32_t HID1);      /* floats */
/* wgrad_fused_sums: dfol_pair_wgrad_fused_f32 AND the sums of dfol_pair_logit_bwd_sums_f32 from the same pass (no pass of their own over
 * pre2): dE [P, ld_de], dbe [P] (or NULL), db2 [HID2] (already summed over the predicates).  Every predicate must own at least 64 pair rows
 * or none, HID2 % 3 == 0 (the running sums of a thread's three columns live in LDS). */
int64_t dfol_pair_wgrad_fused_sums_workspace(int64_t M, int32_t HID2, int32_t HID1, int32_t P);      /* floats */
int dfol_pair_wgrad_fused_sums_f32(const float* pre2, int64_t ld_p2, const float* dx, const int32_t* row_pred, const int64_t* pred_off, int32_t P,
                                   const float* E, int64_t ld_e, const float* scale, const float* Z, int64_t ld_z, int64_t M, int32_t HID2,
                                   int32_t HID1, float* workspace, float* dW, float* dE, int64_t ld_de, float* dbe, float* db2, void* stream);
int dfol_pair_wgrad_fused_f32(const float* pre2, int64_t ld_p2, const float* dx, const int32_t* row_pred, const int64_t* pred_off,
                              const float* E, int64_t ld_e, const float* scale, const float* Z, int64_t ld_z, int64_t M, int32_t HID2,
                              int32_t HID1, float* workspace, float* dW, void* stream);
/* The same four stages over bf16-STORED per-pair activations (the bf16 mode, BASELINE configs[3]): Z, dZ, P2 and dP2 are rows of
 * bfloat16 (row strides in elements, multiples of 4; HID2 % 4 == 0), everything per object / per predicate stays fp32, the arithmetic
 * runs in fp32 registers and results are rounded to nearest even when stored.  Same formulas and reference lines as above. */
int dfol_pair_hidden1_fwd_bf16(const float* U, int64_t ld_u, const float* V, int64_t ld_v, const float* pos, int64_t ld_pos, const float* Wg,
                               const int32_t* obj_off, const int64_t* pair_off, const int32_t* n_obj, int32_t Q, int32_t max_n, int32_t HID1,
                               void* Z_bf16, float* geo, void* stream);
int dfol_pair_hidden1_bwd_bf16(const void* dZ_bf16, const void* Z_bf16, const float* geo, const int32_t* obj_off, const int64_t* pair_off,
                               const int32_t* n_obj, int32_t Q, int32_t max_n, int32_t HID1, float* dU, int64_t ld_du, float* dV,
                               int64_t ld_dv, float* dWg_partial, void* stream);
int dfol_pair_logit_fwd_bf16(const void* P2_bf16, int64_t ld_p2, int32_t HID2, const float* E, int64_t ld_e, const float* be,
                             const int64_t* pred_off, int32_t P, int64_t rows, int64_t max_rows, float* x, void* stream);
int dfol_pair_logit_bwd_bf16(const float* dx, const void* P2_bf16, int64_t ld_p2, int32_t HID2, const float* E, int64_t ld_e,
                             const int64_t* pred_off, int32_t P, void* dP2_bf16, int64_t ld_dp2, float* dE, int64_t ld_de, float* dbe,
                             void* stream);

/* Weight gradient of a dense layer, dW [N, K] = dY^T X with dY [M, N] (row stride ld_dy) and X [M, K] (row stride ld_x): what torch
 * autograd computes for nn.Linear (gqa_interpreter_experiments.py:26-33, 73-74 under trainer.py:436).  fp32 results on the matrix
 * cores; the rows are cut into at most dfol_linear_wgrad_slabs(M, N, K) slabs whose partial results go to `workspace` (slabs *
 * round_up(N * K, 4) floats) and are added in a fixed order: deterministic, no atomics. */
int dfol_linear_wgrad_slabs(int64_t M, int32_t N, int32_t K);
int dfol_linear_wgrad_f32(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, int64_t M, int32_t N, int32_t K,
                          float* workspace, float* dW, void* stream);
/* The same with the bias gradient db [N] = column sums of dY (what autograd's sum over the batch gives nn.Linear's bias) from the same
 * pass over dY; db == NULL skips it.  `workspace`: dfol_linear_wgrad_workspace(M, N, K) floats.  Arithmetic: where dY and X rows are
 * 16-byte aligned the products run on the bf16 matrix pipe with both operands split three ways (fp32 results, as
 * dfol_linear_act_split_f32); otherwise, and always under DFOL_WGRAD_MATH=f32, on the fp32 matrix pipe. */
int64_t dfol_linear_wgrad_workspace(int64_t M, int32_t N, int32_t K);
int dfol_linear_wgrad_bias_f32(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, int64_t M, int32_t N, int32_t K,
                               float* workspace, float* dW, float* db, void* stream);
/* bf16 mode of the weight gradient (see dfol_linear_act_bf16_f32): operands rounded to bf16, one product, fp32 accumulation; db stays
 * an fp32 sum.  Same workspace. */
int dfol_linear_wgrad_bias_bf16(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, int64_t M, int32_t N, int32_t K,
                                float* workspace, float* dW, float* db, void* stream);
/* ... with both operands STORED as bfloat16 (dpre2 [M, N] and Z [M, K] of the bf16 mode; ld in elements, multiples of 4; N, K multiples
 * of 4): no rounding left to do, one product, fp32 accumulation, fp32 dW and db.  Same workspace. */
int dfol_linear_wgrad_bias_bf16_bf16(const void* dY_bf16, int64_t ld_dy, const void* X_bf16, int64_t ld_x, int64_t M, int32_t N, int32_t K,
                                     float* workspace, float* dW, float* db, void* stream);

/* dz = g * act'(.) from the activation's OUTPUT y (flat arrays of n floats): the element-wise factor of the backward of
 * y = act(x W^T + b) - what autograd computes through nn.Sigmoid / nn.ELU / nn.LogSigmoid behind the reference's nn.Linear layers
 * (gqa_interpreter_experiments.py:26-33) - in one launch: Sigmoid g y (1 - y), ELU g (y > 0 ? 1 : y + 1), LogSigmoid g (1 - e^y). */
int dfol_act_bwd_f32(const float* g, const float* y, int64_t n, int32_t act, float* dz, void* stream);

/* ---- backward (training path, trainer.py:429-442) ------------------------------------------------------
 * Gradients of the block operators; formulas in SURVEY.md Appendix B (the reference gets them from torch autograd through
 * batch_base_ops.py:62-215).  g_* outputs that are NULL are skipped.
 * DETERMINISTIC: no atomics.  Every sum over the predicates of a question is taken by the owner of the output element, walking
 * the predicates in order; for that, pred_q must be NON-DECREASING (the predicates of a question are contiguous, as every
 * operator of the reference builds them: util.flatten_list, batch_base_ops.py:324-335).
 */

/* out[q, c] = sum_{p: pred_q[p] = q} src[p, c]  (src [P, NS], out [Q, NS]; columns >= n_obj[q] are 0 when n_obj is given). */
int dfol_reduce_by_question_f32(const float* src, const int32_t* pred_q, const int32_t* n_obj, int32_t P, int32_t Q, int32_t NS,
                                float* out, void* stream);

/* g_prior [Q, NS] is WRITTEN (sum over the question's predicates); g_ll [P, NS]. */
int dfol_filter_bwd_f32(const float* g_out, const float* ll, const int32_t* pred_q, const int32_t* n_obj, const uint8_t* neg,
                        int32_t any_neg, const uint8_t* active, int32_t P, int32_t Q, int32_t NS, float* g_prior, float* g_ll,
                        void* stream);

/* g_prior_s / g_prior_o are PER PREDICATE ([P, NS], written): reduce them with dfol_reduce_by_question_f32. */
int dfol_relate_bwd_f32(const float* prior_s, const float* prior_o, const float* tile, const int32_t* pred_q, const int32_t* n_obj,
                        const float* quant_s, const float* quant_o, const uint8_t* neg, int32_t any_neg, const uint8_t* active,
                        const float* g_post_s, const float* g_post_o, int32_t P, int32_t NS, int32_t orientation,
                        int32_t lone_forall_identity, float* g_prior_s, float* g_prior_o, float* g_tile, void* stream);

int dfol_quantify_bwd_f32(const float* g_lp, const float* att, const float* quant, const int32_t* pred_q, const int32_t* n_obj,
                          int32_t P, int32_t NS, float* g_att, void* stream);

/* g_table is ADDED into, row by row, by the thread that owns the row (one object / one ordered pair): zero it first. */
int dfol_attr_gather_bwd_f32(const float* g_ll, const int32_t* obj_off, const int32_t* pred_q, const int32_t* pred_col, int32_t P,
                             int32_t Q, int32_t NS, float* g_table, int64_t ld_table, void* stream);

int dfol_rel_gather_bwd_f32(const float* g_tile, const int64_t* pair_off, const int32_t* n_obj, const int32_t* pred_q,
                            const int32_t* pred_col, int32_t P, int32_t Q, int32_t NS, int32_t orientation, float* g_table,
                            int64_t ld_table, void* stream);

/* Backward of dfol_attr_ll_f32 (needed-columns attribute likelihood; the reference back-propagates through the full
 * Linear(300 -> 2335) + LogSigmoid of gqa_interpreter_experiments.py:60-77).  gx [P, NS] is scratch (g * sigmoid(-x));
 * d_hidden [O, H] is written for every object; dE [P, H] / db [P] are PER PREDICATE (rows of equal concept are combined by
 * the caller in a fixed order).  Any of d_hidden / dE / db may be NULL. */
int dfol_attr_ll_bwd_f32(const float* g, const float* hidden, int64_t ld_hidden, int32_t H, const float* E, int64_t ld_e,
                         const float* be, const int32_t* obj_off, const int32_t* pred_q, const int32_t* pred_col, int32_t P,
                         int32_t Q, int32_t NS, float* gx, float* d_hidden, int64_t ld_dh, float* dE, int64_t ld_de, float* db,
                         void* stream);

/* from the NORMALISED values y (the softmax weight of option p is exp(y_p)). */
int dfol_option_normalize_bwd_f32(const float* g_y, const float* y, const int32_t* seg_off, int32_t S, const int32_t* pred_q,
                                  const int32_t* n_obj, int32_t NS, int32_t rank, float* g_x, void* stream);

/* ---- the tail of a train step over the flat gradient bucket (round 5; csrc/dfol_optim.hip) -------------------------------------------------
 * Replaces nn.utils.clip_grad_norm_ + torch.optim.Adam.step() of trainer.py:439-441 (seventeen launches of their foreach forms) when every
 * gradient is a view into ONE contiguous fp32 buffer `g` (parallel.GradBucket).  Deterministic, no atomics; equal to torch's result to a few
 * ulp per step (another summation order of the norm, fused multiply-adds).
 * grad_sqnorm: partials[dfol_grad_sqnorm_parts()] <- per-workgroup sums of g^2 (g 16-byte aligned).
 * clip_adam:   total = sqrt(sum partials); coef = min(1, max_norm / (total + 1e-6)) as torch.nn.utils.clip_grad_norm_ (a NaN total gives a NaN coef; max_norm = 0 zeroes the gradients; max_norm < 0: no clipping, coef 1); for tensor t of n_tensors -
 *              param[t], exp_avg[t], exp_avg_sq[t]: device ADDRESSES of its fp32 arrays (numel[t] elements), its gradient at g + goff[t] -
 *              g <- g coef (left behind, as clip_grad_norm_ does), [g += weight_decay p], exp_avg <- lerp(exp_avg, g, 1 - beta1),
 *              exp_avg_sq <- beta2 exp_avg_sq + (1 - beta2) g^2, p <- p - lr / (1 - beta1^step) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^step) + eps).
 *              The work is cut into n_chunks chunks of dfol_clip_adam_chunk() elements: chunk c covers tensor chunk_tensor[c] from element
 *              chunk_start[c].  step: step_ptr[t] = address of tensor t's fp32 step counter on the device (read, +1, written back by a
 *              one-workgroup launch behind the update: torch's capturable Adam), or step_ptr NULL and step_host = the count of THIS update
 *              (>= 1).  norm_out (or NULL) <- total. */
int32_t dfol_grad_sqnorm_parts(void);
int32_t dfol_clip_adam_chunk(void);
int dfol_grad_sqnorm_f32(const float* g, int64_t n, float* partials, void* stream);
int dfol_clip_adam_f32(float* g, const float* partials, const int64_t* param, const int64_t* exp_avg, const int64_t* exp_avg_sq, const int64_t* goff,
                       const int64_t* numel, int32_t n_tensors, const int32_t* chunk_tensor, const int64_t* chunk_start, int32_t n_chunks,
                       const int64_t* step_ptr, float step_host, float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                       float* norm_out, void* stream);

/* ---- fp16 range status (round 5) ------------------------------------------------------------------------------------------------
 * The two-piece fp16 dense kernels split their activations UNSCALED (dfol_linear_act_h2_f32, dfol_linear_logit_h2_f32: |x| > 65504 makes
 * the high piece inf and the row's products NaN).  The reference accepts any fp32 feature (batch_gqa_boxfeatures_pipeline.py:199-213 feeds
 * them to nn.Linear as they are), so instead of answering with NaN these kernels OR a bit into a caller-owned device word when it happens;
 * the host reads the word with the answers and raises, naming `mlp_math: bf16x3` (three bf16 pieces: fp32's exponent range) as the
 * remedy.  The pointer is a THREAD-LOCAL setting of the calling host thread (like dfol_last_error), picked up by the launches that
 * follow; NULL (the default) disables the check.  Cost: one v_max3 per two elements the kernel converts anyway (not measurable: the
 * dense layers of a step take 0.323 ms with it, 0.33 without), one atomic per workgroup in the failing case only.
 * dfol_pair_ll_h2_f32 SATURATES its ELU outputs at 6e4 (units of 1 / ln 2; reaching that needs first-layer weights of magnitude ~58).  The
 * test inside the kernel cost 4 % (its build tick is paced by its instruction count), so since round 6 a small kernel in front of it bounds
 * every image's first-layer sums from above - max_s U[s][k] + max_o V[o][k] + |Wg[k]| . (largest centre distance, pi / 2, 1, 1), reached by
 * a real pair unless both maxima sit on one object - and ORs DFOL_RANGE_PAIR_SATURATED when the bound passes the saturation point or is NaN
 * (launched only while a status word is set; ~6 us at 256 x 100 objects, and it leaves the U|V rows in L2 for the pair kernel).
 */
#define DFOL_RANGE_X_OVERFLOW 1u     /* an input element of a two-piece dense product is beyond fp16's largest finite value (or NaN) */
#define DFOL_RANGE_PAIR_SATURATED 2u /* dfol_pair_ll_h2_f32: a first-layer sum may exceed the ELU saturation point 6e4 / log2(e) = 4.16e4 (or is NaN) */
int dfol_set_range_status(uint32_t* device_word);

/* ---- the native executor of a lowered ProgramBatch (round 5) -------------------------------------------------------------------------
 * Replaces the reference's per-operator Python dispatch for one ProgramBatch - BatchInterpreterBase.forward's build_scene call and execution
 * loop, batch_base_interpreter.py:45-70 and :145-172, with BatchGQAInterpreter._execute, batch_gqa_interpreter.py:72-78, and the operator
 * forwards of batch_gqa_ops.py it reaches (inference: is_training = False, no attention calibration) - by ONE call that enqueues every
 * launch of the batch.  The host lowers the collated batch once (the reference already runs collate in DataLoader workers,
 * data_pipeline.py:893-898) into
 *   instr_host  [n_instr][DFOL_INSTR_WIDTH] int64 on the HOST: opcode + operands (byte offsets into `blob` / `workspace`, -1 = NULL; sizes)
 *   blob        device copy of the batch's small side arrays (geometry, concept columns, negation / validity flags, predicate -> question
 *               maps, segment offsets, quantifiers, gate flags, pair-kernel requests): one upload per batch
 *   workspace   one device arena holding every intermediate and, at its start, the results the host reads back
 * Each instruction is a call of one entry point of this header with pointers resolved from those offsets, so the results are bit for bit
 * the Python operator loop's.  Nothing is allocated or synchronised; the call only enqueues on `stream`.
 */
#define DFOL_INSTR_WIDTH 16
#define DFOL_DENSE_F32 0    /* dfol_linear_act_f32 on the raw weight */
#define DFOL_DENSE_F16X2 1  /* dfol_linear_act_h2_f32 on the dfol_linear_pack_w_f16x2 image */
#define DFOL_DENSE_BF16X3 2 /* dfol_linear_act_split_f32 on the dfol_linear_pack_w_bf16x3 image */
#define DFOL_DENSE_BF16 3   /* dfol_linear_act_bf16_f32 on the dfol_linear_pack_w_bf16 image */
#define DFOL_PAIR_PLAIN 0   /* dfol_pair_ll_f32 */
#define DFOL_PAIR_PACKED 1  /* dfol_pair_ll_packed_f32 */
#define DFOL_PAIR_BF16X3 2  /* dfol_pair_ll_split_f32 */
#define DFOL_PAIR_F16X2 3   /* dfol_pair_ll_h2_f32 */

typedef struct {            /* one nn.Linear + activation of gqa_interpreter_experiments.py:18-36 */
    int32_t kind;           /* DFOL_DENSE_*: the arithmetic of weights >= 65536 elements; smaller ones always run dfol_linear_act_f32 */
    int32_t act;            /* DFOL_ACT_* */
    int32_t N, K;
    const float* weight;    /* [N, ldw] fp32 */
    int64_t ldw;
    const void* packed;     /* the packed image for `kind` (NULL for DFOL_DENSE_F32) */
    const float* bias;      /* [N] or NULL */
} DfolDenseLayer;

typedef struct {            /* the neural modules of build_neural_modules, gqa_interpreter_experiments.py:107-198 (device pointers) */
    int32_t n_featurizer;   /* featurizer_network: raw features [:, :raw_cols - 6] -> [O, D - 4] (written into the object matrix) */
    int32_t n_attribute;    /* attribute_network: object matrix [O, D] -> hidden [O, emb_in] */
    const DfolDenseLayer* featurizer;
    const DfolDenseLayer* attribute;
    DfolDenseLayer uv;      /* the per-object halves of the relation network's first layer: [O, D] -> U | V [O, 2 hid1] */
    int32_t pair_kind;      /* DFOL_PAIR_*: which fused pair kernel evaluates the relation tiles */
    int32_t hid1, hid2, w2_rows;
    const float* wg;        /* [hid1, 4] geometry columns of the first layer */
    const void* w2;         /* second layer: raw padded [w2_rows, ld_w2] (PLAIN) or the kernel's packed image */
    int64_t ld_w2;
    const float* b2;
    const float* emb_w;     /* embedding layer [C, ld_e], bias [C] or NULL (gqa_interpreter_experiments.py:60-77) */
    int64_t ld_e;
    const float* emb_b;
    int32_t emb_in;         /* its input width (= hid2 = the attribute network's output width) */
    int32_t D;              /* object matrix width = featurizer output + 4 box positions */
    /* attention calibration (round 6; all NULL / 0 when the interpreter has no calibrator): the forward [0] and backward [1] nn.LSTMCell of
     * gqa_interpreter_experiments.py:115-138 as dfol_lstm_cell_f32 takes them (TRANSPOSED weights) and the attention-output Linear + Sigmoid */
    const float* lstm_wih_t[2]; /* W_ih^T [lstm_kx, ld >= 4 lstm_h] */
    const float* lstm_whh_t[2]; /* W_hh^T [lstm_h, ld >= 4 lstm_h] */
    int64_t lstm_ld_wih[2], lstm_ld_whh[2];
    const float* lstm_bih[2];   /* [4 lstm_h] or NULL */
    const float* lstm_bhh[2];
    int32_t lstm_kx, lstm_h;    /* input width (17 operator one-hot + 1 type flag + token embedding) and state width */
    const float* att_out_w;     /* [att_out_n, ld_att_out >= 2 lstm_h] */
    int64_t ld_att_out;
    const float* att_out_b;     /* [att_out_n] or NULL */
    int32_t att_out_n;          /* 4 (dfol_modulate_f32's modulations) */
} DfolProgramModel;

typedef struct {            /* the scenes of one ProgramBatch (data_pipeline.py:149) */
    const float* features;  /* [O, ld_features]: raw object features, the last 6 of raw_cols columns are (W, H, x, y, w, h) */
    int64_t ld_features;
    int32_t raw_cols, O;
    int32_t NS, max_n;      /* padded block width (NS % 4 == 0) and the largest object count */
    int64_t n_obj;          /* byte offsets into `blob`: objects per QUESTION [Q] ... */
    int64_t img_n_obj;      /* ... per SCENE [scenes] (the same array unless questions share scenes) ... */
    int64_t obj_off;        /* ... and the first object row of every scene [scenes + 1] (int32) */
} DfolProgramScene;

/* opcodes (operand lists in csrc/dfol_program.hip, written by dfol_vqa_amd/native_plan.py) */
#define DFOL_OP_DENSE 0
#define DFOL_OP_BOX_POSITIONS 1
#define DFOL_OP_FILL 2
#define DFOL_OP_PAIR_LL 3
#define DFOL_OP_ATTR_LL 4
#define DFOL_OP_OPTION_NORMALIZE 5
#define DFOL_OP_FILTER 6
#define DFOL_OP_RELATE_ONE 7
#define DFOL_OP_RELATE 8
#define DFOL_OP_QUANTIFY 9
#define DFOL_OP_GATE 10
#define DFOL_OP_LOGIC 11
#define DFOL_OP_SEGMENT_SUM_ROWS 12
#define DFOL_OP_SEGMENT_OR 13
#define DFOL_OP_IMPLICATION 14
#define DFOL_OP_COMPARE 15
#define DFOL_OP_FIND_MAX_IND 16
#define DFOL_OP_GATHER_TILES 17
#define DFOL_OP_CALIB_FEATURES 18   /* head (blob), n_head, table (blob), E, idx (blob), P, out */
#define DFOL_OP_LSTM_CELL 19        /* which (0 forward, 1 backward network), x (or -1: the token form), h, c, rows, h_out, c_out; token form: head (blob),
                                     * n_head, table (blob), E, idx (blob) - dfol_lstm_cell_tokens_f32 */
#define DFOL_OP_SELECT_ROWS 20      /* x, y, flags (blob, uint8), P, width, out */
#define DFOL_OP_ATT_MODULATIONS 21  /* forward state h (or -1), backward state h (or -1), P, out [P, att_out_n] */
#define DFOL_OP_MODULATE 22         /* att, mods, pred_q (blob), P, out */
#define DFOL_OP_CALIB_WALK 23       /* table (blob: n steps of DFOL_INSTR_WIDTH int64, DFOL_WALK_* below), n, rows: a run of row-wise steps of the calibration
                                     * passes over states of `rows` rows in ONE launch - what DFOL_OP_FILL / SELECT_ROWS / LOGIC (add) / LSTM_CELL (token form) /
                                     * ATT_MODULATIONS launches in a row would compute, bit for bit (a workgroup owns 16 rows and walks the table) */
/* steps of a DFOL_OP_CALIB_WALK table; buffers are [planes][rows][width] floats in the workspace (an LSTM state is h then c: two planes) */
#define DFOL_WALK_FILL 0            /* dst, planes, width, 32-bit pattern */
#define DFOL_WALK_SELECT 1          /* x, y, flags (blob, uint8 [planes * rows]), planes, width, out: out = flags ? x : y per row */
#define DFOL_WALK_ADD 2             /* x, y, -, planes, width, out */
#define DFOL_WALK_LSTM 3            /* which, h, c, h_out, c_out, head (blob), n_head, table (blob), E, idx (blob) */
#define DFOL_WALK_ATT_MODULATIONS 4 /* forward state h, backward state h, out [rows, att_out_n] */

int dfol_run_program(const DfolProgramModel* model, const DfolProgramScene* scene, const int64_t* instr_host, int32_t n_instr,
                     const void* blob, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DFOL_VQA_H */
