// Backward kernels of the soft-logic operators (training path: trainer.py:429-442 back-propagates the loss of
// :181-262 through the interpreter).  Formulas: SURVEY.md Appendix B, checked against the reference's autograd
// through the goldens g6 (tests/test_backward_gpu.py).  Same block layout and launch geometry as the forward.
#include "dfol_common.h"

// d/dx log(max(alpha + c e^x, eps)),  c = 1 - 2 alpha  (the clamp has zero gradient below the floor)
__device__ __forceinline__ float dfol_dpnot(float x, float alpha, float c) {
    const float e = dfol_exp(x);
    const float d = alpha + c * e;
    return d > DFOL_EPS ? c * e / d : 0.f;
}

// chain through  v = min(ll, 0)  and the optional negation  l' = pnot(v, neg)
__device__ __forceinline__ float dfol_dprep(float ll, int any_neg, float alpha_n, float cn) {
    const float v = fminf(ll, 0.f);
    float d = ll < 0.f ? 1.f : 0.f;                        // -relu(-x): subgradient 0 at and above 0 (torch relu)
    if (any_neg) d *= dfol_dpnot(v, alpha_n, cn);
    return d;
}

__device__ __forceinline__ float dfol_prep(float ll, int any_neg, float alpha_n, float cn) {
    float v = fminf(ll, 0.f);
    if (any_neg) v = dfol_pnot(v, alpha_n, cn);
    return v;
}

// Predicates of question q: [p0, p1) by binary search in the non-decreasing pred_q (the predicates of a question are contiguous,
// as every operator builds them).  Every cross-predicate sum of the backward pass is taken by the OWNER of the output element,
// walking this range in order: no atomics, bit-identical from run to run.
__device__ __forceinline__ void dfol_pred_range(const int32_t* __restrict__ pred_q, int P, int q, int& p0, int& p1) {
    int lo = 0, hi = P;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pred_q[mid] < q) lo = mid + 1; else hi = mid;
    }
    p0 = lo;
    hi = P;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pred_q[mid] <= q) lo = mid + 1; else hi = mid;
    }
    p1 = lo;
}

// out[q, c] = sum over the predicates p of question q of src[p, c]  (c < n_obj[q]; padding columns 0)
__global__ void reduce_by_question_kernel(const float* __restrict__ src, const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj,
                                          int P, int NS, float* __restrict__ out) {
    const int q = blockIdx.x;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= NS) return;
    int p0, p1;
    dfol_pred_range(pred_q, P, q, p0, p1);
    float acc = 0.f;
    if (n_obj == nullptr || c < n_obj[q])
        for (int p = p0; p < p1; ++p) acc += src[(int64_t)p * NS + c];
    out[(int64_t)q * NS + c] = acc;
}

extern "C" int dfol_reduce_by_question_f32(const float* src, const int32_t* pred_q, const int32_t* n_obj, int32_t P, int32_t Q, int32_t NS,
                                           float* out, void* stream) {
    DFOL_REQUIRE(P >= 0 && Q >= 0 && NS > 0, "reduce_by_question: bad sizes");
    if (Q == 0) return 0;
    DFOL_REQUIRE(out && (P == 0 || (src && pred_q)), "reduce_by_question: null pointer");
    hipLaunchKernelGGL(reduce_by_question_kernel, dim3(Q, dfol_cdiv(NS, 64)), dim3(64), 0, (hipStream_t)stream, src, pred_q, n_obj, P, NS, out);
    DFOL_LAUNCH_CHECK("reduce_by_question");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// filter:  out[p,o] = prior[q,o] + l'(ll[p,o])
// ---------------------------------------------------------------------------------------------------
__global__ void filter_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ ll, const int32_t* __restrict__ pred_q,
                                  const int32_t* __restrict__ n_obj, const uint8_t* __restrict__ neg, int any_neg,
                                  const uint8_t* __restrict__ active, int NS, float* __restrict__ g_ll) {
    const int p = blockIdx.x;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= NS) return;
    const int q = pred_q[p];
    const int n = n_obj[q];
    const int64_t i = (int64_t)p * NS + c;
    const float g = c < n ? g_out[i] : 0.f;
    const bool act = active == nullptr || active[p];
    if (g_ll) {
        float d = 0.f;
        if (act && c < n) {
            const float alpha = (any_neg && neg[p]) ? 1.f : 0.f;
            d = g * dfol_dprep(ll[i], any_neg, alpha, 1.f - 2.f * alpha);
        }
        g_ll[i] = d;
    }
}

extern "C" int dfol_filter_bwd_f32(const float* g_out, const float* ll, const int32_t* pred_q, const int32_t* n_obj, const uint8_t* neg,
                                   int32_t any_neg, const uint8_t* active, int32_t P, int32_t Q, int32_t NS, float* g_prior, float* g_ll,
                                   void* stream) {
    DFOL_REQUIRE(P >= 0 && Q >= 0 && NS > 0, "filter_bwd: bad sizes");
    if (P == 0 && Q == 0) return 0;
    DFOL_REQUIRE(g_out && ll && pred_q && n_obj && (g_prior || g_ll), "filter_bwd: null pointer");
    DFOL_REQUIRE(!any_neg || neg, "filter_bwd: any_neg set but neg is NULL");
    if (g_ll && P > 0) {
        hipLaunchKernelGGL(filter_bwd_kernel, dim3(P, dfol_cdiv(NS, 64)), dim3(64), 0, (hipStream_t)stream, g_out, ll, pred_q, n_obj, neg,
                           any_neg, active, NS, g_ll);
        DFOL_LAUNCH_CHECK("filter_bwd");
    }
    // d out / d prior = 1 for active and inactive predicates alike: the prior's gradient is the sum of its predicates' g_out
    if (g_prior) return dfol_reduce_by_question_f32(g_out, pred_q, n_obj, P, Q, NS, g_prior, stream);
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// relate: one workgroup per tile, two sweeps (recompute the row/column sums, then the gradients).
// Rows = variable R, columns = variable C, as in the forward kernel.
//   G1[r,c] = gR[r] F_C'(S[r]) F_C'(l'+pC[c]) ; G2[r,c] = gC[c] F_R'(T[c]) F_R'(l'+pR[r])    (off-diagonal)
//   d l' = G1 + G2 ;  d pR[r] = gR[r] + sum_c G2 ;  d pC[c] = gC[c] + sum_r G1
// ---------------------------------------------------------------------------------------------------
// One workgroup per predicate, its RB_WAVES wavefronts on rows r = w, w + RB_WAVES, ...: a row's sums are taken by the wavefront that owns the
// row, a column's partial sums of the wavefronts are added in wavefront order (no atomics).  The kernel is the latency of its dependent row
// iterations (one wavefront per predicate, round 1: 0.25 ms for the 256 predicates of a train step at 100 objects; four: 79 us; sixteen - a
// predicate per CU still - 29 us).
constexpr int RB_WAVES = 16;                                      // (four when the (3 + waves) NS floats of LDS would pass 48 KB)
__global__ __launch_bounds__(64 * RB_WAVES) void relate_bwd_kernel(
    const float* __restrict__ prior_R, const float* __restrict__ prior_C, const float* __restrict__ tile,
    const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj, const float* __restrict__ quant_R,
    const float* __restrict__ quant_C, const uint8_t* __restrict__ neg, int any_neg, const uint8_t* __restrict__ active,
    const float* __restrict__ g_post_R, const float* __restrict__ g_post_C, int P, int NS, int identity_forall,
    float* __restrict__ g_prior_R, float* __restrict__ g_prior_C, float* __restrict__ g_tile) {
    extern __shared__ float relate_bwd_lds[];                         // (3 + RB_WAVES) NS floats: row sums / outer derivatives / per-wavefront column partials
    float* sS = relate_bwd_lds;
    float* sGR = sS + NS;
    float* sGC = sGR + NS;
    float* part_base = sGC + NS;
    auto part = [&](int wave, int c) -> float& { return part_base[wave * NS + c]; };
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x, nw = (int)blockDim.x >> 6, nt = (int)blockDim.x;
    const int p = blockIdx.x;
    const int q = pred_q[p];
    const int n = n_obj[q];
    const float* pR = prior_R + (int64_t)q * NS;
    const float* pC = prior_C + (int64_t)q * NS;
    const float* tp = tile + (int64_t)p * NS * NS;
    float* gt = g_tile ? g_tile + (int64_t)p * NS * NS : nullptr;
    const float* gR = g_post_R ? g_post_R + (int64_t)p * NS : nullptr;
    const float* gC = g_post_C ? g_post_C + (int64_t)p * NS : nullptr;

    if (active && !active[p]) {                            // posterior = prior: the gradient passes straight through
        for (int c = tid; c < NS; c += nt) {
            if (g_prior_R) g_prior_R[(int64_t)p * NS + c] = (gR && c < n) ? gR[c] : 0.f;
            if (g_prior_C) g_prior_C[(int64_t)p * NS + c] = (gC && c < n) ? gC[c] : 0.f;
        }
        if (gt)
            for (int e = tid; e < NS * NS; e += nt) gt[e] = 0.f;
        return;
    }
    const float alpha_n = (any_neg && neg[p]) ? 1.f : 0.f, cn = 1.f - 2.f * alpha_n;
    const float qR = quant_R[p], qC = quant_C[p], kR = 1.f - 2.f * qR, kC = 1.f - 2.f * qC;
    const bool idR = identity_forall && qR == 0.f, idC = identity_forall && qC == 0.f;

    // sweep 1: S[r] = sum_c F_C(l' + pC[c]),  T[c] = sum_r F_R(l' + pR[r])   (lane = column, this wavefront's rows in order)
    for (int c0 = 0; c0 < NS; c0 += 64) {
        const int c = c0 + lane;
        float t_acc = 0.f;
        const float pc = c < n ? pC[c] : 0.f;
        for (int r = w; r < n; r += nw) {
            float s_part = 0.f;
            if (c < n && c != r) {
                const float v = dfol_prep(tp[(int64_t)r * NS + c], any_neg, alpha_n, cn);
                const float u1 = v + pc, u2 = v + pR[r];
                s_part = idC ? u1 : dfol_pnot(u1, qC, kC);
                t_acc += idR ? u2 : dfol_pnot(u2, qR, kR);
            }
            s_part = dfol_wave_sum(s_part);
            if (lane == 0) sS[r] = (c0 == 0 ? 0.f : sS[r]) + s_part;
        }
        if (c < NS) part(w, c) = t_acc;
    }
    __syncthreads();
    // outer derivatives (T[c]: the four partials in wavefront order)
    for (int i = tid; i < n; i += nt) {
        float T = part(0, i);
        for (int j = 1; j < nw; ++j) T += part(j, i);
        sGR[i] = gR ? gR[i] * (idC ? 1.f : dfol_dpnot(sS[i], qC, kC)) : 0.f;
        sGC[i] = gC ? gC[i] * (idR ? 1.f : dfol_dpnot(T, qR, kR)) : 0.f;
    }
    __syncthreads();
    // sweep 2: gradients
    for (int c0 = 0; c0 < NS; c0 += 64) {
        const int c = c0 + lane;
        float dpc = 0.f;
        const float pc = c < n ? pC[c] : 0.f;
        const float gcc = c < n ? sGC[c] : 0.f;
        for (int r = w; r < NS; r += nw) {
            float dl = 0.f, dpr_part = 0.f;
            if (r < n && c < n && c != r) {
                const float raw = tp[(int64_t)r * NS + c];
                const float v = dfol_prep(raw, any_neg, alpha_n, cn);
                const float g1 = sGR[r] * (idC ? 1.f : dfol_dpnot(v + pc, qC, kC));
                const float g2 = gcc * (idR ? 1.f : dfol_dpnot(v + pR[r], qR, kR));
                dl = (g1 + g2) * dfol_dprep(raw, any_neg, alpha_n, cn);
                dpc += g1;
                dpr_part = g2;
            }
            if (gt && c < NS) gt[(int64_t)r * NS + c] = dl;
            if (g_prior_R) {                                   // row totals over the column chunks collect in LDS (sS is free by now)
                dpr_part = dfol_wave_sum(dpr_part);
                if (lane == 0 && r < n) sS[r] = (c0 == 0 ? (gR ? gR[r] : 0.f) : sS[r]) + dpr_part;
            }
        }
        if (c < NS) part(w, c) = dpc;
    }
    __syncthreads();
    for (int c = tid; c < NS; c += nt) {
        if (g_prior_C) {
            float t = part(0, c);
            for (int j = 1; j < nw; ++j) t += part(j, c);
            g_prior_C[(int64_t)p * NS + c] = c < n ? t + (gC ? gC[c] : 0.f) : 0.f;
        }
        if (g_prior_R) g_prior_R[(int64_t)p * NS + c] = c < n ? sS[c] : 0.f;
    }
}

extern "C" int dfol_relate_bwd_f32(const float* prior_s, const float* prior_o, const float* tile, const int32_t* pred_q,
                                   const int32_t* n_obj, const float* quant_s, const float* quant_o, const uint8_t* neg, int32_t any_neg,
                                   const uint8_t* active, const float* g_post_s, const float* g_post_o, int32_t P, int32_t NS,
                                   int32_t orientation, int32_t lone_forall_identity, float* g_prior_s, float* g_prior_o, float* g_tile,
                                   void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0 && NS % 4 == 0 && NS <= 2048, "relate_bwd: bad sizes P=%d NS=%d (NS: a multiple of 4, <= 2048)", P, NS);
    DFOL_REQUIRE(orientation == 0 || orientation == 1, "relate_bwd: bad orientation");
    if (P == 0) return 0;
    DFOL_REQUIRE(prior_s && prior_o && tile && pred_q && n_obj && quant_s && quant_o && (g_post_s || g_post_o), "relate_bwd: null pointer");
    DFOL_REQUIRE(!any_neg || neg, "relate_bwd: any_neg set but neg is NULL");
    const bool sr = orientation == DFOL_TILE_SUBJECT_ROWS;
    const int waves = (size_t)(3 + RB_WAVES) * NS * sizeof(float) <= 48 * 1024 ? RB_WAVES : 4;
    hipLaunchKernelGGL(relate_bwd_kernel, dim3(P), dim3(64 * waves), (size_t)(3 + waves) * NS * sizeof(float), (hipStream_t)stream, sr ? prior_s : prior_o,
                       sr ? prior_o : prior_s, tile, pred_q, n_obj, sr ? quant_s : quant_o, sr ? quant_o : quant_s, neg, any_neg, active,
                       sr ? g_post_s : g_post_o, sr ? g_post_o : g_post_s, P, NS, lone_forall_identity, sr ? g_prior_s : g_prior_o,
                       sr ? g_prior_o : g_prior_s, g_tile);
    DFOL_LAUNCH_CHECK("relate_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// quantify:  lp = F(sum_o F(att[o]))
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quantify_bwd_kernel(const float* __restrict__ g_lp, const float* __restrict__ att,
                                                           const float* __restrict__ quant, const int32_t* __restrict__ pred_q,
                                                           const int32_t* __restrict__ n_obj, int P, int NS, float* __restrict__ g_att) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int lane = threadIdx.x & 63;
    const int n = n_obj[pred_q[p]];
    const float qf = quant[p], k = 1.f - 2.f * qf;
    const float* a = att + (int64_t)p * NS;
    float s = 0.f;
    for (int o = lane; o < n; o += 64) s += dfol_pnot(a[o], qf, k);
    s = dfol_wave_sum(s);
    const float outer = g_lp[p] * dfol_dpnot(s, qf, k);
    for (int o = lane; o < NS; o += 64) g_att[(int64_t)p * NS + o] = o < n ? outer * dfol_dpnot(a[o], qf, k) : 0.f;
}

extern "C" int dfol_quantify_bwd_f32(const float* g_lp, const float* att, const float* quant, const int32_t* pred_q, const int32_t* n_obj,
                                     int32_t P, int32_t NS, float* g_att, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0, "quantify_bwd: bad sizes");
    if (P == 0) return 0;
    DFOL_REQUIRE(g_lp && att && quant && pred_q && n_obj && g_att, "quantify_bwd: null pointer");
    hipLaunchKernelGGL(quantify_bwd_kernel, dim3(dfol_cdiv(P, 4)), dim3(256), 0, (hipStream_t)stream, g_lp, att, quant, pred_q, n_obj, P, NS, g_att);
    DFOL_LAUNCH_CHECK("quantify_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// gathers from the cached tables: the block gradients are added into the table gradients (no atomics)
// ---------------------------------------------------------------------------------------------------
// One thread owns one row of the table gradient (one object / one ordered pair) and walks the predicates of its image in order.
__global__ void attr_gather_bwd_kernel(const float* __restrict__ g_ll, const int32_t* __restrict__ obj_off, const int32_t* __restrict__ pred_q,
                                       const int32_t* __restrict__ pred_col, int P, int NS, float* __restrict__ g_table, int64_t ld) {
    const int q = blockIdx.x;
    const int o = blockIdx.y * blockDim.x + threadIdx.x;
    const int first = obj_off[q], n = obj_off[q + 1] - first;
    if (o >= n) return;
    int p0, p1;
    dfol_pred_range(pred_q, P, q, p0, p1);
    float* row = g_table + (int64_t)(first + o) * ld;
    for (int p = p0; p < p1; ++p) {
        const int col = pred_col[p];
        if (col >= 0) row[col] += g_ll[(int64_t)p * NS + o];
    }
}

extern "C" int dfol_attr_gather_bwd_f32(const float* g_ll, const int32_t* obj_off, const int32_t* pred_q, const int32_t* pred_col, int32_t P,
                                        int32_t Q, int32_t NS, float* g_table, int64_t ld_table, void* stream) {
    DFOL_REQUIRE(P >= 0 && Q >= 0 && NS > 0, "attr_gather_bwd: bad sizes");
    if (P == 0 || Q == 0) return 0;
    DFOL_REQUIRE(g_ll && obj_off && pred_q && pred_col && g_table, "attr_gather_bwd: null pointer");
    hipLaunchKernelGGL(attr_gather_bwd_kernel, dim3(Q, dfol_cdiv(NS, 64)), dim3(64), 0, (hipStream_t)stream, g_ll, obj_off, pred_q, pred_col,
                       P, NS, g_table, ld_table);
    DFOL_LAUNCH_CHECK("attr_gather_bwd");
    return 0;
}

__global__ void rel_gather_bwd_kernel(const float* __restrict__ g_tile, const int64_t* __restrict__ pair_off, const int32_t* __restrict__ n_obj,
                                      const int32_t* __restrict__ pred_q, const int32_t* __restrict__ pred_col, int P, int NS, int transposed,
                                      float* __restrict__ g_table, int64_t ld) {
    const int q = blockIdx.x;
    const int e = blockIdx.y * blockDim.x + threadIdx.x;        // (s, o) slot of the image
    const int n = n_obj[q];
    if (e >= n * n) return;
    const int s = e / n, o = e - s * n;
    if (s == o) return;
    int p0, p1;
    dfol_pred_range(pred_q, P, q, p0, p1);
    const int64_t pair = pair_off[q] + (int64_t)s * (n - 1) + (o > s ? o - 1 : o);
    float* row = g_table + pair * ld;
    const int64_t at = transposed ? (int64_t)o * NS + s : (int64_t)s * NS + o;
    for (int p = p0; p < p1; ++p) {
        const int col = pred_col[p];
        if (col >= 0) row[col] += g_tile[(int64_t)p * NS * NS + at];
    }
}

extern "C" int dfol_rel_gather_bwd_f32(const float* g_tile, const int64_t* pair_off, const int32_t* n_obj, const int32_t* pred_q,
                                       const int32_t* pred_col, int32_t P, int32_t Q, int32_t NS, int32_t orientation, float* g_table,
                                       int64_t ld_table, void* stream) {
    DFOL_REQUIRE(P >= 0 && Q >= 0 && NS > 0 && (orientation == 0 || orientation == 1), "rel_gather_bwd: bad arguments");
    if (P == 0 || Q == 0) return 0;
    DFOL_REQUIRE(g_tile && pair_off && n_obj && pred_q && pred_col && g_table, "rel_gather_bwd: null pointer");
    hipLaunchKernelGGL(rel_gather_bwd_kernel, dim3(Q, dfol_cdiv((int64_t)NS * NS, 256)), dim3(256), 0, (hipStream_t)stream, g_tile, pair_off,
                       n_obj, pred_q, pred_col, P, NS, orientation, g_table, ld_table);
    DFOL_LAUNCH_CHECK("rel_gather_bwd");
    return 0;
}

// option normalisation backward from the normalised values y: softmax weight of option p is e^{y_p}
//   dx_p = g_p - e^{y_p} * sum_{p' in seg} g_p'      (dx_p = g_p where the forward's clamp was active)
__global__ void option_normalize_bwd_kernel(const float* __restrict__ g_y, const float* __restrict__ y, const int32_t* __restrict__ seg_off,
                                            const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj, int NS, int rank,
                                            float* __restrict__ g_x) {
    const int seg = blockIdx.x;
    const int p0 = seg_off[seg], p1 = seg_off[seg + 1];
    if (p1 <= p0) return;
    const int n = n_obj[pred_q[p0]];
    const int e = blockIdx.y * blockDim.x + threadIdx.x;
    const int64_t stride = rank == 2 ? (int64_t)NS * NS : NS;
    if (e >= stride) return;
    bool real;
    if (rank == 2) {
        const int r = e / NS, c = e - r * NS;
        real = r < n && c < n && r != c;
    } else {
        real = e < n;
    }
    if (!real) {
        for (int p = p0; p < p1; ++p) g_x[p * stride + e] = g_y[p * stride + e];
        return;
    }
    float gsum = 0.f, wsum = 0.f;
    for (int p = p0; p < p1; ++p) {
        gsum += g_y[p * stride + e];
        wsum += dfol_exp(y[p * stride + e]);
    }
    const bool clamped = wsum < 0.5f;                       // sum of softmax weights is 1 unless log(max(Z, eps)) hit the floor
    for (int p = p0; p < p1; ++p) {
        const float g = g_y[p * stride + e];
        g_x[p * stride + e] = clamped ? g : g - dfol_exp(y[p * stride + e]) * gsum;
    }
}

extern "C" int dfol_option_normalize_bwd_f32(const float* g_y, const float* y, const int32_t* seg_off, int32_t S, const int32_t* pred_q,
                                             const int32_t* n_obj, int32_t NS, int32_t rank, float* g_x, void* stream) {
    DFOL_REQUIRE(S >= 0 && NS > 0 && (rank == 1 || rank == 2), "option_normalize_bwd: bad arguments");
    if (S == 0) return 0;
    DFOL_REQUIRE(g_y && y && seg_off && pred_q && n_obj && g_x, "option_normalize_bwd: null pointer");
    const int64_t elems = rank == 2 ? (int64_t)NS * NS : NS;
    hipLaunchKernelGGL(option_normalize_bwd_kernel, dim3(S, dfol_cdiv(elems, 128)), dim3(128), 0, (hipStream_t)stream, g_y, y, seg_off,
                       pred_q, n_obj, NS, rank, g_x);
    DFOL_LAUNCH_CHECK("option_normalize_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// needed-columns attribute likelihood, backward:  ll[p][o] = LogSigmoid(x),  x = hidden[o] . E[col_p] + be[col_p]
// (forward: attr_ll_kernel in dfol_dense.hip; the reference back-propagates through Linear(300 -> 2335) + LogSigmoid over ALL
// columns, gqa_interpreter_experiments.py:60-77).  Three small kernels, no atomics:
//   gx[p][o]       = g[p][o] * sigmoid(-x)                                (one wavefront per four objects, as the forward)
//   d hidden[o][:] = sum over the predicates p of o's image of gx[p][o] * E[col_p][:]        (owner: the object)
//   dE[p][:]       = sum_o gx[p][o] * hidden[o][:],   db[p] = sum_o gx[p][o]                 (owner: the predicate; rows of equal
//                    concept are combined by the caller in a fixed order)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attr_ll_bwd_gx_kernel(const float* __restrict__ g, const float* __restrict__ hidden, int64_t ld_h, int H,
                                                             const float* __restrict__ E, int64_t ld_e, const float* __restrict__ be,
                                                             const int32_t* __restrict__ obj_off, const int32_t* __restrict__ pred_q,
                                                             const int32_t* __restrict__ pred_col, int NS, float* __restrict__ gx) {
    const int p = blockIdx.x;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o0 = blockIdx.y * 16 + 4 * wv;
    const int q = pred_q[p], col = pred_col[p];
    const int first = obj_off[q], n = obj_off[q + 1] - first;
    float* out = gx + (int64_t)p * NS;
    if (col < 0 || o0 >= n) {
        if (lane < 4 && o0 + lane < NS) out[o0 + lane] = 0.f;
        return;
    }
    float e[8];                                            // H <= 512
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = (lane + 64 * j < H) ? E[(int64_t)col * ld_e + lane + 64 * j] : 0.f;
    const float bias = be ? be[col] : 0.f;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int o = min(o0 + u, n - 1);
        const float* h = hidden + (int64_t)(first + o) * ld_h;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (lane + 64 * j < H) s[u] = fmaf(h[lane + 64 * j], e[j], s[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = dfol_group_sum<64>(s[u]);     // totals are valid in lanes 48..63
    if (lane >= 60 && o0 + (lane - 60) < NS) {
        const int u = lane - 60;
        const float x = (u == 0 ? s[0] : u == 1 ? s[1] : u == 2 ? s[2] : s[3]) + bias;
        out[o0 + u] = (o0 + u < n) ? g[(int64_t)p * NS + o0 + u] / (1.f + expf(x)) : 0.f;     // d LogSigmoid(x) / dx = sigmoid(-x)
    }
}

__global__ __launch_bounds__(256) void attr_ll_bwd_hidden_kernel(const float* __restrict__ gx, const float* __restrict__ E, int64_t ld_e, int H,
                                                                 const int32_t* __restrict__ obj_off, const int32_t* __restrict__ pred_q,
                                                                 const int32_t* __restrict__ pred_col, int P, int NS,
                                                                 float* __restrict__ d_hidden, int64_t ld_dh) {
    const int q = blockIdx.x;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o = blockIdx.y * 4 + wv;
    const int first = obj_off[q], n = obj_off[q + 1] - first;
    if (o >= n) return;
    int p0, p1;
    dfol_pred_range(pred_q, P, q, p0, p1);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = p0; p < p1; ++p) {
        const int col = pred_col[p];
        if (col < 0) continue;
        const float w = gx[(int64_t)p * NS + o];
        const float* e = E + (int64_t)col * ld_e;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (lane + 64 * j < H) acc[j] = fmaf(w, e[lane + 64 * j], acc[j]);
    }
    float* out = d_hidden + (int64_t)(first + o) * ld_dh;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (lane + 64 * j < H) out[lane + 64 * j] = acc[j];
}

// One workgroup of 1024 threads per predicate: four groups of 256 threads take the objects o = g, g + 4, ... (each a chain of dependent
// row loads: with one group per predicate the kernel was 100 load round trips long, 52 us for 31 MB), their sums are added in group
// order through LDS - fixed order, no atomics.
__global__ __launch_bounds__(1024) void attr_ll_bwd_emb_kernel(const float* __restrict__ gx, const float* __restrict__ hidden, int64_t ld_h, int H,
                                                               const int32_t* __restrict__ obj_off, const int32_t* __restrict__ pred_q, int NS,
                                                               float* __restrict__ dE, int64_t ld_de, float* __restrict__ db) {
    __shared__ float part[3][2 * 256 + 1];
    const int p = blockIdx.x, t = threadIdx.x & 255, grp = threadIdx.x >> 8;
    const int q = pred_q[p];
    const int first = obj_off[q], n = obj_off[q + 1] - first;
    const float* w = gx + (int64_t)p * NS;
    float a0 = 0.f, a1 = 0.f, sb = 0.f;
#pragma unroll 8                                           // (the loads of eight objects in flight; a group's sums stay in object order)
    for (int o = grp; o < n; o += 4) {
        const float wo = w[o];
        const float* h = hidden + (int64_t)(first + o) * ld_h;
        if (t < H) a0 = fmaf(wo, h[t], a0);
        if (t + 256 < H) a1 = fmaf(wo, h[t + 256], a1);
        sb += wo;
    }
    if (grp > 0) {
        part[grp - 1][t] = a0, part[grp - 1][256 + t] = a1;
        if (t == 0) part[grp - 1][512] = sb;
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int j = 0; j < 3; ++j) a0 += part[j][t], a1 += part[j][256 + t], sb += part[j][512];
    if (dE) {
        if (t < H) dE[(int64_t)p * ld_de + t] = a0;
        if (t + 256 < H) dE[(int64_t)p * ld_de + t + 256] = a1;
    }
    if (db && t == 0) db[p] = sb;
}

extern "C" int dfol_attr_ll_bwd_f32(const float* g, const float* hidden, int64_t ld_hidden, int32_t H, const float* E, int64_t ld_e,
                                    const float* be, const int32_t* obj_off, const int32_t* pred_q, const int32_t* pred_col, int32_t P,
                                    int32_t Q, int32_t NS, float* gx, float* d_hidden, int64_t ld_dh, float* dE, int64_t ld_de, float* db,
                                    void* stream) {
    DFOL_REQUIRE(P >= 0 && Q >= 0 && NS > 0 && NS % 4 == 0 && H > 0 && H <= 512, "attr_ll_bwd: bad sizes P=%d NS=%d H=%d (H <= 512)", P, NS, H);
    if (Q == 0) return 0;
    DFOL_REQUIRE(hidden && E && obj_off && (P == 0 || (g && pred_q && pred_col && gx)), "attr_ll_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (P > 0) {
        hipLaunchKernelGGL(attr_ll_bwd_gx_kernel, dim3(P, dfol_cdiv(NS, 16)), dim3(256), 0, st, g, hidden, ld_hidden, H, E, ld_e, be, obj_off,
                           pred_q, pred_col, NS, gx);
        DFOL_LAUNCH_CHECK("attr_ll_bwd (gx)");
    }
    if (d_hidden) {                                        // every object row is written (zero when its image has no valid predicate)
        hipLaunchKernelGGL(attr_ll_bwd_hidden_kernel, dim3(Q, dfol_cdiv(NS, 4)), dim3(256), 0, st, gx, E, ld_e, H, obj_off, pred_q, pred_col, P,
                           NS, d_hidden, ld_dh);
        DFOL_LAUNCH_CHECK("attr_ll_bwd (hidden)");
    }
    if ((dE || db) && P > 0) {
        hipLaunchKernelGGL(attr_ll_bwd_emb_kernel, dim3(P), dim3(1024), 0, st, gx, hidden, ld_hidden, H, obj_off, pred_q, NS, dE, ld_de, db);
        DFOL_LAUNCH_CHECK("attr_ll_bwd (emb)");
    }
    return 0;
}

// dz = g * act'(pre) expressed through the activation's OUTPUT y (what the forward keeps): Sigmoid y (1 - y), ELU (y > 0 ? 1 : y + 1),
// LogSigmoid 1 - e^y.  One flat float4 stream instead of the three to five tensor-op launches autograd's formula takes per layer.
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, int64_t n4, int64_t n, int act,
                                                      float* __restrict__ dz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    auto d = [act](float gv, float yv) __attribute__((always_inline)) {
        if (act == DFOL_ACT_SIGMOID) return gv * yv * (1.0f - yv);
        if (act == DFOL_ACT_ELU) return gv * (yv > 0.f ? 1.0f : yv + 1.0f);
        if (act == DFOL_ACT_LOGSIGMOID) return gv * (1.0f - expf(yv));
        return gv;
    };
    if (i < n4) {
        const float4 a = reinterpret_cast<const float4*>(g)[i], b = reinterpret_cast<const float4*>(y)[i];
        reinterpret_cast<float4*>(dz)[i] = make_float4(d(a.x, b.x), d(a.y, b.y), d(a.z, b.z), d(a.w, b.w));
    } else {
        const int64_t e = 4 * n4 + (i - n4);                // the up to three trailing elements
        if (e < n) dz[e] = d(g[e], y[e]);
    }
}

extern "C" int dfol_act_bwd_f32(const float* g, const float* y, int64_t n, int32_t act, float* dz, void* stream) {
    DFOL_REQUIRE(n >= 0 && act >= DFOL_ACT_NONE && act <= DFOL_ACT_LOGSIGMOID, "act_bwd: bad arguments n=%lld act=%d", (long long)n, act);
    if (n == 0) return 0;
    DFOL_REQUIRE(g && y && dz, "act_bwd: null pointer");
    const bool vec = ((uintptr_t)g % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)dz % 16 == 0);
    const int64_t n4 = vec ? n / 4 : 0, threads = n4 + (n - 4 * n4);
    const int64_t blocks = (threads + 255) / 256;
    DFOL_REQUIRE(blocks < ((int64_t)1 << 31), "act_bwd: too many elements");
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, y, n4, n, act, dz);
    DFOL_LAUNCH_CHECK("act_bwd");
    return 0;
}
