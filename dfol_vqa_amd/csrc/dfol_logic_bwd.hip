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

// ---------------------------------------------------------------------------------------------------
// filter:  out[p,o] = prior[q,o] + l'(ll[p,o])
// ---------------------------------------------------------------------------------------------------
__global__ void filter_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ ll, const int32_t* __restrict__ pred_q,
                                  const int32_t* __restrict__ n_obj, const uint8_t* __restrict__ neg, int any_neg,
                                  const uint8_t* __restrict__ active, int NS, float* __restrict__ g_prior, float* __restrict__ g_ll) {
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
    if (g_prior && c < n) atomicAdd(g_prior + (int64_t)q * NS + c, g);      // several predicates may share one question's prior
}

extern "C" int dfol_filter_bwd_f32(const float* g_out, const float* ll, const int32_t* pred_q, const int32_t* n_obj, const uint8_t* neg,
                                   int32_t any_neg, const uint8_t* active, int32_t P, int32_t NS, float* g_prior, float* g_ll, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0, "filter_bwd: bad sizes");
    if (P == 0) return 0;
    DFOL_REQUIRE(g_out && ll && pred_q && n_obj && (g_prior || g_ll), "filter_bwd: null pointer");
    DFOL_REQUIRE(!any_neg || neg, "filter_bwd: any_neg set but neg is NULL");
    hipLaunchKernelGGL(filter_bwd_kernel, dim3(P, dfol_cdiv(NS, 64)), dim3(64), 0, (hipStream_t)stream, g_out, ll, pred_q, n_obj, neg,
                       any_neg, active, NS, g_prior, g_ll);
    DFOL_LAUNCH_CHECK("filter_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// relate: one wavefront per tile, two sweeps (recompute the row/column sums, then the gradients).
// Rows = variable R, columns = variable C, as in the forward kernel.
//   G1[r,c] = gR[r] F_C'(S[r]) F_C'(l'+pC[c]) ; G2[r,c] = gC[c] F_R'(T[c]) F_R'(l'+pR[r])    (off-diagonal)
//   d l' = G1 + G2 ;  d pR[r] = gR[r] + sum_c G2 ;  d pC[c] = gC[c] + sum_r G1
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relate_bwd_kernel(
    const float* __restrict__ prior_R, const float* __restrict__ prior_C, const float* __restrict__ tile,
    const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj, const float* __restrict__ quant_R,
    const float* __restrict__ quant_C, const uint8_t* __restrict__ neg, int any_neg, const uint8_t* __restrict__ active,
    const float* __restrict__ g_post_R, const float* __restrict__ g_post_C, int P, int NS, int identity_forall,
    float* __restrict__ g_prior_R, float* __restrict__ g_prior_C, float* __restrict__ g_tile) {
    __shared__ float sS[4][256], sT[4][256], sGR[4][256], sGC[4][256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + w;
    if (p >= P) return;
    const int q = pred_q[p];
    const int n = n_obj[q];
    const float* pR = prior_R + (int64_t)q * NS;
    const float* pC = prior_C + (int64_t)q * NS;
    const float* tp = tile + (int64_t)p * NS * NS;
    float* gt = g_tile ? g_tile + (int64_t)p * NS * NS : nullptr;
    const float* gR = g_post_R ? g_post_R + (int64_t)p * NS : nullptr;
    const float* gC = g_post_C ? g_post_C + (int64_t)p * NS : nullptr;

    if (active && !active[p]) {                            // posterior = prior: the gradient passes straight through
        for (int c = lane; c < n; c += 64) {
            if (g_prior_R && gR) atomicAdd(g_prior_R + (int64_t)q * NS + c, gR[c]);
            if (g_prior_C && gC) atomicAdd(g_prior_C + (int64_t)q * NS + c, gC[c]);
        }
        if (gt)
            for (int e = lane; e < NS * NS; e += 64) gt[e] = 0.f;
        return;
    }
    const float alpha_n = (any_neg && neg[p]) ? 1.f : 0.f, cn = 1.f - 2.f * alpha_n;
    const float qR = quant_R[p], qC = quant_C[p], kR = 1.f - 2.f * qR, kC = 1.f - 2.f * qC;
    const bool idR = identity_forall && qR == 0.f, idC = identity_forall && qC == 0.f;

    // sweep 1: S[r] = sum_c F_C(l' + pC[c]),  T[c] = sum_r F_R(l' + pR[r])   (lane = column, rows sequential)
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int c = c0 + lane;
        float t_acc = 0.f;
        const float pc = c < n ? pC[c] : 0.f;
        for (int r = 0; r < n; ++r) {
            float s_part = 0.f;
            if (c < n && c != r) {
                const float v = dfol_prep(tp[(int64_t)r * NS + c], any_neg, alpha_n, cn);
                const float u1 = v + pc, u2 = v + pR[r];
                s_part = idC ? u1 : dfol_pnot(u1, qC, kC);
                t_acc += idR ? u2 : dfol_pnot(u2, qR, kR);
            }
            s_part = dfol_wave_sum(s_part);
            if (lane == 0) sS[w][r] = (c0 == 0 ? 0.f : sS[w][r]) + s_part;
        }
        if (c < n) sT[w][c] = t_acc;
    }
    __builtin_amdgcn_wave_barrier();
    // outer derivatives
    for (int i = lane; i < n; i += 64) {
        sGR[w][i] = gR ? gR[i] * (idC ? 1.f : dfol_dpnot(sS[w][i], qC, kC)) : 0.f;
        sGC[w][i] = gC ? gC[i] * (idR ? 1.f : dfol_dpnot(sT[w][i], qR, kR)) : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    // sweep 2: gradients
    for (int c0 = 0; c0 < NS; c0 += 64) {
        const int c = c0 + lane;
        float dpc = 0.f;
        const float pc = c < n ? pC[c] : 0.f;
        const float gcc = c < n ? sGC[w][c] : 0.f;
        for (int r = 0; r < NS; ++r) {
            float dl = 0.f, dpr_part = 0.f;
            if (r < n && c < n && c != r) {
                const float raw = tp[(int64_t)r * NS + c];
                const float v = dfol_prep(raw, any_neg, alpha_n, cn);
                const float g1 = sGR[w][r] * (idC ? 1.f : dfol_dpnot(v + pc, qC, kC));
                const float g2 = gcc * (idR ? 1.f : dfol_dpnot(v + pR[r], qR, kR));
                dl = (g1 + g2) * dfol_dprep(raw, any_neg, alpha_n, cn);
                dpc += g1;
                dpr_part = g2;
            }
            if (gt && c < NS) gt[(int64_t)r * NS + c] = dl;
            if (g_prior_R) {
                dpr_part = dfol_wave_sum(dpr_part);
                if (lane == 0 && r < n) atomicAdd(g_prior_R + (int64_t)q * NS + r, dpr_part + (c0 == 0 && gR ? gR[r] : 0.f));
            }
        }
        if (g_prior_C && c < n) atomicAdd(g_prior_C + (int64_t)q * NS + c, dpc + (gC ? gC[c] : 0.f));
    }
}

extern "C" int dfol_relate_bwd_f32(const float* prior_s, const float* prior_o, const float* tile, const int32_t* pred_q,
                                   const int32_t* n_obj, const float* quant_s, const float* quant_o, const uint8_t* neg, int32_t any_neg,
                                   const uint8_t* active, const float* g_post_s, const float* g_post_o, int32_t P, int32_t NS,
                                   int32_t orientation, int32_t lone_forall_identity, float* g_prior_s, float* g_prior_o, float* g_tile,
                                   void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0 && NS % 4 == 0 && NS <= 256, "relate_bwd: bad sizes P=%d NS=%d", P, NS);
    DFOL_REQUIRE(orientation == 0 || orientation == 1, "relate_bwd: bad orientation");
    if (P == 0) return 0;
    DFOL_REQUIRE(prior_s && prior_o && tile && pred_q && n_obj && quant_s && quant_o && (g_post_s || g_post_o), "relate_bwd: null pointer");
    DFOL_REQUIRE(!any_neg || neg, "relate_bwd: any_neg set but neg is NULL");
    const bool sr = orientation == DFOL_TILE_SUBJECT_ROWS;
    hipLaunchKernelGGL(relate_bwd_kernel, dim3(dfol_cdiv(P, 4)), dim3(256), 0, (hipStream_t)stream, sr ? prior_s : prior_o,
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
// gathers from the cached tables: scatter-add of the block gradients into the table gradients
// ---------------------------------------------------------------------------------------------------
__global__ void attr_gather_bwd_kernel(const float* __restrict__ g_ll, const int32_t* __restrict__ obj_off, const int32_t* __restrict__ pred_q,
                                       const int32_t* __restrict__ pred_col, int NS, float* __restrict__ g_table, int64_t ld) {
    const int p = blockIdx.x;
    const int o = blockIdx.y * blockDim.x + threadIdx.x;
    const int q = pred_q[p], col = pred_col[p];
    const int first = obj_off[q], n = obj_off[q + 1] - first;
    if (col < 0 || o >= n) return;
    atomicAdd(g_table + (int64_t)(first + o) * ld + col, g_ll[(int64_t)p * NS + o]);
}

extern "C" int dfol_attr_gather_bwd_f32(const float* g_ll, const int32_t* obj_off, const int32_t* pred_q, const int32_t* pred_col, int32_t P,
                                        int32_t NS, float* g_table, int64_t ld_table, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0, "attr_gather_bwd: bad sizes");
    if (P == 0) return 0;
    DFOL_REQUIRE(g_ll && obj_off && pred_q && pred_col && g_table, "attr_gather_bwd: null pointer");
    hipLaunchKernelGGL(attr_gather_bwd_kernel, dim3(P, dfol_cdiv(NS, 64)), dim3(64), 0, (hipStream_t)stream, g_ll, obj_off, pred_q, pred_col,
                       NS, g_table, ld_table);
    DFOL_LAUNCH_CHECK("attr_gather_bwd");
    return 0;
}

__global__ void rel_gather_bwd_kernel(const float* __restrict__ g_tile, const int64_t* __restrict__ pair_off, const int32_t* __restrict__ n_obj,
                                      const int32_t* __restrict__ pred_q, const int32_t* __restrict__ pred_col, int NS, int transposed,
                                      float* __restrict__ g_table, int64_t ld) {
    const int p = blockIdx.x;
    const int e = blockIdx.y * blockDim.x + threadIdx.x;
    if (e >= NS * NS) return;
    const int r = e / NS, c = e - r * NS;
    const int q = pred_q[p], col = pred_col[p];
    const int n = n_obj[q];
    const int s = transposed ? c : r, o = transposed ? r : c;
    if (col < 0 || s >= n || o >= n || s == o) return;
    const int64_t pair = pair_off[q] + (int64_t)s * (n - 1) + (o > s ? o - 1 : o);
    atomicAdd(g_table + pair * ld + col, g_tile[(int64_t)p * NS * NS + e]);
}

extern "C" int dfol_rel_gather_bwd_f32(const float* g_tile, const int64_t* pair_off, const int32_t* n_obj, const int32_t* pred_q,
                                       const int32_t* pred_col, int32_t P, int32_t NS, int32_t orientation, float* g_table, int64_t ld_table,
                                       void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0 && (orientation == 0 || orientation == 1), "rel_gather_bwd: bad arguments");
    if (P == 0) return 0;
    DFOL_REQUIRE(g_tile && pair_off && n_obj && pred_q && pred_col && g_table, "rel_gather_bwd: null pointer");
    hipLaunchKernelGGL(rel_gather_bwd_kernel, dim3(P, dfol_cdiv((int64_t)NS * NS, 256)), dim3(256), 0, (hipStream_t)stream, g_tile, pair_off,
                       n_obj, pred_q, pred_col, NS, orientation, g_table, ld_table);
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
