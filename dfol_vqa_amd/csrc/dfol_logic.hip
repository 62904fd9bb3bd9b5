// Soft-logic kernels of the ∇-FOL interpreter for gfx950 (MI355X): HBM-bound, fp32.
//
// Layout: one block per predicate (include/dfol_vqa.h "BLOCK LAYOUT").  The kernels stream each
// likelihood block exactly once with 16-byte coalesced loads, keep priors and running sums in
// registers, reduce rows with cross-lane shuffles inside a 64-wide wavefront and never use atomics,
// so results are deterministic.
#include "dfol_calib.h"
#include <stdarg.h>

#include <stdlib.h>
#include <string.h>

#include "dfol_common.h"

static thread_local char g_err[512] = "";

void dfol_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dfol_last_error(void) { return g_err; }

static thread_local uint32_t* g_range_status = nullptr;
uint32_t* dfol_range_status_ptr() { return g_range_status; }
extern "C" int dfol_set_range_status(uint32_t* device_word) {
    g_range_status = device_word;
    return 0;
}
extern "C" int dfol_abi_version(void) { return DFOL_ABI_VERSION; }

// =====================================================================================================
// gathers from cached tables
// =====================================================================================================
__global__ void attr_gather_kernel(const float* __restrict__ table, int64_t ld, const int32_t* __restrict__ obj_off,
                                   const int32_t* __restrict__ pred_q, const int32_t* __restrict__ pred_col, int P, int NS,
                                   float dflt, float* __restrict__ ll) {
    const int p = blockIdx.x;
    const int o = blockIdx.y * blockDim.x + threadIdx.x;
    if (o >= NS) return;
    const int q = pred_q[p];
    const int col = pred_col[p];
    const int first = obj_off[q];
    const int n = obj_off[q + 1] - first;
    float v = dflt;
    if (col >= 0 && o < n) v = table[(int64_t)(first + o) * ld + col];
    ll[(int64_t)p * NS + o] = v;
}

extern "C" int dfol_attr_gather_f32(const float* table, int64_t ld_table, const int32_t* obj_off, const int32_t* pred_q,
                                    const int32_t* pred_col, int32_t P, int32_t NS, float default_ll, float* ll, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0 && NS % 4 == 0, "attr_gather: bad sizes P=%d NS=%d (NS must be a positive multiple of 4)", P, NS);
    if (P == 0) return 0;
    DFOL_REQUIRE(table && obj_off && pred_q && pred_col && ll, "attr_gather: null pointer");
    dim3 grid(P, dfol_cdiv(NS, 64));
    hipLaunchKernelGGL(attr_gather_kernel, grid, dim3(64), 0, (hipStream_t)stream, table, ld_table, obj_off, pred_q, pred_col, P,
                       NS, default_ll, ll);
    DFOL_LAUNCH_CHECK("attr_gather");
    return 0;
}

__global__ void rel_gather_kernel(const float* __restrict__ table, int64_t ld, const int64_t* __restrict__ pair_off,
                                  const int32_t* __restrict__ n_obj, const int32_t* __restrict__ pred_q,
                                  const int32_t* __restrict__ pred_col, int NS, int transposed, float dflt,
                                  float* __restrict__ tile) {
    const int p = blockIdx.x;
    const int e = blockIdx.y * blockDim.x + threadIdx.x;   // element of the NS x NS tile
    if (e >= NS * NS) return;
    const int r = e / NS, c = e - r * NS;
    const int q = pred_q[p];
    const int col = pred_col[p];
    const int n = n_obj[q];
    const int s = transposed ? c : r, o = transposed ? r : c;
    float v = dflt;
    if (col >= 0 && s < n && o < n && s != o) {
        const int64_t pair = pair_off[q] + (int64_t)s * (n - 1) + (o > s ? o - 1 : o);   // util.py:96-98 order
        v = table[pair * ld + col];
    }
    tile[(int64_t)p * NS * NS + e] = v;
}

extern "C" int dfol_rel_gather_f32(const float* table, int64_t ld_table, const int64_t* pair_off, const int32_t* n_obj,
                                   const int32_t* pred_q, const int32_t* pred_col, int32_t P, int32_t NS, int32_t orientation,
                                   float default_ll, float* tile, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0 && NS % 4 == 0, "rel_gather: bad sizes P=%d NS=%d", P, NS);
    DFOL_REQUIRE(orientation == 0 || orientation == 1, "rel_gather: bad orientation %d", orientation);
    if (P == 0) return 0;
    DFOL_REQUIRE(table && pair_off && n_obj && pred_q && pred_col && tile, "rel_gather: null pointer");
    dim3 grid(P, dfol_cdiv((int64_t)NS * NS, 256));
    hipLaunchKernelGGL(rel_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, table, ld_table, pair_off, n_obj, pred_q,
                       pred_col, NS, orientation, default_ll, tile);
    DFOL_LAUNCH_CHECK("rel_gather");
    return 0;
}

// =====================================================================================================
// option normalisation
// =====================================================================================================
__global__ void option_normalize_kernel(float* __restrict__ ll, const int32_t* __restrict__ seg_off,
                                        const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj, int NS, int rank) {
    const int seg = blockIdx.x;
    const int p0 = seg_off[seg], p1 = seg_off[seg + 1];
    if (p1 <= p0) return;
    const int n = n_obj[pred_q[p0]];
    const int e = blockIdx.y * blockDim.x + threadIdx.x;
    const int64_t stride = rank == 2 ? (int64_t)NS * NS : NS;
    if (rank == 2) {
        if (e >= NS * NS) return;
        const int r = e / NS, c = e - r * NS;
        if (r >= n || c >= n || r == c) return;
    } else if (e >= n) {
        return;
    }
    float sum = 0.f;
    for (int p = p0; p < p1; ++p) sum += dfol_exp(ll[p * stride + e]);
    const float denom = dfol_slog(sum);
    for (int p = p0; p < p1; ++p) ll[p * stride + e] -= denom;
}

extern "C" int dfol_option_normalize_f32(float* ll, const int32_t* seg_off, int32_t S, const int32_t* pred_q,
                                         const int32_t* n_obj, int32_t NS, int32_t rank, void* stream) {
    DFOL_REQUIRE(S >= 0 && NS > 0 && (rank == 1 || rank == 2), "option_normalize: bad arguments S=%d NS=%d rank=%d", S, NS, rank);
    if (S == 0) return 0;
    DFOL_REQUIRE(ll && seg_off && pred_q && n_obj, "option_normalize: null pointer");
    const int64_t elems = rank == 2 ? (int64_t)NS * NS : NS;
    dim3 grid(S, dfol_cdiv(elems, 128));
    hipLaunchKernelGGL(option_normalize_kernel, grid, dim3(128), 0, (hipStream_t)stream, ll, seg_off, pred_q, n_obj, NS, rank);
    DFOL_LAUNCH_CHECK("option_normalize");
    return 0;
}

// =====================================================================================================
// Filter (arity-1 logic cell)
// =====================================================================================================
// The [P, NS] blocks are streamed as one flat array of float4: a thread handles FILTER_UNR float4s a workgroup-stride apart, all
// loads of the thread issued before the arithmetic.  (One 64-thread workgroup per predicate - the first version - left most lanes
// of a 100-object block idle and ran at 0.48 of the HBM peak once the input no longer fitted the Infinity Cache.)
constexpr int FILTER_UNR = 4;

__global__ __launch_bounds__(256) void filter_fwd_kernel(const float* __restrict__ att_in, const float* __restrict__ ll,
                                                         const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj,
                                                         const uint8_t* __restrict__ neg, int any_neg, const uint8_t* __restrict__ active,
                                                         int64_t total4, int NS4, float* __restrict__ att_out) {
    const int64_t base = (int64_t)blockIdx.x * (256 * FILTER_UNR) + threadIdx.x;
    float4 a[FILTER_UNR], l4[FILTER_UNR];
    int p[FILTER_UNR], c0[FILTER_UNR], n[FILTER_UNR];
    bool act[FILTER_UNR];
#pragma unroll
    for (int u = 0; u < FILTER_UNR; ++u) {
        const int64_t idx = min(base + u * 256, total4 - 1);
        p[u] = (int)(idx / NS4);
        c0[u] = (int)(idx - (int64_t)p[u] * NS4) * 4;
        const int q = pred_q[p[u]];
        n[u] = n_obj[q];
        act[u] = active == nullptr || active[p[u]];
        a[u] = *reinterpret_cast<const float4*>(att_in + ((int64_t)q * NS4 * 4 + c0[u]));
        l4[u] = *reinterpret_cast<const float4*>(ll + idx * 4);
    }
#pragma unroll
    for (int u = 0; u < FILTER_UNR; ++u) {
        if (base + u * 256 >= total4) break;
        float4 out = a[u];
        if (act[u]) {
            const float l[4] = {l4[u].x, l4[u].y, l4[u].z, l4[u].w};
            const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
            float o[4];
            const float alpha = (any_neg && neg[p[u]]) ? 1.f : 0.f;
            const float cc = 1.f - 2.f * alpha;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = fminf(l[j], 0.f);                       // batch_base_ops.py:194
                if (any_neg) v = dfol_pnot(v, alpha, cc);         // :212-213
                o[j] = av[j] + v;                                 // :138
            }
            out = make_float4(o[0], o[1], o[2], o[3]);
        }
        if (c0[u] + 0 >= n[u]) out.x = 0.f;
        if (c0[u] + 1 >= n[u]) out.y = 0.f;
        if (c0[u] + 2 >= n[u]) out.z = 0.f;
        if (c0[u] + 3 >= n[u]) out.w = 0.f;
        *reinterpret_cast<float4*>(att_out + (base + u * 256) * 4) = out;
    }
}

extern "C" int dfol_filter_fwd_f32(const float* att_in, const float* ll, const int32_t* pred_q, const int32_t* n_obj,
                                   const uint8_t* neg, int32_t any_neg, const uint8_t* active, int32_t P, int32_t NS,
                                   float* att_out, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0 && NS % 4 == 0, "filter_fwd: bad sizes P=%d NS=%d", P, NS);
    if (P == 0) return 0;
    DFOL_REQUIRE(att_in && ll && pred_q && n_obj && att_out, "filter_fwd: null pointer");
    DFOL_REQUIRE(!any_neg || neg, "filter_fwd: any_neg set but neg is NULL");
    const int64_t total4 = (int64_t)P * (NS / 4);
    hipLaunchKernelGGL(filter_fwd_kernel, dim3(dfol_cdiv(total4, 256 * FILTER_UNR)), dim3(256), 0, (hipStream_t)stream, att_in, ll, pred_q,
                       n_obj, neg, any_neg, active, total4, NS / 4, att_out);
    DFOL_LAUNCH_CHECK("filter_fwd");
    return 0;
}

// =====================================================================================================
// Relate (arity-2 logic cell): one wavefront per predicate tile
// =====================================================================================================
// Rows of the tile belong to variable R, columns to variable C (which of subject/object is which is the
// host wrapper's business).  A lane owns four consecutive columns (one 16-byte load per row); LPR lanes
// cover a row, so 64/LPR rows are in flight per iteration.
//   post_R[r] = prior_R[r] + F_C( sum_{c != r} F_C(l'[r,c] + prior_C[c]) )      row sums   (cross-lane)
//   post_C[c] = prior_C[c] + F_R( sum_{r != c} F_R(l'[r,c] + prior_R[r]) )      column sums (in-register)
// Fast path for EXISTS/EXISTS predicates.  With E = e^{l'} and probabilities Pc = e^{prior_C}, Pr = e^{prior_R} the two aggregations are
//     1 - prod_c (1 - E[r,c] Pc[c])   and   1 - prod_r (1 - E[r,c] Pr[r]),
// so one v_exp serves both directions, and each is kept in the complement form q <- q + y - q y (dfol_or, dfol_common.h): no logarithm per
// group of factors, no cancellation, and the posterior is prior + log(max(q, eps)).  A prior above log 1 (y > 1) is the one case the
// form does not cover: the wave then reports failure and the caller redoes the predicate with the clamping general code, which keeps
// the reference's result.  The term is built from a = alpha_n + (1 - 2 alpha_n) E (= E, or 1 - E for a NEGATED predicate: e^{l'} after
// :212-213; its inner clamp at eps cannot change y, a * P vanishes either way); diagonal terms are dropped (:112: one select per
// element, which measured 7 % FASTER than an unmasked instantiation with the diagonal's term taken out afterwards); padding rows and
// columns get P = 0.
template <int LPR> struct RelateUnroll { static constexpr int value = LPR == 64 ? 4 : LPR == 32 ? 5 : LPR == 16 ? 3 : LPR == 8 ? 2 : 1; };

template <int LPR, bool WR, bool WC>
__device__ __forceinline__ bool relate_exists_fast(const float* __restrict__ tp, const float* __restrict__ pR,
                                                   const float* __restrict__ pC, int NS, int n, int lane, float* __restrict__ rsum,
                                                   float* __restrict__ oR, float* __restrict__ oC, float alpha_n = 0.f,
                                                   float* __restrict__ part = nullptr, int pitch = 0) {
    // `part` (LPR <= 32): this wavefront's [64][pitch] slab of LDS, reused per chunk of <= 64 rows.  A row's aggregate needs the OR of the partials of the LPR lanes that
    // share the row; as a DPP reduction that is log2(LPR) steps of three instructions per row and lane (15 for N = 100, against 5 for the
    // sum it replaces).  Instead every lane drops its partial into part[row][lane group] and, after the last row, lane c combines the
    // partials of row c serially: one ds_write per row in the loop, and the combining work is spread over all lanes (pitch is odd, so
    // neither access pattern has bank conflicts).  LDS operations of one wavefront execute in order.
    const float cn = 1.f - 2.f * alpha_n;
    constexpr int RPI = 64 / LPR, UNR = RelateUnroll<LPR>::value;
    constexpr float L2E = 1.44269504088896340736f;
    const int cg = lane % LPR, rs = lane / LPR, c0 = cg * 4, cl = min(c0, NS - 4);
    const float4 pc4 = *reinterpret_cast<const float4*>(pC + cl);
    const float pcl[4] = {pc4.x, pc4.y, pc4.z, pc4.w};
    float Pc[4], pmax = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        Pc[j] = (c0 + j < n) ? __builtin_amdgcn_exp2f(pcl[j] * L2E) : 0.f;
        if (WR) pmax = fmaxf(pmax, Pc[j]);
    }
    float cq[4] = {0.f, 0.f, 0.f, 0.f};                        // column complements of this lane's row slot
    constexpr int STEP = RPI * UNR, CH = (64 / STEP) * STEP;     // rows per LDS chunk: whole loop steps, at most one row per lane
    const int live = (n + 3) >> 2;                              // lane groups that hold real columns
    for (int rbase = 0; rbase < n; rbase += (WR && part) ? CH : n) {
        const int rend = (WR && part) ? min(n, rbase + CH) : n;
        for (int r0 = rbase; r0 < rend; r0 += STEP) {
            float4 t[UNR];
            float Pr[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int r = r0 + u * RPI + rs, rc = min(r, n - 1);
                t[u] = *reinterpret_cast<const float4*>(tp + (int64_t)rc * NS + cl);
                Pr[u] = 0.f;
                if (WC) {
                    const float pr = pR[rc];
                    Pr[u] = r < n ? __builtin_amdgcn_exp2f(pr * L2E) : 0.f;
                    pmax = fmaxf(pmax, Pr[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const float l[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
                float E[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) E[j] = __builtin_amdgcn_exp2f(fminf(l[j] * L2E, 0.f));    // :194 (the product is canonical: no NaN-quieting op)
                const int dg = r0 + u * RPI + rs - c0;             // column j of this lane is the diagonal iff dg == j
#pragma unroll
                for (int j = 0; j < 4; ++j) E[j] = (dg == j) ? 0.f : fmaf(cn, E[j], alpha_n);       // self-relations contribute nothing (:112)
                if (WC) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) cq[j] = dfol_or(cq[j], E[j] * Pr[u]);
                }
                if (WR) {
                    const float q01 = dfol_or(E[0] * Pc[0], E[1] * Pc[1]), q23 = dfol_or(E[2] * Pc[2], E[3] * Pc[3]);
                    const int r = r0 + u * RPI + rs;
                    if (part) {
                        if (r < n && c0 < n) part[(r - rbase) * pitch + cg] = dfol_or(q01, q23);
                    } else {
                        const float rq = dfol_group_or<LPR>(dfol_or(q01, q23));
                        if (cg == LPR - 1 && r < n) rsum[r] = rq;
                    }
                }
            }
        }
        if (WR && part) {                                       // the rows of this chunk: lane i combines the partials of row rbase + i
            __builtin_amdgcn_wave_barrier();
            const int c = rbase + lane;
            if (c < rend) {
                const float* pp = part + lane * pitch;
                float rq = 0.f;
                for (int k = 0; k < live; k += 4) {
                    const float v0 = pp[k], v1 = k + 1 < live ? pp[k + 1] : 0.f, v2 = k + 2 < live ? pp[k + 2] : 0.f, v3 = k + 3 < live ? pp[k + 3] : 0.f;
                    rq = dfol_or(rq, dfol_or(dfol_or(v0, v1), dfol_or(v2, v3)));
                }
                rsum[c] = rq;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    const bool ok = !__any(pmax > 1.f);
    if (!ok) return false;
    if (WR) {
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < NS; c += 64) {
            float o = 0.f;
            if (c < n) o = pR[c] + dfol_slog(rsum[c]);      // :112, :133, :138
            oR[c] = o;
        }
    }
    if (WC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int m = 32; m >= LPR; m >>= 1) cq[j] = dfol_or(cq[j], __shfl_xor(cq[j], m, 64));
        }
        if (rs == 0 && c0 < NS) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (c0 + j < n) ? pcl[j] + dfol_slog(cq[j]) : 0.f;
            *reinterpret_cast<float4*>(oC + c0) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    return true;
}

template <int LPR>
__global__ __launch_bounds__(256) void relate_fwd_kernel(
    const float* __restrict__ prior_R, const float* __restrict__ prior_C, const float* __restrict__ tile,
    const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj, const float* __restrict__ quant_R,
    const float* __restrict__ quant_C, const uint8_t* __restrict__ neg, int any_neg, const uint8_t* __restrict__ active,
    const uint8_t* __restrict__ want, int want_R_bit, int want_C_bit, int P, int NS, int flags,
    float* __restrict__ post_R, float* __restrict__ post_C, int part_pitch) {
    const int identity_forall = flags & DFOL_RELATE_LONE_FORALL_IDENTITY;
    constexpr int RPI = 64 / LPR;                       // rows per iteration
    __shared__ float row_sum[4][256];                   // per-wave row sums, flushed with one coalesced store
    const int wave_in_block = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wave_in_block;
    if (p >= P) return;
    const int lane = threadIdx.x & 63;
    const int q = pred_q[p];
    const int n = n_obj[q];
    const int wbits = want ? want[p] : 3;
    const bool wantR = (wbits & want_R_bit) && post_R;
    const bool wantC = (wbits & want_C_bit) && post_C;
    const float* pR = prior_R + (int64_t)q * NS;
    const float* pC = prior_C + (int64_t)q * NS;
    float* oR = post_R ? post_R + (int64_t)p * NS : nullptr;
    float* oC = post_C ? post_C + (int64_t)p * NS : nullptr;

    if (active && !active[p]) {                         // no-op predicate: posterior = prior (batch_base_ops.py:563-564)
        for (int c = lane; c < NS; c += 64) {
            if (oR) oR[c] = (wantR && c < n) ? pR[c] : 0.f;
            if (oC) oC[c] = (wantC && c < n) ? pC[c] : 0.f;
        }
        return;
    }

    // a posterior this predicate does not want is zero-filled, so that downstream selects never see garbage
    if (!wantR && oR)
        for (int c = lane; c < NS; c += 64) oR[c] = 0.f;
    if (!wantC && oC)
        for (int c = lane; c < NS; c += 64) oC[c] = 0.f;

    const int cg = lane % LPR, rs = lane / LPR;
    const int c0 = cg * 4;
    const bool col_live = c0 < n;                        // this lane's float4 touches at least one real column
    const float alpha_n = (any_neg && neg[p]) ? 1.f : 0.f, cn = 1.f - 2.f * alpha_n;
    const float qR = quant_R[p], qC = quant_C[p];
    const float kR = 1.f - 2.f * qR, kC = 1.f - 2.f * qC;
    const bool idR = identity_forall && qR == 0.f, idC = identity_forall && qC == 0.f;

    const float* tp = tile + (int64_t)p * NS * NS;
    if (qR == 1.f && qC == 1.f) {                        // EXISTS / EXISTS: the complement form (see relate_exists_fast)
        float* rsum = row_sum[wave_in_block];
        extern __shared__ float relate_part[];           // LPR <= 32: 4 slabs of [64][pitch] floats (the launch sizes it; 0 bytes otherwise)
        float* part = part_pitch > 0 ? relate_part + (size_t)wave_in_block * 64 * part_pitch : nullptr;
        // (always with the diagonal select: DFOL_RELATE_DIAG_ABSENT is accepted and no longer needed by this kernel)
        const bool ok = (wantR && wantC) ? relate_exists_fast<LPR, true, true>(tp, pR, pC, NS, n, lane, rsum, oR, oC, alpha_n, part, part_pitch)
                        : wantR      ? relate_exists_fast<LPR, true, false>(tp, pR, pC, NS, n, lane, rsum, oR, oC, alpha_n, part, part_pitch)
                        : wantC      ? relate_exists_fast<LPR, false, true>(tp, pR, pC, NS, n, lane, rsum, oR, oC, alpha_n)
                                     : true;
        if (ok) return;
        __builtin_amdgcn_wave_barrier();
    }

    float pc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < NS) {
        const float4 t = *reinterpret_cast<const float4*>(pC + c0);
        pc[0] = t.x; pc[1] = t.y; pc[2] = t.z; pc[3] = t.w;
    }
    float col_acc[4] = {0.f, 0.f, 0.f, 0.f};

    // FOR_ALL / FOR_ALL, un-negated: log(max(e^u, eps)) = u unless u is below log eps, so both aggregations are plain sums of
    // l' + prior with NO transcendental; the smallest term is tracked and a predicate at the clamp goes through the general loop.
    bool sums_done = false;
    if (alpha_n == 0.f && qR == 0.f && qC == 0.f) {
        float chk = 0.f, prmax = -1.f;
        constexpr int UNRF = 4;
        for (int r0 = 0; r0 < n; r0 += RPI * UNRF) {
            float4 tq[UNRF];
            float prq[UNRF];
#pragma unroll
            for (int u = 0; u < UNRF; ++u) {                             // all loads of the step first (clamped, unconditional)
                const int rc = min(r0 + u * RPI + rs, n - 1);
                tq[u] = *reinterpret_cast<const float4*>(tp + (int64_t)rc * NS + min(c0, NS - 4));
                prq[u] = pR[rc];
            }
#pragma unroll
            for (int u = 0; u < UNRF; ++u) {
                const int r = r0 + u * RPI + rs;
                const bool live = r < n && col_live;
                const float l[4] = {tq[u].x, tq[u].y, tq[u].z, tq[u].w};
                const float pr = prq[u];
                prmax = fmaxf(prmax, r < n ? pr : -1.f);
                float row_part = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + j;
                    const bool keep = live && c < n && c != r;
                    const float v = fminf(l[j], 0.f);
                    if (wantR) {
                        const float uu = v + pc[j];
                        chk = fminf(chk, keep ? uu : 0.f);
                        row_part += keep ? uu : 0.f;
                    }
                    if (wantC) {
                        const float uu = v + pr;
                        chk = fminf(chk, keep ? uu : 0.f);
                        col_acc[j] += keep ? uu : 0.f;
                    }
                    if (c < n) prmax = fmaxf(prmax, pc[j]);
                }
                if (wantR) {
                    row_part = dfol_group_sum<LPR>(row_part);
                    if (cg == LPR - 1 && r < n) row_sum[wave_in_block][r] = row_part;
                }
            }
        }
        sums_done = !__any(chk < -46.0f || prmax > 0.f);             // log(1e-20) = -46.05
        if (!sums_done) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) col_acc[j] = 0.f;
        }
    }

    // FOR_ALL / FOR_ALL, NEGATED: l' = log(1 - E), so a row / column sum of l' + prior is the log of a product of (1 - E) factors (each
    // exactly 0, or >= 2^-24: four per lane and row for the row sums, up to five rows for the column sums) plus the sum of the priors;
    // the smallest (1 - E) e^prior is tracked against eps = 1e-20 (the outer clamp), a zero factor (the inner one) shows as -inf.
    if (!sums_done && alpha_n == 1.f && qR == 0.f && qC == 0.f) {
        constexpr float L2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
        constexpr int UNR = 4;
        float Pc[4], pcsum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool cv = c0 + j < n;
            Pc[j] = cv ? __builtin_amdgcn_exp2f(pc[j] * L2E) : 1.f;
            pcsum += cv ? pc[j] : 0.f;
        }
        pcsum = dfol_group_sum<LPR>(pcsum);                               // sum of the column priors (valid in the group's last lane)
        pcsum = __shfl(pcsum, (lane / LPR) * LPR + LPR - 1, 64);
        float chk = 1.f, prmax = -1.f, prsum = 0.f;
        for (int r0 = 0; r0 < n; r0 += RPI * UNR) {
            float cprod[4] = {1.f, 1.f, 1.f, 1.f};
            float4 tq[UNR];
            float prq[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {                              // all loads of the step first
                const int rc = min(r0 + u * RPI + rs, n - 1);
                tq[u] = *reinterpret_cast<const float4*>(tp + (int64_t)rc * NS + min(c0, NS - 4));
                prq[u] = pR[rc];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int r = r0 + u * RPI + rs;
                const bool rlive = r < n, live = rlive && col_live;
                const float4 t = tq[u];
                const float l[4] = {t.x, t.y, t.z, t.w};
                const float pr = prq[u];
                const float Pr = __builtin_amdgcn_exp2f(pr * L2E);
                prmax = fmaxf(prmax, rlive ? pr : -1.f);
                if (cg == 0) prsum += rlive ? pr : 0.f;                   // each row's prior once per wavefront
                float rprod = 1.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + j;
                    const bool keep = live && c < n && c != r;
                    const float a = keep ? 1.f - __builtin_amdgcn_exp2f(fminf(l[j] * L2E, 0.f)) : 1.f;       // e^{l'} (:194, :212-213)
                    if (wantR) {
                        chk = fminf(chk, keep ? a * Pc[j] : 1.f);
                        rprod *= a;
                    }
                    if (wantC) {
                        chk = fminf(chk, keep ? a * Pr : 1.f);
                        cprod[j] *= a;
                    }
                    if (c < n) prmax = fmaxf(prmax, pc[j]);
                }
                if (wantR) {
                    const float part = dfol_group_sum<LPR>(__builtin_amdgcn_logf(rprod));
                    // sum_{c != r} (l' + pC[c]) = ln2 * log2(prod) + (sum of the column priors - pC[r])
                    if (cg == LPR - 1 && rlive) row_sum[wave_in_block][r] = part * LN2 + pcsum - pC[r];
                }
            }
            if (wantC) {
#pragma unroll
                for (int j = 0; j < 4; ++j) col_acc[j] += __builtin_amdgcn_logf(cprod[j]);
            }
        }
        bool bad = chk < 1.2e-20f || prmax > 0.f;
        if (wantC) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) prsum += __shfl_xor(prsum, m, 64);          // every row's prior
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = col_acc[j];
#pragma unroll
                for (int m = 32; m >= LPR; m >>= 1) v += __shfl_xor(v, m, 64);
                bad |= (c0 + j < n) && !(v >= -3.0e38f);
                // reduced here already: the common epilogue below adds the row slots again, so leave the total in slot 0 only
                col_acc[j] = (rs == 0 && c0 + j < n) ? v * LN2 + prsum - pR[c0 + j] : 0.f;      // sum_{r != c} pR[r] = all of them - pR[c]
            }
        }
        if (wantR) {
            __builtin_amdgcn_wave_barrier();
            for (int c = lane; c < n; c += 64) bad |= !(row_sum[wave_in_block][c] >= -3.0e38f);
        }
        sums_done = !__any(bad);
        if (!sums_done) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) col_acc[j] = 0.f;
        }
    }

    for (int r0 = sums_done ? n : 0; r0 < n; r0 += RPI) {
        const int r = r0 + rs;
        const bool live = r < n && col_live;
        float l[4] = {0.f, 0.f, 0.f, 0.f};
        float pr = 0.f;
        if (live) {
            const float4 t = *reinterpret_cast<const float4*>(tp + (int64_t)r * NS + c0);
            l[0] = t.x; l[1] = t.y; l[2] = t.z; l[3] = t.w;
            pr = pR[r];
        }
        float row_part = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            const bool keep = live && c < n && c != r;        // padding and self-relations contribute 0 (:112)
            float v = fminf(l[j], 0.f);                       // :194
            if (any_neg) v = dfol_pnot(v, alpha_n, cn);       // :212-213
            if (wantR) {
                const float u = v + pc[j];                    // :102
                const float t = idC ? u : dfol_pnot(u, qC, kC);   // :108
                row_part += keep ? t : 0.f;
            }
            if (wantC) {
                const float u = v + pr;
                const float t = idR ? u : dfol_pnot(u, qR, kR);
                col_acc[j] += keep ? t : 0.f;
            }
        }
        if (wantR) {
            row_part = dfol_group_sum<LPR>(row_part);
            if (cg == LPR - 1 && r < n) row_sum[wave_in_block][r] = row_part;
        }
    }
    if (wantR) {
        __builtin_amdgcn_wave_barrier();                 // LDS is in-order within a wave; this only pins the schedule
        for (int c = lane; c < NS; c += 64) {
            float o = 0.f;
            if (c < n) {
                const float s = row_sum[wave_in_block][c];
                o = pR[c] + (idC ? s : dfol_pnot(s, qC, kC));          // :133, :138
            }
            oR[c] = o;
        }
    }

    if (wantC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int m = 32; m >= LPR; m >>= 1) col_acc[j] += __shfl_xor(col_acc[j], m, 64);
        }
        if (rs == 0 && c0 < NS) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float s = col_acc[j];
                o[j] = (c0 + j < n) ? pc[j] + (idR ? s : dfol_pnot(s, qR, kR)) : 0.f;
            }
            *reinterpret_cast<float4*>(oC + c0) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ---- NS > 256 ------------------------------------------------------------------------------------------------------------------------
// The kernels above keep a whole tile row in the registers of one wavefront (a lane owns four columns: NS <= 256).  Wider tiles - scenes
// of more than 256 objects, which no configuration of the reference reaches - take these plain forms of the general code: one workgroup
// per predicate, a thread per column for the column sums (rows in order, coalesced), a workgroup reduction per row for the row sums
// (wavefront partials added in wavefront order): deterministic, the same formulas, no fast paths.
__device__ __forceinline__ float relate_prep(float ll, int any_neg, float alpha_n, float cn) {
    float v = fminf(ll, 0.f);                                            // batch_base_ops.py:194
    if (any_neg) v = dfol_pnot(v, alpha_n, cn);                          // :212-213
    return v;
}

__global__ __launch_bounds__(256) void relate_big_fwd_kernel(
    const float* __restrict__ prior_R, const float* __restrict__ prior_C, const float* __restrict__ tile,
    const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj, const float* __restrict__ quant_R,
    const float* __restrict__ quant_C, const uint8_t* __restrict__ neg, int any_neg, const uint8_t* __restrict__ active,
    const uint8_t* __restrict__ want, int want_R_bit, int want_C_bit, int NS, int flags, float* __restrict__ post_R,
    float* __restrict__ post_C) {
    __shared__ float part[4];
    const int p = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int q = pred_q[p], n = n_obj[q];
    const int wbits = want ? want[p] : 3;
    const bool wantR = (wbits & want_R_bit) && post_R, wantC = (wbits & want_C_bit) && post_C;
    const float* pR = prior_R + (int64_t)q * NS;
    const float* pC = prior_C + (int64_t)q * NS;
    float* oR = post_R ? post_R + (int64_t)p * NS : nullptr;
    float* oC = post_C ? post_C + (int64_t)p * NS : nullptr;
    if (active && !active[p]) {                          // no-op predicate: posterior = prior (batch_base_ops.py:563-564)
        for (int c = tid; c < NS; c += 256) {
            if (oR) oR[c] = (wantR && c < n) ? pR[c] : 0.f;
            if (oC) oC[c] = (wantC && c < n) ? pC[c] : 0.f;
        }
        return;
    }
    if (!wantR && oR)
        for (int c = tid; c < NS; c += 256) oR[c] = 0.f;
    if (!wantC && oC)
        for (int c = tid; c < NS; c += 256) oC[c] = 0.f;
    const int identity_forall = flags & DFOL_RELATE_LONE_FORALL_IDENTITY;
    const float alpha_n = (any_neg && neg[p]) ? 1.f : 0.f, cn = 1.f - 2.f * alpha_n;
    const float qR = quant_R[p], qC = quant_C[p], kR = 1.f - 2.f * qR, kC = 1.f - 2.f * qC;
    const bool idR = identity_forall && qR == 0.f, idC = identity_forall && qC == 0.f;
    const float* tp = tile + (int64_t)p * NS * NS;
    if (wantC)
        for (int c = tid; c < NS; c += 256) {
            float acc = 0.f;
            if (c < n) {
                const float pc = pC[c];
                for (int r = 0; r < n; ++r) {
                    if (r == c) continue;                                 // self-relations contribute 0 (:112)
                    const float u = relate_prep(tp[(int64_t)r * NS + c], any_neg, alpha_n, cn) + pR[r];
                    acc += idR ? u : dfol_pnot(u, qR, kR);
                }
                acc = pc + (idR ? acc : dfol_pnot(acc, qR, kR));
            }
            oC[c] = acc;
        }
    if (wantR)
        for (int r = 0; r < NS; ++r) {
            float s = 0.f;
            if (r < n)
                for (int c = tid; c < n; c += 256) {
                    if (c == r) continue;
                    const float u = relate_prep(tp[(int64_t)r * NS + c], any_neg, alpha_n, cn) + pC[c];     // :102
                    s += idC ? u : dfol_pnot(u, qC, kC);                                                 // :108
                }
            s = dfol_wave_sum(s);
            if (lane == 0) part[w] = s;
            __syncthreads();
            if (tid == 0) {
                const float tot = ((part[0] + part[1]) + part[2]) + part[3];
                oR[r] = r < n ? pR[r] + (idC ? tot : dfol_pnot(tot, qC, kC)) : 0.f;                     // :133, :138
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(256) void relate_one_big_kernel(
    const float* __restrict__ x_att, const float* __restrict__ prev_att, const float* __restrict__ tile,
    const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj, const float* __restrict__ quant_prev,
    const uint8_t* __restrict__ neg, int any_neg, const uint8_t* __restrict__ active, int NS, int identity_forall,
    float* __restrict__ post) {
    const int p = blockIdx.x, tid = threadIdx.x;
    const int q = pred_q[p], n = n_obj[q];
    const float* pv = prev_att + (int64_t)q * NS;
    float* out = post + (int64_t)p * NS;
    if (active && !active[p]) {                          // question without this operator: attention passes through
        for (int c = tid; c < NS; c += 256) out[c] = c < n ? pv[c] : 0.f;
        return;
    }
    const bool negated = any_neg && neg[p];
    const float alpha_n = negated ? 1.f : 0.f, cn = 1.f - 2.f * alpha_n;
    const float qf = quant_prev[p], kf = 1.f - 2.f * qf;
    const bool ident = identity_forall && qf == 0.f;
    const bool mask = negated || qf != 1.f;              // (as relate_one_fwd_kernel: only then can the diagonal contribute)
    const float* tp = tile + (int64_t)p * NS * NS;
    for (int c = tid; c < NS; c += 256) {
        float acc = 0.f;
        if (c < n) {
            for (int r = 0; r < n; ++r) {
                if (mask && r == c) continue;
                const float u = relate_prep(tp[(int64_t)r * NS + c], any_neg, alpha_n, cn) + pv[r];
                acc += ident ? u : dfol_pnot(u, qf, kf);
            }
            acc = x_att[(int64_t)p * NS + c] + (ident ? acc : dfol_pnot(acc, qf, kf));
        }
        out[c] = acc;
    }
}

template <int LPR>
static void launch_relate(hipStream_t st, const float* pR, const float* pC, const float* tile, const int32_t* pred_q,
                          const int32_t* n_obj, const float* qR, const float* qC, const uint8_t* neg, int any_neg,
                          const uint8_t* active, const uint8_t* want, int wantRbit, int wantCbit, int P, int NS, int flags,
                          float* oR, float* oC) {
    // row partials of the EXISTS / EXISTS form go through LDS for LPR <= 32 (see relate_exists_fast): [NS][pitch] floats per wavefront, pitch odd
    static const bool rows_dpp = getenv("DFOL_RELATE_ROWS") && !strcmp(getenv("DFOL_RELATE_ROWS"), "dpp");     // A/B switch: the DPP form
    const int pitch = (LPR <= 32 && oR && !rows_dpp) ? ((NS / 4) | 1) : 0;
    const size_t lds = (size_t)4 * 64 * pitch * sizeof(float);             // <= 64 rows per chunk and wavefront: at most 33 KB per workgroup
    hipLaunchKernelGGL(relate_fwd_kernel<LPR>, dim3(dfol_cdiv(P, 4)), dim3(256), lds, st, pR, pC, tile, pred_q, n_obj, qR, qC, neg,
                       any_neg, active, want, wantRbit, wantCbit, P, NS, flags, oR, oC, pitch);
}

extern "C" int dfol_relate_fwd_f32(const float* prior_s, const float* prior_o, const float* tile, const int32_t* pred_q,
                                   const int32_t* n_obj, const float* quant_s, const float* quant_o, const uint8_t* neg,
                                   int32_t any_neg, const uint8_t* active, const uint8_t* want, int32_t P, int32_t NS,
                                   int32_t orientation, int32_t flags, float* post_s, float* post_o, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0 && NS % 4 == 0, "relate_fwd: bad sizes P=%d NS=%d (NS: a multiple of 4)", P, NS);
    DFOL_REQUIRE(orientation == 0 || orientation == 1, "relate_fwd: bad orientation %d", orientation);
    if (P == 0) return 0;
    DFOL_REQUIRE(prior_s && prior_o && tile && pred_q && n_obj && quant_s && quant_o, "relate_fwd: null pointer");
    DFOL_REQUIRE(post_s || post_o, "relate_fwd: no output requested");
    DFOL_REQUIRE(!any_neg || neg, "relate_fwd: any_neg set but neg is NULL");
    // rows/columns of the tile: R = subject, C = object for DFOL_TILE_SUBJECT_ROWS; swapped otherwise
    const bool sr = orientation == DFOL_TILE_SUBJECT_ROWS;
    const float *pR = sr ? prior_s : prior_o, *pC = sr ? prior_o : prior_s;
    const float *qR = sr ? quant_s : quant_o, *qC = sr ? quant_o : quant_s;
    float *oR = sr ? post_s : post_o, *oC = sr ? post_o : post_s;
    const int bR = sr ? DFOL_WANT_SUBJECT : DFOL_WANT_OBJECT, bC = sr ? DFOL_WANT_OBJECT : DFOL_WANT_SUBJECT;
    hipStream_t st = (hipStream_t)stream;
    if (NS > 256) {                                     // wider than a wavefront's registers hold: the plain kernel
        hipLaunchKernelGGL(relate_big_fwd_kernel, dim3(P), dim3(256), 0, st, pR, pC, tile, pred_q, n_obj, qR, qC, neg, any_neg, active, want, bR, bC,
                           NS, flags, oR, oC);
        DFOL_LAUNCH_CHECK("relate_fwd (NS > 256)");
        return 0;
    }
    const int groups = NS / 4;
#define DFOL_RELATE_CASE(L)                                                                                                   \
    launch_relate<L>(st, pR, pC, tile, pred_q, n_obj, qR, qC, neg, any_neg, active, want, bR, bC, P, NS, flags, oR, oC)
    if (groups <= 1) DFOL_RELATE_CASE(1);
    else if (groups <= 2) DFOL_RELATE_CASE(2);
    else if (groups <= 4) DFOL_RELATE_CASE(4);
    else if (groups <= 8) DFOL_RELATE_CASE(8);
    else if (groups <= 16) DFOL_RELATE_CASE(16);
    else if (groups <= 32) DFOL_RELATE_CASE(32);
    else DFOL_RELATE_CASE(64);
#undef DFOL_RELATE_CASE
    DFOL_LAUNCH_CHECK("relate_fwd");
    return 0;
}

// =====================================================================================================
// Relate, single posterior (what GQARelateBatch / verify_rel / choose_rel consume, batch_gqa_ops.py:364-371)
// =====================================================================================================
// The interpreter keeps exactly one posterior of a relate: the one of the freshly selected variable x, given the
// incoming attention `prev` of the other variable.  With the tile stored so that the SUMMED-OUT variable runs along
// rows, that posterior is a pure column reduction:
//     post[c] = x[p][c] + F( sum_{r != c} F( l'[r][c] + prev[q][r] ) ),   F by the quantifier of prev
// Lanes own 4 consecutive columns, 64/LPR rows are processed per step and UNR steps are in flight, so the loop
// body has no cross-lane traffic at all; the row slots are combined once at the end.
// Arithmetic is kept in the log2 domain inside the loop: with L = log2(e),
//     ln(max(q + k e^w, eps)) = ln2 * log2(max(q + k 2^(L w), eps)),   L w = fma(v, L, L prev[r])
// so one element costs min, fma, v_exp, fma, max, v_log, add (the ln2 factor is applied once per column at the end).
// Self-relations: for an un-negated EXISTS predicate the diagonal needs no masking at all — its raw likelihood is the
// absent value -30, so 1 - e^(-30 + prev) rounds to exactly 1.0f and contributes log2(1) = 0, as the reference's explicit
// zeroing does (batch_base_ops.py:112).  Only FOR_ALL / negated predicates take the masked variant (wave-uniform choice).
template <int LPR, int UNR, bool MASK>
__device__ __forceinline__ void relate_one_rows(const float* __restrict__ tp, const float* __restrict__ pv, int NS, int cl, int c0,
                                                int rs, int r0, int n, bool tail, int any_neg, float alpha_n, float cn, float qf,
                                                float kf, bool ident, float (&acc)[4]) {
    constexpr int RPI = 64 / LPR;
    constexpr float L2E = 1.44269504088896340736f;
    float4 t[UNR];
    float pr2[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {                      // all loads of the step first
        const int r = tail ? min(r0 + u * RPI + rs, n - 1) : r0 + u * RPI + rs;
        t[u] = *reinterpret_cast<const float4*>(tp + (int64_t)r * NS + cl);
        pr2[u] = pv[r] * L2E;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int r = r0 + u * RPI + rs;
        const float l[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
        const int d = r - c0;                            // column j of this lane is the diagonal iff d == j
        const bool row_ok = !tail || r < n;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = fminf(l[j], 0.f);
            if (any_neg) v = dfol_pnot(v, alpha_n, cn);
            const float w2 = fmaf(v, L2E, pr2[u]);       // log2(e) * (l' + prev[r])
            float f;
            if (ident) f = w2;
            else f = __builtin_amdgcn_logf(fmaxf(fmaf(kf, __builtin_amdgcn_exp2f(w2), qf), DFOL_EPS));
            if (MASK) f = (row_ok && d != j) ? f : 0.f;
            else if (tail) f = row_ok ? f : 0.f;
            acc[j] += f;
        }
    }
}

// Fast forms of the four predicate kinds: no transcendental pair per element.
//   EXISTS         :  1 - prod_r (1 - E[r,c] Pr[r]) kept as the complement q[c] (dfol_or, dfol_common.h): per element one exp, one
//                     multiply, the diagonal select and the two-instruction OR; no logarithm and no cancellation
//   negated EXISTS :  sum_r log(1 - (1 - E) Pr)        -> the complement form (dfol_or) on y = (1 - E) Pr, diagonal masked (its 1 - E
//                     is 1, not 0); acc holds q, not a log-domain sum, and the caller combines the row slots with OR
//   FOR_ALL        :  sum_r log(max(e^(l' + prev), eps)) = sum_r (l' + prev[r]) as long as no term is below log eps:
//       un-negated :  l' = min(l, 0): NO transcendental at all; the smallest l' + prev is tracked and a value at the clamp
//                     sends the predicate to the general code;
//       negated    :  l' = log(1 - E): the sum over rows is the log of a product of (1 - E) factors (each 0 or >= 2^-24) plus
//                     the sum of the priors; the smallest (1 - E) Pr is tracked against eps.
// (The inner clamp of the negation form, log(max(a, eps)), cannot change a factor: a < eps makes a * Pr vanish against 1 either
// way, and for FOR_ALL it is caught by the tracked minimum.)  Returns true when a clamp may have fired: the caller then redoes
// the predicate with relate_one_rows, which reproduces the reference's clamped value.
template <int LPR, bool NEG, bool FORALL>
__device__ __forceinline__ bool relate_one_masked_fast(const float* __restrict__ tp, const float* __restrict__ pv, int NS, int n,
                                                       int cl, int c0, int rs, float (&acc)[4]) {
    constexpr int RPI = 64 / LPR, UNR = RelateUnroll<LPR>::value;
    constexpr float L2E = 1.44269504088896340736f;
    float prmax = -1.f, chk = (FORALL && !NEG) ? 0.f : 1.f, psum = 0.f;
    for (int r0 = 0; r0 < n; r0 += RPI * UNR) {
        float4 t[UNR];
        float Pr[UNR], pr2[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int r = r0 + u * RPI + rs, rc = min(r, n - 1);
            t[u] = *reinterpret_cast<const float4*>(tp + (int64_t)rc * NS + cl);
            const float pr = pv[rc];
            prmax = fmaxf(prmax, pr);
            pr2[u] = pr * L2E;
            Pr[u] = (FORALL && !NEG) ? 0.f : (r < n ? __builtin_amdgcn_exp2f(pr2[u]) : 0.f);
            if (FORALL && NEG) psum += r < n ? pr2[u] : 0.f;
        }
        float prod[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int r = r0 + u * RPI + rs;
            const int d = r - c0;                            // column j of this lane is the diagonal iff d == j
            const bool row_ok = r < n;
            const float l[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool keep = row_ok && d != j;
                const float v2 = fminf(l[j] * L2E, 0.f);     // :194, log2 domain
                if (FORALL && !NEG) {
                    const float u2 = v2 + pr2[u];
                    chk = fminf(chk, keep ? u2 : 0.f);
                    acc[j] += keep ? u2 : 0.f;
                } else {
                    const float e = __builtin_amdgcn_exp2f(v2);
                    const float a = NEG ? 1.f - e : e;                         // e^{l'} (negated: :212-213)
                    if (FORALL) {
                        chk = fminf(chk, keep ? a * Pr[u] : 1.f);
                        prod[j] *= keep ? a : 1.f;
                    } else {
                        acc[j] = dfol_or(acc[j], (d != j) ? a * Pr[u] : 0.f);  // complement form (dfol_or); padding rows carry Pr = 0
                    }
                }
            }
        }
        if (FORALL && NEG) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += __builtin_amdgcn_logf(prod[j]);
        }
    }
    if (FORALL && NEG) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += psum;          // every row's prior once; the diagonal row's is taken off by the caller
    }
    bool bad = prmax > 0.f;
    if (FORALL) bad |= NEG ? chk < 1.2e-20f : chk < -66.3f;   // eps = 1e-20, log2(eps) = -66.44
    return bad;
}

template <int LPR, int UNR>
__global__ __launch_bounds__(256) void relate_one_fwd_kernel(
    const float* __restrict__ x_att, const float* __restrict__ prev_att, const float* __restrict__ tile,
    const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj, const float* __restrict__ quant_prev,
    const uint8_t* __restrict__ neg, int any_neg, const uint8_t* __restrict__ active, int P, int NS, int identity_forall,
    float* __restrict__ post) {
    constexpr int RPI = 64 / LPR;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int lane = threadIdx.x & 63;
    const int q = pred_q[p];
    const int n = n_obj[q];
    const float* pv = prev_att + (int64_t)q * NS;
    float* out = post + (int64_t)p * NS;
    if (active && !active[p]) {                         // question without this operator: attention passes through
        for (int c = lane; c < NS; c += 64) out[c] = c < n ? pv[c] : 0.f;
        return;
    }
    const int cg = lane % LPR, rs = lane / LPR, c0 = cg * 4;
    const bool negated = any_neg && neg[p];
    const float alpha_n = negated ? 1.f : 0.f, cn = 1.f - 2.f * alpha_n;
    const float qf = quant_prev[p], kf = 1.f - 2.f * qf;
    const bool ident = identity_forall && qf == 0.f;
    const bool mask = negated || qf != 1.f;              // see the note above: only then can the diagonal contribute
    const float* tp = tile + (int64_t)p * NS * NS;
    const int cl = min(c0, NS - 4);                      // lanes beyond the tile width read a valid column and are discarded
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int STEP = RPI * UNR;
    const int n_full = (n / STEP) * STEP;
    if (qf == 1.f || qf == 0.f) {
        const bool forall = qf == 0.f;
        bool bad = forall ? (negated ? relate_one_masked_fast<LPR, true, true>(tp, pv, NS, n, cl, c0, rs, acc)
                                     : relate_one_masked_fast<LPR, false, true>(tp, pv, NS, n, cl, c0, rs, acc))
                          : (negated ? relate_one_masked_fast<LPR, true, false>(tp, pv, NS, n, cl, c0, rs, acc)
                                     : relate_one_masked_fast<LPR, false, false>(tp, pv, NS, n, cl, c0, rs, acc));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int m = 32; m >= LPR; m >>= 1) {
                const float other = __shfl_xor(acc[j], m, 64);
                acc[j] = forall ? acc[j] + other : dfol_or(acc[j], other);
            }
            bad |= (c0 + j < n) && !(acc[j] >= -3.0e38f);
        }
        if (!__any(bad)) {
            if (rs == 0 && c0 < NS) {
                constexpr float L2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
                const float4 xa = *reinterpret_cast<const float4*>(x_att + (int64_t)p * NS + c0);
                const float4 pd = *reinterpret_cast<const float4*>(pv + c0);
                const float xv[4] = {xa.x, xa.y, xa.z, xa.w}, pdv[4] = {pd.x, pd.y, pd.z, pd.w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float s2 = acc[j];
                    if (forall && negated) s2 -= pdv[j] * L2E;                 // the diagonal row's prior is not part of the sum
                    const float sv = s2 * LN2;
                    o[j] = (c0 + j < n) ? xv[j] + (!forall ? dfol_slog(acc[j]) : ident ? sv : dfol_pnot(sv, qf, kf)) : 0.f;
                }
                *reinterpret_cast<float4*>(out + c0) = make_float4(o[0], o[1], o[2], o[3]);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = 0.f;
    }
    if (mask) {
        for (int r0 = 0; r0 < n_full; r0 += STEP)
            relate_one_rows<LPR, UNR, true>(tp, pv, NS, cl, c0, rs, r0, n, false, any_neg, alpha_n, cn, qf, kf, ident, acc);
        if (n_full < n) relate_one_rows<LPR, UNR, true>(tp, pv, NS, cl, c0, rs, n_full, n, true, any_neg, alpha_n, cn, qf, kf, ident, acc);
    } else {
        for (int r0 = 0; r0 < n_full; r0 += STEP)
            relate_one_rows<LPR, UNR, false>(tp, pv, NS, cl, c0, rs, r0, n, false, any_neg, alpha_n, cn, qf, kf, ident, acc);
        if (n_full < n) relate_one_rows<LPR, UNR, false>(tp, pv, NS, cl, c0, rs, n_full, n, true, any_neg, alpha_n, cn, qf, kf, ident, acc);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 32; m >= LPR; m >>= 1) acc[j] += __shfl_xor(acc[j], m, 64);
    if (rs == 0 && c0 < NS) {
        const float4 xa = *reinterpret_cast<const float4*>(x_att + (int64_t)p * NS + c0);
        const float xv[4] = {xa.x, xa.y, xa.z, xa.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = acc[j] * (ident ? 0.69314718055994530942f / 1.0f : 0.69314718055994530942f);   // back from log2 to ln
            o[j] = (c0 + j < n) ? xv[j] + (ident ? s : dfol_pnot(s, qf, kf)) : 0.f;
        }
        *reinterpret_cast<float4*>(out + c0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int dfol_relate_one_fwd_f32(const float* x_att, const float* prev_att, const float* tile, const int32_t* pred_q,
                                       const int32_t* n_obj, const float* quant_prev, const uint8_t* neg, int32_t any_neg,
                                       const uint8_t* active, int32_t P, int32_t NS, int32_t lone_forall_identity, float* post,
                                       void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0 && NS % 4 == 0, "relate_one_fwd: bad sizes P=%d NS=%d (NS: a multiple of 4)", P, NS);
    if (P == 0) return 0;
    DFOL_REQUIRE(x_att && prev_att && tile && pred_q && n_obj && quant_prev && post, "relate_one_fwd: null pointer");
    DFOL_REQUIRE(!any_neg || neg, "relate_one_fwd: any_neg set but neg is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (NS > 256) {
        hipLaunchKernelGGL(relate_one_big_kernel, dim3(P), dim3(256), 0, st, x_att, prev_att, tile, pred_q, n_obj, quant_prev, neg, any_neg, active, NS,
                           lone_forall_identity, post);
        DFOL_LAUNCH_CHECK("relate_one_fwd (NS > 256)");
        return 0;
    }
    const int groups = NS / 4;
#define DFOL_REL1(L, U)                                                                                                       \
    hipLaunchKernelGGL((relate_one_fwd_kernel<L, U>), dim3(dfol_cdiv(P, 4)), dim3(256), 0, st, x_att, prev_att, tile, pred_q, n_obj, \
                       quant_prev, neg, any_neg, active, P, NS, lone_forall_identity, post)
    if (groups <= 1) DFOL_REL1(1, 1);
    else if (groups <= 2) DFOL_REL1(2, 1);
    else if (groups <= 4) DFOL_REL1(4, 1);
    else if (groups <= 8) DFOL_REL1(8, 2);
    else if (groups <= 16) DFOL_REL1(16, 4);
    else if (groups <= 32) DFOL_REL1(32, 4);
    else DFOL_REL1(64, 4);
#undef DFOL_REL1
    DFOL_LAUNCH_CHECK("relate_one_fwd");
    return 0;
}

// =====================================================================================================
// Relate, single posterior, bf16 tiles (BASELINE configs[4]: 256-object scenes, tiles stored in bf16 to halve the HBM bytes)
// =====================================================================================================
// Same contract as relate_one_fwd_kernel; a lane owns EIGHT consecutive columns (one 16-byte load = 8 bf16 per row), so LPR =
// NS / 8 lanes cover a row (NS a multiple of 8, <= 256).  Arithmetic is fp32; only the stored likelihoods are rounded.
__device__ __forceinline__ void bf16x8_to_f32(const uint4 w, float (&l)[8]) {
    const uint32_t v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        l[2 * i] = __uint_as_float(v[i] << 16);
        l[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void relate_one_bf16_kernel(
    const float* __restrict__ x_att, const float* __restrict__ prev_att, const uint16_t* __restrict__ tile,
    const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj, const float* __restrict__ quant_prev,
    const uint8_t* __restrict__ neg, int any_neg, const uint8_t* __restrict__ active, int P, int NS, int identity_forall,
    float* __restrict__ post) {
    constexpr int RPI = 64 / LPR, UNR = LPR >= 32 ? 4 : (LPR >= 8 ? 2 : 1);
    constexpr float L2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int lane = threadIdx.x & 63;
    const int q = pred_q[p];
    const int n = n_obj[q];
    const float* pv = prev_att + (int64_t)q * NS;
    float* out = post + (int64_t)p * NS;
    if (active && !active[p]) {
        for (int c = lane; c < NS; c += 64) out[c] = c < n ? pv[c] : 0.f;
        return;
    }
    const int cg = lane % LPR, rs = lane / LPR, c0 = cg * 8;
    const bool negated = any_neg && neg[p];
    const float alpha_n = negated ? 1.f : 0.f, cn = 1.f - 2.f * alpha_n;
    const float qf = quant_prev[p], kf = 1.f - 2.f * qf;
    const bool ident = identity_forall && qf == 0.f;
    const bool mask = negated || qf != 1.f;
    const uint16_t* tp = tile + (int64_t)p * NS * NS + min(c0, NS - 8);     // lanes beyond the tile width read valid columns and are discarded
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    bool fast_ok = false;
    if (!mask) {                                            // complement form (dfol_or), see relate_one_masked_fast
        float pmax = 0.f;
        for (int r0 = 0; r0 < n; r0 += RPI * UNR) {
            uint4 t[UNR];
            float Pr[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int r = r0 + u * RPI + rs, rc = min(r, n - 1);
                t[u] = *reinterpret_cast<const uint4*>(tp + (int64_t)rc * NS);
                Pr[u] = r < n ? __builtin_amdgcn_exp2f(pv[rc] * L2E) : 0.f;
                pmax = fmaxf(pmax, Pr[u]);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                float l[8];
                bf16x8_to_f32(t[u], l);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[j] = dfol_or(acc[j], (r0 + u * RPI + rs == c0 + j) ? 0.f : __builtin_amdgcn_exp2f(fminf(l[j] * L2E, 0.f)) * Pr[u]);      // :112
            }
        }
        fast_ok = !__any(pmax > 1.f);
        if (fast_ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int m = 32; m >= LPR; m >>= 1) acc[j] = dfol_or(acc[j], __shfl_xor(acc[j], m, 64));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        }
    }
    if (!fast_ok) {                                         // general path: clamps, negation, FOR_ALL, explicit diagonal mask
        for (int r0 = 0; r0 < n; r0 += RPI) {
            const int r = r0 + rs, rc = min(r, n - 1);
            float l[8];
            bf16x8_to_f32(*reinterpret_cast<const uint4*>(tp + (int64_t)rc * NS), l);
            const float pr2 = pv[rc] * L2E;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = fminf(l[j], 0.f);
                if (any_neg) v = dfol_pnot(v, alpha_n, cn);
                const float w2 = fmaf(v, L2E, pr2);
                const float f = ident ? w2 : __builtin_amdgcn_logf(fmaxf(fmaf(kf, __builtin_amdgcn_exp2f(w2), qf), DFOL_EPS));
                acc[j] += (r < n && r != c0 + j) ? f : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int m = 32; m >= LPR; m >>= 1) acc[j] += __shfl_xor(acc[j], m, 64);
    }
    if (rs == 0 && c0 < NS) {
        float o[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 xa = *reinterpret_cast<const float4*>(x_att + (int64_t)p * NS + c0 + 4 * h);
            const float xv[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float s2 = acc[4 * h + j] * LN2;
                float r = 0.f;
                if (c0 + 4 * h + j < n) r = xv[j] + (fast_ok ? dfol_slog(acc[4 * h + j]) : ident ? s2 : dfol_pnot(s2, qf, kf));
                o[4 * h + j] = r;
            }
            *reinterpret_cast<float4*>(out + c0 + 4 * h) = make_float4(o[4 * h], o[4 * h + 1], o[4 * h + 2], o[4 * h + 3]);
        }
    }
}

extern "C" int dfol_relate_one_fwd_bf16(const float* x_att, const float* prev_att, const uint16_t* tile, const int32_t* pred_q,
                                        const int32_t* n_obj, const float* quant_prev, const uint8_t* neg, int32_t any_neg,
                                        const uint8_t* active, int32_t P, int32_t NS, int32_t lone_forall_identity, float* post,
                                        void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0 && NS % 8 == 0 && NS <= 256, "relate_one_fwd_bf16: bad sizes P=%d NS=%d (NS: multiple of 8, <= 256)", P, NS);
    if (P == 0) return 0;
    DFOL_REQUIRE(x_att && prev_att && tile && pred_q && n_obj && quant_prev && post, "relate_one_fwd_bf16: null pointer");
    DFOL_REQUIRE(!any_neg || neg, "relate_one_fwd_bf16: any_neg set but neg is NULL");
    hipStream_t st = (hipStream_t)stream;
    const int groups = NS / 8;
#define DFOL_REL1B(L)                                                                                                          \
    hipLaunchKernelGGL((relate_one_bf16_kernel<L>), dim3(dfol_cdiv(P, 4)), dim3(256), 0, st, x_att, prev_att, tile, pred_q, n_obj, \
                       quant_prev, neg, any_neg, active, P, NS, lone_forall_identity, post)
    if (groups <= 1) DFOL_REL1B(1);
    else if (groups <= 2) DFOL_REL1B(2);
    else if (groups <= 4) DFOL_REL1B(4);
    else if (groups <= 8) DFOL_REL1B(8);
    else if (groups <= 16) DFOL_REL1B(16);
    else DFOL_REL1B(32);
#undef DFOL_REL1B
    DFOL_LAUNCH_CHECK("relate_one_fwd_bf16");
    return 0;
}

// =====================================================================================================
// quantifier aggregation (Exist), gate, small vector ops
// =====================================================================================================
__global__ __launch_bounds__(256) void quantify_fwd_kernel(const float* __restrict__ att, const float* __restrict__ quant,
                                                           const int32_t* __restrict__ pred_q,
                                                           const int32_t* __restrict__ n_obj, int P, int NS,
                                                           float* __restrict__ lp) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int lane = threadIdx.x & 63;
    const int n = n_obj[pred_q[p]];
    const float qf = quant[p], k = 1.f - 2.f * qf;
    const float* a = att + (int64_t)p * NS;
    if (qf == 1.f) {                                                     // EXISTS: the complement form (dfol_or, dfol_common.h)
        float q1 = 0.f;
        for (int o = lane; o < n; o += 64) q1 = dfol_or(q1, fminf(dfol_exp(a[o]), 1.f));
        q1 = dfol_group_or<64>(q1);
        if (lane == 63) lp[p] = dfol_slog(q1);                           // :116-123
        return;
    }
    float s = 0.f;
    for (int o = lane; o < n; o += 64) s += dfol_pnot(a[o], qf, k);      // batch_base_types.py:116
    s = dfol_wave_sum(s);                                                // :118-121 (bom sum)
    if (lane == 0) lp[p] = dfol_pnot(s, qf, k);                          // :123
}

// NS a multiple of 4 (every block the interpreter builds): LPP = the power of two >= NS / 4 lanes cover one predicate with ONE 16-byte
// load each, 64 / LPP predicates share a wavefront and QUANT_UNR such groups are in flight per wavefront (one wavefront per 400-byte
// block - the kernel above - ran at 0.17 of the HBM peak on HBM-sized inputs).  EXISTS predicates aggregate in the complement form
// q <- q + y - q y (dfol_or): no cancellation, one transcendental per element.
constexpr int QUANT_UNR = 4;

template <int LPP>
__global__ __launch_bounds__(256) void quantify_fwd4_kernel(const float* __restrict__ att, const float* __restrict__ quant,
                                                            const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj,
                                                            int P, int NS, float* __restrict__ lp) {
    constexpr int PPW = 64 / LPP;
    constexpr float L2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int sub = lane / LPP, c0 = (lane % LPP) * 4;
    const bool col_ok = c0 < NS;
    float4 v[QUANT_UNR];
    int pp[QUANT_UNR], n[QUANT_UNR];
    float qf[QUANT_UNR];
#pragma unroll
    for (int u = 0; u < QUANT_UNR; ++u) {
        const int p = (wave * QUANT_UNR + u) * PPW + sub;
        pp[u] = p;
        const int pc = min(p, P - 1);
        n[u] = n_obj[pred_q[pc]];
        qf[u] = quant[pc];
        v[u] = *reinterpret_cast<const float4*>(att + (int64_t)pc * NS + (col_ok ? c0 : 0));
    }
#pragma unroll
    for (int u = 0; u < QUANT_UNR; ++u) {
        const float a[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        const float k = 1.f - 2.f * qf[u];
        if (qf[u] == 1.f) {                                              // EXISTS: the complement form (dfol_or, dfol_common.h)
            float q1 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = (col_ok && c0 + j < n[u]) ? fminf(__builtin_amdgcn_exp2f(a[j] * L2E), 1.f) : 0.f;
                q1 = dfol_or(q1, y);
            }
            q1 = dfol_group_or<LPP>(q1);                                 // :118-121 (valid in the last lane of the group)
            if (lane % LPP == LPP - 1 && pp[u] < P) lp[pp[u]] = __builtin_amdgcn_logf(fmaxf(q1, DFOL_EPS)) * LN2;      // :123
        } else {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += (col_ok && c0 + j < n[u]) ? dfol_pnot(a[j], qf[u], k) : 0.f;      // batch_base_types.py:116
            s = dfol_group_sum<LPP>(s);
            if (lane % LPP == LPP - 1 && pp[u] < P) lp[pp[u]] = dfol_pnot(s, qf[u], k);
        }
    }
}

extern "C" int dfol_quantify_fwd_f32(const float* att, const float* quant, const int32_t* pred_q, const int32_t* n_obj,
                                     int32_t P, int32_t NS, float* lp, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0, "quantify_fwd: bad sizes P=%d NS=%d", P, NS);
    if (P == 0) return 0;
    DFOL_REQUIRE(att && quant && pred_q && n_obj && lp, "quantify_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int groups = NS / 4;
#define DFOL_QUANT4(L)                                                                                                                  \
    hipLaunchKernelGGL(quantify_fwd4_kernel<L>, dim3(dfol_cdiv(P, 4 * QUANT_UNR * (64 / L))), dim3(256), 0, st, att, quant, pred_q, n_obj, P, NS, lp)
    if (NS % 4 != 0 || groups > 64)
        hipLaunchKernelGGL(quantify_fwd_kernel, dim3(dfol_cdiv(P, 4)), dim3(256), 0, st, att, quant, pred_q, n_obj, P, NS, lp);
    else if (groups <= 1) DFOL_QUANT4(1);
    else if (groups <= 2) DFOL_QUANT4(2);
    else if (groups <= 4) DFOL_QUANT4(4);
    else if (groups <= 8) DFOL_QUANT4(8);
    else if (groups <= 16) DFOL_QUANT4(16);
    else if (groups <= 32) DFOL_QUANT4(32);
    else DFOL_QUANT4(64);
#undef DFOL_QUANT4
    DFOL_LAUNCH_CHECK("quantify_fwd");
    return 0;
}

// Answer decoding on the device (util.find_max_ind, util.py:64-66): flag[p] = 1 iff predicate p attains the maximum PROBABILITY of its
// question's predicates and that probability exceeds the threshold.  One wavefront per question (its predicates are contiguous:
// seg_off); probabilities by libm expf, compared exactly, as the reference compares torch.exp values (almost_equal with eps = 0).
__global__ __launch_bounds__(256) void find_max_ind_kernel(const float* __restrict__ lp, const int32_t* __restrict__ seg_off, int Q,
                                                           float threshold, uint8_t* __restrict__ flag) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const int lane = threadIdx.x & 63;
    const int p0 = seg_off[q], p1 = seg_off[q + 1];
    float mx = 0.f;                                            // (the reference's dense [P, Q] product holds 0 for other questions' rows)
    for (int p = p0 + lane; p < p1; p += 64) mx = fmaxf(mx, expf(lp[p]));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s, 64));
    for (int p = p0 + lane; p < p1; p += 64) {
        const float v = expf(lp[p]);
        flag[p] = (v == mx && v > threshold) ? 1 : 0;
    }
}

extern "C" int dfol_find_max_ind_f32(const float* lp, const int32_t* seg_off, int32_t Q, float likelihood_threshold, uint8_t* flag,
                                     void* stream) {
    DFOL_REQUIRE(Q >= 0, "find_max_ind: bad sizes Q=%d", Q);
    if (Q == 0) return 0;
    DFOL_REQUIRE(lp && seg_off && flag, "find_max_ind: null pointer");
    hipLaunchKernelGGL(find_max_ind_kernel, dim3(dfol_cdiv(Q, 4)), dim3(256), 0, (hipStream_t)stream, lp, seg_off, Q, likelihood_threshold, flag);
    DFOL_LAUNCH_CHECK("find_max_ind");
    return 0;
}

// hard_mode aggregation (batch_base_types.py:104-112): the sum over objects becomes a minimum.  The reference takes the minimum
// over ALL object columns of (dense batch-object mask * value), so as soon as the batch holds objects of other images a 0 takes part.
__global__ __launch_bounds__(256) void quantify_hard_kernel(const float* __restrict__ att, const float* __restrict__ quant,
                                                            const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj,
                                                            int P, int NS, int total_obj, float* __restrict__ lp) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int lane = threadIdx.x & 63;
    const int n = n_obj[pred_q[p]];
    const float qf = quant[p], k = 1.f - 2.f * qf;
    const float* a = att + (int64_t)p * NS;
    float m = total_obj > n ? 0.f : __builtin_inff();
    for (int o = lane; o < n; o += 64) m = fminf(m, dfol_pnot(a[o], qf, k));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fminf(m, __shfl_xor(m, s, 64));
    if (lane == 0) lp[p] = dfol_pnot(m, qf, k);
}

extern "C" int dfol_quantify_hard_f32(const float* att, const float* quant, const int32_t* pred_q, const int32_t* n_obj, int32_t P,
                                      int32_t NS, int32_t total_obj, float* lp, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0 && total_obj >= 0, "quantify_hard: bad sizes P=%d NS=%d total_obj=%d", P, NS, total_obj);
    if (P == 0) return 0;
    DFOL_REQUIRE(att && quant && pred_q && n_obj && lp, "quantify_hard: null pointer");
    hipLaunchKernelGGL(quantify_hard_kernel, dim3(dfol_cdiv(P, 4)), dim3(256), 0, (hipStream_t)stream, att, quant, pred_q, n_obj, P, NS,
                       total_obj, lp);
    DFOL_LAUNCH_CHECK("quantify_hard");
    return 0;
}

__global__ void gate_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ xq,
                            const float* __restrict__ yq, const float* __restrict__ g, int NS, float* __restrict__ out,
                            float* __restrict__ outq) {
    const int p = blockIdx.x;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    const float gg = g[p];
    if (c < NS) {
        const int64_t i = (int64_t)p * NS + c;
        // batch_base_types.py:159; g is a 0/1 mask, for which the blend is a select (and a select cannot turn a
        // stray inf/NaN in the unselected operand into NaN the way 0 * inf would)
        out[i] = gg == 1.f ? x[i] : (gg == 0.f ? y[i] : x[i] * gg + y[i] * (1.f - gg));
    }
    if (c == 0 && outq) outq[p] = gg == 1.f ? xq[p] : (gg == 0.f ? yq[p] : xq[p] * gg + yq[p] * (1.f - gg));   // :156
}

extern "C" int dfol_gate_f32(const float* x_att, const float* y_att, const float* x_quant, const float* y_quant, const float* g,
                             int32_t P, int32_t NS, float* out_att, float* out_quant, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0, "gate: bad sizes P=%d NS=%d", P, NS);
    if (P == 0) return 0;
    DFOL_REQUIRE(x_att && y_att && g && out_att, "gate: null pointer");
    DFOL_REQUIRE(!out_quant || (x_quant && y_quant), "gate: out_quant requested without input quantifiers");
    hipLaunchKernelGGL(gate_kernel, dim3(P, dfol_cdiv(NS, 64)), dim3(64), 0, (hipStream_t)stream, x_att, y_att, x_quant, y_quant, g,
                       NS, out_att, out_quant);
    DFOL_LAUNCH_CHECK("gate");
    return 0;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int width,
                                   float* __restrict__ out) {
    const int p = blockIdx.x;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c < width) out[(int64_t)p * width + c] = src[(int64_t)idx[p] * width + c];
}

extern "C" int dfol_gather_rows_f32(const float* src, const int32_t* idx, int32_t P, int32_t width, float* out, void* stream) {
    DFOL_REQUIRE(P >= 0 && width > 0, "gather_rows: bad sizes");
    if (P == 0) return 0;
    DFOL_REQUIRE(src && idx && out, "gather_rows: null pointer");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(P, dfol_cdiv(width, 64)), dim3(64), 0, (hipStream_t)stream, src, idx, width, out);
    DFOL_LAUNCH_CHECK("gather_rows");
    return 0;
}

// ---- small pieces of the attention-calibration passes (batch_base_interpreter.py:87-140), for the native executor and the Python operators alike ------
// out[p] = flags[p] ? x[p] : y[p]  (BatchAttentionState.gate with 0 / 1 flags, batch_base_types.py:279-298: g x + (1 - g) y is a row select there)
__global__ void select_rows_kernel(const float* __restrict__ x, const float* __restrict__ y, const uint8_t* __restrict__ flags, int width,
                                   float* __restrict__ out) {
    const int p = blockIdx.x;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c < width) out[(int64_t)p * width + c] = (flags[p] ? x : y)[(int64_t)p * width + c];
}

extern "C" int dfol_select_rows_f32(const float* x, const float* y, const uint8_t* flags, int32_t P, int32_t width, float* out, void* stream) {
    DFOL_REQUIRE(P >= 0 && width > 0, "select_rows: bad sizes");
    if (P == 0) return 0;
    DFOL_REQUIRE(x && y && flags && out, "select_rows: null pointer");
    hipLaunchKernelGGL(select_rows_kernel, dim3(P, dfol_cdiv(width, 64)), dim3(64), 0, (hipStream_t)stream, x, y, flags, width, out);
    DFOL_LAUNCH_CHECK("select_rows");
    return 0;
}

// LSTM input rows of an operator's tokens (batch_base_ops.py:265-273, 437-446, 628-637): out[p] = [head (n_head floats: the operator's one-hot and the
// token-type flag) | table[idx[p]] (E floats: the token's embedding)], or an all-zero row for a no-op token (idx[p] < 0)
__global__ void calib_features_kernel(const float* __restrict__ head, int n_head, const float* __restrict__ table, int E, const int32_t* __restrict__ idx,
                                      float* __restrict__ out) {
    const int p = blockIdx.x, F = n_head + E;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= F) return;
    const int t = idx[p];
    out[(int64_t)p * F + c] = t < 0 ? 0.f : (c < n_head ? head[c] : table[(int64_t)t * E + (c - n_head)]);
}

extern "C" int dfol_calib_features_f32(const float* head, int32_t n_head, const float* table, int32_t E, const int32_t* idx, int32_t P, float* out,
                                       void* stream) {
    DFOL_REQUIRE(P >= 0 && n_head >= 0 && E >= 0 && n_head + E > 0, "calib_features: bad sizes");
    if (P == 0) return 0;
    DFOL_REQUIRE((head || n_head == 0) && (table || E == 0) && idx && out, "calib_features: null pointer");
    hipLaunchKernelGGL(calib_features_kernel, dim3(P, dfol_cdiv(n_head + E, 64)), dim3(64), 0, (hipStream_t)stream, head, n_head, table, E, idx, out);
    DFOL_LAUNCH_CHECK("calib_features");
    return 0;
}

// The attention-output network on the two LSTM states of an operator (BatchOperatorBase._compute_attention_modulations, batch_base_ops.py:275-286, with
// the Linear(2 S -> N) + Sigmoid of gqa_interpreter_experiments.py:119-132): out[p][j] = Sigmoid(b[j] + W[j][:S] . fs[p] + W[j][S:] . bs[p]); a NULL
// state counts as zeros (the reference substitutes zeros_like).  Sixteen lanes per row: lane l takes the inputs l, l + 16, ... (coalesced loads, all in
// flight at once) for up to AM_N outputs, the lanes meet in a butterfly, then the bias.  (First version: one thread per output walking its hundred
// inputs one load after the other, 9 - 14 us per launch at 256 rows; five launches per calibrated forward.)
// The products first, the bias LAST: the bias is -log 9 (gqa_interpreter_experiments.py:124-126) and the hundred products are ~1e-2 each - added one
// by one onto the bias they are each rounded at ulp(2.2), 4e-6 of systematic error in a modulation of 0.1 (found by golden g23).
__global__ __launch_bounds__(256) void attention_modulations_kernel(const float* __restrict__ fs, const float* __restrict__ bs, const float* __restrict__ W,
                                                                    int64_t ld_w, const float* __restrict__ b, int P, int S, int N, float* __restrict__ out) {
    am_rows(fs, bs, W, ld_w, b, P, S, N, out, blockIdx.x * 16, threadIdx.x);
}

extern "C" int dfol_attention_modulations_f32(const float* fs, const float* bs, const float* W, int64_t ld_w, const float* b, int32_t P, int32_t S,
                                              int32_t N, float* out, void* stream) {
    DFOL_REQUIRE(P >= 0 && S > 0 && N > 0 && ld_w >= 2 * S, "attention_modulations: bad sizes P=%d S=%d N=%d", P, S, N);
    if (P == 0) return 0;
    DFOL_REQUIRE(W && out, "attention_modulations: null pointer");
    hipLaunchKernelGGL(attention_modulations_kernel, dim3(dfol_cdiv(P, 16)), dim3(256), 0, (hipStream_t)stream, fs, bs, W, ld_w, b, P, S, N, out);
    DFOL_LAUNCH_CHECK("attention_modulations");
    return 0;
}

__global__ void segment_sum_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ seg_off, int width,
                                        float* __restrict__ out) {
    const int q = blockIdx.x;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= width) return;
    float s = 0.f;
    for (int p = seg_off[q]; p < seg_off[q + 1]; ++p) s += src[(int64_t)p * width + c];
    out[(int64_t)q * width + c] = s;
}

extern "C" int dfol_segment_sum_rows_f32(const float* src, const int32_t* seg_off, int32_t Q, int32_t width, float* out,
                                         void* stream) {
    DFOL_REQUIRE(Q >= 0 && width > 0, "segment_sum_rows: bad sizes");
    if (Q == 0) return 0;
    DFOL_REQUIRE(src && seg_off && out, "segment_sum_rows: null pointer");
    hipLaunchKernelGGL(segment_sum_rows_kernel, dim3(Q, dfol_cdiv(width, 64)), dim3(64), 0, (hipStream_t)stream, src, seg_off,
                       width, out);
    DFOL_LAUNCH_CHECK("segment_sum_rows");
    return 0;
}

// out[ucols[u]][c] (+)= sum over the slots k of segment u of rows[order[k]][c], added in slot order starting from zero - what gather_rows,
// segment_sum_rows and an index_copy into a zeroed matrix compute in three launches (visual_oracle._combine_concept_rows)
__global__ void concept_rows_kernel(const float* __restrict__ rows, int64_t ld_rows, const int32_t* __restrict__ order,
                                    const int32_t* __restrict__ seg_off, const int64_t* __restrict__ ucols, int width, float* __restrict__ out,
                                    int64_t ld_out, int accumulate) {
    const int u = blockIdx.x;
    const int k0 = seg_off[u], k1 = seg_off[u + 1];
    float* __restrict__ dst = out + ucols[u] * ld_out;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= width) return;
    // four rows' loads in flight, added in slot order (the chain of a segment's rows is latency-bound: a load per row, each behind its index)
    float s = 0.f;
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
        const int r0 = order[k], r1 = order[k + 1], r2 = order[k + 2], r3 = order[k + 3];
        const float v0 = rows[(int64_t)r0 * ld_rows + c], v1 = rows[(int64_t)r1 * ld_rows + c], v2 = rows[(int64_t)r2 * ld_rows + c],
                    v3 = rows[(int64_t)r3 * ld_rows + c];
        s = (((s + v0) + v1) + v2) + v3;
    }
    for (; k < k1; ++k) s += rows[(int64_t)order[k] * ld_rows + c];
    dst[c] = accumulate ? dst[c] + s : s;
}

extern "C" int dfol_concept_rows_f32(const float* rows, int64_t ld_rows, const int32_t* order, const int32_t* seg_off, const int64_t* ucols, int32_t U,
                                     int32_t width, float* out, int64_t ld_out, int32_t accumulate, void* stream) {
    DFOL_REQUIRE(U >= 0 && width > 0 && ld_rows >= width && ld_out >= width, "concept_rows: bad sizes");
    if (U == 0) return 0;
    DFOL_REQUIRE(rows && order && seg_off && ucols && out, "concept_rows: null pointer");
    hipLaunchKernelGGL(concept_rows_kernel, dim3(U, dfol_cdiv(width, 64)), dim3(64), 0, (hipStream_t)stream, rows, ld_rows, order, seg_off, ucols, width,
                       out, ld_out, accumulate);
    DFOL_LAUNCH_CHECK("concept_rows");
    return 0;
}

__global__ void logic_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i];
    float r;
    if (op == DFOL_LOGIC_AND) r = x + b[i];                                                            // util.py:29-30
    else if (op == DFOL_LOGIC_OR) r = dfol_slog(dfol_or(dfol_exp(x), dfol_exp(b[i])));                 // util.py:32-33: 1 - (1 - e^a)(1 - e^b) = e^a + e^b - e^a e^b
    else r = dfol_lnot(x);                                                                             // util.py:35-36
    out[i] = r;
}

extern "C" int dfol_logic_f32(int32_t op, const float* a, const float* b, int64_t n, float* out, void* stream) {
    DFOL_REQUIRE(op >= 0 && op <= 2 && n >= 0, "logic: bad arguments op=%d", op);
    if (n == 0) return 0;
    DFOL_REQUIRE(a && out && (op == DFOL_LOGIC_NOT || b), "logic: null pointer");
    hipLaunchKernelGGL(logic_kernel, dim3(dfol_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, op, a, b, n, out);
    DFOL_LAUNCH_CHECK("logic");
    return 0;
}

__global__ void parametric_not_kernel(const float* __restrict__ x, const float* __restrict__ alpha, int width,
                                      float* __restrict__ out) {
    const int r = blockIdx.x;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= width) return;
    const float a = alpha[r];
    out[(int64_t)r * width + c] = dfol_pnot(x[(int64_t)r * width + c], a, 1.f - 2.f * a);
}

extern "C" int dfol_parametric_not_f32(const float* x, const float* alpha, int32_t rows, int32_t width, float* out, void* stream) {
    DFOL_REQUIRE(rows >= 0 && width > 0, "parametric_not: bad sizes");
    if (rows == 0) return 0;
    DFOL_REQUIRE(x && alpha && out, "parametric_not: null pointer");
    hipLaunchKernelGGL(parametric_not_kernel, dim3(rows, dfol_cdiv(width, 64)), dim3(64), 0, (hipStream_t)stream, x, alpha, width, out);
    DFOL_LAUNCH_CHECK("parametric_not");
    return 0;
}

__global__ void segment_or_kernel(const float* __restrict__ lp, const int32_t* __restrict__ seg_off, int Q, float* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    float q1 = 0.f;                                      // log_not(sum_p log_not(lp_p)) in the complement form (dfol_or, dfol_common.h)
    for (int p = seg_off[q]; p < seg_off[q + 1]; ++p) q1 = dfol_or(q1, fminf(dfol_exp(lp[p]), 1.f));
    out[q] = dfol_slog(q1);
}

// The same aggregate AS THE REFERENCE WRITES IT, log_not(sum_p log_not(lp_p)) in fp32 (batch_gqa_ops.py:597-598, 664-665), for the two callers
// that NEGATE it next (all_different :631, two_different :706).  The complement form above is accurate in q; the negation needs 1 - q, which no
// fp32 form of q holds when q -> 1.  There the reference's arithmetic SATURATES - 1 - e^S rounds to exactly 1 once S < -17.3, the aggregate
// is exactly 0, its negation exactly log(1e-20) and the clamp's gradient exactly 0 - while q = 1 - 2^-24 from the complement form turned into
// log(2^-24) = -16.6 with a gradient of order one into the saturated option (golden g19 all_different_small: loss 3.52 against the
// reference's 8.42, fp32 and fp64 alike).  Same rounding behaviour as the reference is the parity that matters for that pair of operators.
__global__ void segment_or_ref_kernel(const float* __restrict__ lp, const int32_t* __restrict__ seg_off, int Q, float* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    float s = 0.f;
    for (int p = seg_off[q]; p < seg_off[q + 1]; ++p) s += dfol_lnot(lp[p]);
    out[q] = dfol_lnot(s);
}

extern "C" int dfol_segment_or_ref_f32(const float* lp, const int32_t* seg_off, int32_t Q, float* out, void* stream) {
    DFOL_REQUIRE(Q >= 0, "segment_or_ref: bad sizes");
    if (Q == 0) return 0;
    DFOL_REQUIRE(lp && seg_off && out, "segment_or_ref: null pointer");
    hipLaunchKernelGGL(segment_or_ref_kernel, dim3(dfol_cdiv(Q, 64)), dim3(64), 0, (hipStream_t)stream, lp, seg_off, Q, out);
    DFOL_LAUNCH_CHECK("segment_or_ref");
    return 0;
}

extern "C" int dfol_segment_or_f32(const float* lp, const int32_t* seg_off, int32_t Q, float* out, void* stream) {
    DFOL_REQUIRE(Q >= 0, "segment_or: bad sizes");
    if (Q == 0) return 0;
    DFOL_REQUIRE(lp && seg_off && out, "segment_or: null pointer");
    hipLaunchKernelGGL(segment_or_kernel, dim3(dfol_cdiv(Q, 64)), dim3(64), 0, (hipStream_t)stream, lp, seg_off, Q, out);
    DFOL_LAUNCH_CHECK("segment_or");
    return 0;
}

__global__ void implication_kernel(const float* __restrict__ prior, const float* __restrict__ x, const int32_t* __restrict__ pred_q,
                                   const int32_t* __restrict__ n_obj, int NS, float* __restrict__ out) {
    const int p = blockIdx.x;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= NS) return;
    const int q = pred_q[p];
    float r = 0.f;
    if (c < n_obj[q]) r = dfol_lnot(prior[(int64_t)q * NS + c] + dfol_lnot(x[(int64_t)p * NS + c]));   // batch_gqa_ops.py:588-589
    out[(int64_t)p * NS + c] = r;
}

extern "C" int dfol_implication_f32(const float* prior, const float* x, const int32_t* pred_q, const int32_t* n_obj, int32_t P,
                                    int32_t NS, float* out, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0, "implication: bad sizes");
    if (P == 0) return 0;
    DFOL_REQUIRE(prior && x && pred_q && n_obj && out, "implication: null pointer");
    hipLaunchKernelGGL(implication_kernel, dim3(P, dfol_cdiv(NS, 64)), dim3(64), 0, (hipStream_t)stream, prior, x, pred_q, n_obj, NS, out);
    DFOL_LAUNCH_CHECK("implication");
    return 0;
}

__global__ void compare_kernel(const float* __restrict__ lp1, const float* __restrict__ lp2, const float* __restrict__ is_less, int Q,
                               float* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const float a = lp1[q], b = lp2[q];
    const float m = fmaxf(a, b);
    const float lse = m + logf(expf(a - m) + expf(b - m));               // nn.LogSoftmax(dim=1), batch_gqa_ops.py:735
    const float al = is_less[q], k = 1.f - 2.f * al;
    out[2 * q + 0] = dfol_pnot(a - lse, al, k);                          // :737-738
    out[2 * q + 1] = dfol_pnot(b - lse, al, k);
}

extern "C" int dfol_compare_f32(const float* lp1, const float* lp2, const float* is_less, int32_t Q, float* out, void* stream) {
    DFOL_REQUIRE(Q >= 0, "compare: bad sizes");
    if (Q == 0) return 0;
    DFOL_REQUIRE(lp1 && lp2 && is_less && out, "compare: null pointer");
    hipLaunchKernelGGL(compare_kernel, dim3(dfol_cdiv(Q, 64)), dim3(64), 0, (hipStream_t)stream, lp1, lp2, is_less, Q, out);
    DFOL_LAUNCH_CHECK("compare");
    return 0;
}


// =====================================================================================================
// attention calibration: BatchVariableSet.apply_modulations, batch_base_types.py:170-179 (4-column modulations)
//   alpha = 10 m0, beta = 10 m1, c = 10 m2, d = m3
//   t  = alpha * a + slog(c) + slog(d)
//   a' = t - slog( exp(beta * lnot(a) + slog(1 - d)) + exp(t) )
// =====================================================================================================
__global__ void modulate_kernel(const float* __restrict__ att, const float* __restrict__ mods, const int32_t* __restrict__ pred_q,
                                const int32_t* __restrict__ n_obj, int NS, float* __restrict__ out) {
    const int p = blockIdx.x;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= NS) return;
    const int64_t i = (int64_t)p * NS + c;
    float r = 0.f;
    if (c < n_obj[pred_q[p]]) {
        const float alpha = mods[4 * p + 0] * 10.f, beta = mods[4 * p + 1] * 10.f, cc = mods[4 * p + 2] * 10.f, d = mods[4 * p + 3];
        const float a = att[i];
        const float t = alpha * a + logf(fmaxf(cc, DFOL_EPS)) + logf(fmaxf(d, DFOL_EPS));
        const float na = logf(fmaxf(1.f - expf(a), DFOL_EPS));
        const float u = beta * na + logf(fmaxf(1.f - d, DFOL_EPS));
        r = t - logf(fmaxf(expf(u) + expf(t), DFOL_EPS));
    }
    out[i] = r;
}

extern "C" int dfol_modulate_f32(const float* att, const float* mods, const int32_t* pred_q, const int32_t* n_obj, int32_t P, int32_t NS,
                                 float* out, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0, "modulate: bad sizes");
    if (P == 0) return 0;
    DFOL_REQUIRE(att && mods && pred_q && n_obj && out, "modulate: null pointer");
    hipLaunchKernelGGL(modulate_kernel, dim3(P, dfol_cdiv(NS, 64)), dim3(64), 0, (hipStream_t)stream, att, mods, pred_q, n_obj, NS, out);
    DFOL_LAUNCH_CHECK("modulate");
    return 0;
}

// Backward of apply_modulations (one wavefront per predicate): with S = e^u + e^t, r = t - log(max(S, eps)),
//   dr/dt = 1 - [S > eps] e^t / S,  dr/du = -[S > eps] e^u / S,  dt/da = alpha,  du/da = beta D(a)  (D = d log_not, 0 where its clamp is active)
//   g_att = g (dr/dt alpha + dr/du beta D(a));  g_mods[p] = (10 sum g dr/dt a, 10 sum g dr/du log_not(a), 10 sum g dr/dt [c > eps] / c,
//                                                            sum g (dr/dt [d > eps] / d - dr/du [1 - d > eps] / (1 - d)))
// sums over the predicate's objects in lane order, then across lanes: no atomics, repeatable bit for bit.
__global__ __launch_bounds__(64) void modulate_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ att, const float* __restrict__ mods,
                                                          const int32_t* __restrict__ pred_q, const int32_t* __restrict__ n_obj, int NS,
                                                          float* __restrict__ g_att, float* __restrict__ g_mods) {
    const int p = blockIdx.x, lane = threadIdx.x;
    const int n = n_obj[pred_q[p]];
    const float alpha = mods[4 * p + 0] * 10.f, beta = mods[4 * p + 1] * 10.f, cc = mods[4 * p + 2] * 10.f, d = mods[4 * p + 3];
    const float lc = logf(fmaxf(cc, DFOL_EPS)), ld = logf(fmaxf(d, DFOL_EPS)), l1d = logf(fmaxf(1.f - d, DFOL_EPS));
    const float dc = cc > DFOL_EPS ? 1.f / cc : 0.f, dd = d > DFOL_EPS ? 1.f / d : 0.f, d1d = (1.f - d) > DFOL_EPS ? 1.f / (1.f - d) : 0.f;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int c = lane; c < NS; c += 64) {
        const int64_t i = (int64_t)p * NS + c;
        float ga = 0.f;
        if (c < n) {
            const float a = att[i], g = g_out[i];
            const float ea = expf(a), om = 1.f - ea;
            const float na = logf(fmaxf(om, DFOL_EPS)), Da = om > DFOL_EPS ? -ea / om : 0.f;
            const float t = alpha * a + lc + ld, u = beta * na + l1d;
            const float et = expf(t), eu = expf(u), S = et + eu;
            const float inv = S > DFOL_EPS ? 1.f / S : 0.f;
            const float rt = 1.f - et * inv, ru = -eu * inv;
            ga = g * (rt * alpha + ru * beta * Da);
            s0 += g * rt * a;
            s1 += g * ru * na;
            s2 += g * rt * dc;
            s3 += g * (rt * dd - ru * d1d);
        }
        g_att[i] = ga;
    }
    s0 = dfol_wave_sum(s0), s1 = dfol_wave_sum(s1), s2 = dfol_wave_sum(s2), s3 = dfol_wave_sum(s3);
    if (lane == 0) {
        g_mods[4 * p + 0] = 10.f * s0, g_mods[4 * p + 1] = 10.f * s1, g_mods[4 * p + 2] = 10.f * s2, g_mods[4 * p + 3] = s3;
    }
}

extern "C" int dfol_modulate_bwd_f32(const float* g_out, const float* att, const float* mods, const int32_t* pred_q, const int32_t* n_obj,
                                     int32_t P, int32_t NS, float* g_att, float* g_mods, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0, "modulate_bwd: bad sizes");
    if (P == 0) return 0;
    DFOL_REQUIRE(g_out && att && mods && pred_q && n_obj && g_att && g_mods, "modulate_bwd: null pointer");
    hipLaunchKernelGGL(modulate_bwd_kernel, dim3(P), dim3(64), 0, (hipStream_t)stream, g_out, att, mods, pred_q, n_obj, NS, g_att, g_mods);
    DFOL_LAUNCH_CHECK("modulate_bwd");
    return 0;
}

// =====================================================================================================
// LSTM cell pointwise stage of the attention-calibration passes (batch_base_interpreter.py:87-140 run nn.LSTMCell(318 -> 50) once
// per operator and direction): gates = x W_ih^T + b_ih + h W_hh^T + b_hh come from two dfol_linear_act_f32 launches, this kernel
// does  c' = sigmoid(f) c + sigmoid(i) tanh(g),  h' = sigmoid(o) tanh(c')  with torch's gate order (i, f, g, o).
// =====================================================================================================
__global__ void lstm_pointwise_kernel(const float* __restrict__ ig, const float* __restrict__ hg, const float* __restrict__ c, int rows,
                                      int H, float* __restrict__ hy, float* __restrict__ cy) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * H) return;
    const int r = idx / H, j = idx - r * H;
    const float* a = ig + (int64_t)r * 4 * H;
    const float* b = hg + (int64_t)r * 4 * H;
    const float gi = a[j] + b[j], gf = a[H + j] + b[H + j], gg = a[2 * H + j] + b[2 * H + j], go = a[3 * H + j] + b[3 * H + j];
    const float si = 1.0f / (1.0f + expf(-gi)), sf = 1.0f / (1.0f + expf(-gf)), so = 1.0f / (1.0f + expf(-go));
    const float cn = sf * c[idx] + si * tanhf(gg);
    cy[idx] = cn;
    hy[idx] = so * tanhf(cn);
}

// The whole cell in one launch: grid (row blocks of LC_ROWS, hidden-unit slices of LC_UNITS), the pieces in dfol_calib.h.
__global__ __launch_bounds__(LC_THREADS) void lstm_cell_kernel(LcCell p) {
    extern __shared__ __attribute__((aligned(16))) float lc_s[];   // [KX + H][LC_ROWS] inputs, then [LC_SLICES][LC_ROWS][LC_PS] partial gates
    const int r0 = blockIdx.x * LC_ROWS;
    lc_stage(p, r0, lc_s);
    lc_units(p, r0, blockIdx.y * LC_UNITS, lc_s, lc_s + LC_ROWS * (p.KX + p.H));
}

static int lstm_cell_launch(const char* what, const float* x, int64_t ld_x, int32_t KX, LcTokens tk, const float* h, int64_t ld_h, const float* c,
                            const float* Wih, int64_t ld_wih, const float* Whh, int64_t ld_whh, const float* bih, const float* bhh, int32_t rows, int32_t H,
                            float* h_out, float* c_out, float* gates, void* stream) {
    DFOL_REQUIRE(rows >= 0 && H > 0 && KX > 0, "%s: bad sizes rows=%d H=%d KX=%d", what, rows, H, KX);
    const size_t lds = sizeof(float) * lc_lds_floats(KX, H);
    DFOL_REQUIRE(lds <= 64 * 1024, "%s: input width %d + hidden %d too large for the staging buffer", what, KX, H);
    if (rows == 0) return 0;
    DFOL_REQUIRE(h && c && Wih && Whh && h_out && c_out, "%s: null pointer", what);
    hipLaunchKernelGGL(lstm_cell_kernel, dim3(dfol_cdiv(rows, LC_ROWS), dfol_cdiv(H, LC_UNITS)), dim3(LC_THREADS), lds, (hipStream_t)stream,
                       LcCell{x, ld_x, KX, h, ld_h, c, Wih, ld_wih, Whh, ld_whh, bih, bhh, rows, H, h_out, c_out, gates, tk});
    DFOL_LAUNCH_CHECK(what);
    return 0;
}

extern "C" int dfol_lstm_cell_f32(const float* x, int64_t ld_x, int32_t KX, const float* h, int64_t ld_h, const float* c, const float* Wih,
                                  int64_t ld_wih, const float* Whh, int64_t ld_whh, const float* bih, const float* bhh, int32_t rows,
                                  int32_t H, float* h_out, float* c_out, void* stream) {
    DFOL_REQUIRE(x || rows == 0, "lstm_cell: null pointer");
    return lstm_cell_launch("lstm_cell", x, ld_x, KX, LcTokens{}, h, ld_h, c, Wih, ld_wih, Whh, ld_whh, bih, bhh, rows, H, h_out, c_out, nullptr, stream);
}

// The cell on an operator's TOKENS: row p of x is [head (n_head floats: the operator's one-hot and the token-type flag) | table[idx[p]] (E floats)] or
// all zeros for a no-op token (idx[p] < 0) - dfol_calib_features_f32's rows, built while they are staged instead of written and read back.
extern "C" int dfol_lstm_cell_tokens_f32(const float* head, int32_t n_head, const float* table, int32_t E, const int32_t* idx, const float* h, int64_t ld_h,
                                         const float* c, const float* Wih, int64_t ld_wih, const float* Whh, int64_t ld_whh, const float* bih, const float* bhh,
                                         int32_t rows, int32_t H, float* h_out, float* c_out, void* stream) {
    DFOL_REQUIRE(n_head >= 0 && E >= 0 && n_head + E > 0, "lstm_cell_tokens: bad sizes");
    DFOL_REQUIRE(rows == 0 || ((head || n_head == 0) && (table || E == 0) && idx), "lstm_cell_tokens: null pointer");
    return lstm_cell_launch("lstm_cell_tokens", nullptr, 0, n_head + E, LcTokens{head, n_head, table, E, idx}, h, ld_h, c, Wih, ld_wih, Whh, ld_whh, bih, bhh,
                            rows, H, h_out, c_out, nullptr, stream);
}

extern "C" int dfol_lstm_cell_train_f32(const float* x, int64_t ld_x, int32_t KX, const float* h, int64_t ld_h, const float* c, const float* Wih,
                                        int64_t ld_wih, const float* Whh, int64_t ld_whh, const float* bih, const float* bhh, int32_t rows,
                                        int32_t H, float* h_out, float* c_out, float* gates, void* stream) {
    DFOL_REQUIRE((x && gates) || rows == 0, "lstm_cell_train: null pointer");
    return lstm_cell_launch("lstm_cell_train", x, ld_x, KX, LcTokens{}, h, ld_h, c, Wih, ld_wih, Whh, ld_whh, bih, bhh, rows, H, h_out, c_out, gates, stream);
}

// Backward of the cell's pointwise stage (torch's lstm_cell_backward): from the activated gates (i, f, g, o), the old and new cell state
// and the gradients of h' and c' to the gradient of the PRE-activation gates [rows, 4H] (i, f, g, o order) and of the old cell state.
// The four products with the weights (dx, dh, dW_ih, dW_hh, db) are dfol_linear_act_f32 / dfol_linear_wgrad_bias_f32 calls on d_gates.
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev, const float* __restrict__ c_new,
                                     const float* __restrict__ d_hy, const float* __restrict__ d_cy, int rows, int H,
                                     float* __restrict__ d_gates, float* __restrict__ d_c_prev) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * H) return;
    const int r = idx / H, j = idx - r * H;
    const float* g = gates + (int64_t)r * 4 * H;
    const float gi = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
    const float tc = tanhf(c_new[idx]);
    const float dh = d_hy ? d_hy[idx] : 0.f;
    const float dct = (d_cy ? d_cy[idx] : 0.f) + dh * go * (1.f - tc * tc);
    float* dg = d_gates + (int64_t)r * 4 * H;
    dg[j] = dct * gg * gi * (1.f - gi);
    dg[H + j] = dct * c_prev[idx] * gf * (1.f - gf);
    dg[2 * H + j] = dct * gi * (1.f - gg * gg);
    dg[3 * H + j] = dh * tc * go * (1.f - go);
    d_c_prev[idx] = dct * gf;
}

extern "C" int dfol_lstm_cell_bwd_f32(const float* gates, const float* c_prev, const float* c_new, const float* d_hy, const float* d_cy,
                                      int32_t rows, int32_t H, float* d_gates, float* d_c_prev, void* stream) {
    DFOL_REQUIRE(rows >= 0 && H > 0, "lstm_cell_bwd: bad sizes rows=%d H=%d", rows, H);
    if (rows == 0) return 0;
    DFOL_REQUIRE(gates && c_prev && c_new && d_gates && d_c_prev && (d_hy || d_cy), "lstm_cell_bwd: null pointer");
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(dfol_cdiv((int64_t)rows * H, 256)), dim3(256), 0, (hipStream_t)stream, gates, c_prev, c_new,
                       d_hy, d_cy, rows, H, d_gates, d_c_prev);
    DFOL_LAUNCH_CHECK("lstm_cell_bwd");
    return 0;
}

extern "C" int dfol_lstm_pointwise_f32(const float* igates, const float* hgates, const float* c, int32_t rows, int32_t H, float* h_out,
                                       float* c_out, void* stream) {
    DFOL_REQUIRE(rows >= 0 && H > 0, "lstm_pointwise: bad sizes rows=%d H=%d", rows, H);
    if (rows == 0) return 0;
    DFOL_REQUIRE(igates && hgates && c && h_out && c_out, "lstm_pointwise: null pointer");
    hipLaunchKernelGGL(lstm_pointwise_kernel, dim3(dfol_cdiv((int64_t)rows * H, 256)), dim3(256), 0, (hipStream_t)stream, igates, hgates, c,
                       rows, H, h_out, c_out);
    DFOL_LAUNCH_CHECK("lstm_pointwise");
    return 0;
}
