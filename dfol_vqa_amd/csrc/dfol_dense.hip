// Dense contractions of the visual oracle for gfx950: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Y = act(X W^T + b) with both operands K-contiguous (torch nn.Linear layout).  128x128 block tile,
// four wavefronts in a 2x2 arrangement, each owning a 64x64 sub-tile as 2x2 MFMA tiles of 32x32.
// K is consumed in chunks of 32 staged through LDS.  The MFMA's K index is only a summation index, so a
// lane takes the 16 *contiguous* k of its half (k = 16*(lane>>5) + t at step t) instead of the
// interleaved k = 2t + (lane>>5): that turns 16 ds_read_b32 per operand tile into 4 conflict-free
// ds_read_b128 (row pitch 36 floats).
#include <stdlib.h>

#include "dfol_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32, PITCH = 36;

template <int ALIGN, bool FULL = false>
__device__ __forceinline__ float4 load4(const float* __restrict__ base, int64_t ld, int row, int col, int rows, int cols) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!FULL && (row >= rows || col >= cols)) return v;
    const float* p = base + (int64_t)row * ld + col;
    if (FULL) {
        if (ALIGN == 4) return *reinterpret_cast<const float4*>(p);
        if (ALIGN == 2) {
            const float2 a = *reinterpret_cast<const float2*>(p), b = *reinterpret_cast<const float2*>(p + 2);
            return make_float4(a.x, a.y, b.x, b.y);
        }
        return make_float4(p[0], p[1], p[2], p[3]);
    }
    if (col + 3 < cols) {
        if (ALIGN == 4) return *reinterpret_cast<const float4*>(p);
        if (ALIGN == 2) {
            const float2 a = *reinterpret_cast<const float2*>(p), b = *reinterpret_cast<const float2*>(p + 2);
            return make_float4(a.x, a.y, b.x, b.y);
        }
        return make_float4(p[0], p[1], p[2], p[3]);
    }
    v.x = p[0];
    if (col + 1 < cols) v.y = p[1];
    if (col + 2 < cols) v.z = p[2];
    return v;
}

template <int ACT>
__device__ __forceinline__ float activate(float x) {
    if (ACT == DFOL_ACT_SIGMOID) return 1.0f / (1.0f + expf(-x));
    if (ACT == DFOL_ACT_ELU) return x > 0.f ? x : expm1f(x);
    if (ACT == DFOL_ACT_LOGSIGMOID) return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
    return x;
}

// BMT = rows of the block tile: 128 (each wavefront 64x64) or 64 (each wavefront 32x64).  The small tile doubles the number of
// workgroups; the host picks it when the 128-row grid would leave CUs idle (M = 9216: 288 workgroups for 256 CUs).
// FULL: 0 = every operand load is bounds-checked; 1 = M and N are multiples of the tile, so the loads of every K chunk that lies
// inside K are unchecked (the last, partial chunk still is); 2 = K is a multiple of the chunk too: no checks at all.
template <int ACT, int ALIGN_X, int ALIGN_W, int BMT, int FULL>
__global__ __launch_bounds__(256) void linear_act_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ W,
                                                         int64_t ldw, const float* __restrict__ bias, float* __restrict__ Y,
                                                         int64_t ldy, int M, int N, int K) {
    constexpr int TI = BMT / 64;                            // 32-row MFMA tiles per wavefront along M
    constexpr int PA = BMT / 32;                            // loader passes over the A tile
    __shared__ __attribute__((aligned(16))) float As[BMT * PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[BN * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order is not needed here: W (<= 2.8 MB) stays L2-resident on every XCD.
    const int m0 = blockIdx.y * BMT, n0 = blockIdx.x * BN;
    const int lrow = tid >> 3, lk = (tid & 7) * 4;        // loader: 32 rows x 8 float4 per pass

    floatx16 acc[TI][2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[PA], rb[4];
#pragma unroll
    for (int i = 0; i < PA; ++i) ra[i] = load4<ALIGN_X, false>(X, ldx, m0 + lrow + 32 * i, lk, M, K);
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = load4<ALIGN_W, false>(W, ldw, n0 + lrow + 32 * i, lk, N, K);

    const int half = lane >> 5, r32 = lane & 31;
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PA; ++i) *reinterpret_cast<float4*>(&As[(lrow + 32 * i) * PITCH + lk]) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&Bs[(lrow + 32 * i) * PITCH + lk]) = rb[i];
        __syncthreads();
        if (FULL == 2 ? k0 + BK < K : (FULL == 1 && k0 + 2 * BK <= K)) {      // the next chunk lies entirely inside K: unchecked loads
#pragma unroll
            for (int i = 0; i < PA; ++i) ra[i] = load4<ALIGN_X, true>(X, ldx, m0 + lrow + 32 * i, k0 + BK + lk, M, K);
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = load4<ALIGN_W, true>(W, ldw, n0 + lrow + 32 * i, k0 + BK + lk, N, K);
        } else if (FULL != 2 && k0 + BK < K) {
#pragma unroll
            for (int i = 0; i < PA; ++i) ra[i] = load4<ALIGN_X, false>(X, ldx, m0 + lrow + 32 * i, k0 + BK + lk, M, K);
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = load4<ALIGN_W, false>(W, ldw, n0 + lrow + 32 * i, k0 + BK + lk, N, K);
        }
        float a[TI][16], b[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* pb = &Bs[(wn * 64 + t * 32 + r32) * PITCH + half * 16];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 fb = *reinterpret_cast<const float4*>(pb + 4 * v);
                b[t][4 * v + 0] = fb.x; b[t][4 * v + 1] = fb.y; b[t][4 * v + 2] = fb.z; b[t][4 * v + 3] = fb.w;
            }
        }
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            const float* pa = &As[(wm * 32 * TI + t * 32 + r32) * PITCH + half * 16];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 fa = *reinterpret_cast<const float4*>(pa + 4 * v);
                a[t][4 * v + 0] = fa.x; a[t][4 * v + 1] = fa.y; a[t][4 * v + 2] = fa.z; a[t][4 * v + 3] = fa.w;
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + r32;
            if (n >= N) continue;
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 32 * TI + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (m < M) Y[(int64_t)m * ldy + n] = activate<ACT>(acc[i][j][e] + bv);
            }
        }
}

int align_of(const float* p, int64_t ld) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    if (a % 16 == 0 && ld % 4 == 0) return 4;
    if (a % 8 == 0 && ld % 2 == 0) return 2;
    return 1;
}

template <int ACT, int BMT>
void launch_linear(hipStream_t st, int ax, int aw, const float* X, int64_t ldx, const float* W, int64_t ldw,
                   const float* bias, float* Y, int64_t ldy, int M, int N, int K) {
    const dim3 grid(dfol_cdiv(N, BN), dfol_cdiv(M, BMT));
    const int full = (M % BMT == 0 && N % BN == 0) ? (K % BK == 0 ? 2 : 1) : 0;
#define DFOL_LIN(AX, AW) \
    do {                                                                                                                         \
        if (full == 2) hipLaunchKernelGGL((linear_act_kernel<ACT, AX, AW, BMT, 2>), grid, dim3(256), 0, st, X, ldx, W, ldw, bias, Y, ldy, M, N, K); \
        else if (full == 1) hipLaunchKernelGGL((linear_act_kernel<ACT, AX, AW, BMT, 1>), grid, dim3(256), 0, st, X, ldx, W, ldw, bias, Y, ldy, M, N, K); \
        else hipLaunchKernelGGL((linear_act_kernel<ACT, AX, AW, BMT, 0>), grid, dim3(256), 0, st, X, ldx, W, ldw, bias, Y, ldy, M, N, K);     \
    } while (0)
    if (ax == 4 && aw == 4) DFOL_LIN(4, 4);
    else if (ax == 2 && aw == 4) DFOL_LIN(2, 4);
    else if (ax == 4 && aw == 2) DFOL_LIN(4, 2);
    else if (ax == 2 && aw == 2) DFOL_LIN(2, 2);
    else DFOL_LIN(1, 1);
#undef DFOL_LIN
}

}  // namespace

extern "C" int dfol_linear_act_f32(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y,
                                   int64_t ldy, int32_t M, int32_t N, int32_t K, int32_t act, void* stream) {
    DFOL_REQUIRE(M >= 0 && N > 0 && K > 0, "linear_act: bad sizes M=%d N=%d K=%d", M, N, K);
    DFOL_REQUIRE(ldx >= K && ldw >= K && ldy >= N, "linear_act: leading dimensions too small");
    DFOL_REQUIRE(act >= 0 && act <= 3, "linear_act: unknown activation %d", act);
    if (M == 0) return 0;
    DFOL_REQUIRE(X && W && Y, "linear_act: null pointer");
    DFOL_REQUIRE(dfol_cdiv(M, 64) <= 65535, "linear_act: M=%d too large for one launch", M);
    hipStream_t st = (hipStream_t)stream;
    const int ax = align_of(X, ldx), aw = align_of(W, ldw);
    // 64-row tiles when the 128-row grid cannot give every CU a few workgroups (env DFOL_LINEAR_BM = 64 / 128 forces one)
    static const int forced = getenv("DFOL_LINEAR_BM") ? atoi(getenv("DFOL_LINEAR_BM")) : 0;
    const bool small = forced ? forced == 64 : (int64_t)dfol_cdiv(M, BM) * dfol_cdiv(N, BN) < 1200;
#define DFOL_LINB(A)                                                                             \
    do {                                                                                         \
        if (small) launch_linear<A, 64>(st, ax, aw, X, ldx, W, ldw, bias, Y, ldy, M, N, K);      \
        else launch_linear<A, 128>(st, ax, aw, X, ldx, W, ldw, bias, Y, ldy, M, N, K);           \
    } while (0)
    switch (act) {
        case DFOL_ACT_NONE: DFOL_LINB(DFOL_ACT_NONE); break;
        case DFOL_ACT_SIGMOID: DFOL_LINB(DFOL_ACT_SIGMOID); break;
        case DFOL_ACT_ELU: DFOL_LINB(DFOL_ACT_ELU); break;
        default: DFOL_LINB(DFOL_ACT_LOGSIGMOID); break;
    }
#undef DFOL_LINB
    DFOL_LAUNCH_CHECK("linear_act");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
__global__ void box_positions_kernel(const float* __restrict__ raw, int64_t ld_raw, int raw_cols, int O, float* __restrict__ obj,
                                     int64_t ld_obj, int pos_col) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= O) return;
    const float* t = raw + (int64_t)o * ld_raw + raw_cols - 6;      // (W, H, x, y, w, h)
    const float Wc = fmaxf(t[0], 1.f), Hc = fmaxf(t[1], 1.f);      // clamp(1), batch_gqa_boxfeatures_pipeline.py:209
    float* d = obj + (int64_t)o * ld_obj + pos_col;
    d[0] = t[2] / Wc;
    d[1] = t[3] / Hc;
    d[2] = t[4] / Wc;
    d[3] = t[5] / Hc;
}

extern "C" int dfol_box_positions_f32(const float* raw, int64_t ld_raw, int32_t raw_cols, int32_t O, float* obj, int64_t ld_obj,
                                      int32_t pos_col, void* stream) {
    DFOL_REQUIRE(O >= 0 && raw_cols >= 6 && ld_raw >= raw_cols && pos_col >= 0 && ld_obj >= pos_col + 4, "box_positions: bad sizes");
    if (O == 0) return 0;
    DFOL_REQUIRE(raw && obj, "box_positions: null pointer");
    hipLaunchKernelGGL(box_positions_kernel, dim3(dfol_cdiv(O, 256)), dim3(256), 0, (hipStream_t)stream, raw, ld_raw, raw_cols, O,
                       obj, ld_obj, pos_col);
    DFOL_LAUNCH_CHECK("box_positions");
    return 0;
}

// One block per (image, subject); threads sweep the (object, feature) plane of that subject's pair rows.
__global__ void pair_features_kernel(const float* __restrict__ obj, int64_t ld_obj, int D, const int32_t* __restrict__ obj_off,
                                     const int64_t* __restrict__ pair_off, int max_n, float* __restrict__ pair, int64_t ld_pair) {
    const int q = blockIdx.x / max_n, s = blockIdx.x % max_n;
    const int first = obj_off[q], n = obj_off[q + 1] - first;
    if (s >= n) return;
    const float* fs = obj + (int64_t)(first + s) * ld_obj;
    const int width = 2 * D + 4;
    for (int e = threadIdx.x; e < (n - 1) * width; e += blockDim.x) {
        const int k = e / width, f = e - k * width;           // k-th partner of s
        const int o = k < s ? k : k + 1;
        const float* fo = obj + (int64_t)(first + o) * ld_obj;
        float v;
        if (f < D) v = fs[f];
        else if (f < 2 * D) v = fo[f - D];
        else {
            const float x1 = fs[D - 4], y1 = fs[D - 3], w1 = fs[D - 2], h1 = fs[D - 1];
            const float x2 = fo[D - 4], y2 = fo[D - 3], w2 = fo[D - 2], h2 = fo[D - 1];
            const float dx = x1 + w1 / 2.0f - x2 - w2 / 2.0f, dy = y1 + h1 / 2.0f - y2 - h2 / 2.0f;   // :271-272
            const float dist = sqrtf(dx * dx + dy * dy);
            const int g = f - 2 * D;
            if (g == 0) v = dist;
            else if (g == 1) v = asinf(dy / fmaxf(dist, 1e-10f));                                   // :275
            else if (g == 2) v = (x2 - x1 > 0.f) ? 1.f : ((x2 - x1 < 0.f) ? -1.f : 0.f);            // :276
            else v = (y2 - y1 > 0.f) ? 1.f : ((y2 - y1 < 0.f) ? -1.f : 0.f);                        // :277
        }
        pair[(pair_off[q] + (int64_t)s * (n - 1) + k) * ld_pair + f] = v;
    }
}

extern "C" int dfol_pair_features_f32(const float* obj, int64_t ld_obj, int32_t D, const int32_t* obj_off, const int64_t* pair_off,
                                      int32_t Q, int32_t max_n, float* pair, int64_t ld_pair, void* stream) {
    DFOL_REQUIRE(Q >= 0 && D >= 4 && max_n >= 0 && ld_obj >= D && ld_pair >= 2 * D + 4, "pair_features: bad sizes");
    if (Q == 0 || max_n <= 1) return 0;
    DFOL_REQUIRE(obj && obj_off && pair_off && pair, "pair_features: null pointer");
    hipLaunchKernelGGL(pair_features_kernel, dim3((unsigned)Q * max_n), dim3(256), 0, (hipStream_t)stream, obj, ld_obj, D, obj_off,
                       pair_off, max_n, pair, ld_pair);
    DFOL_LAUNCH_CHECK("pair_features");
    return 0;
}

// =====================================================================================================
// Needed-columns oracle: only the likelihoods a program asks for are ever computed.
// =====================================================================================================
// Attribute likelihood of one (image, concept) request: ll[p][o] = LogSigmoid(hidden[o] . E[col] + be[col]).
// A workgroup owns 16 objects of one predicate (four per wavefront, all in flight at once); lanes stride the hidden
// dimension (coalesced), the per-object dot products are reduced with DPP (no LDS traffic).
__global__ __launch_bounds__(256) void attr_ll_kernel(const float* __restrict__ hidden, int64_t ld_h, int H,
                                                      const float* __restrict__ E, int64_t ld_e, const float* __restrict__ be,
                                                      const int32_t* __restrict__ obj_off, const int32_t* __restrict__ pred_q,
                                                      const int32_t* __restrict__ pred_col, int P, int NS, float dflt,
                                                      float* __restrict__ ll) {
    const int p = blockIdx.x;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o0 = blockIdx.y * 16 + 4 * wv;                // this wavefront's four objects
    const int q = pred_q[p], col = pred_col[p];
    const int first = obj_off[q], n = obj_off[q + 1] - first;
    float* out = ll + (int64_t)p * NS;
    if (col < 0 || o0 >= n) {                               // no-op token, or padding columns: the absent value
        if (lane < 4 && o0 + lane < NS) out[o0 + lane] = dflt;
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
        out[o0 + u] = (o0 + u < n) ? fminf(x, 0.f) - log1pf(expf(-fabsf(x))) : dflt;       // nn.LogSigmoid
    }
}

extern "C" int dfol_attr_ll_f32(const float* hidden, int64_t ld_hidden, int32_t H, const float* E, int64_t ld_e, const float* be,
                                const int32_t* obj_off, const int32_t* pred_q, const int32_t* pred_col, int32_t P, int32_t NS,
                                float default_ll, float* ll, void* stream) {
    DFOL_REQUIRE(P >= 0 && NS > 0 && NS % 4 == 0 && H > 0 && H <= 512, "attr_ll: bad sizes P=%d NS=%d H=%d (H <= 512)", P, NS, H);
    if (P == 0) return 0;
    DFOL_REQUIRE(hidden && E && obj_off && pred_q && pred_col && ll, "attr_ll: null pointer");
    hipLaunchKernelGGL(attr_ll_kernel, dim3(P, dfol_cdiv(NS, 16)), dim3(256), 0, (hipStream_t)stream, hidden, ld_hidden, H, E, ld_e, be,
                       obj_off, pred_q, pred_col, P, NS, default_ll, ll);
    DFOL_LAUNCH_CHECK("attr_ll");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Fused pair kernel.  For every ordered object pair (s, o) of an image it rebuilds the pair MLP of the
// reference without ever materialising the [pairs, 1036] pair matrix or the [pairs, 2335] table:
//   z = ELU( U[s] + V[o] + Wg . geo(s,o) )            U = W1[:, :D] obj + b1,  V = W1[:, D:2D] obj   (per OBJECT, one GEMM)
//   h = Sigmoid( W2 z + b2 )                          exact-fp32 MFMA, 128 pairs x HID2 per workgroup
//   tile[t][s][o] = LogSigmoid( h . E[col_t] + be[col_t] )   only for the (image, concept) requests of the program
// A workgroup owns 128 consecutive slots e = s*n + o of one image; a wavefront owns 32 of them across ALL
// HID2 columns (NB accumulator tiles of 32x32), so the final dot products with the requested embedding rows
// need only a 32-lane shuffle reduction.
namespace {

constexpr int PK_PITCH = 36;

// KEXACT: HID1 is a multiple of 32 AND W2 is allocated with 32*NB rows (rows >= HID2 zero): no guards in the main loop
template <int NB, bool KEXACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void pair_ll_kernel(
    const float* __restrict__ UV, int64_t ld_uv, int HID1, const float* __restrict__ pos, int64_t ld_pos,
    const float* __restrict__ Wg, const float* __restrict__ W2, int64_t ld_w2, const float* __restrict__ b2, int HID2,
    const float* __restrict__ E, int64_t ld_e, const float* __restrict__ be, const int32_t* __restrict__ n_obj,
    const int32_t* __restrict__ obj_off, int Q, int tiles_per_image, const int32_t* __restrict__ req_col,
    const int32_t* __restrict__ req_tile, const uint8_t* __restrict__ req_orient, int K, int NS, float dflt,
    float* __restrict__ tiles) {
    __shared__ __attribute__((aligned(16))) float Bs[32 * NB * PK_PITCH];
    __shared__ __attribute__((aligned(16))) float Wgs[256 * 4];
    const int q = blockIdx.x / tiles_per_image, tb = blockIdx.x - q * tiles_per_image;
    const int n = n_obj[q];
    if (tb * 128 >= n * n) return;
    bool any = false;
    for (int k = 0; k < K; ++k) any |= req_col[(int64_t)k * Q + q] >= 0;
    if (!any) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r32 = lane & 31;
    const int first = obj_off[q];
    const int e_slot = tb * 128 + wave * 32 + r32;
    const bool valid = e_slot < n * n;
    const int s = valid ? e_slot / n : 0, o = valid ? e_slot - s * n : 0;

    // geometry of this lane's pair: batch_gqa_boxfeatures_pipeline.py:260-279
    float geo[4];
    {
        const float* ps = pos + (int64_t)(first + s) * ld_pos;
        const float* po = pos + (int64_t)(first + o) * ld_pos;
        const float x1 = ps[0], y1 = ps[1], w1 = ps[2], h1 = ps[3], x2 = po[0], y2 = po[1], w2 = po[2], h2 = po[3];
        const float dx = x1 + w1 / 2.0f - x2 - w2 / 2.0f, dy = y1 + h1 / 2.0f - y2 - h2 / 2.0f;
        const float dist = sqrtf(dx * dx + dy * dy);
        geo[0] = dist;
        geo[1] = asinf(dy / fmaxf(dist, 1e-10f));
        geo[2] = (x2 - x1 > 0.f) ? 1.f : ((x2 - x1 < 0.f) ? -1.f : 0.f);
        geo[3] = (y2 - y1 > 0.f) ? 1.f : ((y2 - y1 < 0.f) ? -1.f : 0.f);
    }
    for (int i = tid; i < HID1; i += 256) *reinterpret_cast<float4*>(&Wgs[i * 4]) = *reinterpret_cast<const float4*>(Wg + i * 4);

    const float* Urow = UV + (int64_t)(first + s) * ld_uv;
    const float* Vrow = UV + (int64_t)(first + o) * ld_uv + HID1;

    floatx16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    const int lrow = tid >> 3, lk = (tid & 7) * 4;          // W2 loader: 32 rows x 8 float4 per pass, NB passes
    float4 rb[NB];
    auto load_w2 = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int nrow = lrow + 32 * i;
            if (KEXACT) {
                rb[i] = *reinterpret_cast<const float4*>(W2 + (int64_t)nrow * ld_w2 + k0 + lk);
            } else {
                const int k = min(k0 + lk, HID1 - 4);
                const float4 v = *reinterpret_cast<const float4*>(W2 + (int64_t)min(nrow, HID2 - 1) * ld_w2 + k);
                rb[i] = (nrow < HID2 && k0 + lk < HID1) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    float4 ru[4], rv[4];
    auto load_uv = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = k0 + 16 * half + 4 * j, k = KEXACT ? kk : min(kk, HID1 - 4);
            ru[j] = *reinterpret_cast<const float4*>(Urow + k);      // beyond HID1 the value is discarded below
            rv[j] = *reinterpret_cast<const float4*>(Vrow + k);
        }
    };
    // A operand of a chunk: 16 contiguous k of this lane's half, z = ELU(U[s] + V[o] + Wg . geo)
    auto make_a = [&](int k0, float (&a)[16]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float uu[4] = {ru[j].x, ru[j].y, ru[j].z, ru[j].w}, vv[4] = {rv[j].x, rv[j].y, rv[j].z, rv[j].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kk = k0 + 16 * half + 4 * j + c, k = KEXACT ? kk : min(kk, HID1 - 1);
                const float4 g = *reinterpret_cast<const float4*>(&Wgs[k * 4]);
                float z = uu[c] + vv[c] + (g.x * geo[0] + g.y * geo[1] + g.z * geo[2] + g.w * geo[3]);
                z = z > 0.f ? z : dfol_exp(z) - 1.0f;       // nn.ELU (hardware exp: abs error < 1e-7 on a hidden activation)
                a[4 * j + c] = (KEXACT || kk < HID1) ? z : 0.f;
            }
        }
    };
    auto store_w2 = [&]() {
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(&Bs[(lrow + 32 * i) * PK_PITCH + lk]) = rb[i];
    };

    // Software pipeline: while the MFMAs of chunk c run, the A operand of chunk c+1 is built (VALU) and the
    // operands of chunk c+2 are in flight (VMEM).
    float a_cur[16], a_next[16];
    load_w2(0);
    load_uv(0);
    __syncthreads();                                        // Wgs visible
    store_w2();
    make_a(0, a_cur);
    __syncthreads();
    if (KEXACT) {
        // Branch-free steady state: chunk indices beyond the end are clamped (the redundant work of the last
        // iteration is discarded), so every iteration is one straight-line scheduling region in which the
        // A-operand VALU work of the next chunk is interleaved with this chunk's MFMAs.
        const int last = HID1 - 32;
        load_w2(min(32, last));
        load_uv(min(32, last));
        for (int k0 = 0; k0 < HID1; k0 += 32) {
            make_a(min(k0 + 32, last), a_next);
#pragma unroll
            for (int tg = 0; tg < 4; ++tg) {
                float4 b4[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) b4[i] = *reinterpret_cast<const float4*>(&Bs[(i * 32 + r32) * PK_PITCH + 16 * half + 4 * tg]);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * tg + 0], b4[i].x, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * tg + 1], b4[i].y, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * tg + 2], b4[i].z, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * tg + 3], b4[i].w, acc[i], 0, 0, 0);
                }
            }
            __syncthreads();                                // every wave is done reading this chunk of W2
            store_w2();
            __syncthreads();
            load_w2(min(k0 + 64, last));
            load_uv(min(k0 + 64, last));
#pragma unroll
            for (int t = 0; t < 16; ++t) a_cur[t] = a_next[t];
        }
    } else {
    if (32 < HID1) {
        load_w2(32);
        load_uv(32);
    }
    for (int k0 = 0; k0 < HID1; k0 += 32) {
        const bool more = k0 + 32 < HID1;
        if (more) make_a(k0 + 32, a_next);
#pragma unroll
        for (int tg = 0; tg < 4; ++tg) {
            float4 b4[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) b4[i] = *reinterpret_cast<const float4*>(&Bs[(i * 32 + r32) * PK_PITCH + 16 * half + 4 * tg]);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * tg + 0], b4[i].x, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * tg + 1], b4[i].y, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * tg + 2], b4[i].z, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * tg + 3], b4[i].w, acc[i], 0, 0, 0);
            }
        }
        if (more) {
            __syncthreads();                                // every wave is done reading this chunk of W2
            store_w2();
            __syncthreads();
            if (k0 + 64 < HID1) {
                load_w2(k0 + 64);
                load_uv(k0 + 64);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) a_cur[t] = a_next[t];
        }
    }

    }   // !KEXACT

    // h = Sigmoid(acc + b2); C layout: column = i*32 + r32, row(e) = (e & 3) + 8 * (e >> 2) + 4 * half
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int ncol = i * 32 + r32;
        const bool in = ncol < HID2;
        const float bv = in ? b2[ncol] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = in ? __frcp_rn(1.0f + dfol_exp(-(acc[i][e] + bv))) : 0.f;   // nn.Sigmoid
    }
    const int64_t tile_sz = (int64_t)NS * NS;
    for (int k = 0; k < K; ++k) {
        const int col = req_col[(int64_t)k * Q + q];
        if (col < 0) continue;                              // uniform over the workgroup
        float part[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) part[e] = 0.f;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int ncol = i * 32 + r32;
            const float ev = ncol < HID2 ? E[(int64_t)col * ld_e + ncol] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) part[e] = fmaf(acc[i][e], ev, part[e]);
        }
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1)
#pragma unroll
            for (int e = 0; e < 16; ++e) part[e] += __shfl_xor(part[e], m, 64);
        // every lane of a half now holds the 16 row sums of that half; lane j < 16 writes row j of its half
        if (r32 < 16) {
            float v = part[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) v = (r32 == e) ? part[e] : v;
            const int row = (r32 & 3) + 8 * (r32 >> 2) + 4 * half;
            const int ee = tb * 128 + wave * 32 + row;
            if (ee < n * n) {
                const int ss = ee / n, oo = ee - ss * n;
                const float x = v + (be ? be[col] : 0.f);
                const float val = (ss == oo) ? dflt : fminf(x, 0.f) - log1pf(expf(-fabsf(x)));   // nn.LogSigmoid; diagonal stays absent
                float* t = tiles + (int64_t)req_tile[(int64_t)k * Q + q] * tile_sz;
                if (req_orient && req_orient[(int64_t)k * Q + q]) t[(int64_t)oo * NS + ss] = val;
                else t[(int64_t)ss * NS + oo] = val;
            }
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// Second geometry of the fused pair kernel for the full-size oracle: 16x16x4 MFMA tiles, 8 wavefronts per
// workgroup (two per SIMD).  A wavefront owns 16 pairs across all hidden columns (NB16 accumulator tiles of
// 16x16 = 4 registers each), so its accumulators need 80 registers instead of 160 and two wavefronts fit on
// every SIMD: while one builds its next A operand (VALU) or waits at the barrier, the other keeps the matrix
// pipe busy.  Requires HID1 % 32 == 0 and W2 allocated with 16*NB16 rows.
// LDS layout of the W2 chunk: row pitch 36 floats; the 8-float k-group of (row, kq) is stored at group
// kq ^ flip(row), flip = 1 for rows 4..11 (mod 16): a ds_read_b128 lane group always mixes lanes of two adjacent
// hardware k indices (rows 0-3,12-15 of one with rows 4-11 of the next), and the flip puts them on the same group,
// which makes every such read conflict-free.
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NB16>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void pair_ll16_kernel(
    const float* __restrict__ UV, int64_t ld_uv, int HID1, const float* __restrict__ pos, int64_t ld_pos,
    const float* __restrict__ Wg, const float* __restrict__ W2, int64_t ld_w2, const float* __restrict__ b2, int HID2,
    const float* __restrict__ E, int64_t ld_e, const float* __restrict__ be, const int32_t* __restrict__ n_obj,
    const int32_t* __restrict__ obj_off, int Q, int tiles_per_image, const int32_t* __restrict__ req_col,
    const int32_t* __restrict__ req_tile, const uint8_t* __restrict__ req_orient, int K, int NS, float dflt,
    float* __restrict__ tiles) {
    constexpr int ROWS = (16 * NB16 + 63) / 64 * 64;        // W2 rows staged per chunk (the loader works in passes of 64 rows)
    constexpr int PASSES = ROWS / 64;
    constexpr int PARTS = 4, HALF = (NB16 + PARTS - 1) / PARTS;   // column tiles are visited in groups (register budget)
    __shared__ __attribute__((aligned(16))) float Bs[2][ROWS * PK_PITCH];      // double-buffered W2 chunk: one barrier per chunk
    __shared__ __attribute__((aligned(16))) float Wgs[256 * 4];
    const int q = blockIdx.x / tiles_per_image, tb = blockIdx.x - q * tiles_per_image;
    const int n = n_obj[q];
    if (tb * 128 >= n * n) return;
    bool any = false;
    for (int k = 0; k < K; ++k) any |= req_col[(int64_t)k * Q + q] >= 0;
    if (!any) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 4, r16 = lane & 15;
    const int first = obj_off[q];
    const int e_slot = tb * 128 + wave * 16 + r16;
    const bool valid = e_slot < n * n;
    const int s = valid ? e_slot / n : 0, o = valid ? e_slot - s * n : 0;
    float geo[4];
    {
        const float* ps = pos + (int64_t)(first + s) * ld_pos;
        const float* po = pos + (int64_t)(first + o) * ld_pos;
        const float x1 = ps[0], y1 = ps[1], w1 = ps[2], h1 = ps[3], x2 = po[0], y2 = po[1], w2 = po[2], h2 = po[3];
        const float dx = x1 + w1 / 2.0f - x2 - w2 / 2.0f, dy = y1 + h1 / 2.0f - y2 - h2 / 2.0f;
        const float dist = sqrtf(dx * dx + dy * dy);
        geo[0] = dist;
        geo[1] = asinf(dy / fmaxf(dist, 1e-10f));
        geo[2] = (x2 - x1 > 0.f) ? 1.f : ((x2 - x1 < 0.f) ? -1.f : 0.f);
        geo[3] = (y2 - y1 > 0.f) ? 1.f : ((y2 - y1 < 0.f) ? -1.f : 0.f);
    }
    for (int i = tid; i < HID1; i += 512) *reinterpret_cast<float4*>(&Wgs[i * 4]) = *reinterpret_cast<const float4*>(Wg + i * 4);
    const float* Urow = UV + (int64_t)(first + s) * ld_uv;
    const float* Vrow = UV + (int64_t)(first + o) * ld_uv + HID1;

    floatx4 acc[NB16];
#pragma unroll
    for (int i = 0; i < NB16; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};

    // W2 loader: 64 rows x 8 float4 per pass; element (row, k) goes to k-group (k/8) ^ flip(row)
    const int lrow = tid >> 3, lkq = (tid & 7) >> 1, lwithin = ((tid & 7) & 1) * 4;
    const int lflip = ((lrow & 15) >> 2) == 1 || ((lrow & 15) >> 2) == 2 ? 1 : 0;       // the same for lrow + 64 i
    const int lds_off = lrow * PK_PITCH + 8 * (lkq ^ lflip) + lwithin;
    float4 rb[PASSES];
    auto load_w2 = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) rb[i] = *reinterpret_cast<const float4*>(W2 + (int64_t)(lrow + 64 * i) * ld_w2 + k0 + (tid & 7) * 4);
    };
    auto store_w2 = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) *reinterpret_cast<float4*>(&Bs[buf][64 * i * PK_PITCH + lds_off]) = rb[i];
    };
    float4 ru[2], rv[2];
    auto load_uv = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ru[j] = *reinterpret_cast<const float4*>(Urow + k0 + 8 * kh + 4 * j);
            rv[j] = *reinterpret_cast<const float4*>(Vrow + k0 + 8 * kh + 4 * j);
        }
    };
    auto make_a = [&](int k0, float (&a)[8]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float uu[4] = {ru[j].x, ru[j].y, ru[j].z, ru[j].w}, vv[4] = {rv[j].x, rv[j].y, rv[j].z, rv[j].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 g = *reinterpret_cast<const float4*>(&Wgs[(k0 + 8 * kh + 4 * j + c) * 4]);
                float z = uu[c] + vv[c] + (g.x * geo[0] + g.y * geo[1] + g.z * geo[2] + g.w * geo[3]);
                a[4 * j + c] = z > 0.f ? z : dfol_exp(z) - 1.0f;      // nn.ELU
            }
        }
    };
    const int rflip = (r16 >> 2) == 1 || (r16 >> 2) == 2 ? 1 : 0;
    const int boff = r16 * PK_PITCH + 8 * (kh ^ rflip);

    float a_cur[8], a_next[8];
    const int last = HID1 - 32;
    load_w2(0);
    load_uv(0);
    __syncthreads();                                        // Wgs visible
    store_w2(0);
    make_a(0, a_cur);
    load_w2(min(32, last));
    load_uv(min(32, last));
    __syncthreads();
    int buf = 0;
    const bool early = wave < 4;
    for (int k0 = 0; k0 < HID1; k0 += 32, buf ^= 1) {
        store_w2(buf ^ 1);                                  // chunk k0+32 (in registers since the previous iteration)
        // The two wavefronts that share a SIMD (w and w+4) run their VALU phase at opposite ends of the iteration,
        // so one of them always has MFMAs to issue while the other builds its next A operand.
        if (early) {
            make_a(min(k0 + 32, last), a_next);
            load_uv(min(k0 + 64, last));
            __builtin_amdgcn_sched_barrier(0);
        }
        load_w2(min(k0 + 64, last));                        // chunk k0+64 flies during this chunk's MFMAs
        const float* brow = &Bs[buf][boff];
#pragma unroll
        for (int h = 0; h < 2; ++h)                          // k steps 4h .. 4h+3 of this lane's 8
#pragma unroll
            for (int part = 0; part < PARTS; ++part) {
                float4 b4[HALF];
#pragma unroll
                for (int i = 0; i < HALF; ++i)
                    if (part * HALF + i < NB16) b4[i] = *reinterpret_cast<const float4*>(brow + (part * HALF + i) * 16 * PK_PITCH + 4 * h);
                // k step outermost: consecutive MFMAs hit different accumulators (a 16x16x4 MFMA issues every 32 cycles
                // but its result is ready for a dependent one only after 40)
#pragma unroll
                for (int i = 0; i < HALF; ++i)
                    if (part * HALF + i < NB16) acc[part * HALF + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[4 * h + 0], b4[i].x, acc[part * HALF + i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < HALF; ++i)
                    if (part * HALF + i < NB16) acc[part * HALF + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[4 * h + 1], b4[i].y, acc[part * HALF + i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < HALF; ++i)
                    if (part * HALF + i < NB16) acc[part * HALF + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[4 * h + 2], b4[i].z, acc[part * HALF + i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < HALF; ++i)
                    if (part * HALF + i < NB16) acc[part * HALF + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[4 * h + 3], b4[i].w, acc[part * HALF + i], 0, 0, 0);
            }
        if (!early) {
            __builtin_amdgcn_sched_barrier(0);
            make_a(min(k0 + 32, last), a_next);
            load_uv(min(k0 + 64, last));
        }
        __syncthreads();                                    // chunk k0 fully read, chunk k0+32 fully written
#pragma unroll
        for (int t = 0; t < 8; ++t) a_cur[t] = a_next[t];
    }

    // h = Sigmoid(acc + b2); 16x16 C layout: column = i*16 + r16, row(e) = 4 * kh + e
    // (padding columns >= HID2: clamped addresses instead of guarded loads, their activation is forced to 0)
#pragma unroll
    for (int i = 0; i < NB16; ++i) {
        const int ncol = i * 16 + r16;
        const float bv = b2[min(ncol, HID2 - 1)];
        const float keep = ncol < HID2 ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = keep * __builtin_amdgcn_rcpf(1.0f + dfol_exp(-(acc[i][e] + bv)));
    }
    const int64_t tile_sz = (int64_t)NS * NS;
    for (int k = 0; k < K; ++k) {
        const int col = req_col[(int64_t)k * Q + q];
        if (col < 0) continue;
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        const float* erow = E + (int64_t)col * ld_e;
#pragma unroll
        for (int i = 0; i < NB16; ++i) {
            const float ev = erow[min(i * 16 + r16, HID2 - 1)];
#pragma unroll
            for (int e = 0; e < 4; ++e) part[e] = fmaf(acc[i][e], ev, part[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) part[e] = dfol_group_sum<16>(part[e]);
        if (r16 < 4) {                                      // lane e of each 16-lane set writes row 4*kh + e
            const float v = r16 == 0 ? part[0] : (r16 == 1 ? part[1] : (r16 == 2 ? part[2] : part[3]));
            const int ee = tb * 128 + wave * 16 + 4 * kh + r16;
            if (ee < n * n) {
                const int ss = ee / n, oo = ee - ss * n;
                const float x = v + (be ? be[col] : 0.f);
                const float val = (ss == oo) ? dflt : fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
                float* t = tiles + (int64_t)req_tile[(int64_t)k * Q + q] * tile_sz;
                if (req_orient && req_orient[(int64_t)k * Q + q]) t[(int64_t)oo * NS + ss] = val;
                else t[(int64_t)ss * NS + oo] = val;
            }
        }
    }
}

}  // namespace

extern "C" int dfol_pair_ll_f32(const float* UV, int64_t ld_uv, int32_t HID1, const float* pos, int64_t ld_pos, const float* Wg,
                                const float* W2, int64_t ld_w2, int32_t w2_rows_alloc, const float* b2, int32_t HID2, const float* E,
                                int64_t ld_e, const float* be, const int32_t* n_obj, const int32_t* obj_off, int32_t Q, int32_t max_n,
                                const int32_t* req_col, const int32_t* req_tile, const uint8_t* req_orient, int32_t K, int32_t NS,
                                float default_ll, float* tiles, void* stream) {
    DFOL_REQUIRE(Q >= 0 && K >= 0 && NS > 0 && NS % 4 == 0 && max_n >= 0 && max_n <= NS, "pair_ll: bad sizes Q=%d K=%d NS=%d max_n=%d", Q, K, NS, max_n);
    DFOL_REQUIRE(HID1 > 0 && HID1 <= 256 && HID1 % 4 == 0 && ld_uv % 4 == 0 && ld_w2 % 4 == 0, "pair_ll: HID1=%d must be a multiple of 4, <= 256, rows 16-byte aligned", HID1);
    DFOL_REQUIRE(HID2 > 0 && HID2 <= 320 && w2_rows_alloc >= HID2, "pair_ll: HID2=%d must be <= 320 and <= w2_rows_alloc=%d", HID2, w2_rows_alloc);
    if (Q == 0 || K == 0 || max_n < 2) return 0;
    DFOL_REQUIRE(UV && pos && Wg && W2 && b2 && E && n_obj && obj_off && req_col && req_tile && tiles, "pair_ll: null pointer");
    DFOL_REQUIRE(((uintptr_t)UV % 16 == 0) && ((uintptr_t)W2 % 16 == 0) && ((uintptr_t)Wg % 16 == 0), "pair_ll: operands must be 16-byte aligned");
    const int tpi = dfol_cdiv((int64_t)max_n * max_n, 128);
    const dim3 grid((unsigned)Q * tpi);
    hipStream_t st = (hipStream_t)stream;
#define DFOL_PAIR(NBV, KX)                                                                                                      \
    hipLaunchKernelGGL((pair_ll_kernel<NBV, KX>), grid, dim3(256), 0, st, UV, ld_uv, HID1, pos, ld_pos, Wg, W2, ld_w2, b2, HID2, E, \
                       ld_e, be, n_obj, obj_off, Q, tpi, req_col, req_tile, req_orient, K, NS, default_ll, tiles)
    const int nb = HID2 <= 32 ? 1 : (HID2 <= 128 ? 4 : 10);
    const bool kx = HID1 % 32 == 0 && w2_rows_alloc >= 32 * nb;
    static const int variant = getenv("DFOL_PAIR_VARIANT") ? atoi(getenv("DFOL_PAIR_VARIANT")) : 16;
    if (kx && nb == 10 && variant == 16) {                  // full-size oracle: 16x16x4 tiles, two wavefronts per SIMD
        if (HID2 <= 304)                                    // 300 hidden units need 19 column tiles of 16, not 20
            hipLaunchKernelGGL((pair_ll16_kernel<19>), grid, dim3(512), 0, st, UV, ld_uv, HID1, pos, ld_pos, Wg, W2, ld_w2, b2, HID2, E, ld_e,
                               be, n_obj, obj_off, Q, tpi, req_col, req_tile, req_orient, K, NS, default_ll, tiles);
        else
            hipLaunchKernelGGL((pair_ll16_kernel<20>), grid, dim3(512), 0, st, UV, ld_uv, HID1, pos, ld_pos, Wg, W2, ld_w2, b2, HID2, E, ld_e,
                               be, n_obj, obj_off, Q, tpi, req_col, req_tile, req_orient, K, NS, default_ll, tiles);
        DFOL_LAUNCH_CHECK("pair_ll");
        return 0;
    }
    if (HID2 <= 32) { if (kx) DFOL_PAIR(1, true); else DFOL_PAIR(1, false); }
    else if (HID2 <= 128) { if (kx) DFOL_PAIR(4, true); else DFOL_PAIR(4, false); }
    else { if (kx) DFOL_PAIR(10, true); else DFOL_PAIR(10, false); }
#undef DFOL_PAIR
    DFOL_LAUNCH_CHECK("pair_ll");
    return 0;
}
