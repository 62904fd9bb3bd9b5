// Weight gradient of a dense layer on the matrix cores:  dW[n][k] = sum_m dY[m][n] * X[m][k]      (dW = dY^T X, "TN" product)
//
// Training path (the reference gets it from torch autograd through nn.Linear, gqa_interpreter_experiments.py:26-33, 73-74, under
// trainer.py:436).  The contraction runs over the ROWS of two row-major matrices - millions of rows for the pair MLP (one row per
// ordered object pair) against a tiny [HID2, HID1] result - the shape vendor GEMMs handle worst (7.5 ms plain / 3.4 ms as a batched
// product for [300 x 2.5M] x [2.5M x 256]).
//
// v_mfma_f32_32x32x2_f32 takes ONE float per lane and operand: A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31].
// With i = output row n, j = output column k and the MFMA's k = the row m, both operands are what a lane reads with a plain coalesced
// load: 32 consecutive floats of row m (lanes 0..31) and of row m + 1 (lanes 32..63).  No LDS, no transposes, exact fp32 (the
// matrix pipe's fmaf chain).  A wavefront owns TN x TK tiles of 32 x 32 (TN * TK * 16 accumulator registers, one wavefront per
// SIMD), so a step of two rows costs TN + TK loads for TN * TK MFMAs of 64 cycles; the next step's operands are in flight under them.
// The rows are cut into slabs, one workgroup (four wavefronts = four output blocks of the same rows, sharing them through L1 / L2)
// per slab and block group; every slab writes its partial [N, K] block and a second kernel adds the slabs in a FIXED order:
// no atomics, bit-identical from run to run.
#include "dfol_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TN, int TK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad_tn_kernel(
    const float* __restrict__ dY, int64_t ld_dy, const float* __restrict__ X, int64_t ld_x, int M, int N, int K, int rows_per_slab,
    int nb_n, int nb_k, float* __restrict__ part) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blk = blockIdx.y * 4 + wave;
    if (blk >= nb_n * nb_k) return;
    const int bn = blk / nb_k, bk = blk - bn * nb_k;
    const int n0 = bn * 32 * TN, k0 = bk * 32 * TK;
    const int slab = blockIdx.x;
    const int m_begin = slab * rows_per_slab, m_end = min(M, m_begin + rows_per_slab);
    const int col = lane & 31, half = lane >> 5;

    const float* pa[TN];
    const float* pb[TK];
    bool oka[TN], okb[TK];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int n = n0 + 32 * t + col;
        oka[t] = n < N;
        pa[t] = dY + min(n, N - 1);
    }
#pragma unroll
    for (int u = 0; u < TK; ++u) {
        const int k = k0 + 32 * u + col;
        okb[u] = k < K;
        pb[u] = X + min(k, K - 1);
    }
    f32x16 acc[TN][TK];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int u = 0; u < TK; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;

    float a[TN], b[TK], an[TN], bn_[TK];
    auto load = [&](int m, float (&ra)[TN], float (&rb)[TK]) {
        const int row = m + half;
        const bool live = row < m_end;
        const int64_t r = min(row, M - 1);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const float v = pa[t][r * ld_dy];
            ra[t] = (live && oka[t]) ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < TK; ++u) {
            const float v = pb[u][r * ld_x];
            rb[u] = (live && okb[u]) ? v : 0.f;
        }
    };
    if (m_begin < m_end) load(m_begin, a, b);
    for (int m = m_begin; m < m_end; m += 2) {
        if (m + 2 < m_end) load(m + 2, an, bn_);            // in flight under this step's MFMAs
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int u = 0; u < TK; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[u], acc[t][u], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TN; ++t) a[t] = an[t];
#pragma unroll
        for (int u = 0; u < TK; ++u) b[u] = bn_[u];
    }
    // C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* out = part + (int64_t)slab * ((((int64_t)N * K) + 3) & ~(int64_t)3);
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int u = 0; u < TK; ++u) {
            const int k = k0 + 32 * u + col;
            if (k >= K) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = n0 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * half;
                if (n < N) out[(int64_t)n * K + k] = acc[t][u][i];
            }
        }
}

// dW[e] = sum over the slabs, in slab order
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int slabs, int64_t elems, float* __restrict__ dW) {
    const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= elems) return;
    const int64_t stride = (elems + 3) & ~(int64_t)3;       // slabs start 16-byte aligned
    if (e + 4 <= elems) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < slabs; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)i * stride + e);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(dW + e) = s;
    } else {
        for (int64_t j = e; j < elems; ++j) {
            float s = 0.f;
            for (int i = 0; i < slabs; ++i) s += part[(int64_t)i * stride + j];
            dW[j] = s;
        }
    }
}

// Number of row slabs the launch of dfol_linear_wgrad_f32 uses (the caller sizes the workspace with it): enough workgroups to
// fill the chip a few times over, at least 64 rows per slab.
extern "C" int dfol_linear_wgrad_slabs(int64_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const bool wide = (N > 256 && N <= 320) || N % 160 == 0;           // 5 x 32 rows per block: 300 -> 320, not 384
    const int tn = wide ? 5 : 4;
    const int blocks = dfol_cdiv(N, 32 * tn) * dfol_cdiv(K, 128);
    const int groups = dfol_cdiv(blocks, 4);
    int slabs = dfol_cdiv(1024, groups);
    const int64_t max_slabs = (M + 63) / 64;
    if (slabs > max_slabs) slabs = (int)max_slabs;
    return slabs < 1 ? 1 : slabs;
}

extern "C" int dfol_linear_wgrad_f32(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, int64_t M, int32_t N, int32_t K,
                                     float* workspace, float* dW, void* stream) {
    DFOL_REQUIRE(M > 0 && M < (1ll << 31) && N > 0 && K > 0, "linear_wgrad: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
    DFOL_REQUIRE(dY && X && workspace && dW, "linear_wgrad: null pointer");
    const bool wide = (N > 256 && N <= 320) || N % 160 == 0;
    const int tn = wide ? 5 : 4;
    const int nb_n = dfol_cdiv(N, 32 * tn), nb_k = dfol_cdiv(K, 128);
    const int groups = dfol_cdiv(nb_n * nb_k, 4);
    const int slabs = dfol_linear_wgrad_slabs(M, N, K);
    int rows_per_slab = dfol_cdiv(M, slabs);
    rows_per_slab += rows_per_slab & 1;                                 // two rows per MFMA step
    hipStream_t st = (hipStream_t)stream;
    if (wide)
        hipLaunchKernelGGL((wgrad_tn_kernel<5, 4>), dim3(slabs, groups), dim3(256), 0, st, dY, ld_dy, X, ld_x, (int)M, N, K, rows_per_slab, nb_n,
                           nb_k, workspace);
    else
        hipLaunchKernelGGL((wgrad_tn_kernel<4, 4>), dim3(slabs, groups), dim3(256), 0, st, dY, ld_dy, X, ld_x, (int)M, N, K, rows_per_slab, nb_n,
                           nb_k, workspace);
    DFOL_LAUNCH_CHECK("linear_wgrad");
    const int64_t elems = (int64_t)N * K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dfol_cdiv(dfol_cdiv(elems, 4), 256)), dim3(256), 0, st, workspace, slabs, elems, dW);
    DFOL_LAUNCH_CHECK("linear_wgrad (reduce)");
    return 0;
}
