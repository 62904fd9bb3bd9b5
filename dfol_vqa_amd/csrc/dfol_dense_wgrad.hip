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

// The same product with 16-byte loads and a four-step prefetch ring (N % 4 == 0 and K % 4 == 0: every layer of the reference's model).
// The order of the MFMA's k index is free (it is summed over) and so is the assignment of output rows / columns to tile rows / columns,
// as long as the epilogue writes with the same map.  So a lane loads FOUR consecutive floats of its row - columns 4c .. 4c+3 for
// c = lane & 31 - and register j of that float4 is the operand of tile j, whose 32 rows are the columns {4i + j}: one
// global_load_dwordx4 feeds four tiles.  A wavefront owns a 128 x 128 block of dW (4 x 4 tiles = all 256 accumulator registers; five
// row tiles do not fit: the allocator then shuffles accumulators through scratch inside the loop); the last row block of N = 300 has 44
// live rows and runs with two row tiles (TN template), so 40 tiles are computed where 37.5 are needed.  The four wavefronts of a
// workgroup take four consecutive row slabs of the SAME block.  One wavefront per SIMD cannot hide HBM latency by switching wavefronts,
// so four steps (eight rows) of operands are in flight under the MFMAs: with a single step ahead the first version of this kernel
// ran at a fifth of the matrix pipe's rate, waiting for memory.
// AVEC = true: the float4 row map above for dY (always four row tiles).  AVEC = false: the narrow last row block (e.g. 44 of 300): a lane
// loads ONE float per row tile, tile t holding the 32 consecutive columns n0 + 32 t + c, so that only TN = ceil(width / 32) tiles run.
template <int TN, bool AVEC>
__device__ __forceinline__ void wgrad_tn4_block(const float* __restrict__ dY, int64_t ld_dy, const float* __restrict__ X, int64_t ld_x, int M,
                                                int N, int K, int m_begin, int m_end, int n0, int k0, int lane, float* __restrict__ out) {
    static_assert(!AVEC || TN == 4, "the float4 row map always fills four tiles");
    constexpr int TK = 4, D = 4, NA = AVEC ? 1 : TN;
    const int col = lane & 31, half = lane >> 5;
    const int kb = k0 + 4 * col;
    const bool okb = kb < K;                                             // (K a multiple of 4: a float4 is all in or all out)
    const float* pb = X + (okb ? kb : 0);
    const float* pa[NA];
    bool oka[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t) {
        const int n = AVEC ? n0 + 4 * col : n0 + 32 * t + col;
        oka[t] = n < N;
        pa[t] = dY + (oka[t] ? n : 0);
    }

    f32x16 acc[TN][TK];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int u = 0; u < TK; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;

    float4 a4[D], b4[D];
    float a1[D][NA];
    // Loads are UNCONDITIONAL (clamped addresses) and land raw in the ring; invalid lanes are zeroed with a bit mask when the step
    // is USED.  (A select on the loaded value makes the compiler predicate the load itself, and masking at load time makes it wait for
    // every load right where it is issued: either way the prefetch ring drains - s_waitcnt vmcnt(0) - on every step.)
    auto keep = [](float v, uint32_t mask) __attribute__((always_inline)) { return __uint_as_float(__float_as_uint(v) & mask); };
    auto load = [&](int m, int s) __attribute__((always_inline)) {
        const int64_t r = min(m + half, M - 1);
        if (AVEC) {
            a4[s] = *reinterpret_cast<const float4*>(pa[0] + r * ld_dy);
        } else {
#pragma unroll
            for (int t = 0; t < NA; ++t) a1[s][t] = pa[t][r * ld_dy];
        }
        b4[s] = *reinterpret_cast<const float4*>(pb + r * ld_x);
    };
    auto mfma = [&](int m, int s) __attribute__((always_inline)) {
        const uint32_t live = (m + half) < m_end ? 0xffffffffu : 0u;
        const uint32_t mb = okb ? live : 0u;
        float av[TN];
        if (AVEC) {
            const uint32_t ma = oka[0] ? live : 0u;
            av[0] = keep(a4[s].x, ma), av[1] = keep(a4[s].y, ma), av[2] = keep(a4[s].z, ma), av[3] = keep(a4[s].w, ma);
        } else {
#pragma unroll
            for (int t = 0; t < TN; ++t) av[t] = keep(a1[s][t], oka[t] ? live : 0u);
        }
        const float bv[4] = {keep(b4[s].x, mb), keep(b4[s].y, mb), keep(b4[s].z, mb), keep(b4[s].w, mb)};
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int u = 0; u < TK; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[u], acc[t][u], 0, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < D; ++s) load(m_begin + 2 * s, s);              // rows beyond the slab load a clamped address and count as zero
    __builtin_amdgcn_sched_barrier(0);
    for (int m = m_begin; m < m_end; m += 2 * D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            mfma(m + 2 * s, s);
            __builtin_amdgcn_sched_barrier(0);          // (left alone, the scheduler sinks the refill next to its use four steps later)
            load(m + 2 * (s + D), s);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D tile: column j = lane & 31, row i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); tile (t, u) holds k = k0 + 4 j + u and
    // n = n0 + 4 i + t (AVEC) or n0 + 32 t + i
    if (okb) {
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
                const int n = AVEC ? n0 + 4 * row + t : n0 + 32 * t + row;
                if (n < N) *reinterpret_cast<float4*>(out + (int64_t)n * K + kb) = make_float4(acc[t][0][i], acc[t][1][i], acc[t][2][i], acc[t][3][i]);
            }
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad_tn4_kernel(
    const float* __restrict__ dY, int64_t ld_dy, const float* __restrict__ X, int64_t ld_x, int M, int N, int K, int rows_per_slab, int nb_n,
    int nb_k, float* __restrict__ part) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bn = blockIdx.y / nb_k, bk = blockIdx.y - bn * nb_k;
    const int slab = blockIdx.x * 4 + wave;
    const int m_begin = slab * rows_per_slab, m_end = min(M, m_begin + rows_per_slab);
    float* out = part + (int64_t)slab * ((((int64_t)N * K) + 3) & ~(int64_t)3);
    const int n0 = bn * 128, k0 = bk * 128, width = N - n0;             // width < 128 only in the last row block
    if (width > 96) wgrad_tn4_block<4, true>(dY, ld_dy, X, ld_x, M, N, K, m_begin, m_end, n0, k0, lane, out);
    else if (width > 64) wgrad_tn4_block<3, false>(dY, ld_dy, X, ld_x, M, N, K, m_begin, m_end, n0, k0, lane, out);
    else if (width > 32) wgrad_tn4_block<2, false>(dY, ld_dy, X, ld_x, M, N, K, m_begin, m_end, n0, k0, lane, out);
    else wgrad_tn4_block<1, false>(dY, ld_dy, X, ld_x, M, N, K, m_begin, m_end, n0, k0, lane, out);
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
// fill the chip a few times over, at least 64 rows per slab, a multiple of four (one slab per wavefront of a workgroup).
extern "C" int dfol_linear_wgrad_slabs(int64_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int blocks = dfol_cdiv(N, 128) * dfol_cdiv(K, 128);
    int wgs = dfol_cdiv(768, blocks);                                   // workgroups per block of dW
    const int64_t max_wgs = (M + 255) / 256;
    if (wgs > max_wgs) wgs = (int)max_wgs;
    if (wgs < 1) wgs = 1;
    return 4 * wgs;
}

extern "C" int dfol_linear_wgrad_f32(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, int64_t M, int32_t N, int32_t K,
                                     float* workspace, float* dW, void* stream) {
    DFOL_REQUIRE(M > 0 && M < (1ll << 31) && N > 0 && K > 0, "linear_wgrad: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
    DFOL_REQUIRE(dY && X && workspace && dW, "linear_wgrad: null pointer");
    const int nb_n = dfol_cdiv(N, 128), nb_k = dfol_cdiv(K, 128);
    const int slabs = dfol_linear_wgrad_slabs(M, N, K);
    int rows_per_slab = dfol_cdiv(M, slabs);
    rows_per_slab = (rows_per_slab + 7) & ~7;                           // two rows per MFMA step, four steps per ring turn
    hipStream_t st = (hipStream_t)stream;
    const bool vec4 = N % 4 == 0 && K % 4 == 0 && ld_dy % 4 == 0 && ((uintptr_t)dY % 16 == 0);      // (X rows may be 8-byte aligned only)
    if (vec4)
        hipLaunchKernelGGL(wgrad_tn4_kernel, dim3(slabs / 4, nb_n * nb_k), dim3(256), 0, st, dY, ld_dy, X, ld_x, (int)M, N, K, rows_per_slab, nb_n,
                           nb_k, workspace);
    else                                                                // odd widths: one float per lane and load, the four wavefronts of
        hipLaunchKernelGGL((wgrad_tn_kernel<4, 4>), dim3(slabs, dfol_cdiv(nb_n * nb_k, 4)), dim3(256), 0, st, dY, ld_dy, X, ld_x, (int)M, N, K,
                           rows_per_slab, nb_n, nb_k, workspace);      // a workgroup on four blocks of the same rows
    DFOL_LAUNCH_CHECK("linear_wgrad");
    const int64_t elems = (int64_t)N * K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dfol_cdiv(dfol_cdiv(elems, 4), 256)), dim3(256), 0, st, workspace, slabs, elems, dW);
    DFOL_LAUNCH_CHECK("linear_wgrad (reduce)");
    return 0;
}
