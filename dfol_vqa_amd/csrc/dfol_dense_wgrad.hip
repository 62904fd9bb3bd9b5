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

#include <stdlib.h>
#include <string.h>

#include <type_traits>

#ifndef W3B_ALIGN
#define W3B_ALIGN 1
#endif
#ifndef W3B_D
#define W3B_D 4                 // steps of rows in flight in the bf16-storage weight gradient (1..4)
#endif

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

// ---------------------------------------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix pipe with fp32 results: the exact three-way operand split of csrc/dfol_pair_split.hip (x = h + m + l,
// six of the nine piece products, fp32 accumulation) - 6/16 of the fp32 pipe's time.  v_mfma_f32_32x32x16_bf16 wants EIGHT consecutive
// k per lane and operand, and k is the row index here, so an operand register holds eight ROWS of one column: a lane loads the float4
// (columns 4c .. 4c+3) of its eight rows m0 + 8 (lane >> 5) + 0..7 - plain coalesced 16-byte loads again - and component t of the
// eight registers, split and packed, is the operand of tile t (same free row / column map as wgrad_tn4_kernel).  Per step of 16 rows
// and wavefront: 16 loads, 96 MFMAs (3072 cycles) and ~350 VALU instructions of splitting at the top of the step; the next step's rows are
// in flight under the MFMAs.
typedef __bf16 w3_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t w3_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void w3_split(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = __float_as_uint(x);
    const float r = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r);
    l = __float_as_uint(r - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ uint32_t w3_pack(uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }
__device__ __forceinline__ void w3_split8(const float (&v)[8], w3_u32x4& h, w3_u32x4& m, w3_u32x4& l) {
    uint32_t ph[8], pm[8], pl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w3_split(v[j], ph[j], pm[j], pl[j]);
    h = w3_u32x4{w3_pack(ph[0], ph[1]), w3_pack(ph[2], ph[3]), w3_pack(ph[4], ph[5]), w3_pack(ph[6], ph[7])};
    m = w3_u32x4{w3_pack(pm[0], pm[1]), w3_pack(pm[2], pm[3]), w3_pack(pm[4], pm[5]), w3_pack(pm[6], pm[7])};
    l = w3_u32x4{w3_pack(pl[0], pl[1]), w3_pack(pl[2], pl[3]), w3_pack(pl[4], pl[5]), w3_pack(pl[6], pl[7])};
}

// Addressing: the rows of a step come through buffer descriptors rebuilt per step from wave-uniform values (base = the step's first
// row, size = what is left of the wavefront's row range), with per-lane byte offsets that never change: no address arithmetic on the
// vector ALU, and rows past the range read as zero from the bounds check (a zero A row switches the row off in every product), so
// there is no tail code either.
// Workgroup = four wavefronts on FOUR BLOCKS OF THE SAME ROWS (consecutive blocks in row-block-major order: 2 x 2 blocks of dW for the
// pair layer); they start together and do the same work per step, so dY and X rows are fetched from HBM once per workgroup and hit
// in L1/L2 for the other wavefronts (a barrier per step to keep them aligned costs 9 % and buys nothing).  (One wavefront per block with four row ranges per workgroup, the first version, read dY twice and
// X three times for the pair layer: 13.9 GB per launch, HBM-bound at 2.96 ms.)  A last group of one or two blocks splits its rows in two
// halves over the four wavefronts; the second half's accumulators go through LDS and are added by the first half's wavefront - a
// fixed order.
typedef uint32_t w3_u32x2 __attribute__((ext_vector_type(2)));

// (every input through readfirstlane: hipcc wraps each buffer load in a waterfall loop unless the descriptor is provably wave-uniform)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t w3_descriptor(const void* base, int64_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7fffffff ? bytes : 0x7fffffff));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

typedef __bf16 w3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float w3_f32x2 __attribute__((ext_vector_type(2)));
// eight fp32 -> eight bf16, round to nearest even (v_cvt_pk_bf16_f32): the operand of the bf16 mode (NP = 1)
__device__ __forceinline__ w3_u32x4 w3_rne8(const float (&v)[8]) {
    w3_u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(w3_f32x2{v[2 * j], v[2 * j + 1]}, w3_bf16x2));
    return r;
}

// AV: floats of dY a lane loads per row = row tiles of the block (4: 128 output rows; 2: a narrow last row block, up to 64 output rows)
// NP: pieces per operand - 3: fp32 results (six piece products); 1: the bf16 mode (operands rounded to bf16, one product, fp32 accumulation)
template <int AV, int NP>
__device__ __forceinline__ void wgrad_tn3_accumulate(const float* __restrict__ dY, int64_t ld_dy, const float* __restrict__ X, int64_t ld_x,
                                                     int rows, int steps, int N, int K, int n0, int k0, int lane, bool want_bias,
                                                     f32x16 (&acc)[AV][4], float (&bs)[AV]) {
    // dY, X: first row of this wavefront's range (wave-uniform); rows: rows in the range (<= 0: nothing to add)
    // bs[t] (want_bias, wave-uniform): this lane's part of the bias gradient, the sum of dY[m][n0 + AV col + t] over its rows
    const int col = lane & 31, half = lane >> 5;
    // columns beyond the matrix read a clamped column: they only feed output rows / columns that are never stored
    const int ca = min(n0 + AV * col, N - AV), cb = min(k0 + 4 * col, K - 4);
    int va[8], vb[8];                                                              // byte offsets of this lane's eight rows within a step
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        va[r] = (int)(((8 * half + r) * ld_dy + ca) * 4);
        vb[r] = (int)(((8 * half + r) * ld_x + cb) * 4);
    }
#pragma unroll
    for (int t = 0; t < AV; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
#pragma unroll
    for (int t = 0; t < AV; ++t) bs[t] = 0.f;

    float ra[8][AV];                                                               // the step's dY and X rows: dead once the pieces are built,
    w3_u32x4 rb[8];                                                                // then refilled with the next step's (in flight under the MFMAs)
    auto load = [&](int s) __attribute__((always_inline)) {
        const int left = rows - 16 * s;                                            // rows of the range from this step on
        const int64_t bytes_a = left > 0 ? ((int64_t)(left - 1) * ld_dy + N) * 4 : 0, bytes_b = left > 0 ? ((int64_t)(left - 1) * ld_x + K) * 4 : 0;
        const auto da = w3_descriptor(dY + (int64_t)s * 16 * ld_dy, bytes_a), db = w3_descriptor(X + (int64_t)s * 16 * ld_x, bytes_b);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (AV == 4) {
                const w3_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(da, va[r], 0, 0);
                ra[r][0] = __uint_as_float(v.x), ra[r][1] = __uint_as_float(v.y), ra[r][2] = __uint_as_float(v.z), ra[r][3] = __uint_as_float(v.w);
            } else {
                const w3_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(da, va[r], 0, 0);
                ra[r][0] = __uint_as_float(v.x), ra[r][1] = __uint_as_float(v.y);
            }
            rb[r] = __builtin_amdgcn_raw_buffer_load_b128(db, vb[r], 0, 0);
        }
    };
    constexpr int PA6[6] = {2, 0, 1, 1, 0, 0}, PB6[6] = {0, 2, 1, 0, 1, 0};       // (A piece, B piece): l*h, h*l, m*m, m*h, h*m, h*h
    // One step: all pieces first (a tile at a time: the split's temporaries are 32 registers per tile), then the next step's loads, then
    // 24 AV MFMAs back to back.  (Building the B pieces of column tile u + 1 between the MFMAs of tile u was tried first: with 256
    // accumulators the allocator then parks VGPRs in the accumulator file and cycles every tile through one AGPR tuple.)
    load(0);
    __builtin_amdgcn_sched_barrier(0);
    for (int s = 0; s < steps; ++s) {
        w3_u32x4 ap[AV][NP], bp[4][NP];
        if (want_bias) {
#pragma unroll
            for (int t = 0; t < AV; ++t)
                bs[t] += ((ra[0][t] + ra[1][t]) + (ra[2][t] + ra[3][t])) + ((ra[4][t] + ra[5][t]) + (ra[6][t] + ra[7][t]));
        }
#pragma unroll
        for (int t = 0; t < AV; ++t) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = ra[r][t];
            if constexpr (NP == 1) ap[t][0] = w3_rne8(v);
            else w3_split8(v, ap[t][0], ap[t][1], ap[t][2]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = __uint_as_float(rb[r][u]);
            if constexpr (NP == 1) bp[u][0] = w3_rne8(v);
            else w3_split8(v, bp[u][0], bp[u][1], bp[u][2]);
            __builtin_amdgcn_sched_barrier(0);
        }
        load(s + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < AV; ++t)
#pragma unroll
                for (int x = NP == 1 ? 5 : 0; x < 6; ++x)                       // (the bf16 mode keeps the last product: piece 0 x piece 0)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(w3_bf16x8, ap[t][PA6[x]]),
                                                                        __builtin_bit_cast(w3_bf16x8, bp[u][PB6[x]]), acc[t][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same with dY and X STORED as bfloat16 (the bf16 mode's per-pair activations: dpre2 [pairs, HID2] and Z [pairs, HID1]): a lane loads
// AV resp. 4 consecutive bf16 of its eight rows (8- / 4-byte buffer loads), and an operand register pair is two rows' halves of one word
// (v_perm_b32) - no conversions, no rounding, half the bytes.  The bias sums widen the same registers to fp32.
template <int AV, int NP>
__device__ __forceinline__ void wgrad_tn3_accumulate(const uint16_t* __restrict__ dY, int64_t ld_dy, const uint16_t* __restrict__ X, int64_t ld_x,
                                                     int rows, int steps, int N, int K, int n0, int k0, int lane, bool want_bias,
                                                     f32x16 (&acc)[AV][4], float (&bs)[AV]) {
    static_assert(NP == 1, "bf16 storage belongs to the bf16 mode");
    constexpr int WA = AV / 2;                                                     // 32-bit words of dY per row and lane
    const int col = lane & 31, half = lane >> 5;
    const int ca = min(n0 + AV * col, N - AV), cb = min(k0 + 4 * col, K - 4);
    int va[8], vb[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        va[r] = (int)(((8 * half + r) * ld_dy + ca) * 2);
        vb[r] = (int)(((8 * half + r) * ld_x + cb) * 2);
    }
#pragma unroll
    for (int t = 0; t < AV; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
#pragma unroll
    for (int t = 0; t < AV; ++t) bs[t] = 0.f;

    // A step is 16 MFMAs here (512 cycles of pipe) against the 96 of the three-piece kernel: one step of loads in flight does not cover
    // HBM latency any more, so the rows of W3B_D steps ahead are (a ring of register sets, 32 registers each).
    constexpr int D = W3B_D;
    uint32_t ra[D][8][WA];
    w3_u32x2 rb[D][8];
    auto load = [&](int s, auto set_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        const int left = rows - 16 * s;
        const int64_t bytes_a = left > 0 ? ((int64_t)(left - 1) * ld_dy + N) * 2 : 0, bytes_b = left > 0 ? ((int64_t)(left - 1) * ld_x + K) * 2 : 0;
        const auto da = w3_descriptor(dY + (int64_t)s * 16 * ld_dy, bytes_a), db = w3_descriptor(X + (int64_t)s * 16 * ld_x, bytes_b);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (AV == 4) {
                const w3_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(da, va[r], 0, 0);
                ra[S][r][0] = v.x, ra[S][r][WA - 1] = v.y;
            } else {
                ra[S][r][0] = __builtin_amdgcn_raw_buffer_load_b32(da, va[r], 0, 0);
            }
            rb[S][r] = __builtin_amdgcn_raw_buffer_load_b64(db, vb[r], 0, 0);
        }
    };
    // eight rows' element `odd` of one word each -> the operand's four registers (row 2j in the low half)
    auto gather = [](uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5, uint32_t w6, uint32_t w7, bool odd)
        __attribute__((always_inline)) {
        return odd ? w3_u32x4{__builtin_amdgcn_perm(w1, w0, 0x07060302u), __builtin_amdgcn_perm(w3, w2, 0x07060302u),
                              __builtin_amdgcn_perm(w5, w4, 0x07060302u), __builtin_amdgcn_perm(w7, w6, 0x07060302u)}
                   : w3_u32x4{__builtin_amdgcn_perm(w1, w0, 0x05040100u), __builtin_amdgcn_perm(w3, w2, 0x05040100u),
                              __builtin_amdgcn_perm(w5, w4, 0x05040100u), __builtin_amdgcn_perm(w7, w6, 0x05040100u)};
    };
    auto one_step = [&](int s, auto set_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        w3_u32x4 ap[AV], bp[4];
        if (want_bias) {
#pragma unroll
            for (int t = 0; t < AV; ++t) {
                float f[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) f[r] = __uint_as_float((t & 1) ? (ra[S][r][t >> 1] & 0xffff0000u) : (ra[S][r][t >> 1] << 16));
                bs[t] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
            }
        }
#pragma unroll
        for (int t = 0; t < AV; ++t)
            ap[t] = gather(ra[S][0][t >> 1], ra[S][1][t >> 1], ra[S][2][t >> 1], ra[S][3][t >> 1], ra[S][4][t >> 1], ra[S][5][t >> 1], ra[S][6][t >> 1],
                           ra[S][7][t >> 1], t & 1);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            bp[u] = gather(rb[S][0][u >> 1], rb[S][1][u >> 1], rb[S][2][u >> 1], rb[S][3][u >> 1], rb[S][4][u >> 1], rb[S][5][u >> 1], rb[S][6][u >> 1],
                           rb[S][7][u >> 1], u & 1);
        __builtin_amdgcn_sched_barrier(0);
        load(s + D, set_tag);                                                      // (past the range: zero bytes, nothing fetched)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < AV; ++t)
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(w3_bf16x8, ap[t]), __builtin_bit_cast(w3_bf16x8, bp[u]), acc[t][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    load(0, std::integral_constant<int, 0>());
    if constexpr (D > 1) load(1, std::integral_constant<int, 1>());
    if constexpr (D > 2) load(2, std::integral_constant<int, 2>());
    if constexpr (D > 3) load(3, std::integral_constant<int, 3>());
    __builtin_amdgcn_sched_barrier(0);
    for (int s = 0; s < steps; s += D) {                                           // (steps past the range multiply zero rows: exact zeros)
#if W3B_ALIGN
        // the four wavefronts of a workgroup read the SAME rows (different column blocks): kept within a ring turn of each other, the
        // second reader of a row finds it in L1 / L2.  Left free-running they drift apart and every wavefront fetches its rows from HBM:
        // 2.11 x the algorithmic bytes at 6.4 TB/s - this kernel, unlike its three-piece form, is HBM-bound.  (Wavefronts that have
        // returned do not count for s_barrier.)  Two further attempts at the byte count, both measured and dropped: handing the four row panels
        // of a 2 x 2 group through LDS (one fetch per panel and workgroup) - same FETCH_SIZE (4.65 GB per launch in isolation: the re-reads
        // were L2 hits already) and same 0.97 ms; dispatching a slab's two groups next to each other ((groups, slabs) grid) - same bytes,
        // 1.1 ms.  What the counters show as 1.65 - 2.1 x is structural: the second workgroup class reads X again and dY's 600-byte rows
        // straddle 128-byte lines.
        __builtin_amdgcn_s_barrier();
#endif
        one_step(s, std::integral_constant<int, 0>());
        if constexpr (D > 1) one_step(s + 1, std::integral_constant<int, 1>());
        if constexpr (D > 2) one_step(s + 2, std::integral_constant<int, 2>());
        if constexpr (D > 3) one_step(s + 3, std::integral_constant<int, 3>());
    }
}

template <int AV, int NP, typename TIN>
__device__ __forceinline__ void wgrad_tn3_block(const TIN* __restrict__ dY, int64_t ld_dy, const TIN* __restrict__ X, int64_t ld_x, int rows,
                                                int steps, int N, int K, int n0, int k0, int lane, int role, float* __restrict__ lds,
                                                float* __restrict__ out, float* __restrict__ db_out) {
    // role 0: accumulate and store; 1: accumulate, add the partner's accumulators from `lds`, store; 2: accumulate into `lds` (the partner)
    // db_out (k0 == 0 wavefronts, or null): this slab's part of the bias gradient
    f32x16 acc[AV][4];
    float bs[AV];
    const bool want_bias = db_out != nullptr && k0 == 0;
    wgrad_tn3_accumulate<AV, NP>(dY, ld_dy, X, ld_x, rows, steps, N, K, n0, k0, lane, want_bias, acc, bs);
    if (role == 2) {
#pragma unroll
        for (int t = 0; t < AV; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) lds[((t * 4 + u) * 16 + i) * 64 + lane] = acc[t][u][i];
#pragma unroll
        for (int t = 0; t < AV; ++t) lds[(256 + t) * 64 + lane] = bs[t];
    }
    __syncthreads();
    if (role == 2) return;
    if (role == 1) {
#pragma unroll
        for (int t = 0; t < AV; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][u][i] += lds[((t * 4 + u) * 16 + i) * 64 + lane];
#pragma unroll
        for (int t = 0; t < AV; ++t) bs[t] += lds[(256 + t) * 64 + lane];
    }
    // D tile: column j = lane & 31, row i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); tile (t, u) holds n = n0 + AV i + t, k = k0 + 4 j + u
    const int half = lane >> 5, kb = k0 + 4 * (lane & 31);
    if (kb < K) {
#pragma unroll
        for (int t = 0; t < AV; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = n0 + AV * ((i & 3) + 8 * (i >> 2) + 4 * half) + t;
                if (n < N) *reinterpret_cast<float4*>(out + (int64_t)n * K + kb) = make_float4(acc[t][0][i], acc[t][1][i], acc[t][2][i], acc[t][3][i]);
            }
    }
    if (want_bias) {                                                               // lanes l and l + 32 hold the two row halves of a column
#pragma unroll
        for (int t = 0; t < AV; ++t) {
            const float other = __shfl_xor(bs[t], 32);
            const int n = n0 + AV * (lane & 31) + t;
            if (half == 0 && n < N) db_out[n] = bs[t] + other;
        }
    }
}

constexpr int W3_LDS_SLOT = 64 * (256 + 4);        // floats a row-half partner hands over: 256 accumulators and 4 bias sums per lane

// grid (row slabs, groups of four blocks); dynamic LDS: two slots when the last group splits its rows (see the launcher), else none
template <int NP, typename TIN = float>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad_tn3_kernel(
    const TIN* __restrict__ dY, int64_t ld_dy, const TIN* __restrict__ X, int64_t ld_x, int M, int N, int K, int rows_per_slab, int nb_k,
    int nb, float* __restrict__ part, float* __restrict__ db_part) {
    extern __shared__ float w3_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int first = blockIdx.y * 4, in_group = min(4, nb - first);               // blocks of this group
    const bool split = in_group <= 2;                                              // then wavefront w: block w % in_group, row half w / in_group
    int blk = first + wave, role = 0, half_id = 0;
    if (split) {
        half_id = wave / in_group;
        blk = first + wave - half_id * in_group;
        role = half_id >= 2 ? 3 : (half_id == 1 ? 2 : 1);                          // (one block: wavefronts 2, 3 have nothing to do)
    } else if (blk >= nb) role = 3;
    const int m_begin = blockIdx.x * rows_per_slab, m_end = min(M, m_begin + rows_per_slab);
    int rows = m_end - m_begin, steps = (rows + 15) >> 4, m_first = m_begin;
    if (split) {
        const int first_half = ((rows + 31) >> 5) << 4;                            // rows of the first half: a multiple of 16, the larger one
        steps = first_half >> 4;
        if (half_id == 1) m_first = m_begin + first_half, rows -= first_half;
        else rows = min(rows, first_half);
    }
    if (role == 3) {                                                               // (the one barrier of wgrad_tn3_block)
        __syncthreads();
        return;
    }
    const int bn = blk / nb_k, bk = blk - bn * nb_k;
    const int n0 = bn * 128, k0 = bk * 128;
    float* out = part + (int64_t)blockIdx.x * ((((int64_t)N * K) + 3) & ~(int64_t)3);
    float* lds = w3_lds + (split ? wave - half_id * in_group : 0) * W3_LDS_SLOT;   // wavefronts w and w + in_group: the same block, one slot
    float* db_out = db_part ? db_part + (int64_t)blockIdx.x * ((N + 3) & ~3) : nullptr;
    const TIN* a = dY + (int64_t)m_first * ld_dy;
    const TIN* b = X + (int64_t)m_first * ld_x;
    if (N - n0 > 64) wgrad_tn3_block<4, NP, TIN>(a, ld_dy, b, ld_x, rows, steps, N, K, n0, k0, lane, role, lds, out, db_out);
    else wgrad_tn3_block<2, NP, TIN>(a, ld_dy, b, ld_x, rows, steps, N, K, n0, k0, lane, role, lds, out, db_out);
}

// dW[e] = sum over the slabs in a fixed order: a workgroup takes 16 groups of four elements x 16 slab phases (phase p adds slabs p,
// p + 16, ... in order), then the phases are added in order.  (One thread per element group walking all slabs, the first version, was
// 75 workgroups of serial loads for the pair layer.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int slabs, int64_t elems, float* __restrict__ dW) {
    __shared__ float4 sums[16][16];
    const int g = threadIdx.x & 15, phase = threadIdx.x >> 4;
    const int64_t e = ((int64_t)blockIdx.x * 16 + g) * 4;
    const int64_t stride = (elems + 3) & ~(int64_t)3;       // slabs start 16-byte aligned; the padding floats of a last group are never stored
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < elems)
        for (int i = phase; i < slabs; i += 16) {
            const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)i * stride + e);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    sums[phase][g] = s;
    __syncthreads();
    if (phase != 0 || e >= elems) return;
#pragma unroll
    for (int p = 1; p < 16; ++p) {
        const float4 v = sums[p][g];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (e + 4 <= elems) *reinterpret_cast<float4*>(dW + e) = s;
    else {
        const float t[4] = {s.x, s.y, s.z, s.w};
        for (int64_t k = e; k < elems; ++k) dW[k] = t[k - e];
    }
}

// Bias gradient beside the fp32-pipe kernels (the bf16x3 kernel takes it from the dY rows it loads anyway): column sums of a row slab,
// four row phases per workgroup added in a fixed order.  grid (slabs, ceil(N / 64)).
__global__ __launch_bounds__(256) void wgrad_colsum_kernel(const float* __restrict__ dY, int64_t ld_dy, int M, int N, int rows_per_slab,
                                                           float* __restrict__ db_part) {
    __shared__ float sums[4][64];
    const int c = threadIdx.x & 63, phase = threadIdx.x >> 6, n = blockIdx.y * 64 + c;
    const int m_begin = blockIdx.x * rows_per_slab, m_end = min(M, m_begin + rows_per_slab);
    float s = 0.f;
    if (n < N)
        for (int m = m_begin + phase; m < m_end; m += 4) s += dY[(int64_t)m * ld_dy + n];
    sums[phase][c] = s;
    __syncthreads();
    if (phase == 0 && n < N) db_part[(int64_t)blockIdx.x * ((N + 3) & ~3) + n] = ((sums[0][c] + sums[1][c]) + sums[2][c]) + sums[3][c];
}

// Row slabs of the bf16x3 kernel: one round of workgroups for the full groups of four blocks (a workgroup owns a CU: one wavefront per
// SIMD), at least 256 rows (16 steps) per slab.
static int wgrad_tn3_slabs(int64_t M, int nb) {
    const int full = nb / 4;
    int64_t s = full > 0 ? 256 / full : 256;
    s = std::min<int64_t>(s, (M + 255) / 256);
    return (int)std::max<int64_t>(s, 1);
}
static int wgrad_tn4_slabs(int64_t M, int nb) {
    int wgs = dfol_cdiv(768, nb);                                       // workgroups per block of dW
    const int64_t max_wgs = (M + 255) / 256;
    if (wgs > max_wgs) wgs = (int)max_wgs;
    if (wgs < 1) wgs = 1;
    return 4 * wgs;                                                     // one slab per wavefront of a workgroup
}

// Row slabs the caller sizes the workspace of dfol_linear_wgrad_f32 for (slabs x N x K floats, rounded up to 4 floats per slab): the
// launch uses this many or fewer.
extern "C" int dfol_linear_wgrad_slabs(int64_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int nb = dfol_cdiv(N, 128) * dfol_cdiv(K, 128);
    return std::max(wgrad_tn3_slabs(M, nb), wgrad_tn4_slabs(M, nb));
}

// Floats of workspace dfol_linear_wgrad_bias_f32 needs: the slabs' partial dW blocks, then their partial bias gradients
extern "C" int64_t dfol_linear_wgrad_workspace(int64_t M, int32_t N, int32_t K) {
    const int64_t slabs = dfol_linear_wgrad_slabs(M, N, K);
    return slabs * (((((int64_t)N * K) + 3) & ~(int64_t)3) + ((N + 3) & ~3));
}

static int wgrad_launch(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, int64_t M, int32_t N, int32_t K, float* workspace,
                        float* dW, float* db, void* stream, bool bf16_mode) {
    DFOL_REQUIRE(M > 0 && M < (1ll << 31) && N > 0 && K > 0, "linear_wgrad: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
    DFOL_REQUIRE(dY && X && workspace && dW, "linear_wgrad: null pointer");
    float* db_part = db ? workspace + (int64_t)dfol_linear_wgrad_slabs(M, N, K) * ((((int64_t)N * K) + 3) & ~(int64_t)3) : nullptr;
    const int nb_n = dfol_cdiv(N, 128), nb_k = dfol_cdiv(K, 128);
    const int nb = nb_n * nb_k;
    int slabs = wgrad_tn4_slabs(M, nb);
    int rows_per_slab = dfol_cdiv(M, slabs);
    rows_per_slab = (rows_per_slab + 7) & ~7;                           // two rows per MFMA step, four steps per ring turn
    hipStream_t st = (hipStream_t)stream;
    const bool vec4 = N % 4 == 0 && K % 4 == 0 && ld_dy % 4 == 0 && ((uintptr_t)dY % 16 == 0);      // (X rows may be 8-byte aligned only)
    const char* math = getenv("DFOL_WGRAD_MATH");                       // "f32": the fp32 matrix pipe (exact products); default: bf16x3
    const bool f32_pipe = !bf16_mode && math && !strcmp(math, "f32");
    bool bias_done = false;
    if (vec4 && !f32_pipe && N >= 4 && K >= 4) {                        // fp32 results from the bf16 matrix pipe (X rows: any 4-byte alignment)
        slabs = wgrad_tn3_slabs(M, nb);
        rows_per_slab = (dfol_cdiv(M, slabs) + 15) & ~15;               // sixteen rows per MFMA step
        DFOL_REQUIRE(16 * std::max(ld_dy, ld_x) * 4 < (1ll << 31), "linear_wgrad: row stride too large (%lld)", (long long)std::max(ld_dy, ld_x));
        const int groups = dfol_cdiv(nb, 4);
        const size_t lds = nb % 4 == 1 || nb % 4 == 2 ? 2 * W3_LDS_SLOT * sizeof(float) : 0;  // the last group splits its rows (two slots)
        static const hipError_t lds_ok = hipFuncSetAttribute((const void*)wgrad_tn3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             2 * W3_LDS_SLOT * sizeof(float));
        static const hipError_t lds_ok1 = hipFuncSetAttribute((const void*)wgrad_tn3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                              2 * W3_LDS_SLOT * sizeof(float));
        DFOL_REQUIRE(lds_ok == hipSuccess && lds_ok1 == hipSuccess, "linear_wgrad: cannot reserve 130 KB of LDS (%s)", hipGetErrorString(lds_ok));
        if (bf16_mode)
            hipLaunchKernelGGL(wgrad_tn3_kernel<1>, dim3(slabs, groups), dim3(256), lds, st, dY, ld_dy, X, ld_x, (int)M, N, K, rows_per_slab, nb_k,
                               nb, workspace, db_part);
        else
            hipLaunchKernelGGL(wgrad_tn3_kernel<3>, dim3(slabs, groups), dim3(256), lds, st, dY, ld_dy, X, ld_x, (int)M, N, K, rows_per_slab, nb_k,
                               nb, workspace, db_part);
        bias_done = true;
    } else if (vec4)
        hipLaunchKernelGGL(wgrad_tn4_kernel, dim3(slabs / 4, nb_n * nb_k), dim3(256), 0, st, dY, ld_dy, X, ld_x, (int)M, N, K, rows_per_slab, nb_n,
                           nb_k, workspace);
    else                                                                // odd widths: one float per lane and load, the four wavefronts of
        hipLaunchKernelGGL((wgrad_tn_kernel<4, 4>), dim3(slabs, dfol_cdiv(nb_n * nb_k, 4)), dim3(256), 0, st, dY, ld_dy, X, ld_x, (int)M, N, K,
                           rows_per_slab, nb_n, nb_k, workspace);      // a workgroup on four blocks of the same rows
    DFOL_LAUNCH_CHECK("linear_wgrad");
    const int64_t elems = (int64_t)N * K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dfol_cdiv(dfol_cdiv(elems, 4), 16)), dim3(256), 0, st, workspace, slabs, elems, dW);
    DFOL_LAUNCH_CHECK("linear_wgrad (reduce)");
    if (db) {
        if (!bias_done) {
            hipLaunchKernelGGL(wgrad_colsum_kernel, dim3(slabs, dfol_cdiv(N, 64)), dim3(256), 0, st, dY, ld_dy, (int)M, N, rows_per_slab, db_part);
            DFOL_LAUNCH_CHECK("linear_wgrad (column sums)");
        }
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dfol_cdiv(dfol_cdiv(N, 4), 16)), dim3(256), 0, st, db_part, slabs, (int64_t)N, db);
        DFOL_LAUNCH_CHECK("linear_wgrad (bias reduce)");
    }
    return 0;
}

extern "C" int dfol_linear_wgrad_bias_f32(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, int64_t M, int32_t N, int32_t K,
                                          float* workspace, float* dW, float* db, void* stream) {
    return wgrad_launch(dY, ld_dy, X, ld_x, M, N, K, workspace, dW, db, stream, false);
}

// bf16 mode (BASELINE configs[3]): both operands rounded to bf16, one product per pair, fp32 accumulation; the bias gradient stays an
// fp32 sum.  Shapes the bf16x3 kernel does not take (widths that are not multiples of 4) run in fp32 as above.
extern "C" int dfol_linear_wgrad_bias_bf16(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, int64_t M, int32_t N, int32_t K,
                                           float* workspace, float* dW, float* db, void* stream) {
    return wgrad_launch(dY, ld_dy, X, ld_x, M, N, K, workspace, dW, db, stream, true);
}

extern "C" int dfol_linear_wgrad_f32(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, int64_t M, int32_t N, int32_t K,
                                     float* workspace, float* dW, void* stream) {
    return wgrad_launch(dY, ld_dy, X, ld_x, M, N, K, workspace, dW, nullptr, stream, false);
}

// bf16 mode with bf16 STORAGE of both operands (the per-pair activations dpre2 [M, N] and Z [M, K] as bfloat16, rows 4-byte / 8-byte
// aligned: ld_dy % 2 == 0, ld_x % 4 == 0 in elements, N and K multiples of 4): one product per pair, fp32 accumulation, fp32 dW and db.
extern "C" int dfol_linear_wgrad_bias_bf16_bf16(const void* dY_bf16, int64_t ld_dy, const void* X_bf16, int64_t ld_x, int64_t M, int32_t N, int32_t K,
                                                float* workspace, float* dW, float* db, void* stream) {
    DFOL_REQUIRE(M > 0 && M < (1ll << 31) && N >= 4 && K >= 4 && N % 4 == 0 && K % 4 == 0, "linear_wgrad_bf16_bf16: bad sizes M=%lld N=%d K=%d (N, K multiples of 4)",
                 (long long)M, N, K);
    DFOL_REQUIRE(dY_bf16 && X_bf16 && workspace && dW, "linear_wgrad_bf16_bf16: null pointer");
    DFOL_REQUIRE(ld_dy % 4 == 0 && ld_x % 4 == 0 && ((uintptr_t)dY_bf16 % 8 == 0) && ((uintptr_t)X_bf16 % 8 == 0), "linear_wgrad_bf16_bf16: rows must be 8-byte aligned");
    float* db_part = db ? workspace + (int64_t)dfol_linear_wgrad_slabs(M, N, K) * ((((int64_t)N * K) + 3) & ~(int64_t)3) : nullptr;
    const int nb_k = dfol_cdiv(K, 128), nb = dfol_cdiv(N, 128) * nb_k;
    const int slabs = wgrad_tn3_slabs(M, nb);
    const int rows_per_slab = (dfol_cdiv(M, slabs) + 15) & ~15;
    DFOL_REQUIRE(16 * std::max(ld_dy, ld_x) * 2 < (1ll << 31), "linear_wgrad_bf16_bf16: row stride too large (%lld)", (long long)std::max(ld_dy, ld_x));
    const int groups = dfol_cdiv(nb, 4);
    const size_t lds = nb % 4 == 1 || nb % 4 == 2 ? 2 * W3_LDS_SLOT * sizeof(float) : 0;
    static const hipError_t lds_ok = hipFuncSetAttribute((const void*)wgrad_tn3_kernel<1, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         2 * W3_LDS_SLOT * sizeof(float));
    DFOL_REQUIRE(lds_ok == hipSuccess, "linear_wgrad_bf16_bf16: cannot reserve 130 KB of LDS (%s)", hipGetErrorString(lds_ok));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((wgrad_tn3_kernel<1, uint16_t>), dim3(slabs, groups), dim3(256), lds, st, (const uint16_t*)dY_bf16, ld_dy, (const uint16_t*)X_bf16,
                       ld_x, (int)M, N, K, rows_per_slab, nb_k, nb, workspace, db_part);
    DFOL_LAUNCH_CHECK("linear_wgrad_bf16_bf16");
    const int64_t elems = (int64_t)N * K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dfol_cdiv(dfol_cdiv(elems, 4), 16)), dim3(256), 0, st, workspace, slabs, elems, dW);
    DFOL_LAUNCH_CHECK("linear_wgrad_bf16_bf16 (reduce)");
    if (db) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dfol_cdiv(dfol_cdiv(N, 4), 16)), dim3(256), 0, st, db_part, slabs, (int64_t)N, db);
        DFOL_LAUNCH_CHECK("linear_wgrad_bf16_bf16 (bias reduce)");
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4: the pair layer's weight gradient with dpre2 PRODUCED in the kernel (the logit layer's backward folded in, like
// dfol_pair_dz_fused_f32 does for the input gradient):   dW2[j][c] = sum_r dpre2[r][j] Z[r][c],   db2[j] = sum_r dpre2[r][j],
// dpre2[r][j] = dx[r] E[p(r)][j] h (1 - h), h = Sigmoid(pre2[r][j]).  pre2 and Z are read ONCE (the bf16x3 kernel above reads Z once per
// workgroup class and dpre2 - which no longer exists - on top: 1.6 x the algorithmic bytes), and the operands are two fp16 pieces, three
// products on v_mfma_f32_32x32x16_f16 instead of six bf16 ones.
//
// One workgroup (8 wavefronts, two per SIMD) per row slab, ALL of dW2 in its accumulators: wavefront (a, b) owns the 5 x 2 tiles of 32 x 32
// at rows 160 a .. and columns 64 b .. (160 accumulator registers), H2 <= 320, H1 <= 256.  The operands go through LDS: per macro step of
// 32 rows, thread (column quad c, row octet o) loads the four columns of its eight rows (plain 16-byte loads), builds dpre2 resp. takes Z,
// splits into fp16 pieces and writes, per column, the 16-byte MFMA operand entry (eight consecutive rows = eight consecutive k of the
// MFMA) of both pieces; every wavefront then reads its tiles' entries (conflict-free ds_read_b128).  Two LDS buffers (2 x 72 KB), one
// barrier per macro step; the next macro step's rows are in flight under the MFMAs.
// A row's gradient has no natural scale and the contraction runs over the rows, so ONE power of two S for the whole launch (from the
// caller: S max_r |dx[r]| max|E[p(r)]| / 4 in [2^13, 2^14)) scales dpre2 into fp16's range: an element's error is
// max(2^-23 |a|, 2^-39 max_r(|dx[r]| max|E[p(r)]|)) - rows that far below the largest do not move the sum.  Z is split unscaled (ELU outputs).
// one fp32 -> its bfloat16 (round to nearest even) as the bits of the fp32 it widens back to
__device__ __forceinline__ uint32_t w3_pack1(float x) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(w3_f32x2{x, 0.f}, w3_bf16x2)) << 16;
}
typedef _Float16 pw_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 pw_f16x8 __attribute__((ext_vector_type(8)));

constexpr int PW_TA = 10, PW_TB = 8;                                   // 32-wide tiles of H2 (<= 320) and of H1 (<= 256)
constexpr int PW_A_ENT = 2 * 2 * PW_TA * 64, PW_B_ENT = 2 * 2 * PW_TB * 64;   // 16-byte entries per buffer: [k-step][piece][tile][lane]
constexpr int PW_BUF = PW_A_ENT + PW_B_ENT;

__device__ __forceinline__ void pw_split2(float x0, float x1, uint32_t& h, uint32_t& l) {
    const w3_f32x2 x = {x0, x1};
    const pw_f16x2 hh = __builtin_convertvector(x, pw_f16x2);
    const w3_f32x2 r = x - __builtin_convertvector(hh, w3_f32x2);
    h = __builtin_bit_cast(uint32_t, hh);
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, pw_f16x2));
}
__device__ __forceinline__ void pw_split8(const float (&v)[8], w3_u32x4& h, w3_u32x4& l) {
    uint32_t hh[4], ll[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pw_split2(v[2 * j], v[2 * j + 1], hh[j], ll[j]);
    h = w3_u32x4{hh[0], hh[1], hh[2], hh[3]};
    l = w3_u32x4{ll[0], ll[1], ll[2], ll[3]};
}
__device__ __forceinline__ float pw_dsigmoid(float x) {
    const float h = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
    return h * (1.0f - h);
}

// CPT: columns of dpre2 per thread - 3 when H2 is a multiple of 3 (HID2 = 300: 100 x 4 = 400 building threads with 24 prefetched registers
// each; with 4 - 300 threads, 32 registers - the allocator runs out under the MFMAs and spills a prefetched row right behind its load).
//
// Schedule (second version; the first ran "build all of step s, barrier, multiply step s" and took the SUM of the two phases - 1.2 ms
// + 0.77 ms at 256 x 100 objects - because every wavefront of the CU was in the same phase): the MFMAs of step s are interleaved, in
// the instruction stream of every wavefront, with the building of step s + 1 into the other LDS buffer - two 32 x 32 tile rows of MFMAs
// (12 instructions, 384 cycles of pipe), then one chunk of vector work (a pair of dpre2 rows, or the Z columns), and so on - so the
// SIMD's two wavefronts find each other's gaps.  A pair of rows leaves for LDS as soon as it is built (4-byte pieces of the 16-byte
// operand entries) and its registers are refilled at once with the same rows of step s + 2: one register set is both the prefetch ring
// and the work space.
typedef uint32_t w3_u32x3 __attribute__((ext_vector_type(3)));
// SUMS: the sums of the logit layer's backward from the same pass (every h and dx is in registers here): per thread the running sums of
// dx h (dE of the current predicate) and of dpre2 (db2) over its rows; the dE sums are flushed where the thread's rows cross into the
// next predicate - its own first row past the boundary, or, for threads whose rows of the boundary step all lie before it, the step
// after - into partials indexed by (slab + predicate, row octet), which pair_sums_reduce_kernel adds in a fixed order.  Needs every
// predicate to own at least 32 rows (or none): a macro step then holds at most one boundary (the launcher's caller checks 64).
// BIO (the bf16 mode): pre2 and Z are rows of bfloat16 (8-byte aligned rows, strides in elements), four columns of dpre2 per thread, ONE
// bf16 piece per operand and one product on v_mfma_f32_32x32x16_bf16; dpre2 is rounded to bfloat16 exactly as dfol_pair_logit_bwd_bf16
// stores it (no scaling: bf16 has fp32's exponent range; `scale` is not read).
template <int CPT, bool SUMS, bool BIO = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void pair_wgrad_fused_kernel(
    const void* __restrict__ P2v, int64_t ld_p2, const float* __restrict__ G, const int32_t* __restrict__ RP, const int64_t* __restrict__ pred_off,
    const float* __restrict__ E, int64_t ld_e, const float* __restrict__ scale, const void* __restrict__ Zv, int64_t ld_z, int M, int H2, int H1,
    int rows_per_slab, float* __restrict__ part, float* __restrict__ de_part, float* __restrict__ db_part) {
    static_assert(!BIO || CPT == 4, "bf16 storage: four columns (8 bytes) per thread and row");
    typedef typename std::conditional<BIO, uint16_t, float>::type TIN;
    const TIN* __restrict__ P2 = reinterpret_cast<const TIN*>(P2v);
    const TIN* __restrict__ Z = reinterpret_cast<const TIN*>(Zv);
    constexpr int EB = sizeof(TIN), NPC = BIO ? 1 : 2;                // bytes per stored element; pieces per operand
    constexpr int A_ENT = 2 * NPC * PW_TA * 64, B_ENT = 2 * NPC * PW_TB * 64, BUFK = A_ENT + B_ENT;      // 16-byte entries: [k-step][piece][tile][lane]
    extern __shared__ __attribute__((aligned(16))) w3_u32x4 pw_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m_begin = blockIdx.x * rows_per_slab, m_end = min(M, m_begin + rows_per_slab);
    const int steps = (m_end - m_begin + 31) >> 5;
    const int QA = H2 / CPT, QB = H1 >> 1;
    const float S = BIO ? 1.0f : scale[0], invS = BIO ? 1.0f : scale[1];

    // Roles per macro step of 32 rows.  dpre2: thread (column group ca, row octet oa) for tid < 4 QA - eight rows x CPT columns, the
    // expensive part (a Sigmoid derivative per element).  Z: thread (column pair cz, row octet oz) for tid < 4 QB - eight rows x two columns.
    const bool is_a = tid < 4 * QA;
    const int ca = is_a ? tid % QA : 0, oa = is_a ? tid / QA : 0;
    const bool is_z = tid < 4 * QB;
    const int cz = is_z ? tid % QB : 0, oz = is_z ? tid / QB : 0;

    for (int i = tid; i < 2 * BUFK; i += 512) pw_lds[i] = w3_u32x4{0u, 0u, 0u, 0u};      // (columns past the matrices stay zero for good)

    f32x16 acc[5][2];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;

    // The rows of a macro step come through buffer descriptors rebuilt per step from wave-uniform values (base = the step's first row,
    // size = what is left of the slab), with per-lane byte offsets that never change and the row within the octet as the scalar offset:
    // no 64-bit address arithmetic on the vector ALU, and rows past the slab read as zero (dx = 0 switches such a row off).
    typename std::conditional<BIO, w3_u32x2, typename std::conditional<CPT == 3, w3_u32x3, w3_u32x4>::type>::type xa[8];
    typename std::conditional<BIO, uint32_t, w3_u32x2>::type xz[8];
    // The predicate of the step's first row and the row its range ends at, wave-uniform (the rows of a predicate are contiguous and
    // row_pred is non-decreasing): an octet that ends before that row reads the predicate's embedding columns through a uniform base; the
    // few octets at a boundary look their rows up one by one.
    int p_cur = 0, end_cur = 0;                                        // (M < 2^31)
    if (steps > 0) {
        p_cur = __builtin_amdgcn_readfirstlane(RP[m_begin]);
        end_cur = __builtin_amdgcn_readfirstlane((int)pred_off[p_cur + 1]);
    }
    // The thread's constants (byte offsets of its loads, its columns) live in LDS and are re-read where they are used: registers that would
    // otherwise sit idle under the MFMAs, where accumulators + operand fragments + the row ring fill the 256 of a wavefront.
    int* cst = reinterpret_cast<int*>(pw_lds + 2 * BUFK) + tid;       // [2][512] words, then (SUMS) the threads' running sums [2 CPT][512]
    const int rs_p = (int)(ld_p2 * EB), rs_z = (int)(ld_z * EB);
    // the 16-byte LDS entry of column `col` (tile col >> 5, row col & 31 of the MFMA operand) for the eight rows of octet o
    auto entry = [&](int col, int o, int tiles) __attribute__((always_inline)) { return ((o >> 1) * NPC * tiles + (col >> 5)) * 64 + (col & 31) + 32 * (o & 1); };
    cst[0] = ca | (oa << 16);                                          // (two words per thread; everything else is derived where it is used)
    cst[512] = cz | (oz << 16);
    auto k_oa = [&]() __attribute__((always_inline)) { return cst[0] >> 16; };
    auto k_col0 = [&]() __attribute__((always_inline)) { return CPT * (cst[0] & 0xffff); };                    // the thread's first column of dpre2
    auto k_goff = [&]() __attribute__((always_inline)) { return 32 * (cst[0] >> 16); };                        // byte offset of its octet's dx
    auto k_voff_a = [&]() __attribute__((always_inline)) { const int w = cst[0]; return (8 * (w >> 16) * (int)ld_p2 + CPT * (w & 0xffff)) * EB; };
    auto k_voff_z = [&]() __attribute__((always_inline)) { const int w = cst[512]; return (8 * (w >> 16) * (int)ld_z + 2 * (w & 0xffff)) * EB; };
    auto k_at_z = [&]() __attribute__((always_inline)) { const int w = cst[512]; return A_ENT + entry(2 * (w & 0xffff), w >> 16, PW_TB); };   // entry of its first Z column

    struct Desc { __amdgpu_buffer_rsrc_t p, z, g; };
    auto descriptors = [&](int s) __attribute__((always_inline)) {     // step s (past the slab: empty ranges, every load returns zero)
        const int first = m_begin + 32 * s, left = max(m_end - first, 0);
        Desc d;
        d.p = w3_descriptor(P2 + (int64_t)first * ld_p2, left > 0 ? ((int64_t)(left - 1) * ld_p2 + H2) * EB : 0);
        d.z = w3_descriptor(Z + (int64_t)first * ld_z, left > 0 ? ((int64_t)(left - 1) * ld_z + H1) * EB : 0);
        d.g = w3_descriptor(G + first, (int64_t)left * 4);
        return d;
    };
    auto load_a = [&](const Desc& d, int i) __attribute__((always_inline)) {           // row i of the thread's octet
        const int voff_a = k_voff_a();
        if constexpr (BIO) xa[i] = __builtin_amdgcn_raw_buffer_load_b64(d.p, voff_a, i * rs_p, 0);
        else if constexpr (CPT == 3) xa[i] = __builtin_amdgcn_raw_buffer_load_b96(d.p, voff_a, i * rs_p, 0);
        else xa[i] = __builtin_amdgcn_raw_buffer_load_b128(d.p, voff_a, i * rs_p, 0);
    };
    auto load_z = [&](const Desc& d) __attribute__((always_inline)) {
        const int voff_z = k_voff_z();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (BIO) xz[i] = __builtin_amdgcn_raw_buffer_load_b32(d.z, voff_z, i * rs_z, 0);
            else xz[i] = __builtin_amdgcn_raw_buffer_load_b64(d.z, voff_z, i * rs_z, 0);
        }
    };

    // ---- building step s (its rows are in xa / xz) into `buf`, a chunk at a time; `next`: the descriptors of step s + 1, whose rows refill
    // the registers as they are released
    // SUMS: the running sums live in LDS (the accumulators leave no register that survives the MFMAs: held in registers they were
    // spilled to scratch and reloaded four times a step, 2.44 ms against 1.77 for the kernel without them) and are read, added to and
    // written back by every pair of rows
    float* sums = reinterpret_cast<float*>(pw_lds + 2 * BUFK) + 1024 + tid;       // [2 CPT][512]: dE sums, then db2 sums
    if constexpr (SUMS) {
#pragma unroll
        for (int t = 0; t < 2 * CPT; ++t) sums[512 * t] = 0.f;
    }
    auto flush_de = [&](int p) __attribute__((always_inline)) {        // this thread's dE sums of predicate p -> partial row (slab + p, octet); cleared
        if constexpr (SUMS) {
            float* o = de_part + (((int64_t)blockIdx.x + p) * 4 + k_oa()) * H2 + k_col0();
#pragma unroll
            for (int t = 0; t < CPT; ++t) o[t] = sums[512 * t] * invS, sums[512 * t] = 0.f;
        }
    };
    bool same = false;
    float ev[CPT];
    w3_u32x2 gq;                                                       // dx of the row pair that comes next
    auto begin_a = [&](int s, const Desc& d) __attribute__((always_inline)) {          // before the first pair of rows
        if (is_a) {
            const int goff = k_goff(), col0 = k_col0();
            const int first = m_begin + 32 * s + (goff >> 2);         // the octet's first row
            same = first + 7 < m_end && first + 7 < end_cur;          // (all eight rows exist and belong to p_cur)
#pragma unroll
            for (int t = 0; t < CPT; ++t) ev[t] = 0.f;
            if (same) {
#pragma unroll
                for (int t = 0; t < CPT; ++t) ev[t] = E[(int64_t)p_cur * ld_e + col0 + t];
            }
            gq = __builtin_amdgcn_raw_buffer_load_b64(d.g, goff, 0, 0);
        }
    };
    auto pair_a = [&](int s, int k, const Desc& d, const Desc& next, uint32_t* __restrict__ buf32) __attribute__((always_inline)) {   // rows 2 k, 2 k + 1
        if (is_a) {
            const int goff = k_goff(), col0 = k_col0(), o = k_oa();
            const int first = m_begin + 32 * s + (goff >> 2);
            float v[2][CPT];
            float sde[SUMS ? CPT : 1], sdb[SUMS ? CPT : 1];            // this pair's share of the running sums (added to LDS once, below)
#pragma unroll
            for (int t = 0; t < (SUMS ? CPT : 1); ++t) sde[t] = sdb[t] = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = 2 * k + r;
                float e[CPT];
#pragma unroll
                for (int t = 0; t < CPT; ++t) e[t] = ev[t];
                if (!same) {                                           // (an octet across two predicates, or cut by the slab's end)
                    const float* er = E + (int64_t)RP[min(first + i, m_end - 1)] * ld_e + col0;
#pragma unroll
                    for (int t = 0; t < CPT; ++t) e[t] = er[t];
                }
                const float gs = __uint_as_float(gq[r]) * S;          // (rows past the slab were read as zero)
                if constexpr (SUMS) {
                    if (!same) {                                       // this thread's first row past the predicate's end: its sums so far belong to p_cur
                        const int rr = first + i;
                        if (rr >= end_cur && (i == 0 || rr - 1 < end_cur)) {
#pragma unroll
                            for (int t = 0; t < CPT; ++t) sums[512 * t] += sde[t], sde[t] = 0.f;       // (the pair's first row, if it lies before the end)
                            flush_de(p_cur);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < CPT; ++t) {
                    float xv;
                    if constexpr (BIO) xv = __uint_as_float((t & 1) ? (xa[i][t >> 1] & 0xffff0000u) : (xa[i][t >> 1] << 16));
                    else xv = __uint_as_float(xa[i][t]);
                    const float hh = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * xv));
                    if constexpr (BIO) v[r][t] = __uint_as_float(w3_pack1(gs * e[t] * hh * (1.0f - hh)));      // (the value dfol_pair_logit_bwd_bf16 stores)
                    else v[r][t] = (gs * e[t]) * (hh * (1.0f - hh));
                    if constexpr (SUMS) {
                        sde[t] = fmaf(gs, hh, sde[t]);
                        sdb[t] += v[r][t];
                    }
                }
            }
            if constexpr (SUMS) {
#pragma unroll
                for (int t = 0; t < CPT; ++t) sums[512 * t] += sde[t], sums[512 * (CPT + t)] += sdb[t];
            }
            if (k < 3) gq = __builtin_amdgcn_raw_buffer_load_b64(d.g, goff, 8 * (k + 1), 0);
            load_a(next, 2 * k);                                       // the same rows of the next step: in flight for a whole step
            load_a(next, 2 * k + 1);
#pragma unroll
            for (int t = 0; t < CPT; ++t) {
                const int at = entry(col0 + t, o, PW_TA) * 4 + k;      // 4-byte piece k of the entry: rows 2 k, 2 k + 1
                if constexpr (BIO) {
                    buf32[at] = (__float_as_uint(v[0][t]) >> 16) | (__float_as_uint(v[1][t]) & 0xffff0000u);      // (rounded above: the high halves are the bf16 bits)
                } else {
                    uint32_t h, l;
                    pw_split2(v[0][t], v[1][t], h, l);
                    buf32[at] = h;
                    buf32[at + PW_TA * 64 * 4] = l;
                }
            }
        }
    };
    auto columns_z = [&](const Desc& next, w3_u32x4* __restrict__ buf) __attribute__((always_inline)) {
        if (is_z) {
            const int at = k_at_z();
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if constexpr (BIO) {                                   // eight rows' halfword c of one word each -> the entry's four words (row 2 j in the low half)
                    w3_u32x4 en;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        en[j] = c ? ((xz[2 * j] >> 16) | (xz[2 * j + 1] & 0xffff0000u)) : ((xz[2 * j] & 0xffffu) | (xz[2 * j + 1] << 16));
                    buf[at + c] = en;
                } else {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = __uint_as_float(xz[i][c]);
                    w3_u32x4 h, l;
                    pw_split8(v, h, l);
                    buf[at + c] = h;
                    buf[at + c + PW_TB * 64] = l;
                }
            }
            load_z(next);
        }
    };
    const int ta0 = 5 * (wave >> 2), tb0 = 2 * (wave & 3);
    // one tile row of the step's MFMAs: k-step ks, A tile t against both B tiles (fragments read just before)
    auto tile_row = [&](const w3_u32x4* __restrict__ buf, int ks, int t) __attribute__((always_inline)) {
        const w3_u32x4* Ab = buf + ks * NPC * PW_TA * 64 + lane;
        const w3_u32x4* Bb = buf + A_ENT + ks * NPC * PW_TB * 64 + lane;
        if constexpr (BIO) {
            const w3_bf16x8 a = __builtin_bit_cast(w3_bf16x8, Ab[(ta0 + t) * 64]);
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(w3_bf16x8, Bb[(tb0 + u) * 64]), acc[t][u], 0, 0, 0);
            return;
        }
        const pw_f16x8 ah = __builtin_bit_cast(pw_f16x8, Ab[(ta0 + t) * 64]), al = __builtin_bit_cast(pw_f16x8, Ab[((NPC - 1) * PW_TA + ta0 + t) * 64]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {                                  // smallest terms first
            const pw_f16x8 bh = __builtin_bit_cast(pw_f16x8, Bb[(tb0 + u) * 64]), bl = __builtin_bit_cast(pw_f16x8, Bb[((NPC - 1) * PW_TB + tb0 + u) * 64]);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t][u], 0, 0, 0);
        }
    };
    auto fence = []() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

    // prologue: step 0 is built on its own (nothing to multiply yet); its chunks already refill the ring with step 1
    if (steps > 0) {
        const Desc d0 = descriptors(0), d1 = descriptors(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (is_a) load_a(d0, i);
        if (is_z) load_z(d0);
        __syncthreads();                                               // the zeroed buffers
        begin_a(0, d0);
#pragma unroll
        for (int k = 0; k < 4; ++k) pair_a(0, k, d0, d1, reinterpret_cast<uint32_t*>(pw_lds));
        columns_z(d1, pw_lds);
    }
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const w3_u32x4* cur = pw_lds + (s & 1) * BUFK;
        w3_u32x4* nxt = pw_lds + ((s + 1) & 1) * BUFK;
        uint32_t* nxt32 = reinterpret_cast<uint32_t*>(nxt);
        const bool more = s + 1 < steps;                               // (uniform) is there a step s + 1 to build
        const int p_old = p_cur, end_old = end_cur;
        while (more && m_begin + 32 * (s + 1) >= end_cur) {            // (scalar; predicates without rows are stepped over)
            ++p_cur;
            end_cur = __builtin_amdgcn_readfirstlane((int)pred_off[p_cur + 1]);
        }
        if constexpr (SUMS) {
            // the predicate ended inside step s: the threads whose octet of step s reached its end have flushed there, the others flush now
            if (p_cur != p_old && is_a && m_begin + 32 * s + 8 * k_oa() + 7 < end_old) flush_de(p_old);
        }
        const Desc d1 = descriptors(s + 1), d2 = descriptors(s + 2);
        if (more) begin_a(s + 1, d1);
        fence();
        tile_row(cur, 0, 0); tile_row(cur, 0, 1);
        fence();
        if (more) pair_a(s + 1, 0, d1, d2, nxt32);
        fence();
        tile_row(cur, 0, 2); tile_row(cur, 0, 3);
        fence();
        if (more) pair_a(s + 1, 1, d1, d2, nxt32);
        fence();
        tile_row(cur, 0, 4); tile_row(cur, 1, 0);
        fence();
        if (more) pair_a(s + 1, 2, d1, d2, nxt32);
        fence();
        tile_row(cur, 1, 1); tile_row(cur, 1, 2);
        fence();
        if (more) pair_a(s + 1, 3, d1, d2, nxt32);
        fence();
        tile_row(cur, 1, 3);
        fence();
        if (more) columns_z(d2, nxt);
        fence();
        tile_row(cur, 1, 4);
        fence();
        __syncthreads();                                               // step s + 1 is complete in LDS; everyone is past the MFMAs of step s
    }

    if constexpr (SUMS) {
        if (steps > 0 && is_a) {
            const bool crossed = m_begin + 32 * (steps - 1) + 8 * k_oa() + 7 >= end_cur;      // (this thread's sums already belong to the next predicate)
            if (!crossed) flush_de(p_cur);
            if (end_cur < m_end) {                                     // (uniform) the slab's last rows open another predicate: its partial row is written by everyone
                int p_nxt = p_cur + 1;
                while ((int)pred_off[p_nxt + 1] <= end_cur) ++p_nxt;   // (predicates without rows)
                flush_de(__builtin_amdgcn_readfirstlane(p_nxt));
            }
            float* o = db_part + ((int64_t)blockIdx.x * 4 + k_oa()) * H2 + k_col0();
#pragma unroll
            for (int t = 0; t < CPT; ++t) o[t] = sums[512 * (CPT + t)] * invS;
        }
    }
    // D tile: column j = lane & 31, row i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int64_t stride = (((int64_t)H2 * H1) + 3) & ~(int64_t)3;
    float* out = part + (int64_t)blockIdx.x * stride;
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = 32 * (tb0 + u) + (lane & 31);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = 32 * (ta0 + t) + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                if (n < H2 && k < H1) out[(int64_t)n * H1 + k] = acc[t][u][i] * invS;
            }
        }
}

static int pw_slabs(int64_t M) { return (int)std::max<int64_t>(1, std::min<int64_t>(256, (M + 255) / 256)); }

extern "C" int64_t dfol_pair_wgrad_fused_workspace(int64_t M, int32_t H2, int32_t H1) {
    if (M <= 0 || H2 <= 0 || H1 <= 0) return 0;
    return (int64_t)pw_slabs(M) * ((((int64_t)H2 * H1) + 3) & ~(int64_t)3);
}

// The partial sums of pair_wgrad_fused_kernel<.., true> in a fixed order.  Block p < P: dE[p][:] from the partial rows (slab + p, octet) of
// the slabs that hold rows of p, and dbe[p] = the sum of dx over p's rows; blocks P .. P + 7: db2 from every slab's four octet rows.
__global__ __launch_bounds__(320) void pair_sums_reduce_kernel(const float* __restrict__ de_part, const float* __restrict__ db_part,
                                                               const float* __restrict__ dx, const int64_t* __restrict__ pred_off, int P, int H2,
                                                               int slabs, int rows_per_slab, float* __restrict__ dE, int64_t ld_de,
                                                               float* __restrict__ dbe, float* __restrict__ db2) {
    __shared__ float red[320];
    const int p = blockIdx.x, tid = threadIdx.x;
    if (p >= P) {                                                      // db2: 8 blocks of 40 columns, 8 row phases each (one thread walking all 4 x slabs rows of a column took 0.25 ms)
        const int c = (p - P) * 40 + tid % 40, ph = tid / 40;
        float acc = 0.f;
        if (c < H2)
            for (int r = ph; r < slabs * 4; r += 8) acc += db_part[(int64_t)r * H2 + c];
        red[tid] = acc;
        __syncthreads();
        if (ph == 0 && c < H2) {
            float t = 0.f;
            for (int i = 0; i < 8; ++i) t += red[i * 40 + tid];
            db2[c] = t;
        }
        return;
    }
    const int64_t r0 = pred_off[p], r1 = pred_off[p + 1];
    float acc = 0.f;
    if (r1 > r0 && tid < H2) {
        const int s0 = (int)(r0 / rows_per_slab), s1 = (int)((r1 - 1) / rows_per_slab);
        for (int s = s0; s <= s1; ++s)
            for (int o = 0; o < 4; ++o) acc += de_part[(((int64_t)s + p) * 4 + o) * H2 + tid];
    }
    if (tid < H2) dE[(int64_t)p * ld_de + tid] = acc;
    if (dbe) {
        float g = 0.f;
        for (int64_t r = r0 + tid; r < r1; r += 320) g += dx[r];
        red[tid] = g;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int i = 0; i < 320; ++i) t += red[i];
            dbe[p] = t;
        }
    }
}

// floats of workspace for dfol_pair_wgrad_fused_sums_f32: the slabs' dW2 partials, then the dE partial rows ((slabs + P) x 4 x HID2)
// and the db2 partial rows (slabs x 4 x HID2)
extern "C" int64_t dfol_pair_wgrad_fused_sums_workspace(int64_t M, int32_t H2, int32_t H1, int32_t P) {
    if (M <= 0 || H2 <= 0 || H1 <= 0 || P < 0) return 0;
    const int64_t slabs = pw_slabs(M);
    return slabs * ((((int64_t)H2 * H1) + 3) & ~(int64_t)3) + (2 * slabs + P) * 4 * (int64_t)H2;
}

static int pw_launch(const void* pre2, int64_t ld_p2, const float* dx, const int32_t* row_pred, const int64_t* pred_off, int32_t P, const float* E,
                     int64_t ld_e, const float* scale, const void* Z, int64_t ld_z, int64_t M, int32_t H2, int32_t H1, float* workspace, float* dW,
                     float* dE, int64_t ld_de, float* dbe, float* db2, bool sums, void* stream, bool bio = false);

// scale: device pointer to {S, 1 / S}, S a power of two with S max_r(|dx[r]| max|E[row_pred[r]]|) / 4 <= 2^14 (see above)
// row_pred [M]: NON-DECREASING valid rows of E (the pair rows of a predicate are contiguous), pred_off [P + 1]: the first pair row of every
// predicate (row_pred[r] = p for pred_off[p] <= r < pred_off[p + 1], pred_off[P] = M); a row without a gradient carries dx = 0.
// (The bias gradient db2 - the column sums of dpre2 - comes from dfol_pair_logit_bwd_f32's db2 output: that kernel has the registers for it.)
extern "C" int dfol_pair_wgrad_fused_f32(const float* pre2, int64_t ld_p2, const float* dx, const int32_t* row_pred, const int64_t* pred_off,
                                         const float* E, int64_t ld_e, const float* scale, const float* Z, int64_t ld_z, int64_t M, int32_t H2,
                                         int32_t H1, float* workspace, float* dW, void* stream) {
    return pw_launch(pre2, ld_p2, dx, row_pred, pred_off, 0, E, ld_e, scale, Z, ld_z, M, H2, H1, workspace, dW, nullptr, 0, nullptr, nullptr, false, stream);
}

// ... and the sums of the logit layer's backward from the same pass (no separate pass over pre2): dE [P, ld_de], dbe [P] (or NULL),
// db2 [HID2].  EVERY predicate must own at least 64 pair rows or none (the caller checks; dfol_pair_logit_bwd_sums_f32 otherwise);
// workspace: dfol_pair_wgrad_fused_sums_workspace floats.
extern "C" int dfol_pair_wgrad_fused_sums_f32(const float* pre2, int64_t ld_p2, const float* dx, const int32_t* row_pred, const int64_t* pred_off,
                                              int32_t P, const float* E, int64_t ld_e, const float* scale, const float* Z, int64_t ld_z, int64_t M,
                                              int32_t H2, int32_t H1, float* workspace, float* dW, float* dE, int64_t ld_de, float* dbe, float* db2,
                                              void* stream) {
    DFOL_REQUIRE(P > 0 && dE && db2 && ld_de >= H2 && H2 <= 320, "pair_wgrad_fused_sums: bad arguments P=%d", P);
    return pw_launch(pre2, ld_p2, dx, row_pred, pred_off, P, E, ld_e, scale, Z, ld_z, M, H2, H1, workspace, dW, dE, ld_de, dbe, db2, true, stream);
}

// ... over bf16-STORED pre2 and Z (the bf16 mode; strides in elements, multiples of 4, 8-byte aligned rows): one bf16 piece per operand,
// dpre2 rounded as dfol_pair_logit_bwd_bf16 stores it; db2 = the column sums of those rounded values.  HID2 % 4 == 0 is enough here.
extern "C" int dfol_pair_wgrad_fused_sums_bf16(const void* pre2_bf16, int64_t ld_p2, const float* dx, const int32_t* row_pred, const int64_t* pred_off,
                                               int32_t P, const float* E, int64_t ld_e, const void* Z_bf16, int64_t ld_z, int64_t M, int32_t H2,
                                               int32_t H1, float* workspace, float* dW, float* dE, int64_t ld_de, float* dbe, float* db2, void* stream) {
    DFOL_REQUIRE(P > 0 && dE && db2 && ld_de >= H2 && H2 <= 320, "pair_wgrad_fused_sums_bf16: bad arguments P=%d", P);
    return pw_launch(pre2_bf16, ld_p2, dx, row_pred, pred_off, P, E, ld_e, nullptr, Z_bf16, ld_z, M, H2, H1, workspace, dW, dE, ld_de, dbe, db2, true, stream, true);
}

static int pw_launch(const void* pre2, int64_t ld_p2, const float* dx, const int32_t* row_pred, const int64_t* pred_off, int32_t P, const float* E,
                     int64_t ld_e, const float* scale, const void* Z, int64_t ld_z, int64_t M, int32_t H2, int32_t H1, float* workspace, float* dW,
                     float* dE, int64_t ld_de, float* dbe, float* db2, bool sums, void* stream, bool bio) {
    DFOL_REQUIRE(M > 0 && M < (1ll << 31) - 64 && H2 >= 4 && H1 >= 4 && H2 % 4 == 0 && H1 % 4 == 0 && H2 <= 32 * PW_TA && H1 <= 32 * PW_TB,
                 "pair_wgrad_fused: bad sizes M=%lld H2=%d H1=%d (multiples of 4, H2 <= %d, H1 <= %d)", (long long)M, H2, H1, 32 * PW_TA, 32 * PW_TB);
    DFOL_REQUIRE(pre2 && dx && row_pred && pred_off && E && (scale || bio) && Z && workspace && dW, "pair_wgrad_fused: null pointer");
    DFOL_REQUIRE(ld_p2 % 4 == 0 && ld_z % 4 == 0 && ld_e % 4 == 0 && ((uintptr_t)pre2 % (bio ? 8 : 16) == 0) && ((uintptr_t)Z % (bio ? 8 : 16) == 0) &&
                 ((uintptr_t)E % 16 == 0), "pair_wgrad_fused: rows of pre2, Z and E must be 16-byte (bf16 storage: 8-byte) aligned");
    DFOL_REQUIRE(8 * std::max(ld_p2, ld_z) * 4 * 4 < (1ll << 31), "pair_wgrad_fused: row stride too large (%lld)", (long long)std::max(ld_p2, ld_z));
    const int slabs = pw_slabs(M);
    const int rows_per_slab = (dfol_cdiv(M, slabs) + 31) & ~31;
    size_t lds = (size_t)2 * PW_BUF * 16 + (2 + 6) * 512 * 4;                // (fp32 storage with CPT = 4 and the sums - 2 + 8 words per thread - does not fit)
    const int64_t elems = (int64_t)H2 * H1;
    float* de_part = workspace + (int64_t)slabs * ((elems + 3) & ~(int64_t)3);
    float* db_part = de_part + ((int64_t)slabs + P) * 4 * H2;
    hipStream_t st = (hipStream_t)stream;
#define DFOL_PW(C, S)                                                                                                                           \
    {                                                                                                                                          \
        static const hipError_t ok = hipFuncSetAttribute((const void*)pair_wgrad_fused_kernel<C, S>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                        (int)((size_t)2 * PW_BUF * 16 + (2 + 6) * 512 * 4));                                     \
        DFOL_REQUIRE(ok == hipSuccess, "pair_wgrad_fused: cannot reserve %zu bytes of LDS (%s)", lds, hipGetErrorString(ok));                      \
        hipLaunchKernelGGL((pair_wgrad_fused_kernel<C, S>), dim3(slabs), dim3(512), lds, st, pre2, ld_p2, dx, row_pred, pred_off, E, ld_e, scale, Z, \
                           ld_z, (int)M, H2, H1, rows_per_slab, workspace, de_part, db_part);                                                   \
    }
    if (bio) {
        lds = (size_t)PW_BUF * 16 + (2 + 8) * 512 * 4;                  // (one piece per operand: the two buffers are half the size)
        static const hipError_t ok = hipFuncSetAttribute((const void*)pair_wgrad_fused_kernel<4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)((size_t)PW_BUF * 16 + (2 + 8) * 512 * 4));
        DFOL_REQUIRE(ok == hipSuccess && sums, "pair_wgrad_fused (bf16 storage): cannot reserve %zu bytes of LDS (%s)", lds, hipGetErrorString(ok));
        hipLaunchKernelGGL((pair_wgrad_fused_kernel<4, true, true>), dim3(slabs), dim3(512), lds, st, pre2, ld_p2, dx, row_pred, pred_off, E, ld_e, scale, Z,
                           ld_z, (int)M, H2, H1, rows_per_slab, workspace, de_part, db_part);
    } else if (H2 % 3 == 0) {                                          // (4 (H2 / 3) <= 427 threads build dpre2)
        if (sums) DFOL_PW(3, true) else DFOL_PW(3, false)
    } else {
        DFOL_REQUIRE(!sums, "pair_wgrad_fused_sums: HID2=%d must be a multiple of 3 (the running sums of four columns per thread do not fit the LDS)", H2);
        DFOL_PW(4, false)
    }
#undef DFOL_PW
    DFOL_LAUNCH_CHECK("pair_wgrad_fused");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dfol_cdiv(dfol_cdiv(elems, 4), 16)), dim3(256), 0, st, workspace, slabs, elems, dW);
    DFOL_LAUNCH_CHECK("pair_wgrad_fused (reduce)");
    if (sums) {
        hipLaunchKernelGGL(pair_sums_reduce_kernel, dim3(P + 8), dim3(320), 0, st, (const float*)de_part, (const float*)db_part, dx, pred_off, P, H2, slabs,
                           rows_per_slab, dE, ld_de, dbe, db2);
        DFOL_LAUNCH_CHECK("pair_wgrad_fused (sums reduce)");
    }
    return 0;
}
