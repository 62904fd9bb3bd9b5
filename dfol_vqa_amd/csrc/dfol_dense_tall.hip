// The two TALL products of a train step of the pair MLP (trainer.py:429-442 through gqa_interpreter_experiments.py:26-33): millions of rows
// (one per ordered object pair) against a small weight matrix - pre2 = Z W2^T + b2 ([2.5 M, 256] x [256, 300]) and dZ = dpre2 W2
// ([2.5 M, 300] x [300, 256]) - as ONE persistent workgroup per CU that walks row blocks (gfx950).
//
// csrc/dfol_dense_split.hip runs these shapes as 128 x 128 output tiles of 8 - 10 k-steps each, two workgroups per CU: every tile pays a
// cold start and a store phase, X is split once per column block (three times for 300 columns), and the products moved their 5.6 GB at
// 2.7 - 2.9 TB/s (1.96 / 2.13 ms at 256 x 100 objects) where the step's plain streaming kernels reach 5.4 - 6 TB/s.  Here a workgroup of
// eight wavefronts owns 128 rows x ALL columns (N <= 320): the X rows are read and split once, the weight tiles of all column blocks are
// staged per k-step, wavefront (wm, wn) multiplies rows 64 wm .. by columns 16 NTW wn .. (4 x NTW accumulator tiles of 16 x 16), and
// the step stream runs on across row blocks - the X rows of the next block's first steps are in flight (four steps of registers) under
// the last steps and the stores of the current one.  Two LDS buffers, one barrier per step:  multiply step g from one buffer, then split
// and store step g + 1 into the other.
//
// Same arithmetic as dfol_linear_act_h2_f32 (two fp16 pieces per operand, three products, the packed image of dfol_linear_pack_w_f16x2,
// the same order of the products): results are bit for bit those of csrc/dfol_dense_split.hip.
//   MODE 0: Y = X W^T + b
//   MODE 1: the A operand is produced (dpre2 from pre2, dx and the rows' embedding rows, see dfol_pair_dz_fused_f32); the per-row scaling
//           comes precomputed (tall_row_scale_kernel: gs[r] = 2^e_r dx[r], rsinv[r] = 2^-e_r)
//   MODE 2: MODE 0 plus the logit layer's partial sums from the epilogue (see dfol_linear_logit_h2_f32), one slot per column group (4)
#include "dfol_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TL_XS = 1408;                                           // floats of per-column / per-row epilogue inputs in LDS (see `xs`)
constexpr int TL_BM = 128, TL_BK = 32, TL_XD = 4;                     // rows of a block, k of a step, steps of X rows in flight

__device__ __forceinline__ int tl_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }
__device__ __forceinline__ void tl_split2h(float x0, float x1, uint32_t& h, uint32_t& l) {
    const f32x2 x = {x0, x1};
    const f16x2 hh = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(hh, f32x2);
    h = __builtin_bit_cast(uint32_t, hh);
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}
__device__ __forceinline__ void tl_split8h(const float4& a, const float4& b, u32x4& h, u32x4& l) {
    uint32_t hh[4], ll[4];
    tl_split2h(a.x, a.y, hh[0], ll[0]);
    tl_split2h(a.z, a.w, hh[1], ll[1]);
    tl_split2h(b.x, b.y, hh[2], ll[2]);
    tl_split2h(b.z, b.w, hh[3], ll[3]);
    h = u32x4{hh[0], hh[1], hh[2], hh[3]};
    l = u32x4{ll[0], ll[1], ll[2], ll[3]};
}
__device__ __forceinline__ float tl_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x)); }
__device__ __forceinline__ float tl_dsigmoid(float x) {
    const float h = tl_sigmoid(x);
    return h * (1.0f - h);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tl_bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t tl_rne2(float x0, float x1) {     // two fp32 -> two bf16, round to nearest even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x0, x1}, tl_bf16x2));
}
__device__ __forceinline__ float4 tl_widen(const u32x2& v) {
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}

struct TlExtra {
    const float* gs;                // MODE 1: 2^e_r dx[r] [M]
    const float* rsinv;             // MODE 1: 2^-e_r [M]
    const int32_t* row_pred;        // MODE 1, 2: [M] the row of E a pair row reads (MODE 1: -1 = no gradient; MODE 2: non-decreasing)
    const float* E;                 // [P, ld_e]
    int64_t ld_e;
    int accumulate;                 // MODE 1: Y += instead of Y =
    float* x_part;                  // MODE 2: [4][ld_xp]
    int64_t ld_xp;
    // MODE 1, MULTI (several readers of one trunk in ONE pass: dpre2[r][j] = h (1 - h) sum_k dx_k[r] E_k[row_pred[r]][j]): reader k's scaled dx at
    // gs + k gs_stride, its embedding rows at E + k e_stride; the rows share one scale (the bound of the sum) and one row -> embedding row map
    int nr;
    int64_t gs_stride, e_stride;
};
constexpr int TL_MAXR = 4;          // readers per MULTI launch

// MODE 1's per-row scaling: e_r puts the bound |dx[r]| max|E[p]| / 4 into [2^13, 2^14)  (the rule of csrc/dfol_dense_split.hip, LsProducer)
// bound_max (or null): the largest bound of the launch, as the bits of a non-negative float (their order is the integers' order, and a
// maximum does not depend on the order of its updates: the atomic keeps the run repeatable); zeroed by the launcher
__global__ __launch_bounds__(256) void tall_row_scale_kernel(const float* __restrict__ dx, const int32_t* __restrict__ row_pred,
                                                             const float* __restrict__ emax, int M, float* __restrict__ gs,
                                                             float* __restrict__ rsinv, uint32_t* __restrict__ bound_max) {
    __shared__ float wmax[4];
    float m = 0.f;
    for (int r = blockIdx.x * 256 + (int)threadIdx.x; r < M; r += gridDim.x * 256) {      // (a fixed grid: one atomic per workgroup, ~1000 per launch)
        const int p = row_pred[r];
        const float gg = p >= 0 ? dx[r] : 0.f;
        const float bound = fabsf(gg) * emax[max(p, 0)] * 0.25f;
        int e = 0;
        if (bound > 0.f && bound < 3.0e38f) {
            int x;
            (void)frexpf(bound, &x);
            e = 14 - x;
            e = e < -100 ? -100 : (e > 100 ? 100 : e);
            m = fmaxf(m, bound);                                      // (a non-finite bound leaves the scale alone; the products show the NaN)
        }
        gs[r] = ldexpf(gg, e);
        rsinv[r] = ldexpf(1.0f, -e);
    }
    if (!bound_max) return;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (m > 0.f) atomicMax(bound_max, __float_as_uint(m));
    }
}

// NTW: 16-column tiles per wavefront (4 wavefronts across the columns: N <= 64 NTW); NBN = ceil(64 NTW / 128) column blocks of the packed image
// BIO (the bf16 mode, round 4 late): X and Y are rows of bfloat16 (8-byte aligned rows, K % 4 == 0, N % 4 == 0), operands ONE bf16 piece
// (the image of dfol_linear_pack_w_bf16), one product on v_mfma_f32_16x16x32_bf16, fp32 accumulation, results rounded to nearest even -
// bit for bit dfol_linear_act_bf16_bf16.  MODE 1 then needs no row scaling (bf16 has fp32's exponent range): gs = dx, and dpre2 is
// rounded to bf16 exactly as dfol_pair_logit_bwd_bf16 stores it.  The output block leaves through LDS (2-byte stores straight from the
// accumulator layout cost a third of the tiled kernel): a staging array behind the buffers, one extra barrier per block.
template <int NTW, int MODE, bool BIO = false, bool MULTI = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void tall_h2_kernel(
    const void* __restrict__ Xv, int64_t ldx, const u32x4* __restrict__ Wp, const float* __restrict__ bias, void* __restrict__ Yv, int64_t ldy,
    int M, int N, int K, int ksteps, int nbn, TlExtra ex) {
    constexpr bool PROD = MODE == 1, LOGIT = MODE == 2;
    static_assert(!MULTI || (PROD && !BIO), "several readers: the fp32 dZ product only");
    typedef typename std::conditional<BIO, uint16_t, float>::type TX;
    const TX* __restrict__ X = reinterpret_cast<const TX*>(Xv);
    TX* __restrict__ Y = reinterpret_cast<TX*>(Yv);
    constexpr int NBN = (64 * NTW + 127) / 128;
    constexpr int NP = BIO ? 1 : 2;                                   // pieces per operand
    constexpr int TILE = NP * 128 * 4;                                // 16-byte pieces of one [pieces][128 rows][4 k-groups] tile
    constexpr int BUF = TILE + NBN * TILE;                            // one LDS buffer: the A tile, then the B tiles of all column blocks
    extern __shared__ __attribute__((aligned(16))) u32x4 tl_sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kh = lane >> 4, r16 = lane & 15;
    const int wm = wave >> 2, wn = wave & 3;
    const int arow = tid >> 2, aq = tid & 3;                          // staging: one row of the block, 8 consecutive k per step
    const int nblocks = (M + TL_BM - 1) / TL_BM, stride = gridDim.x;
    const int first = blockIdx.x;
    if (first >= nblocks) return;
    const int nmine = (nblocks - first + stride - 1) / stride;
    const int T = nmine * ksteps;                                     // steps of this workgroup's stream

    floatx4 acc[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // ---- X rows: a ring of TL_XD register sets, the loader's position runs ahead of the multiplier's across block boundaries
    typedef typename std::conditional<BIO, u32x2, float4>::type XR;   // four consecutive k of a row as loaded
    XR xa[TL_XD][2];
    int lb = first, lks = 0;                                          // block and step the next load belongs to
    const TX* xrow = X + (int64_t)min(lb * TL_BM + arow, M - 1) * ldx + aq * 8;
    auto load_x = [&](auto set_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        const int k = lks * TL_BK + aq * 8;
        const int c0 = min(k, K - 4) - aq * 8, c1 = min(k + 4, K - 4) - aq * 8;          // (K % 4 == 0; clamped, zeroed when used)
        xa[S][0] = *reinterpret_cast<const XR*>(xrow + c0);
        xa[S][1] = *reinterpret_cast<const XR*>(xrow + c1);
        if (++lks == ksteps) {                                        // (uniform) on to the next block of this workgroup; past the last: clamped rows, never used
            lks = 0;
            lb += stride;
            xrow = X + (int64_t)min((int64_t)lb * TL_BM + arow, (int64_t)M - 1) * ldx + aq * 8;
        }
    };
    // ---- weight tiles of a step: all column blocks, one step ahead in registers
    u32x4 wb[NP * NBN];
    int wks = 0;
    auto load_w = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NP * NBN; ++i) {
            const int idx = i * 512 + tid, nb = idx / TILE, within = idx % TILE;
            wb[i] = Wp[((int64_t)min(nb, nbn - 1) * ksteps + wks) * TILE + within];
        }
        if (++wks == ksteps) wks = 0;
    };
    auto store_b = [&](int off) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NP * NBN; ++i) tl_sm[off + TILE + i * 512 + tid] = wb[i];
    };
    // ---- MODE 1: the producer's state for the block whose steps are being stored; the next block's row values are prefetched a block ahead
    float gs_cur = 0.f, gs_next = 0.f;
    float gsm_cur[MULTI ? TL_MAXR : 1], gsm_next[MULTI ? TL_MAXR : 1];           // MULTI: the readers' scaled dx of the row
    int p_next = -1;
    const float* ep = ex.E;
    float4 ea[2];
    int sb = first, sks = 0;                                          // block and step the next store belongs to
    auto fetch_row_state = [&](int b) __attribute__((always_inline)) {
        if constexpr (PROD) {
            const int64_t r = min((int64_t)b * TL_BM + arow, (int64_t)M - 1);
            const bool live = (int64_t)b * TL_BM + arow < M;
            p_next = live ? ex.row_pred[r] : -1;
            if constexpr (MULTI) {
#pragma unroll
                for (int q = 0; q < TL_MAXR; ++q) gsm_next[q] = (live && q < ex.nr) ? ex.gs[(int64_t)q * ex.gs_stride + r] : 0.f;
            } else {
                gs_next = live ? ex.gs[r] : 0.f;
            }
        }
    };
    auto switch_row_state = [&]() __attribute__((always_inline)) {   // at the first step of a block
        if constexpr (PROD) {
            if constexpr (MULTI) {
#pragma unroll
                for (int q = 0; q < TL_MAXR; ++q) gsm_cur[q] = p_next >= 0 ? gsm_next[q] : 0.f;
            } else {
                gs_cur = p_next >= 0 ? gs_next : 0.f;
            }
            ep = ex.E + (int64_t)max(p_next, 0) * ex.ld_e + aq * 8;
            fetch_row_state(sb + stride);
        }
    };
    auto load_e = [&]() __attribute__((always_inline)) {              // the embedding pieces of the step about to be stored
        if constexpr (PROD) {
            const int k = sks * TL_BK + aq * 8;
            const int c0 = min(k, K - 4) - aq * 8, c1 = min(k + 4, K - 4) - aq * 8;
            if constexpr (MULTI) {                                    // the row's combined coefficients sum_k (2^e dx_k) E_k: the factor of h (1 - h)
                float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
#pragma unroll
                for (int q = 0; q < TL_MAXR; ++q) {
                    if (q < ex.nr) {
                        const float g = gsm_cur[q];
                        const float4 e0 = *reinterpret_cast<const float4*>(ep + (int64_t)q * ex.e_stride + c0);
                        const float4 e1 = *reinterpret_cast<const float4*>(ep + (int64_t)q * ex.e_stride + c1);
                        s0.x = fmaf(g, e0.x, s0.x), s0.y = fmaf(g, e0.y, s0.y), s0.z = fmaf(g, e0.z, s0.z), s0.w = fmaf(g, e0.w, s0.w);
                        s1.x = fmaf(g, e1.x, s1.x), s1.y = fmaf(g, e1.y, s1.y), s1.z = fmaf(g, e1.z, s1.z), s1.w = fmaf(g, e1.w, s1.w);
                    }
                }
                ea[0] = s0, ea[1] = s1;
            } else {
                ea[0] = *reinterpret_cast<const float4*>(ep + c0);
                ea[1] = *reinterpret_cast<const float4*>(ep + c1);
            }
        }
    };
    auto store_a = [&](auto set_tag, int off) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        const int k = sks * TL_BK + aq * 8;
        const int at = off + arow * 4 + (aq ^ tl_swz(arow));
        if constexpr (BIO) {
            const u32x2 zz = u32x2{0u, 0u};
            u32x2 v0 = k < K ? xa[S][0] : zz, v1 = k + 4 < K ? xa[S][1] : zz;
            if constexpr (PROD) {                                     // ((dx E) h) (1 - h), rounded to nearest even: the values dfol_pair_logit_bwd_bf16 stores
                auto dp = [&](const u32x2& xb, const float4& e) __attribute__((always_inline)) {
                    const float4 x = tl_widen(xb);
                    const float h0 = tl_sigmoid(x.x), h1 = tl_sigmoid(x.y), h2 = tl_sigmoid(x.z), h3 = tl_sigmoid(x.w);
                    return u32x2{tl_rne2(gs_cur * e.x * h0 * (1.0f - h0), gs_cur * e.y * h1 * (1.0f - h1)),
                                 tl_rne2(gs_cur * e.z * h2 * (1.0f - h2), gs_cur * e.w * h3 * (1.0f - h3))};
                };
                v0 = k < K ? dp(xa[S][0], ea[0]) : zz;
                v1 = k + 4 < K ? dp(xa[S][1], ea[1]) : zz;
            }
            tl_sm[at] = u32x4{v0.x, v0.y, v1.x, v1.y};
        } else {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 v0 = k < K ? xa[S][0] : z, v1 = k + 4 < K ? xa[S][1] : z;
            if constexpr (PROD) {
                auto dp = [&](const float4& x, const float4& e) __attribute__((always_inline)) {
                    if constexpr (MULTI)                              // (the readers' dx are inside e already: load_e)
                        return make_float4(e.x * tl_dsigmoid(x.x), e.y * tl_dsigmoid(x.y), e.z * tl_dsigmoid(x.z), e.w * tl_dsigmoid(x.w));
                    else
                        return make_float4((gs_cur * e.x) * tl_dsigmoid(x.x), (gs_cur * e.y) * tl_dsigmoid(x.y), (gs_cur * e.z) * tl_dsigmoid(x.z),
                                           (gs_cur * e.w) * tl_dsigmoid(x.w));
                };
                v0 = k < K ? dp(xa[S][0], ea[0]) : z;
                v1 = k + 4 < K ? dp(xa[S][1], ea[1]) : z;
            }
            u32x4 ph, pl;
            tl_split8h(v0, v1, ph, pl);
            tl_sm[at] = ph;
            tl_sm[at + TL_BM * 4] = pl;
        }
        if (++sks == ksteps) {
            sks = 0;
            sb += stride;
        }
    };

    // ---- the step's MFMAs
    const int aoff = (wm * 64 + r16) * 4 + (kh ^ tl_swz(r16));
    int boff[NTW];                                                    // fragment of column tile j: block (col >> 7), row col & 127 of its tile
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int col0 = wn * (16 * NTW) + j * 16;
        boff[j] = TILE + (col0 >> 7) * TILE + ((col0 & 127) + r16) * 4 + (kh ^ tl_swz(r16));
    }
    constexpr int PA3[3] = {1, 0, 0}, PB3[3] = {0, 1, 0};             // xl wh, xh wl, xh wh (smallest first)
    auto multiply = [&](int off) __attribute__((always_inline)) {
#pragma unroll
        for (int ih = 0; ih < 4; ih += 2) {                           // two row tiles at a time (register budget)
            u32x4 a[2][NP];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) a[i][p] = tl_sm[off + p * TL_BM * 4 + (ih + i) * 64 + aoff];
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                u32x4 b[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) b[p] = tl_sm[off + p * 128 * 4 + boff[j]];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if constexpr (BIO) {
                        acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i][0]), __builtin_bit_cast(bf16x8, b[0]), acc[ih + i][j], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int x = 0; x < 3; ++x)
                            acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i][PA3[x] % NP]), __builtin_bit_cast(f16x8, b[PB3[x] % NP]), acc[ih + i][j], 0, 0, 0);
                    }
                }
            }
        }
    };

    // ---- what the epilogue needs beside the accumulators lives in LDS (4 KB behind the two buffers), put there steps ahead: the weight rows'
    // scales and the bias (once), and per block the rows' un-scaling factors (MODE 1) resp. the block's embedding row (MODE 2: a block nearly
    // always lies inside one predicate - row_pred is non-decreasing and a predicate owns n (n - 1) rows; a block across a boundary looks its
    // rows up one by one).  Loaded in the epilogue itself these were dependent round trips with the whole CU waiting: the first version
    // of the logit epilogue cost 1.2 ms over 77 blocks per workgroup.
    float* xs = reinterpret_cast<float*>(tl_sm + 2 * BUF);            // [0, 320) scales, [320, 640) bias, [640, 960) embedding row, [960, 1088) row factors,
    constexpr int PITCH = 64 * NTW + 8;                               // [1088, 1408) the embedding row of the block's LAST row; BIO: halfwords per staged output row
    uint16_t* stage = reinterpret_cast<uint16_t*>(xs + TL_XS);
    if (tid < 320) {
        const float* tail = reinterpret_cast<const float*>(Wp + (int64_t)nbn * ksteps * TILE);
        xs[tid] = BIO ? 1.0f : (tid < nbn * 128 ? tail[tid] : 0.f);     // (the one-piece image has no row scales)
        xs[320 + tid] = (bias && tid < N) ? bias[tid] : 0.f;
    }
    int lg_p0 = -1, lg_pl = -2;                                       // (scalar registers) predicates of the block's first and last row
    bool lg_uni = false;
    float stage_reg = 0.f, stage_reg2 = 0.f;
    int p0_reg = -1, pl_reg = -2;                                     // (requested at step 0, read at step 1: no wait at the request)
    auto epilogue_stage = [&](int b, int step) __attribute__((always_inline)) {      // called at steps 0, 1, 2 of block b (ksteps >= 4)
        const int m0 = b * TL_BM;
        if (step == 0) {
            if constexpr (LOGIT) {
                p0_reg = ex.row_pred[m0];
                pl_reg = ex.row_pred[min(m0 + TL_BM, M) - 1];
            }
            if constexpr (PROD && !BIO) {
                if (tid < TL_BM) stage_reg = ex.rsinv[min(m0 + tid, M - 1)];
            }
        } else if (step == 1) {
            if constexpr (LOGIT) {
                lg_p0 = __builtin_amdgcn_readfirstlane(p0_reg);
                lg_pl = __builtin_amdgcn_readfirstlane(pl_reg);
                lg_uni = lg_p0 >= 0 && lg_p0 == lg_pl;
                // (a block across ONE predicate boundary - every tenth block at 36 objects per image - reads the two rows it can meet)
                if (tid < 320) {
                    stage_reg = (lg_p0 >= 0 && tid < N) ? ex.E[(int64_t)lg_p0 * ex.ld_e + tid] : 0.f;
                    stage_reg2 = (!lg_uni && lg_pl >= 0 && tid < N) ? ex.E[(int64_t)lg_pl * ex.ld_e + tid] : 0.f;
                }
            }
            if constexpr (PROD && !BIO) {
                if (tid < TL_BM) xs[960 + tid] = stage_reg;
            }
        } else {
            if constexpr (LOGIT) {
                if (tid < 320) xs[640 + tid] = stage_reg, xs[1088 + tid] = stage_reg2;
            }
        }
    };
    auto epilogue = [&](int b) __attribute__((always_inline)) {       // block b's accumulators -> Y (and the logit partial sums), then cleared
        const int m0 = b * TL_BM;
        float cs[NTW], bv[NTW], le[NTW], le1[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int col = wn * (16 * NTW) + j * 16 + r16;
            cs[j] = xs[col], bv[j] = xs[320 + col];
            le[j] = LOGIT ? xs[640 + col] : 0.f;
            le1[j] = LOGIT ? xs[1088 + col] : 0.f;
        }
        auto rows = [&](auto uni_tag) __attribute__((always_inline)) {
            constexpr bool UNI = decltype(uni_tag)::value;            // LOGIT: the whole block reads one embedding row (le[])
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int lrow = wm * 64 + i * 16 + 4 * kh + e, row = m0 + lrow;
                    const bool row_ok = row < M;
                    const int64_t rc = min(row, M - 1);
                    float rsc = 1.0f;
                    if constexpr (PROD && !BIO) rsc = xs[960 + lrow];
                    int pr = -1;
                    if constexpr (LOGIT && !UNI) pr = row_ok ? ex.row_pred[rc] : -1;
                    // (not UNI: the row's embedding row is the block's first row's, its last row's, or - a predicate of fewer than 128
                    // rows inside the block - one that is read from memory element by element)
                    const bool at0 = pr == lg_p0, at1 = pr == lg_pl, far = pr >= 0 && !at0 && !at1;
                    float sum = 0.f;
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        const int col = wn * (16 * NTW) + j * 16 + r16;
                        float v = acc[i][j][e] * cs[j];
                        if constexpr (PROD) {
                            v *= rsc;
                            if (ex.accumulate) {
                                if constexpr (BIO) v += __uint_as_float((uint32_t)Y[rc * ldy + min(col, N - 1)] << 16);
                                else v += Y[rc * ldy + min(col, N - 1)];
                            }
                        } else {
                            v += bv[j];
                        }
                        if constexpr (BIO) {                          // rounded to nearest even, staged for the 8-byte stores below; what follows sees the stored value
                            const uint32_t hv = tl_rne2(v, 0.f) & 0xffffu;
                            stage[lrow * PITCH + col] = (uint16_t)hv;
                            v = __uint_as_float(hv << 16);
                        } else {
                            if (row_ok && col < N) Y[(int64_t)row * ldy + col] = v;
                        }
                        if constexpr (LOGIT) {
                            if constexpr (UNI) sum += tl_sigmoid(v) * le[j];          // (le = 0 in the columns past N)
                            else if (far) {
                                if (col < N) sum += tl_sigmoid(v) * ex.E[(int64_t)pr * ex.ld_e + col];
                            } else if (pr >= 0) sum += tl_sigmoid(v) * (at0 ? le[j] : le1[j]);
                        }
                        acc[i][j][e] = 0.f;
                    }
                    if constexpr (LOGIT) {
#pragma unroll
                        for (int sh = 1; sh < 16; sh <<= 1) sum += __shfl_xor(sum, sh, 64);
                        if (r16 == 0 && row_ok) ex.x_part[(int64_t)wn * ex.ld_xp + row] = sum;
                    }
                }
        };
        if (LOGIT && lg_uni) rows(std::true_type());                  // (uniform)
        else rows(std::false_type());
        if constexpr (BIO) {
            __syncthreads();                                          // the block's rows are staged
            const int q4 = N >> 2;                                    // 8-byte pieces of a row (N % 4 == 0)
            for (int c = tid; c < TL_BM * q4; c += 512) {
                const int row = c / q4, q = c - row * q4;
                if (m0 + row < M)
                    *reinterpret_cast<u32x2*>(Y + (int64_t)(m0 + row) * ldy + 4 * q) = *reinterpret_cast<const u32x2*>(stage + row * PITCH + 4 * q);
            }
        }
    };

    // ---- the stream
    const std::integral_constant<int, 0> S0;
    const std::integral_constant<int, 1> S1;
    const std::integral_constant<int, 2> S2;
    const std::integral_constant<int, 3> S3;
    fetch_row_state(first);
    load_x(S0);
    load_w();
    load_x(S1);
    load_x(S2);
    load_x(S3);
    switch_row_state();
    load_e();
    store_b(0);
    load_w();
    store_a(S0, 0);
    load_x(S0);
    __syncthreads();
    int g = 0, cks = 0, cb = first;                                   // the step being multiplied: index in the stream, step and block
    auto body = [&](auto next_tag) __attribute__((always_inline)) {  // next_tag: the X set of step g + 1
        const int cur = (g & 1) * BUF, nxt = BUF - cur;
        if constexpr (PROD) {
            if (sks == 0) switch_row_state();                         // (uniform) step g + 1 opens a block
        }
        if (cks < 3) epilogue_stage(cb, cks);                         // (uniform)
        load_e();
        store_b(nxt);                                                 // (the weight tiles of step g + 1, requested a step ago)
        load_w();
        __builtin_amdgcn_sched_barrier(0);
        multiply(cur);                                                // (the MFMA phase at a raised issue priority - s_setprio 1 or 3 around it - measured: no change, 1.87 - 1.89 ms either way)
        __builtin_amdgcn_sched_barrier(0);
        // (the SIMD's two wavefronts taking these two halves in opposite order - a branch on wm around both orders - was measured at 5.4 ms
        // against 1.4: the doubled body no longer fits the instruction cache / the register budget; one scheduling region for both halves with
        // a sched_group_barrier pattern of one MFMA, 2 - 4 vector instructions, one LDS read: no change, 1.85 -> 1.88 ms)
        store_a(next_tag, nxt);                                       // under the tail of the MFMAs; frees the register set ...
        load_x(next_tag);                                             // ... for the rows four steps on
        __syncthreads();                                              // buffer nxt complete, buffer cur fully read
        ++g;
        if (++cks == ksteps) {                                        // (uniform) the block is done
            epilogue(cb);
            cks = 0;
            cb += stride;
        }
    };
    // (Code size: with the epilogue inlined behind each of the four bodies the forward-with-logits kernel is 109 KB, more than the 64 KB
    // instruction cache.  Two ways to have it once were measured and lost: a scalar switch around the four bodies, one step per trip - the
    // allocator then spills 140 - 580 bytes per lane; the epilogue behind the fourth body only, for step counts that are multiples of
    // four - 96 bytes of spills, 2.39 ms against 1.9.)
    while (g < T) {
        body(S1);
        if (g < T) body(S2);
        if (g < T) body(S3);
        if (g < T) body(S0);
    }
}

template <int MODE, bool BIO = false, bool MULTI = false>
static int tall_launch(const void* X, int64_t ldx, const void* W_split, const float* bias, void* Y, int64_t ldy, int32_t M, int32_t N, int32_t K,
                       const TlExtra& ex, void* stream, const char* name) {
    const int ksteps = dfol_cdiv(K, TL_BK), nbn = dfol_cdiv(N, 128);
    const int ntw = dfol_cdiv(N, 64);                                 // 16-column tiles per wavefront: 4 (N <= 256) or 5 (N <= 320)
    DFOL_REQUIRE(ntw <= 5, "%s: N=%d must be <= 320", name, N);
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    const int nblocks = dfol_cdiv(M, TL_BM);
    const int grid = std::min(nblocks, cus);
    const int nbn_lds = ntw <= 4 ? 2 : 3;
    constexpr int TILE = (BIO ? 1 : 2) * 128 * 4;
    const size_t lds = (size_t)2 * (TILE + nbn_lds * TILE) * 16 + TL_XS * 4 + (BIO ? (size_t)TL_BM * (64 * (ntw <= 4 ? 4 : 5) + 8) * 2 : 0);
    hipStream_t st = (hipStream_t)stream;
#define DFOL_TALL(NT)                                                                                                                       \
    {                                                                                                                                      \
        static const hipError_t ok = hipFuncSetAttribute((const void*)tall_h2_kernel<NT, MODE, BIO, MULTI>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                        (int)((size_t)2 * (TILE + ((64 * NT + 127) / 128) * TILE) * 16 + TL_XS * 4 +             \
                                                              (BIO ? (size_t)TL_BM * (64 * NT + 8) * 2 : 0)));                                    \
        DFOL_REQUIRE(ok == hipSuccess, "%s: cannot reserve %zu bytes of LDS (%s)", name, lds, hipGetErrorString(ok));                       \
        hipLaunchKernelGGL((tall_h2_kernel<NT, MODE, BIO, MULTI>), dim3(grid), dim3(512), lds, st, X, ldx, (const u32x4*)W_split, bias, Y, ldy, M, N, K, \
                           ksteps, nbn, ex);                                                                                               \
    }
    if (ntw <= 4) DFOL_TALL(4) else DFOL_TALL(5)
#undef DFOL_TALL
    DFOL_LAUNCH_CHECK(name);
    return 0;
}

}  // namespace

// The shapes the persistent kernel takes (the callers fall back to csrc/dfol_dense_split.hip otherwise)
extern "C" int dfol_linear_tall_supported(int64_t M, int32_t N, int32_t K) {        // (K >= 100: four steps per block, see epilogue_stage)
    return M >= 16384 && M < (1ll << 31) - 256 && N >= 16 && N <= 320 && K >= 100 && K % 4 == 0;
}

// Y = X W^T + b, fp32 results from two fp16 pieces per operand (bit for bit dfol_linear_act_h2_f32 with no activation); M >= 16384, N <= 320.
// x_part != NULL: also the logit layer's partial sums, FOUR slots: x_part[s][r] = sum over the s-th quarter of the (padded) columns of
// Sigmoid(Y[r][j]) E[row_pred[r]][j] (row_pred non-decreasing, see dfol_linear_logit_h2_f32).
extern "C" int dfol_linear_tall_h2_f32(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy, int32_t M, int32_t N,
                                       int32_t K, const int32_t* row_pred, const float* E, int64_t ld_e, float* x_part, int64_t ld_xp, void* stream) {
    DFOL_REQUIRE(dfol_linear_tall_supported(M, N, K) && ldx % 4 == 0 && ldx >= K && ldy >= N, "linear_tall_h2: bad sizes M=%d N=%d K=%d (M >= 16384, N <= 320, K, ldx multiples of 4)", M, N, K);
    DFOL_REQUIRE(X && W_split && Y, "linear_tall_h2: null pointer");
    DFOL_REQUIRE(((uintptr_t)X % 16 == 0) && ((uintptr_t)W_split % 16 == 0), "linear_tall_h2: X and W_split must be 16-byte aligned");
    if (x_part) {
        DFOL_REQUIRE(row_pred && E && ld_e >= N && ld_xp >= M, "linear_tall_h2: the logit partial sums need row_pred, E [P, >= N] and x_part [4, >= M]");
        const TlExtra ex = {nullptr, nullptr, row_pred, E, ld_e, 0, x_part, ld_xp};
        return tall_launch<2>(X, ldx, W_split, bias, Y, ldy, M, N, K, ex, stream, "linear_tall_h2 (logit)");
    }
    const TlExtra ex = {nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0};
    return tall_launch<0>(X, ldx, W_split, bias, Y, ldy, M, N, K, ex, stream, "linear_tall_h2");
}

// {S, 1 / S} for dfol_pair_wgrad_fused_f32 from the launch's largest bound: S a power of two with S bound in [2^13, 2^14) (one when
// the bound is zero)
__global__ void tall_wgrad_scale_kernel(const float* __restrict__ bound_max, float* __restrict__ scale) {
    const float b = bound_max[0];
    int e = 0;
    if (b > 0.f && b < 3.0e38f) {
        int x;
        (void)frexpf(b, &x);
        e = 14 - x;
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
    }
    scale[0] = ldexpf(1.0f, e);
    scale[1] = ldexpf(1.0f, -e);
}

// dZ (+)= dpre2 W2 as dfol_pair_dz_fused_f32 (same operands, same results bit for bit), persistent form; workspace: 2 M + 4 floats (the
// rows' scaled dx and un-scaling factors; then max_r |dx[r]| emax[row_pred[r]] / 4 and {S, 1 / S} of it, what dfol_pair_wgrad_fused_f32
// takes as `scale`: workspace + 2 M + 1)
extern "C" int dfol_pair_dz_tall_f32(const float* pre2, int64_t ld_p2, const float* dx, const int32_t* row_pred, const float* E, int64_t ld_e,
                                     const float* emax, const void* W2t_split, float* dZ, int64_t ld_dz, int32_t M, int32_t H1, int32_t H2,
                                     int32_t accumulate, float* workspace, void* stream) {
    DFOL_REQUIRE(dfol_linear_tall_supported(M, H1, H2) && ld_p2 % 4 == 0 && ld_p2 >= H2 && ld_e % 4 == 0 && ld_e >= H2 && ld_dz >= H1,
                 "pair_dz_tall: bad sizes M=%d H1=%d H2=%d", M, H1, H2);
    DFOL_REQUIRE(pre2 && dx && row_pred && E && emax && W2t_split && dZ && workspace, "pair_dz_tall: null pointer");
    DFOL_REQUIRE(((uintptr_t)pre2 % 16 == 0) && ((uintptr_t)E % 16 == 0) && ((uintptr_t)W2t_split % 16 == 0), "pair_dz_tall: pre2, E and the packed weights must be 16-byte aligned");
    float* gs = workspace;
    float* rsinv = workspace + M;
    float* bound_max = workspace + 2 * (int64_t)M;
    DFOL_REQUIRE(hipMemsetAsync(bound_max, 0, 4, (hipStream_t)stream) == hipSuccess, "pair_dz_tall: hipMemsetAsync failed");
    hipLaunchKernelGGL(tall_row_scale_kernel, dim3(std::min(dfol_cdiv(M, 256), 1024)), dim3(256), 0, (hipStream_t)stream, dx, row_pred, emax, M, gs, rsinv,
                       reinterpret_cast<uint32_t*>(bound_max));
    hipLaunchKernelGGL(tall_wgrad_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const float*)bound_max, bound_max + 1);
    DFOL_LAUNCH_CHECK("pair_dz_tall (row scales)");
    const TlExtra ex = {gs, rsinv, row_pred, E, ld_e, accumulate, nullptr, 0};
    return tall_launch<1>(pre2, ld_p2, W2t_split, nullptr, dZ, ld_dz, M, H1, H2, ex, stream, "pair_dz_tall");
}

// Several readers of one trunk (relate hops of a program, option slots of choose_rel) in ONE pass: dZ (+)= dpre2 W2 with
// dpre2[r][j] = h (1 - h) sum_k dx_k[r] E_k[row_pred[r]][j] - each reader on its own costs a pass over pre2 and, from the second on, a read-modify-write
// of dZ (3.1 ms against 1.8 at 256 x 100 objects).  The readers share the row -> embedding row map (every pair row of the batch under one predicate per
// reader, in order); the row's scale is that of the bound sum_k |dx_k[r]| max|E_k[p]| / 4.
__global__ __launch_bounds__(256) void tall_row_scale_multi_kernel(const float* __restrict__ dx, int64_t dx_stride, const int32_t* __restrict__ row_pred,
                                                                   const float* __restrict__ emax, int P, int nr, int M, float* __restrict__ gs,
                                                                   float* __restrict__ rsinv) {
    for (int r = blockIdx.x * 256 + (int)threadIdx.x; r < M; r += gridDim.x * 256) {
        const int p = row_pred[r];
        float g[TL_MAXR], bound = 0.f;
#pragma unroll
        for (int q = 0; q < TL_MAXR; ++q) {
            g[q] = (p >= 0 && q < nr) ? dx[(int64_t)q * dx_stride + r] : 0.f;
            bound += fabsf(g[q]) * (q < nr ? emax[(int64_t)q * P + max(p, 0)] : 0.f) * 0.25f;
        }
        int e = 0;
        if (bound > 0.f && bound < 3.0e38f) {
            int x;
            (void)frexpf(bound, &x);
            e = 14 - x;
            e = e < -100 ? -100 : (e > 100 ? 100 : e);
        }
#pragma unroll
        for (int q = 0; q < TL_MAXR; ++q)
            if (q < nr) gs[(int64_t)q * M + r] = ldexpf(g[q], e);
        rsinv[r] = ldexpf(1.0f, -e);
    }
}

// dx [nr][dx_stride >= M], E [nr][P][ld_e] (reader k's rows at E + k P ld_e), emax [nr][P] = max_j |E_k[p][j]|; workspace: (nr + 1) M floats
extern "C" int dfol_pair_dz_tall_multi_f32(const float* pre2, int64_t ld_p2, const float* dx, int64_t dx_stride, int32_t nr, const int32_t* row_pred,
                                           const float* E, int64_t ld_e, int32_t P, const float* emax, const void* W2t_split, float* dZ, int64_t ld_dz,
                                           int32_t M, int32_t H1, int32_t H2, int32_t accumulate, float* workspace, void* stream) {
    DFOL_REQUIRE(nr >= 1 && nr <= TL_MAXR && P >= 1 && dx_stride >= M, "pair_dz_tall_multi: %d readers (1..%d), P=%d", nr, TL_MAXR, P);
    DFOL_REQUIRE(dfol_linear_tall_supported(M, H1, H2) && ld_p2 % 4 == 0 && ld_p2 >= H2 && ld_e % 4 == 0 && ld_e >= H2 && ld_dz >= H1,
                 "pair_dz_tall_multi: bad sizes M=%d H1=%d H2=%d", M, H1, H2);
    DFOL_REQUIRE(pre2 && dx && row_pred && E && emax && W2t_split && dZ && workspace, "pair_dz_tall_multi: null pointer");
    DFOL_REQUIRE(((uintptr_t)pre2 % 16 == 0) && ((uintptr_t)E % 16 == 0) && ((uintptr_t)W2t_split % 16 == 0),
                 "pair_dz_tall_multi: pre2, E and the packed weights must be 16-byte aligned");
    float* gs = workspace;
    float* rsinv = workspace + (int64_t)nr * M;
    hipLaunchKernelGGL(tall_row_scale_multi_kernel, dim3(std::min(dfol_cdiv(M, 256), 1024)), dim3(256), 0, (hipStream_t)stream, dx, dx_stride, row_pred, emax, P, nr,
                       M, gs, rsinv);
    DFOL_LAUNCH_CHECK("pair_dz_tall_multi (row scales)");
    const TlExtra ex = {gs, rsinv, row_pred, E, ld_e, accumulate, nullptr, 0, nr, (int64_t)M, (int64_t)P * ld_e};
    return tall_launch<1, false, true>(pre2, ld_p2, W2t_split, nullptr, dZ, ld_dz, M, H1, H2, ex, stream, "pair_dz_tall_multi");
}

// The bf16 mode's forms (bf16-STORED activations, one bf16 piece per operand, results rounded to nearest even): bit for bit
// dfol_linear_act_bf16_bf16 (no activation) resp. dfol_pair_logit_bwd_bf16 followed by it.  N % 4 == 0, rows 8-byte aligned (strides in elements).
extern "C" int dfol_linear_tall_bf16_bf16(const void* X_bf16, int64_t ldx, const void* W_bf16, const float* bias, void* Y_bf16, int64_t ldy, int32_t M,
                                          int32_t N, int32_t K, const int32_t* row_pred, const float* E, int64_t ld_e, float* x_part, int64_t ld_xp,
                                          void* stream) {
    DFOL_REQUIRE(dfol_linear_tall_supported(M, N, K) && N % 4 == 0 && ldx % 4 == 0 && ldx >= K && ldy % 4 == 0 && ldy >= N,
                 "linear_tall_bf16_bf16: bad sizes M=%d N=%d K=%d (M >= 16384, N <= 320, N, K, strides multiples of 4)", M, N, K);
    DFOL_REQUIRE(X_bf16 && W_bf16 && Y_bf16, "linear_tall_bf16_bf16: null pointer");
    DFOL_REQUIRE(((uintptr_t)X_bf16 % 8 == 0) && ((uintptr_t)Y_bf16 % 8 == 0) && ((uintptr_t)W_bf16 % 16 == 0), "linear_tall_bf16_bf16: X, Y must be 8-byte and W 16-byte aligned");
    if (x_part) {
        DFOL_REQUIRE(row_pred && E && ld_e >= N && ld_xp >= M, "linear_tall_bf16_bf16: the logit partial sums need row_pred, E [P, >= N] and x_part [4, >= M]");
        const TlExtra ex = {nullptr, nullptr, row_pred, E, ld_e, 0, x_part, ld_xp};
        return tall_launch<2, true>(X_bf16, ldx, W_bf16, bias, Y_bf16, ldy, M, N, K, ex, stream, "linear_tall_bf16_bf16 (logit)");
    }
    const TlExtra ex = {nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0};
    return tall_launch<0, true>(X_bf16, ldx, W_bf16, bias, Y_bf16, ldy, M, N, K, ex, stream, "linear_tall_bf16_bf16");
}

extern "C" int dfol_pair_dz_tall_bf16(const void* pre2_bf16, int64_t ld_p2, const float* dx, const int32_t* row_pred, const float* E, int64_t ld_e,
                                      const void* W2t_bf16, void* dZ_bf16, int64_t ld_dz, int32_t M, int32_t H1, int32_t H2, int32_t accumulate,
                                      void* stream) {
    DFOL_REQUIRE(dfol_linear_tall_supported(M, H1, H2) && H1 % 4 == 0 && ld_p2 % 4 == 0 && ld_p2 >= H2 && ld_e % 4 == 0 && ld_e >= H2 && ld_dz % 4 == 0 && ld_dz >= H1,
                 "pair_dz_tall_bf16: bad sizes M=%d H1=%d H2=%d", M, H1, H2);
    DFOL_REQUIRE(pre2_bf16 && dx && row_pred && E && W2t_bf16 && dZ_bf16, "pair_dz_tall_bf16: null pointer");
    DFOL_REQUIRE(((uintptr_t)pre2_bf16 % 8 == 0) && ((uintptr_t)dZ_bf16 % 8 == 0) && ((uintptr_t)E % 16 == 0) && ((uintptr_t)W2t_bf16 % 16 == 0),
                 "pair_dz_tall_bf16: pre2, dZ must be 8-byte, E and the packed weights 16-byte aligned");
    const TlExtra ex = {dx, nullptr, row_pred, E, ld_e, accumulate, nullptr, 0};
    return tall_launch<1, true>(pre2_bf16, ld_p2, W2t_bf16, nullptr, dZ_bf16, ld_dz, M, H1, H2, ex, stream, "pair_dz_tall_bf16");
}
