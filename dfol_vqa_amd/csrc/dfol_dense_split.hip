// Y = act(X W^T + b) with fp32 results from the bf16 matrix pipes (gfx950): the dense layers of the full-size oracle
// (featurizer 2048 -> 512, attribute / pair first layers 516 -> 256 / 512) on the same exact three-way operand split as the
// fused pair kernel (csrc/dfol_pair_split.hip): x = h + m + l with bf16 pieces, six piece products per fp32 product accumulated
// in fp32 by v_mfma_f32_16x16x32_bf16, the dropped products below 2^-23 |x w|.  6/16 of the fp32 matrix pipe's time.
//
// W is split and packed once per weight version (dfol_linear_pack_w_bf16x3): [N/128 column blocks][K/32 steps][3 pieces][128 rows]
// [4 k-groups] x 16 bytes, rows >= N and k >= K zero, k-groups swizzled by (row >> 2) & 3, so that a workgroup's B tile of one
// k-step is 24 KB of contiguous memory in fragment order.  X is fp32 in HBM: every thread loads the 2 x 8 consecutive k of its two
// rows into registers, splits them (VALU) and writes the three pieces to LDS in the same swizzled layout, so all fragment reads are
// conflict-free ds_read_b128.  Both tiles are staged through registers - the B tile one step ahead, the X rows two steps ahead - and
// vmcnt retires in order, so B is requested before X and waiting for B leaves the X loads in flight (see `step` below; LDS-DMA for
// the B tile was tried first and is described there).
//
// Workgroup: 256 threads = 2 x 2 wavefronts, 128 x 128 output tile, wavefront tile 64 x 64 (16 accumulator tiles, 64 registers);
// 48 KB of LDS: two workgroups per CU.  Consecutive workgroups of one XCD share the X row block (blockIdx is re-mapped so that the
// column blocks of a row block run on the same XCD and hit its L2).
// -DDFOL_DENSE_TRACE: clock64 stamps of one workgroup (tools/scratch/trace_dense.py).
#include "dfol_common.h"

#include <stdlib.h>

#include <type_traits>

#ifdef DFOL_DENSE_TRACE
__device__ long long dfol_dense_trace_buf[4 * 64];
#define LTRACE(slot) do { if (blockIdx.x == 300 && lane == 0 && (slot) < 64) dfol_dense_trace_buf[wave * 64 + (slot)] = clock64(); } while (0)
#else
#define LTRACE(slot)
#endif

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int LS_BM = 128, LS_BN = 128, LS_BK = 32;
// Two measured variants of the step loop for the two-piece kernels, both OFF: -DLS_DOUBLE_BUFFER=1 (two LDS buffers, one barrier per step)
// and with it -DLS_XRING=4 (four steps of X rows in flight instead of two).  At the train step's tall products (2.5 M rows, K = 256 / 300;
// tools/lab/time_tall.py) the single-buffered loop takes 1.96 / 2.13 ms, the double-buffered one 2.02 / 2.27 ms, with the deeper ring
// 2.14 / 2.35 ms: these products move 5.6 GB each at ~2.8 TB/s and neither barriers nor the depth of the prefetch bound them (rocprofv3
// counters, tools/lab/pmc_tall.sh: 46 % of the wavefront cycles wait on memory counters, 15 % issue MFMAs; without the MFMAs the
// kernels are 0.15 ms faster, without the stores 0.5 ms).
#ifndef LS_DOUBLE_BUFFER
#define LS_DOUBLE_BUFFER 0
#endif
// -DLS_B_GLOBAL=1 (two-piece kernels): the B fragments go from the packed image (it is in fragment order) straight into the MFMA operand
// registers, one step ahead - no LDS traffic for B.  The step loop of the two-piece kernels is LDS-bound otherwise: per step and
// wavefront 8 writes and 24 reads of 16 bytes x 64 lanes for 48 MFMAs, 256 KB per CU and step pair = 2048 cycles of the LDS pipe against
// 1536 of the matrix pipe (without any global traffic the forward tall product still takes 1.38 of its 1.96 ms).
// (measured at the same two products: 1.92 / 2.39 ms against 1.95 / 2.13 - the producer kernel spills with the fragment registers; OFF)
#ifndef LS_B_GLOBAL
#define LS_B_GLOBAL 0
#endif
#ifndef LS_XRING
#define LS_XRING 2                // steps of X rows in flight in the double-buffered form (register sets of 16 per 128-row block)
#endif

__device__ __forceinline__ int ls_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }
__device__ __forceinline__ void ls_split(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = __float_as_uint(x);
    const float r = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r);
    l = __float_as_uint(r - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ uint32_t ls_pack(uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// 8 consecutive fp32 -> the three 8 x bf16 pieces
__device__ __forceinline__ void ls_split8(const float4& a, const float4& b, u32x4& h, u32x4& m, u32x4& l) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t ph[8], pm[8], pl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ls_split(v[j], ph[j], pm[j], pl[j]);
    h = u32x4{ls_pack(ph[0], ph[1]), ls_pack(ph[2], ph[3]), ls_pack(ph[4], ph[5]), ls_pack(ph[6], ph[7])};
    m = u32x4{ls_pack(pm[0], pm[1]), ls_pack(pm[2], pm[3]), ls_pack(pm[4], pm[5]), ls_pack(pm[6], pm[7])};
    l = u32x4{ls_pack(pl[0], pl[1]), ls_pack(pl[2], pl[3]), ls_pack(pl[4], pl[5]), ls_pack(pl[6], pl[7])};
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// NP = 2, round 4: TWO fp16 pieces per operand x = h + l (h = fp16(x), l = fp16(x - h), round to nearest even; x - h is exact) and
// THREE products xl wh + xh wl + xh wh on v_mfma_f32_16x16x32_f16 (csrc/dfol_pair_h2.hip has the accuracy story: 22 - 23 significand bits
// per operand, the dropped product below 2^-22; every row of W is scaled by its own power of two at pack time so that its low pieces
// are normal fp16 numbers, and the epilogue multiplies the accumulator by 2^-e_n).  X is split UNSCALED: an element's error is
// max(2^-22 |x|, 2^-25) - fp32-class for activations of order 1 (features, Sigmoid / ELU outputs: the forward products), and NOT for
// operands of arbitrary magnitude (gradients): the backward products stay on the three bf16 pieces, whose exponent range is fp32's.
// |x| > 65504 overflows fp16 (the result is NaN, loudly).
__device__ __forceinline__ void ls_split2h(float x0, float x1, uint32_t& h, uint32_t& l) {
    const f32x2 x = {x0, x1};
    const f16x2 hh = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(hh, f32x2);
    h = __builtin_bit_cast(uint32_t, hh);
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}
__device__ __forceinline__ void ls_split8h(const float4& a, const float4& b, u32x4& h, u32x4& l) {
    uint32_t hh[4], ll[4];
    ls_split2h(a.x, a.y, hh[0], ll[0]);
    ls_split2h(a.z, a.w, hh[1], ll[1]);
    ls_split2h(b.x, b.y, hh[2], ll[2]);
    ls_split2h(b.z, b.w, hh[3], ll[3]);
    h = u32x4{hh[0], hh[1], hh[2], hh[3]};
    l = u32x4{ll[0], ll[1], ll[2], ll[3]};
}
// two fp32 -> two bf16, round to nearest even (v_cvt_pk_bf16_f32): the operand form of the bf16 mode (NP = 1)
__device__ __forceinline__ uint32_t ls_rne2(float x0, float x1) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x0, x1}, bf16x2)); }

// np = 2: one wavefront per (padded) row of W: e_n puts the row's largest magnitude into [2^13, 2^14); tail[n] = 2^-e_n, tail[rows + n] = e_n.
__global__ void linear_row_scale_kernel(const float* __restrict__ W, int64_t ldw, int N, int K, int rows, float* __restrict__ tail) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= rows) return;
    float m = 0.f;
    if (n < N)
        for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(W[(int64_t)n * ldw + k]));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    int e = 0;
    if (m > 0.f && m < 3.0e38f) {
        int x;
        (void)frexpf(m, &x);
        e = 14 - x;
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
    }
    if (lane == 0) {
        tail[n] = ldexpf(1.0f, -e);
        reinterpret_cast<int32_t*>(tail)[rows + n] = e;
    }
}

// One thread per 16-byte piece of the packed image.  np = 3: the exact bf16 split; np = 2: the two fp16 pieces of the row-scaled weights
// (`tail`: linear_row_scale_kernel's exponents); np = 1: the bf16 mode's image (one piece, rounded to nearest)
__global__ void linear_pack_w_split_kernel(const float* __restrict__ W, int64_t ldw, int N, int K, int ksteps, int nblocks, int np,
                                           const float* __restrict__ tail, u32x4* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int tile_pieces = np * LS_BN * 4;
    if (idx >= (int64_t)nblocks * ksteps * tile_pieces) return;
    const int rem = (int)(idx % tile_pieces);
    const int64_t tile = idx / tile_pieces;
    const int ks = (int)(tile % ksteps), nb = (int)(tile / ksteps);
    const int p = rem / (LS_BN * 4), rr = rem - p * LS_BN * 4, r = rr >> 2, slot = rr & 3;
    const int kq = slot ^ ls_swz(r), n = nb * LS_BN + r, k0 = ks * LS_BK + kq * 8;
    uint32_t piece[8];
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        w[j] = (n < N && k0 + j < K) ? W[(int64_t)n * ldw + k0 + j] : 0.f;
        uint32_t h, m, l;
        ls_split(w[j], h, m, l);
        piece[j] = p == 0 ? h : (p == 1 ? m : l);
    }
    if (np == 2) {
        const int e = reinterpret_cast<const int32_t*>(tail)[nblocks * LS_BN + n];
        uint32_t q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t h, l;
            ls_split2h(ldexpf(w[2 * j], e), ldexpf(w[2 * j + 1], e), h, l);
            q[j] = p == 0 ? h : l;
        }
        out[idx] = u32x4{q[0], q[1], q[2], q[3]};
    } else if (np == 1) out[idx] = u32x4{ls_rne2(w[0], w[1]), ls_rne2(w[2], w[3]), ls_rne2(w[4], w[5]), ls_rne2(w[6], w[7])};
    else out[idx] = u32x4{ls_pack(piece[0], piece[1]), ls_pack(piece[2], piece[3]), ls_pack(piece[4], piece[5]), ls_pack(piece[6], piece[7])};
}

// PROD (round 4): the A operand is PRODUCED, not loaded - the backward of the pair MLP's logit layer folded into the input-gradient
// product dZ = dpre2 W2 (csrc/dfol_pair_train.hip, logit_bwd: dpre2[r][j] = dx[r] E[p(r)][j] h (1 - h), h = Sigmoid(pre2[r][j])).  X is
// pre2; a thread keeps dx[r] and the embedding row of r's predicate for its two rows, loads the row's E pieces beside the X pieces and
// writes the two fp16 pieces of 2^e_r dpre2 to LDS: dpre2 (3 GB written and read back per train step at 256 x 100 objects) never
// exists.  e_r puts the row's BOUND |dx[r]| max|E[p]| / 4 into [2^13, 2^14) - gradients have no natural scale, fp16 pieces need one -
// so an element's error is max(2^-23 |a|, 2^-39 |dx[r]| max|E[p]|); the epilogue multiplies the row by 2^-e_r (exact).
struct LsProducer {
    const float* g;                 // dx [M]: gradient of the row's logit
    const int32_t* row_pred;        // [M]: the row of E the pair row reads (its predicate), or -1: no gradient
    const float* E;                 // [P, K] embedding rows, 16-byte aligned rows
    int64_t ld_e;
    const float* emax;              // [P]: max |E[p][:]|
    int accumulate;                 // Y += instead of Y = (a second use of the same hidden layer adds its input gradient)
    // MODE 2 (LOGIT), the forward counterpart: the epilogue ALSO leaves, per row and 64-column half block, the partial sum
    // sum_j Sigmoid(Y[r][j]) E[row_pred[r]][j] - the logit layer's forward (csrc/dfol_pair_train.hip, logit_fwd) without its pass over Y
    float* x_part;                  // [2 column blocks of N][ld_xp >= M]
    int64_t ld_xp;
    // every mode: the caller's fp16-range status word (dfol_set_range_status) or NULL - the two-piece kernels OR DFOL_RANGE_X_OVERFLOW into it
    // when an X element is beyond fp16's largest finite value (its high piece is inf and the products NaN)
    uint32_t* status;
};
__device__ __forceinline__ float ls_dsigmoid(float x) {
    const float h = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
    return h * (1.0f - h);
}

template <int ACT>
__device__ __forceinline__ float ls_act(float x) {
    // branch-free forms on the hardware exp / log / rcp (1 ulp each; absolute error < 2e-7 on these ranges)
    if (ACT == DFOL_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + dfol_exp(-x));
    if (ACT == DFOL_ACT_ELU) return fmaxf(x, dfol_exp(fminf(x, 0.f)) - 1.0f);
    if (ACT == DFOL_ACT_LOGSIGMOID) return fminf(x, 0.f) - dfol_log(1.0f + dfol_exp(-fabsf(x)));
    return x;
}

// XV: floats per X load (4: rows 16-byte aligned; 2: rows 8-byte aligned, e.g. the 2054-column raw feature matrix)
// NT: column tiles of 16 per wavefront - 4, or 2 for a last column block of at most 64 valid columns (N = 300 = 128 + 128 + 44: the third
// block would otherwise spend a full block's MFMAs on 44 columns, 28 % of the pair layer's forward product)
// NP: pieces per operand - 3: fp32 results (six piece products per step and accumulator); 1: the bf16 mode (operands rounded to bf16, one
// product, fp32 accumulation - what a "bf16 forward" computes; BASELINE configs[3])
// RT: row tiles of 16 per wavefront - 4 (a 128-row block: the default) or 2 (a 64-row block, for products whose 128-row tiling would leave
// most of the chip without a workgroup: the featurizer of 36-object scenes is 288 blocks of 128 rows on 512 workgroup slots, and a
// batch of shared scenes 60).  An output element sees the same products in the same order whatever the block height, so the choice
// changes no result bit (tests/test_kernels_gpu.py::test_linear_act_split_block_height_changes_no_bit) - a sharded run still equals
// the single-process run.
// BIO: bf16 storage on both sides (NP = 1 only: the bf16 mode's per-pair activations): X and Y are rows of bfloat16 (8-byte aligned rows,
// K % 4 == 0); a thread's 8 consecutive k are two 8-byte loads that go to LDS as they are - no conversion - and the epilogue rounds the
// fp32 accumulators to nearest even.  ldx / ldy count ELEMENTS.
template <int ACT, int XV, int NT, int NP, int RT, bool BIO, int MODE = 0>
__device__ __forceinline__ void ls_tile(u32x4* __restrict__ As, u32x4* __restrict__ Bs, const void* __restrict__ Xv, int64_t ldx,
                                        const u32x4* __restrict__ Wp, const float* __restrict__ bias, void* __restrict__ Yv, int64_t ldy, int M, int N,
                                        int K, int ksteps, int mb, int nb, int nbn, const LsProducer& prod = LsProducer(), float* __restrict__ Rs = nullptr) {
    static_assert(!BIO || NP == 1, "bf16 storage belongs to the bf16 mode");
    constexpr bool PROD = MODE == 1, LOGIT = MODE == 2;
    constexpr bool BG = LS_B_GLOBAL && !LS_DOUBLE_BUFFER && NP == 2 && !BIO;     // B fragments straight from the packed image
    constexpr bool DB = LS_DOUBLE_BUFFER && NP == 2 && !BIO;          // two LDS buffers (2 x 32 KB for a 128-row block: still two workgroups per CU)
    static_assert(!LOGIT || (!BIO && ACT == DFOL_ACT_NONE), "the logit partial sums belong to the fp32 pre-activation output");
    static_assert(!PROD || (NP == 2 && !BIO && XV == 4 && ACT == DFOL_ACT_NONE), "the produced operand is two fp16 pieces of fp32 rows");
    typedef typename std::conditional<BIO, uint16_t, float>::type TX;
    const TX* __restrict__ X = reinterpret_cast<const TX*>(Xv);
    TX* __restrict__ Y = reinterpret_cast<TX*>(Yv);
    constexpr int BM = 32 * RT, RH = RT / 2;                          // rows of the block; 64-row halves staged per thread
    const int m0 = mb * BM, n0 = nb * LS_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kh = lane >> 4, r16 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int WN = 16 * NT;                                       // columns of a wavefront

    floatx4 acc[RT][NT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // A staging: rows (tid >> 2) and (tid >> 2) + 64, k-group tid & 3 (8 consecutive k): four threads read 128 contiguous bytes
    const int arow = tid >> 2, aq = tid & 3;
    const TX* xp0 = X + (int64_t)min(m0 + arow, M - 1) * ldx + aq * 8;
    const TX* xp1 = X + (int64_t)min(m0 + arow + 64, M - 1) * ldx + aq * 8;      // (RT = 4 only)
    float gs[RT / 2];                                                 // PROD: 2^e_r dx[r] of the thread's rows, and their embedding rows
    const float* ep[RT / 2];
    if constexpr (PROD) {
#pragma unroll
        for (int h = 0; h < RT / 2; ++h) {
            const int row = m0 + arow + 64 * h, rc = min(row, M - 1);
            const int p = prod.row_pred[rc];
            const bool live = row < M && p >= 0;
            const int pc = max(p, 0);
            const float gg = live ? prod.g[rc] : 0.f;
            const float bound = fabsf(gg) * prod.emax[pc] * 0.25f;
            int e = 0;
            if (bound > 0.f && bound < 3.0e38f) {
                int x;
                (void)frexpf(bound, &x);
                e = 14 - x;
                e = e < -100 ? -100 : (e > 100 ? 100 : e);
            }
            gs[h] = ldexpf(gg, e);
            ep[h] = prod.E + (int64_t)pc * prod.ld_e + aq * 8;
            if (aq == 0) Rs[arow + 64 * h] = ldexpf(1.0f, -e);        // read by the epilogue, many barriers later
        }
    }
    // X registers: two steps in flight (HBM latency is longer than one step of 96 MFMAs).  The loads are unconditional - addresses
    // clamped, out-of-range k zeroed afterwards - so that every wavefront issues exactly 4 per step and the vmcnt arithmetic below holds.
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef typename std::conditional<BIO, u32x2, float4>::type XR;  // four consecutive k of a row as loaded
    constexpr int XD = DB ? LS_XRING : 2;                           // X register sets = steps of X in flight
    XR xa[XD][RH][2];                                               // [set][row half][k half]
    float4 ea[PROD ? RH : 1][2];                                    // PROD: the same pieces of the rows' embedding rows - L2 hits, ONE step ahead
    float xmax = 0.f;                                               // NP = 2: the largest |x| this thread split (one v_max3 per two elements)
    auto load_x = [&](int ks, auto set_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        const int k = ks * LS_BK + aq * 8;
        const int c0 = min(k, K - 4) - aq * 8, c1 = min(k + 4, K - 4) - aq * 8;      // K % 4 == 0: a group of four is wholly in or out
        auto ld4 = [&](const TX* p) __attribute__((always_inline)) {
            if constexpr (BIO) {
                return *reinterpret_cast<const u32x2*>(p);
            } else {
                if (XV == 4) return *reinterpret_cast<const float4*>(p);
                const float2 lo = *reinterpret_cast<const float2*>(p), hi = *reinterpret_cast<const float2*>(p + 2);
                return make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        };
        xa[S][0][0] = ld4(xp0 + c0);
        xa[S][0][1] = ld4(xp0 + c1);
        if constexpr (RH == 2) {
            xa[S][1][0] = ld4(xp1 + c0);
            xa[S][1][1] = ld4(xp1 + c1);
        }
    };
    auto load_e = [&](int ks) __attribute__((always_inline)) {
        if constexpr (PROD) {
            const int k = ks * LS_BK + aq * 8;
            const int c0 = min(k, K - 4) - aq * 8, c1 = min(k + 4, K - 4) - aq * 8;
#pragma unroll
            for (int h = 0; h < RH; ++h) {
                ea[h][0] = *reinterpret_cast<const float4*>(ep[h] + c0);
                ea[h][1] = *reinterpret_cast<const float4*>(ep[h] + c1);
            }
        }
    };
    auto store_a = [&](int ks, auto set_tag, int off = 0) __attribute__((always_inline)) {     // off: the LDS buffer (double-buffered form)
        constexpr int S = decltype(set_tag)::value;
        const int k = ks * LS_BK + aq * 8;
#pragma unroll
        for (int h = 0; h < RH; ++h) {
            const int row = arow + 64 * h;
            const int at = off + row * 4 + (aq ^ ls_swz(row));
            if constexpr (BIO) {
                const u32x2 zz = u32x2{0u, 0u};
                const u32x2 v0 = k < K ? xa[S][h][0] : zz, v1 = k + 4 < K ? xa[S][h][1] : zz;
                As[at] = u32x4{v0.x, v0.y, v1.x, v1.y};
            } else {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 v0 = k < K ? xa[S][h][0] : z, v1 = k + 4 < K ? xa[S][h][1] : z;
                if constexpr (PROD) {
                    auto dp = [&](const float4& x, const float4& e) __attribute__((always_inline)) {
                        return make_float4((gs[h] * e.x) * ls_dsigmoid(x.x), (gs[h] * e.y) * ls_dsigmoid(x.y), (gs[h] * e.z) * ls_dsigmoid(x.z),
                                           (gs[h] * e.w) * ls_dsigmoid(x.w));
                    };
                    v0 = k < K ? dp(xa[S][h][0], ea[h][0]) : z;
                    v1 = k + 4 < K ? dp(xa[S][h][1], ea[h][1]) : z;
                }
                if (NP == 1) {
                    As[at] = u32x4{ls_rne2(v0.x, v0.y), ls_rne2(v0.z, v0.w), ls_rne2(v1.x, v1.y), ls_rne2(v1.z, v1.w)};
                } else if (NP == 2) {
                    u32x4 ph, pl;
                    if constexpr (!PROD) {                      // (the produced operand is scaled into range by construction)
                        xmax = fmaxf(fmaxf(xmax, fabsf(v0.x)), fabsf(v0.y));
                        xmax = fmaxf(fmaxf(xmax, fabsf(v0.z)), fabsf(v0.w));
                        xmax = fmaxf(fmaxf(xmax, fabsf(v1.x)), fabsf(v1.y));
                        xmax = fmaxf(fmaxf(xmax, fabsf(v1.z)), fabsf(v1.w));
                    }
                    ls_split8h(v0, v1, ph, pl);
                    As[at] = ph;
                    As[BM * 4 + at] = pl;
                } else {
                    u32x4 ph, pm, pl;
                    ls_split8(v0, v1, ph, pm, pl);
                    As[at] = ph;
                    As[BM * 4 + at] = pm;
                    As[2 * BM * 4 + at] = pl;
                }
            }
        }
    };
    constexpr int TILE_PIECES = NP * LS_BN * 4;                        // 16-byte pieces of one B tile in the packed image
    const u32x4* wtile = Wp + (int64_t)nb * ksteps * TILE_PIECES;
    // B tile of the next step: six 16-byte pieces per thread, in registers until the tile in LDS has been consumed.  (LDS-DMA would
    // save the registers, but the compiler treats an in-flight global_load_lds as a pending FLAT access and turns EVERY later wait -
    // also the one for the X registers - into vmcnt(0), which would drain the two-step X prefetch at every step.  Reading the B
    // fragments straight from global memory into the MFMA operand registers - the packed image is in fragment order - was measured
    // too: no LDS traffic for B at all, same speed at K = 2048 and 15 % slower at K = 516.)
    u32x4 wb[TILE_PIECES / 256];
    auto load_w = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TILE_PIECES / 256; ++i) wb[i] = wtile[(int64_t)ks * TILE_PIECES + 256 * i + tid];
    };
    auto store_b = [&](int off = 0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TILE_PIECES / 256; ++i) Bs[off + 256 * i + tid] = wb[i];
    };

    const int aoff = (wm * (16 * RT) + r16) * 4 + (kh ^ ls_swz(r16));
    const int boff = (wn * WN + r16) * 4 + (kh ^ ls_swz(r16));

    u32x4 bq[1][BG ? NT : 1][BG ? NP : 1];                           // BG: the B fragments of the step (requested at its top, under the split of the X rows)
    auto load_bq = [&](int ks, auto set_tag) __attribute__((always_inline)) {
        if constexpr (BG) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p) bq[0][j][p] = wtile[(int64_t)ks * TILE_PIECES + p * LS_BN * 4 + j * 64 + boff];
        }
    };
    constexpr int PA6[6] = {2, 0, 1, 1, 0, 0}, PB6[6] = {0, 2, 1, 0, 1, 0};
    constexpr int X0 = NP == 1 ? 5 : 0;                               // the bf16 mode keeps the last product only (piece 0 x piece 0)
    constexpr int PA3[3] = {1, 0, 0}, PB3[3] = {0, 1, 0};             // NP = 2: xl wh, xh wl, xh wh (smallest first)
    typedef typename std::conditional<NP == 2, f16x8, bf16x8>::type FR;
    auto multiply = [&](int off, auto bq_tag) __attribute__((always_inline)) {  // the step's MFMAs on the tiles of LDS buffer `off`
#pragma unroll
        for (int ih = 0; ih < RT; ih += 2) {                // two row tiles at a time (register budget)
            FR a[2][NP];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) a[i][p] = __builtin_bit_cast(FR, As[off + p * BM * 4 + (ih + i) * 64 + aoff]);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                FR b[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if constexpr (BG) b[p] = __builtin_bit_cast(FR, bq[0][j][p]);
                    else b[p] = __builtin_bit_cast(FR, Bs[off + p * LS_BN * 4 + j * 64 + boff]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if constexpr (NP == 2) {
#pragma unroll
                        for (int x = 0; x < 3; ++x)         // three dependent MFMAs per accumulator, smallest terms first
                            acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][PA3[x]], b[PB3[x]], acc[ih + i][j], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int x = X0; x < 6; ++x)        // six dependent MFMAs per accumulator, smallest terms first
                            acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA6[x]], b[PB6[x]], acc[ih + i][j], 0, 0, 0);
                    }
                }
            }
        }
    };
    // Step ks: X(ks) was requested two steps ago, the B tile one step ago, and X(ks+1) after it: vmcnt retires in order, so the
    // wait for the B registers leaves the four loads of X(ks+1) in flight.
    auto step = [&](int ks, auto set_tag, auto has_b, auto has_x) __attribute__((always_inline)) {
        constexpr bool HAS_B = decltype(has_b)::value, HAS_X = decltype(has_x)::value;     // is there a B tile ks+1 / an X step ks+2
        LTRACE(4 * ks);
        if constexpr (BG) load_bq(ks, set_tag);             // (L2 hits: there by the time the X rows are split, stored and the barrier passed)
        store_a(ks, set_tag);
        if constexpr (!BG) store_b();
        LTRACE(4 * ks + 1);
        __syncthreads();                                    // A pieces and B tile ks visible
        LTRACE(4 * ks + 2);
        if constexpr (!BG) {
            if (HAS_B) load_w(ks + 1);
        }
        if (HAS_B) load_e(ks + 1);                          // (consumed by this step's store_a already: one register set)
        if (HAS_X) load_x(ks + 2, set_tag);
        __builtin_amdgcn_sched_barrier(0);                  // requests first; and the next step's split must not drift up here
        multiply(0, set_tag);
        __builtin_amdgcn_sched_barrier(0);                  // (it would wait for X(ks+1) in the middle of the MFMAs)
        LTRACE(4 * ks + 3);
        __syncthreads();                                    // A and B tile ks fully read
    };
    // The steady-state body has no conditional memory operations (the compiler's wait counts stay exact); the last three steps
    // are peeled.
    const std::integral_constant<int, 0> S0;
    const std::integral_constant<int, 1> S1;
    const std::true_type yes;
    const std::false_type no;
    LTRACE(60);
    load_x(0, S0);
    if constexpr (!BG) load_w(0);
    load_e(0);
    load_x(min(1, ksteps - 1), S1);
    int ks = 0;
    if constexpr (DB) {
        // Two LDS buffers, ONE barrier per step: step ks multiplies buffer ks & 1 while the tiles of step ks + 1 are split and written
        // to the other one - by whichever wavefront gets there, under the MFMAs of the others (the single-buffered form below stops
        // every wavefront twice per step: 3300 cycles per step of 768 cycles of MFMAs at K = 256 .. 300).  All loads are unconditional
        // (steps past the end re-read the last one: L2 hits, never used), so the compiler's wait counts stay exact; a store past the
        // end goes to the buffer nobody reads any more.
        constexpr int BUF = NP * BM * 4 + NP * LS_BN * 4;
        const int last = ksteps - 1;
        const std::integral_constant<int, 2 % XD> S2;
        const std::integral_constant<int, 3 % XD> S3;
        if constexpr (XD == 4) {                            // (steps 0 and 1 were requested above)
            load_x(min(2, last), S2);
            load_x(min(3, last), S3);
        }
        store_a(0, S0, 0);
        store_b(0);
        load_w(min(1, last));
        load_e(min(1, last));
        load_x(min(XD, last), S0);
        __syncthreads();
        auto body = [&](int k, auto next_tag) __attribute__((always_inline)) {      // k: the step multiplied; next_tag: the X set of step k + 1
            const int cur = (k & 1) * BUF, nxt = BUF - cur;
            LTRACE(4 * k);
            store_a(k + 1, next_tag, nxt);                  // (k + 1 past the end: all columns >= K, zeros)
            store_b(nxt);
            load_w(min(k + 2, last));
            load_e(min(k + 2, last));
            load_x(min(k + 1 + XD, last), next_tag);
            LTRACE(4 * k + 1);
            __builtin_amdgcn_sched_barrier(0);
            multiply(cur, std::integral_constant<int, 0>());
            __builtin_amdgcn_sched_barrier(0);
            LTRACE(4 * k + 3);
            __syncthreads();                                // buffer nxt complete, buffer cur fully read
        };
        if constexpr (XD == 4) {
            for (; ks + 3 < ksteps; ks += 4) {
                body(ks, S1);
                body(ks + 1, S2);
                body(ks + 2, S3);
                body(ks + 3, S0);
            }
            if (ks < ksteps) body(ks, S1);                  // (the tags continue where the loop stopped: ks is a multiple of 4)
            if (ks + 1 < ksteps) body(ks + 1, S2);
            if (ks + 2 < ksteps) body(ks + 2, S3);
        } else {
            for (; ks + 1 < ksteps; ks += 2) {
                body(ks, S1);
                body(ks + 1, S0);
            }
            if (ks < ksteps) body(ks, S1);
        }
        ks = ksteps;
    }
    for (; ks + 3 < ksteps; ks += 2) {
        step(ks, S0, yes, yes);
        step(ks + 1, S1, yes, yes);
    }
    const int rem = DB ? 0 : ksteps - ks;
    if (rem == 0) {
    } else if (rem == 3) {
        step(ks, S0, yes, yes);
        step(ks + 1, S1, yes, no);
        step(ks + 2, S0, no, no);
    } else if (rem == 2) {
        step(ks, S0, yes, no);
        step(ks + 1, S1, no, no);
    } else {
        step(ks, S0, no, no);
    }

    // epilogue: lane holds column r16 and rows 4 kh + e of every 16 x 16 tile.  Interior tiles take the branch-free path.
    LTRACE(61);
    if constexpr (NP == 2 && !PROD) {                               // an X element beyond fp16's range: say so (its products are NaN)
        if (prod.status != nullptr && !(xmax <= 65504.0f)) atomicOr(prod.status, (uint32_t)DFOL_RANGE_X_OVERFLOW);
    }
    // LOGIT: the rows of a block nearly always belong to ONE predicate (row_pred is non-decreasing: a predicate owns n (n - 1) consecutive
    // rows) - then every thread needs the same few embedding values for all its rows, requested here, ahead of the epilogue's arithmetic;
    // a block across a boundary looks its rows up one by one.
    int lg_p0 = -1;
    bool lg_uni = false;
    float4 lg_e4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float lg_ej[NT];
    if constexpr (LOGIT) {
        lg_p0 = __builtin_amdgcn_readfirstlane(prod.row_pred[m0]);
        lg_uni = lg_p0 >= 0 && lg_p0 == __builtin_amdgcn_readfirstlane(prod.row_pred[min(m0 + BM, M) - 1]);
        if (lg_uni) {
            if (NT == 4 && n0 + LS_BN <= N) lg_e4 = *reinterpret_cast<const float4*>(prod.E + (int64_t)lg_p0 * prod.ld_e + n0 + 4 * (tid & 31));
#pragma unroll
            for (int j = 0; j < NT; ++j) lg_ej[j] = prod.E[(int64_t)lg_p0 * prod.ld_e + min(n0 + wn * WN + j * 16 + r16, N - 1)];
        }
    }
    float bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = bias ? bias[min(n0 + wn * WN + j * 16 + r16, N - 1)] : 0.f;
    if constexpr (NP == 2) {                                          // un-scale the rows of W: acc <- acc 2^-e_n (exact), then the bias
        const float* cs = reinterpret_cast<const float*>(Wp + (int64_t)nbn * ksteps * TILE_PIECES);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float c = cs[n0 + wn * WN + j * 16 + r16];          // (the tail is padded to whole column blocks)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] *= c;
        }
    }
    if constexpr (PROD) {                                             // un-scale the produced rows: acc <- acc 2^-e_r (exact)
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = Rs[wm * (16 * RT) + i * 16 + 4 * kh + e];
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j][e] *= c;
            }
    }
    if constexpr (BIO) {
        // bf16 output: 2-byte stores straight from the accumulator layout (16 lanes x 2 bytes per row and instruction) cost a third of the
        // kernel (1.02 ms against 0.73 without any store); the tile goes through LDS instead - the step loop's last barrier has freed it - and
        // leaves as 8-byte pieces, 32 consecutive threads on the 256 contiguous bytes of a row (N % 4 == 0: a piece is wholly in or out).
        constexpr int PITCH = LS_BN + 8;                                  // halfwords per staged row (272 bytes: 8-byte aligned, rows 4 banks apart)
        uint16_t* stage = reinterpret_cast<uint16_t*>(As);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    stage[(wm * (16 * RT) + i * 16 + 4 * kh + e) * PITCH + wn * WN + j * 16 + r16] = (uint16_t)ls_rne2(ls_act<ACT>(acc[i][j][e] + bv[j]), 0.f);
        __syncthreads();
        typedef uint32_t u32x2s __attribute__((ext_vector_type(2)));
        const int cols = min(N - n0, NT == 4 ? LS_BN : 2 * WN);           // valid columns of this block
#pragma unroll
        for (int it = 0; it < BM * (LS_BN / 4) / 256; ++it) {
            const int c = tid + 256 * it, row = c / (LS_BN / 4), cc = c % (LS_BN / 4);
            if (m0 + row < M && 4 * cc < cols)
                *reinterpret_cast<u32x2s*>(Y + (int64_t)(m0 + row) * ldy + n0 + 4 * cc) = *reinterpret_cast<const u32x2s*>(stage + row * PITCH + 4 * cc);
        }
        return;
    }
    if constexpr (!BIO) {
        // Interior tiles leave through LDS: straight from the accumulator layout a store instruction writes 16 lanes x 4 bytes of four
        // rows (64 per thread and tile; the stores were a quarter of the tall products' time: 1.96 -> 1.46 ms without them), staged, a
        // thread stores 16 bytes and 32 consecutive threads one 512-byte row segment.  Two passes of half the block's rows (the loop's
        // last barrier has freed the operand tiles); rows 132 floats apart: the four row groups of a wavefront land 16 banks apart.
        const bool interior = NT == 4 && m0 + BM <= M && n0 + LS_BN <= N && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0;
        if (interior) {
            constexpr int PITCH = LS_BN + 4, IPP = RT / 2;
            float* stage = reinterpret_cast<float*>(As);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (p) __syncthreads();                               // the first pass's rows have been read
#pragma unroll
                for (int ii = 0; ii < IPP; ++ii)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            stage[(wm * (16 * IPP) + ii * 16 + 4 * kh + e) * PITCH + wn * WN + j * 16 + r16] = ls_act<ACT>(acc[p * IPP + ii][j][e] + bv[j]);
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 16 * RT * (LS_BN / 4) / 256; ++it) {
                    const int c = tid + 256 * it, row = c >> 5, c4 = c & 31;
                    const int half = row / (16 * IPP), within = row - half * (16 * IPP);
                    float4 v = *reinterpret_cast<const float4*>(stage + row * PITCH + 4 * c4);
                    float* dst = reinterpret_cast<float*>(Y) + (int64_t)(m0 + half * (16 * RT) + p * (16 * IPP) + within) * ldy + n0 + 4 * c4;
                    if constexpr (PROD) {
                        if (prod.accumulate) {
                            const float4 o = *reinterpret_cast<const float4*>(dst);
                            v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
                        }
                    }
                    *reinterpret_cast<float4*>(dst) = v;
                    if constexpr (LOGIT) {                            // 16 consecutive lanes hold the 64 columns of a half block of one row
                        const int grow = m0 + half * (16 * RT) + p * (16 * IPP) + within;
                        float4 e = lg_e4;                             // (c4 = tid & 31 in every trip)
                        bool live = true;
                        if (!lg_uni) {
                            const int pr = prod.row_pred[grow];
                            live = pr >= 0;
                            e = *reinterpret_cast<const float4*>(prod.E + (int64_t)max(pr, 0) * prod.ld_e + n0 + 4 * c4);
                        }
                        float sum = ls_act<DFOL_ACT_SIGMOID>(v.x) * e.x + ls_act<DFOL_ACT_SIGMOID>(v.y) * e.y + ls_act<DFOL_ACT_SIGMOID>(v.z) * e.z +
                                    ls_act<DFOL_ACT_SIGMOID>(v.w) * e.w;
                        sum = live ? sum : 0.f;
#pragma unroll
                        for (int sh = 1; sh < 16; sh <<= 1) sum += __shfl_xor(sum, sh, 64);
                        if ((c4 & 15) == 0) prod.x_part[(int64_t)(2 * nb + (c4 >> 4)) * prod.ld_xp + grow] = sum;
                    }
                }
            }
            LTRACE(62);
            return;
        }
    }
    TX* yp = Y + (int64_t)(m0 + wm * (16 * RT) + 4 * kh) * ldy + n0 + wn * WN + r16;
    auto out = [](float v) __attribute__((always_inline)) { return v; };
    if constexpr (PROD) {
        if (prod.accumulate) {                                        // (rows / columns past the matrix: clamped reads, never stored)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t r = min(m0 + wm * (16 * RT) + i * 16 + 4 * kh + e, M - 1);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j][e] += Y[r * ldy + min(n0 + wn * WN + j * 16 + r16, N - 1)];
                }
        }
    }
    if (NT == 4 && m0 + BM <= M && n0 + LS_BN <= N) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NT; ++j) yp[(int64_t)(i * 16 + e) * ldy + j * 16] = out(ls_act<ACT>(acc[i][j][e] + bv[j]));
    } else {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool row_ok = m0 + wm * (16 * RT) + i * 16 + 4 * kh + e < M;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float v = ls_act<ACT>(acc[i][j][e] + bv[j]);
                    if (row_ok && n0 + wn * WN + j * 16 + r16 < N) yp[(int64_t)(i * 16 + e) * ldy + j * 16] = out(v);
                }
            }
    }
    if constexpr (LOGIT) {                                            // (edge tiles: straight from the accumulator layout; columns past N add nothing)
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int grow = m0 + wm * (16 * RT) + i * 16 + 4 * kh + e;
                const int pr = lg_uni ? lg_p0 : (grow < M ? prod.row_pred[grow] : -1);
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = n0 + wn * WN + j * 16 + r16;
                    if (pr >= 0 && col < N)
                        sum += ls_act<DFOL_ACT_SIGMOID>(acc[i][j][e] + bv[j]) * (lg_uni ? lg_ej[j] : prod.E[(int64_t)pr * prod.ld_e + col]);
                }
#pragma unroll
                for (int sh = 1; sh < 16; sh <<= 1) sum += __shfl_xor(sum, sh, 64);
                // (a 64-row, two-tile wavefront covers 32 columns: its partial goes to the half block its columns lie in; NT = 2 only in the last block)
                if (r16 == 0 && grow < M) {
                    if (NT == 4) prod.x_part[(int64_t)(2 * nb + wn) * prod.ld_xp + grow] = sum;
                    else prod.x_part[(int64_t)(2 * nb + wn) * prod.ld_xp + grow] = sum;
                }
            }
    }
    LTRACE(62);
}

#ifndef DFOL_BIO_WAVES
#define DFOL_BIO_WAVES 3
#endif
// (bf16 storage: a tile of K = 256 is 8 steps of 16 MFMAs per wavefront, so the two ends of a tile - cold start, stores - outweigh its
// steps; the one-piece tiles are 16 KB of LDS and the lighter register set fits three workgroups per CU.  Measured at 256 x 100 objects,
// forward / input-gradient product: 2 per CU 1.23 / 1.10 ms, 3 per CU 1.13 / 1.09 ms, 4 per CU (128 registers: 88 spilled) 1.26 / 1.56 ms)
template <int ACT, int XV, int NP, int RT, bool BIO = false, int MODE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BIO ? DFOL_BIO_WAVES : 2, BIO ? DFOL_BIO_WAVES : (RT == 2 ? 3 : 2)))) void linear_act_split_kernel(
    const void* __restrict__ X, int64_t ldx, const u32x4* __restrict__ Wp, const float* __restrict__ bias, void* __restrict__ Y,
    int64_t ldy, int M, int N, int K, int ksteps, int nbn, int nblocks, LsProducer prod) {
    constexpr int LP = BIO ? 1 : (NP == 2 ? 2 : 3);                         // (the bf16 mode's fp32-storage kernels keep the 48 KB of NP = 3: same occupancy as before)
    constexpr int A_PIECES = LP * 32 * RT * 4, B_PIECES = LP * LS_BN * 4;   // [piece][row][k-group] 24 KB (12 KB for 64-row blocks); the B tile of the step 24 KB
    // the output tile staged for its stores: bf16 storage 34 KB / 17 KB; fp32: half the block's rows, 132 floats apart (33 KB / 16.5 KB)
    constexpr int STAGE_PIECES = BIO ? 32 * RT * (LS_BN + 8) * 2 / 16 : 16 * RT * (LS_BN + 4) * 4 / 16;
    constexpr int R_PIECES = MODE == 1 ? 32 * RT / 4 : 0;                   // PROD: 2^-e_r of the block's rows
    constexpr int AB_PIECES = (LS_DOUBLE_BUFFER && NP == 2 && !BIO ? 2 : 1) * (A_PIECES + B_PIECES);
    constexpr int T_PIECES = AB_PIECES > STAGE_PIECES ? AB_PIECES : STAGE_PIECES;
    __shared__ __attribute__((aligned(16))) u32x4 Sm[T_PIECES + R_PIECES];
    u32x4* As = Sm;
    u32x4* Bs = Sm + A_PIECES;
    float* Rs = reinterpret_cast<float*>(Sm + T_PIECES);
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so id % 8 is the XCD; give each XCD a contiguous run of
    // logical tiles (column blocks of a row block are consecutive): the X rows are fetched into that XCD's L2 once
    int bid = blockIdx.x;
    if (nblocks % 8 == 0) bid = (bid & 7) * (nblocks >> 3) + (bid >> 3);
    const int mb = bid / nbn, nb = bid - mb * nbn;
    if (N - nb * LS_BN > 64) ls_tile<ACT, XV, 4, NP, RT, BIO, MODE>(As, Bs, X, ldx, Wp, bias, Y, ldy, M, N, K, ksteps, mb, nb, nbn, prod, Rs);
    else ls_tile<ACT, XV, 2, NP, RT, BIO, MODE>(As, Bs, X, ldx, Wp, bias, Y, ldy, M, N, K, ksteps, mb, nb, nbn, prod, Rs);
}

}  // namespace

#ifdef DFOL_DENSE_TRACE
extern "C" int dfol_dense_trace_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dfol_dense_trace_buf), sizeof(dfol_dense_trace_buf)); }
#endif

static int ls_pack_w(const float* W, int64_t ldw, int32_t N, int32_t K, void* W_split, void* stream, int np, const char* name) {
    DFOL_REQUIRE(W && W_split && N > 0 && K > 0 && ldw >= K, "%s: bad arguments N=%d K=%d", name, N, K);
    DFOL_REQUIRE((uintptr_t)W_split % 16 == 0, "%s: output must be 16-byte aligned", name);
    const int ksteps = dfol_cdiv(K, LS_BK), nbn = dfol_cdiv(N, LS_BN);
    const int64_t total = (int64_t)nbn * ksteps * np * LS_BN * 4;
    float* tail = reinterpret_cast<float*>(reinterpret_cast<u32x4*>(W_split) + total);      // np = 2: 2^-e_n and e_n per padded row
    if (np == 2)
        hipLaunchKernelGGL(linear_row_scale_kernel, dim3(nbn * LS_BN / 4), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, nbn * LS_BN, tail);
    hipLaunchKernelGGL(linear_pack_w_split_kernel, dim3((unsigned)dfol_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, ksteps,
                       nbn, np, (const float*)tail, (u32x4*)W_split);
    DFOL_LAUNCH_CHECK(name);
    return 0;
}

extern "C" int dfol_linear_pack_w_bf16x3(const float* W, int64_t ldw, int32_t N, int32_t K, void* W_split, void* stream) {
    return ls_pack_w(W, ldw, N, K, W_split, stream, 3, "linear_pack_w_bf16x3");
}
extern "C" int dfol_linear_pack_w_bf16(const float* W, int64_t ldw, int32_t N, int32_t K, void* W_bf16, void* stream) {
    return ls_pack_w(W, ldw, N, K, W_bf16, stream, 1, "linear_pack_w_bf16");
}
extern "C" int64_t dfol_linear_w_f16x2_bytes(int32_t N, int32_t K) {
    const int64_t nbn = dfol_cdiv(N, LS_BN);
    return nbn * dfol_cdiv(K, LS_BK) * 2 * LS_BN * 4 * 16 + nbn * LS_BN * 8;
}
extern "C" int dfol_linear_pack_w_f16x2(const float* W, int64_t ldw, int32_t N, int32_t K, void* W_split, void* stream) {
    return ls_pack_w(W, ldw, N, K, W_split, stream, 2, "linear_pack_w_f16x2");
}

template <int NP>
static int ls_launch_rows(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy, int32_t M, int32_t N, int32_t K,
                          int32_t act, void* stream, bool small);

template <int NP>
static int ls_launch(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy, int32_t M, int32_t N, int32_t K,
                     int32_t act, void* stream) {
    DFOL_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 4 == 0 && ldx % 2 == 0 && ldx >= K && ldy >= N, "linear_act_split: bad sizes M=%d N=%d K=%d (K %% 4, ldx %% 2)", M, N, K);
    if (M == 0) return 0;
    DFOL_REQUIRE(X && W_split && Y, "linear_act_split: null pointer");
    DFOL_REQUIRE(((uintptr_t)X % 8 == 0) && ((uintptr_t)W_split % 16 == 0), "linear_act_split: X must be 8-byte and W_split 16-byte aligned");
    if (NP == 2 && dfol_linear_wide_supported(M, N, K))            // wide outputs: X fetched and split once (csrc/dfol_dense_wide.hip), same bits
        return dfol_linear_wide_h2_f32(X, ldx, W_split, bias, Y, ldy, M, N, K, act, stream);
    const int nbn = dfol_cdiv(N, LS_BN);
    // 64-row blocks when 128-row blocks would not even give every CU two workgroups (the block height changes no result bit, see ls_tile);
    // DFOL_DENSE_BM=128 / 64 forces one for A/B runs
    static const int force_bm = getenv("DFOL_DENSE_BM") ? atoi(getenv("DFOL_DENSE_BM")) : 0;
    const int64_t nb128 = (int64_t)dfol_cdiv(M, LS_BM) * nbn;
    // (Mixing heights inside a product of 1.5 rounds - 128-row blocks for the full round, 64-row blocks for the rest, as two launches -
    // was measured at 256 x 100 objects: 0.478 ms against 0.475 ms for the four dense layers; not kept.)
    return ls_launch_rows<NP>(X, ldx, W_split, bias, Y, ldy, M, N, K, act, stream, force_bm ? force_bm == 64 : nb128 < 512);
}

template <int NP>
static int ls_launch_rows(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy, int32_t M, int32_t N, int32_t K,
                          int32_t act, void* stream, bool small) {
    const bool x16 = (uintptr_t)X % 16 == 0 && ldx % 4 == 0;
    const int ksteps = dfol_cdiv(K, LS_BK), nbn = dfol_cdiv(N, LS_BN);
    const int nbm = dfol_cdiv(M, small ? 64 : LS_BM);
    DFOL_REQUIRE((int64_t)nbm * nbn < ((int64_t)1 << 31), "linear_act_split: too many tiles");
    const int nblocks = nbm * nbn;
    LsProducer plain = LsProducer();
    plain.status = NP == 2 ? dfol_range_status_ptr() : nullptr;
#define DFOL_LS_K(A, XVV, RTT)                                                                                                              \
    hipLaunchKernelGGL((linear_act_split_kernel<A, XVV, NP, RTT>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, X, ldx, (const u32x4*)W_split, \
                       bias, Y, ldy, M, N, K, ksteps, nbn, nblocks, plain)
#define DFOL_LS(A)                                                                                                                          \
    if (x16) { if (small) DFOL_LS_K(A, 4, 2); else DFOL_LS_K(A, 4, 4); }                                                                    \
    else { if (small) DFOL_LS_K(A, 2, 2); else DFOL_LS_K(A, 2, 4); }
    switch (act) {
        case DFOL_ACT_NONE: DFOL_LS(DFOL_ACT_NONE); break;
        case DFOL_ACT_SIGMOID: DFOL_LS(DFOL_ACT_SIGMOID); break;
        case DFOL_ACT_ELU: DFOL_LS(DFOL_ACT_ELU); break;
        case DFOL_ACT_LOGSIGMOID: DFOL_LS(DFOL_ACT_LOGSIGMOID); break;
        default: DFOL_REQUIRE(false, "linear_act_split: unknown activation %d", act);
    }
#undef DFOL_LS
#undef DFOL_LS_K
    DFOL_LAUNCH_CHECK("linear_act_split");
    return 0;
}

extern "C" int dfol_linear_act_split_f32(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy,
                                         int32_t M, int32_t N, int32_t K, int32_t act, void* stream) {
    return ls_launch<3>(X, ldx, W_split, bias, Y, ldy, M, N, K, act, stream);
}
extern "C" int dfol_linear_act_h2_f32(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy,
                                      int32_t M, int32_t N, int32_t K, int32_t act, void* stream) {
    return ls_launch<2>(X, ldx, W_split, bias, Y, ldy, M, N, K, act, stream);
}
extern "C" int dfol_linear_act_bf16_f32(const float* X, int64_t ldx, const void* W_bf16, const float* bias, float* Y, int64_t ldy,
                                        int32_t M, int32_t N, int32_t K, int32_t act, void* stream) {
    return ls_launch<1>(X, ldx, W_bf16, bias, Y, ldy, M, N, K, act, stream);
}

// bf16 storage on both sides (the bf16 mode's per-pair activations: Z -> pre2 and dpre2 -> dZ): X [M, K] and Y [M, N] bfloat16, rows 8-byte
// aligned (ldx, ldy in elements, multiples of 4), no activation other than the four the fp32-storage kernel has; fp32 accumulation and bias
extern "C" int dfol_linear_act_bf16_bf16(const void* X_bf16, int64_t ldx, const void* W_bf16, const float* bias, void* Y_bf16, int64_t ldy,
                                         int32_t M, int32_t N, int32_t K, int32_t act, void* stream) {
    DFOL_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 4 == 0 && ldx % 4 == 0 && ldx >= K && ldy >= N, "linear_act_bf16_bf16: bad sizes M=%d N=%d K=%d (K %% 4, ldx %% 4)", M, N, K);
    if (M == 0) return 0;
    DFOL_REQUIRE(X_bf16 && W_bf16 && Y_bf16, "linear_act_bf16_bf16: null pointer");
    DFOL_REQUIRE(((uintptr_t)X_bf16 % 8 == 0) && ((uintptr_t)W_bf16 % 16 == 0), "linear_act_bf16_bf16: X must be 8-byte and W 16-byte aligned");
    // the output tile leaves as 8-byte pieces of four columns: a piece must be wholly inside a row, and 8-byte aligned
    DFOL_REQUIRE(N % 4 == 0 && ldy % 4 == 0 && (uintptr_t)Y_bf16 % 8 == 0, "linear_act_bf16_bf16: N=%d and ldy=%lld must be multiples of 4 and Y 8-byte aligned", N, (long long)ldy);
    const int ksteps = dfol_cdiv(K, LS_BK), nbn = dfol_cdiv(N, LS_BN);
    static const int force_bm = getenv("DFOL_DENSE_BM") ? atoi(getenv("DFOL_DENSE_BM")) : 0;
    const bool small = force_bm ? force_bm == 64 : (int64_t)dfol_cdiv(M, LS_BM) * nbn < 512;
    const int nbm = dfol_cdiv(M, small ? 64 : LS_BM);
    DFOL_REQUIRE((int64_t)nbm * nbn < ((int64_t)1 << 31), "linear_act_bf16_bf16: too many tiles");
    const int nblocks = nbm * nbn;
#define DFOL_LSB(A)                                                                                                                             \
    if (small) hipLaunchKernelGGL((linear_act_split_kernel<A, 4, 1, 2, true>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, X_bf16, ldx,  \
                                  (const u32x4*)W_bf16, bias, Y_bf16, ldy, M, N, K, ksteps, nbn, nblocks, LsProducer());                        \
    else hipLaunchKernelGGL((linear_act_split_kernel<A, 4, 1, 4, true>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, X_bf16, ldx,        \
                            (const u32x4*)W_bf16, bias, Y_bf16, ldy, M, N, K, ksteps, nbn, nblocks, LsProducer())
    switch (act) {
        case DFOL_ACT_NONE: DFOL_LSB(DFOL_ACT_NONE); break;
        case DFOL_ACT_SIGMOID: DFOL_LSB(DFOL_ACT_SIGMOID); break;
        case DFOL_ACT_ELU: DFOL_LSB(DFOL_ACT_ELU); break;
        case DFOL_ACT_LOGSIGMOID: DFOL_LSB(DFOL_ACT_LOGSIGMOID); break;
        default: DFOL_REQUIRE(false, "linear_act_bf16_bf16: unknown activation %d", act);
    }
#undef DFOL_LSB
    DFOL_LAUNCH_CHECK("linear_act_bf16_bf16");
    return 0;
}

// dZ (+)= dpre2 W2 with dpre2 produced inside the kernel from pre2, dx, the rows' predicates and their embedding rows (see LsProducer):
// W2t_split = dfol_linear_pack_w_f16x2 of W2^T [H1, H2]; pre2 [M, H2] and E [P, H2] with 16-byte aligned rows (H2 % 4 == 0).
extern "C" int dfol_pair_dz_fused_f32(const float* pre2, int64_t ld_p2, const float* dx, const int32_t* row_pred, const float* E, int64_t ld_e,
                                      const float* emax, const void* W2t_split, float* dZ, int64_t ld_dz, int32_t M, int32_t H1, int32_t H2,
                                      int32_t accumulate, void* stream) {
    DFOL_REQUIRE(M >= 0 && H1 > 0 && H2 > 0 && H2 % 4 == 0 && ld_p2 % 4 == 0 && ld_p2 >= H2 && ld_e % 4 == 0 && ld_e >= H2 && ld_dz >= H1,
                 "pair_dz_fused: bad sizes M=%d H1=%d H2=%d (H2, row strides: multiples of 4)", M, H1, H2);
    if (M == 0) return 0;
    DFOL_REQUIRE(pre2 && dx && row_pred && E && emax && W2t_split && dZ, "pair_dz_fused: null pointer");
    DFOL_REQUIRE(((uintptr_t)pre2 % 16 == 0) && ((uintptr_t)E % 16 == 0) && ((uintptr_t)W2t_split % 16 == 0), "pair_dz_fused: pre2, E and the packed weights must be 16-byte aligned");
    const int N = H1, K = H2, ksteps = dfol_cdiv(K, LS_BK), nbn = dfol_cdiv(N, LS_BN);
    static const int force_bm = getenv("DFOL_DENSE_BM") ? atoi(getenv("DFOL_DENSE_BM")) : 0;
    // (the producer's temporaries on top of a 128-row block's registers spill - 332 bytes of scratch per lane - and the 128-row blocks are
    // still the faster ones: 2.15 ms against 2.43 ms at 256 x 100 objects, the weight tile being fetched half as often)
    const bool small = force_bm ? force_bm == 64 : (int64_t)dfol_cdiv(M, LS_BM) * nbn < 512;
    const int nbm = dfol_cdiv(M, small ? 64 : LS_BM);
    DFOL_REQUIRE((int64_t)nbm * nbn < ((int64_t)1 << 31), "pair_dz_fused: too many tiles");
    const int nblocks = nbm * nbn;
    const LsProducer prod = {dx, row_pred, E, ld_e, emax, accumulate, nullptr, 0};
    if (small)
        hipLaunchKernelGGL((linear_act_split_kernel<DFOL_ACT_NONE, 4, 2, 2, false, 1>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, pre2, ld_p2,
                           (const u32x4*)W2t_split, (const float*)nullptr, dZ, ld_dz, M, N, K, ksteps, nbn, nblocks, prod);
    else
        hipLaunchKernelGGL((linear_act_split_kernel<DFOL_ACT_NONE, 4, 2, 4, false, 1>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, pre2, ld_p2,
                           (const u32x4*)W2t_split, (const float*)nullptr, dZ, ld_dz, M, N, K, ksteps, nbn, nblocks, prod);
    DFOL_LAUNCH_CHECK("pair_dz_fused");
    return 0;
}

// Y = X W^T + b on two fp16 pieces as dfol_linear_act_h2_f32 (no activation), and the logit layer's forward from the same pass:
// x_part[s][r], s < 2 ceil(N / 128), = the sum over the s-th 64-column half block of Sigmoid(Y[r][j]) E[row_pred[r]][j] (row_pred < 0: 0).
// row_pred: NON-DECREASING over the rows (a predicate owns consecutive rows; -1 only before the first predicate's rows).
// The caller adds the slots of a row (and the predicate's bias).  X 16-byte aligned rows; E [P, N] rows 16-byte aligned.
extern "C" int dfol_linear_logit_h2_f32(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy, int32_t M,
                                        int32_t N, int32_t K, const int32_t* row_pred, const float* E, int64_t ld_e, float* x_part, int64_t ld_xp,
                                        void* stream) {
    DFOL_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 4 == 0 && ldx % 4 == 0 && ldx >= K && ldy >= N && ld_e % 4 == 0 && ld_e >= N && ld_xp >= M,
                 "linear_logit_h2: bad sizes M=%d N=%d K=%d (K, ldx, ld_e multiples of 4)", M, N, K);
    if (M == 0) return 0;
    DFOL_REQUIRE(X && W_split && Y && row_pred && E && x_part, "linear_logit_h2: null pointer");
    DFOL_REQUIRE(((uintptr_t)X % 16 == 0) && ((uintptr_t)W_split % 16 == 0) && ((uintptr_t)E % 16 == 0), "linear_logit_h2: X, W_split and E must be 16-byte aligned");
    const int ksteps = dfol_cdiv(K, LS_BK), nbn = dfol_cdiv(N, LS_BN);
    static const int force_bm = getenv("DFOL_DENSE_BM") ? atoi(getenv("DFOL_DENSE_BM")) : 0;
    const bool small = force_bm ? force_bm == 64 : (int64_t)dfol_cdiv(M, LS_BM) * nbn < 512;
    const int nbm = dfol_cdiv(M, small ? 64 : LS_BM);
    DFOL_REQUIRE((int64_t)nbm * nbn < ((int64_t)1 << 31), "linear_logit_h2: too many tiles");
    const int nblocks = nbm * nbn;
    const LsProducer lg = {nullptr, row_pred, E, ld_e, nullptr, 0, x_part, ld_xp, dfol_range_status_ptr()};
    if (small)
        hipLaunchKernelGGL((linear_act_split_kernel<DFOL_ACT_NONE, 4, 2, 2, false, 2>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, X, ldx,
                           (const u32x4*)W_split, bias, Y, ldy, M, N, K, ksteps, nbn, nblocks, lg);
    else
        hipLaunchKernelGGL((linear_act_split_kernel<DFOL_ACT_NONE, 4, 2, 4, false, 2>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, X, ldx,
                           (const u32x4*)W_split, bias, Y, ldy, M, N, K, ksteps, nbn, nblocks, lg);
    DFOL_LAUNCH_CHECK("linear_logit_h2");
    return 0;
}
