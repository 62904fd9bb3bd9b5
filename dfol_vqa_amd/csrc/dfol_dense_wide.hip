// The WIDE forward products of a batch (the featurizer, batch_gqa_boxfeatures_pipeline.py:199-281: [objects, 2048] x [2048, 512] + Sigmoid;
// the stacked first layer of the pair MLP, classifier_oracle.py:145-156 split per object: [objects, 516] x [516, 512]) as ONE persistent
// workgroup per CU that owns 128 rows x ALL columns (256 < N <= 512), gfx950.
//
// csrc/dfol_dense_split.hip runs these shapes as 128 x 128 output tiles: with four column blocks every row of X is fetched and split into
// its fp16 pieces FOUR times (three of the fetches are cache hits: the counters show 302 MB from memory for the 210 MB feature matrix of a
// featurizer launch; this kernel: 247 MB, profiles/r05_pmc_fetch_*.txt), and every tile pays a cold start.
// Here a workgroup of eight wavefronts reads and splits its X rows once per k-step and multiplies them with the weight tiles of all four
// column blocks, staged through LDS as two column HALVES per k-step (2 x 32 KB; all four would not fit beside two A buffers):
//
//   half 0 of step g:  multiply A[g] x B(g, columns 0..255)    | store B(g, columns 256..511), request B(g + 1, columns 0..255)
//   half 1 of step g:  multiply A[g] x B(g, columns 256..511)  | store B(g + 1, columns 0..255), split and store A[g + 1], request X rows 3 steps on
//
// one barrier per half.  Wavefront (wm, wn) owns rows 64 wm .. and, in each half, columns 64 wn ..: 4 x 8 accumulator tiles of 16 x 16.
// The step stream runs on across row blocks (the X rows of the next block's first steps are in flight under the last steps and the stores
// of the current one), as in csrc/dfol_dense_tall.hip.
//
// Same arithmetic as dfol_linear_act_h2_f32 (two fp16 pieces per operand, three products, the packed image of dfol_linear_pack_w_f16x2, the
// same order of the products, the same epilogue): results are bit for bit those of csrc/dfol_dense_split.hip
// (tests/test_kernels_gpu.py::test_linear_wide_equals_the_tiled_kernel_bit_for_bit).
#include "dfol_common.h"

#include <stdlib.h>

#include <algorithm>
#include <type_traits>

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int WD_BM = 128, WD_BK = 32, WD_XD = 3;                      // rows of a block, k of a step, steps of X rows in flight (a step is ~1.3 us)
constexpr int WD_TILE = 2 * 128 * 4;                                   // 16-byte pieces of one [2 pieces][128 rows][4 k-groups] tile (16 KB)
constexpr int WD_NMAX = 512;
constexpr size_t WD_LDS = (size_t)6 * WD_TILE * 16 + 2 * WD_NMAX * 4 + 8 * 4096;  // A0, A1, B0 (two column blocks), B1; row scales and bias; the epilogue's 4 KB per wavefront

__device__ __forceinline__ int wd_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }
__device__ __forceinline__ void wd_split2h(float x0, float x1, uint32_t& h, uint32_t& l) {
    const f32x2 x = {x0, x1};
    const f16x2 hh = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(hh, f32x2);
    h = __builtin_bit_cast(uint32_t, hh);
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}
__device__ __forceinline__ void wd_split8h(const float4& a, const float4& b, u32x4& h, u32x4& l) {
    uint32_t hh[4], ll[4];
    wd_split2h(a.x, a.y, hh[0], ll[0]);
    wd_split2h(a.z, a.w, hh[1], ll[1]);
    wd_split2h(b.x, b.y, hh[2], ll[2]);
    wd_split2h(b.z, b.w, hh[3], ll[3]);
    h = u32x4{hh[0], hh[1], hh[2], hh[3]};
    l = u32x4{ll[0], ll[1], ll[2], ll[3]};
}

template <int ACT>
__device__ __forceinline__ float wd_act(float x) {                     // (the forms of csrc/dfol_dense_split.hip: ls_act)
    if (ACT == DFOL_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + dfol_exp(-x));
    if (ACT == DFOL_ACT_ELU) return fmaxf(x, dfol_exp(fminf(x, 0.f)) - 1.0f);
    if (ACT == DFOL_ACT_LOGSIGMOID) return fminf(x, 0.f) - dfol_log(1.0f + dfol_exp(-fabsf(x)));
    return x;
}

// XV: floats per X load (4: rows 16-byte aligned; 2: rows 8-byte aligned - the 2054-column raw feature matrix)
template <int ACT, int XV>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wide_h2_kernel(
    const float* __restrict__ X, int64_t ldx, const u32x4* __restrict__ Wp, const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy, int M,
    int N, int K, int ksteps, int nbn, uint32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) u32x4 wd_sm[];
    constexpr int A0 = 0, B0 = 2 * WD_TILE;                            // A buffer b at b * WD_TILE; B buffer b at B0 + b * 2 * WD_TILE
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kh = lane >> 4, r16 = lane & 15;
    const int wm = wave >> 2, wn = wave & 3;
    const int arow = tid >> 2, aq = tid & 3;                           // staging: one row of the block, 8 consecutive k per step
    const int nblocks = (M + WD_BM - 1) / WD_BM, stride = gridDim.x;
    const int first = blockIdx.x;
    if (first >= nblocks) return;
    const int nmine = (nblocks - first + stride - 1) / stride;
    const int T = nmine * ksteps;                                      // steps of this workgroup's stream

    floatx4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // ---- X rows: a ring of WD_XD register sets; the loader's position runs ahead of the multiplier's across block boundaries
    float4 xa[WD_XD][2];
    int lb = first, lks = 0;
    const float* xrow = X + (int64_t)min(lb * WD_BM + arow, M - 1) * ldx + aq * 8;
    auto load_x = [&](auto set_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        const int k = lks * WD_BK + aq * 8;
        const int c0 = min(k, K - 4) - aq * 8, c1 = min(k + 4, K - 4) - aq * 8;          // (K % 4 == 0; clamped, zeroed when used)
        auto ld4 = [&](const float* p) __attribute__((always_inline)) {
            if (XV == 4) return *reinterpret_cast<const float4*>(p);
            const float2 lo = *reinterpret_cast<const float2*>(p), hi = *reinterpret_cast<const float2*>(p + 2);
            return make_float4(lo.x, lo.y, hi.x, hi.y);
        };
        xa[S][0] = ld4(xrow + c0);
        xa[S][1] = ld4(xrow + c1);
        if (++lks == ksteps) {                                         // (uniform) on to the next block of this workgroup; past the last: clamped rows, never used
            lks = 0;
            lb += stride;
            xrow = X + (int64_t)min((int64_t)lb * WD_BM + arow, (int64_t)M - 1) * ldx + aq * 8;
        }
    };
    // ---- weight tiles: the two column blocks of a half, one half-step ahead in registers
    u32x4 wb[4];
    int wks = 0, wh = 0;
    auto load_w = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = i * 512 + tid, nb = 2 * wh + idx / WD_TILE, within = idx % WD_TILE;
            wb[i] = Wp[((int64_t)min(nb, nbn - 1) * ksteps + wks) * WD_TILE + within];
        }
        if (++wh == 2) {
            wh = 0;
            if (++wks == ksteps) wks = 0;
        }
    };
    auto store_b = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wd_sm[B0 + buf * 2 * WD_TILE + i * 512 + tid] = wb[i];
    };
    float xmax = 0.f;                                                  // the largest |x| this thread split (dfol_set_range_status)
    int sks = 0;
    auto store_a = [&](auto set_tag, int buf) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        const int k = sks * WD_BK + aq * 8;
        const int at = A0 + buf * WD_TILE + arow * 4 + (aq ^ wd_swz(arow));
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v0 = k < K ? xa[S][0] : z, v1 = k + 4 < K ? xa[S][1] : z;
        xmax = fmaxf(fmaxf(xmax, fabsf(v0.x)), fabsf(v0.y));
        xmax = fmaxf(fmaxf(xmax, fabsf(v0.z)), fabsf(v0.w));
        xmax = fmaxf(fmaxf(xmax, fabsf(v1.x)), fabsf(v1.y));
        xmax = fmaxf(fmaxf(xmax, fabsf(v1.z)), fabsf(v1.w));
        u32x4 ph, pl;
        wd_split8h(v0, v1, ph, pl);
        wd_sm[at] = ph;
        wd_sm[at + WD_BM * 4] = pl;
        if (++sks == ksteps) sks = 0;
    };

    // ---- a half-step's MFMAs: rows 64 wm .., columns 256 h + 64 wn .. (column block wn >> 1 of the half's two, rows 64 (wn & 1) .. of its tile)
    const int aoff = (wm * 64 + r16) * 4 + (kh ^ wd_swz(r16));
    const int boff = (wn >> 1) * WD_TILE + ((wn & 1) * 64 + r16) * 4 + (kh ^ wd_swz(r16));
    constexpr int PA3[3] = {1, 0, 0}, PB3[3] = {0, 1, 0};              // xl wh, xh wl, xh wh (smallest first)
    u32x4 afr[4][2];                                                   // the step's A fragments: read once, used by both halves
    auto load_a = [&](int abuf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) afr[i][p] = wd_sm[A0 + abuf * WD_TILE + aoff + p * WD_BM * 4 + i * 64];
    };
    auto multiply = [&](int bbuf, auto half_tag) __attribute__((always_inline)) {
        constexpr int JB = 4 * decltype(half_tag)::value;
        const int bo = B0 + bbuf * 2 * WD_TILE + boff;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 b[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) b[p] = wd_sm[bo + p * 128 * 4 + j * 64];
#pragma unroll
            for (int x = 0; x < 3; ++x)                                // (per accumulator the three products keep their order; its next MFMA is four issues on)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][JB + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, afr[i][PA3[x]]), __builtin_bit_cast(f16x8, b[PB3[x]]),
                                                                            acc[i][JB + j], 0, 0, 0);
        }
    };

    // ---- the weight rows' scales and the bias, once, behind the buffers
    float* xs = reinterpret_cast<float*>(wd_sm + 6 * WD_TILE);         // [0, 512) scales, [512, 1024) bias
    {
        const float* tail = reinterpret_cast<const float*>(Wp + (int64_t)nbn * ksteps * WD_TILE);
        xs[tid] = tid < nbn * 128 ? tail[tid] : 0.f;
        xs[WD_NMAX + tid] = (bias && tid < N) ? bias[tid] : 0.f;
    }
    // The epilogue goes through 4 KB of LDS per wavefront, one 16 x 64 piece of the block at a time: the accumulator layout (a lane holds four
    // ROWS of one column) becomes rows of 64 consecutive columns, so a store instruction covers 256 contiguous bytes, and the scale / bias /
    // activation arithmetic is a 16-trip loop instead of 128 unrolled copies (the loop body of the stream is inlined three times: with the
    // unrolled epilogue a kernel was 47 - 60 KB of code, next to a 64 KB instruction cache).  Only the owning wavefront touches its piece.
    float* stage = xs + 2 * WD_NMAX + wave * 1024;
    auto epilogue = [&](int b) __attribute__((always_inline)) {        // block b's accumulators -> Y, then cleared
        const int m0 = b * WD_BM + wm * 64;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = h * 256 + wn * 64 + lane;
            const float cs = xs[col], bv = xs[WD_NMAX + col];
            float* yp = Y + col;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        stage[(4 * kh + e) * 64 + j * 16 + r16] = acc[i][4 * h + j][e];
                        acc[i][4 * h + j][e] = 0.f;
                    }
#pragma unroll 4
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + i * 16 + r;
                    const float v = wd_act<ACT>(stage[r * 64 + lane] * cs + bv);          // (cs is a power of two: the product is exact, as in the tiled kernel)
                    if (row < M && col < N) yp[(int64_t)row * ldy] = v;
                }
            }
        }
    };

    // ---- the stream
    const std::integral_constant<int, 0> S0;
    const std::integral_constant<int, 1> S1;
    const std::integral_constant<int, 2> S2;
    load_x(S0);
    load_w();                                                          // (step 0, half 0)
    load_x(S1);
    load_x(S2);
    store_b(0);
    load_w();                                                          // (step 0, half 1)
    store_a(S0, 0);
    load_x(S0);
    __syncthreads();
    int g = 0, cks = 0, cb = first;                                    // the step being multiplied: index in the stream, step and block
    // (The SIMD's two wavefronts - one of each row half - taking the staging work and the multiply of a window in OPPOSITE orders, so that one
    // always has MFMAs to issue while the other one's vector / LDS / memory instructions go out, was built - one non-inlined function per
    // order, called once per wavefront - and measured at 325 us against 166 for either order alone: as in csrc/dfol_dense_tall.hip.)
    auto body = [&](auto next_tag) __attribute__((always_inline)) {    // next_tag: the X set of step g + 1
        const int cur = g & 1, nxt = cur ^ 1;
        store_b(1);                                                    // (step g, half 1: requested a half-step ago)
        load_w();                                                      // (step g + 1, half 0)
        load_a(cur);
        multiply(0, S0);
        __syncthreads();                                               // B1 complete, B0 fully read
        store_b(0);                                                    // (step g + 1, half 0)
        load_w();                                                      // (step g + 1, half 1)
        store_a(next_tag, nxt);                                        // frees the register set ...
        load_x(next_tag);                                              // ... for the rows three steps on
        multiply(1, S1);
        __syncthreads();                                               // A[nxt], B0 complete; A[cur], B1 fully read
        ++g;
        if (++cks == ksteps) {                                         // (uniform) the block is done
            epilogue(cb);
            cks = 0;
            cb += stride;
        }
    };
    while (g < T) {
        body(S1);
        if (g < T) body(S2);
        if (g < T) body(S0);
    }
    if (status != nullptr && !(xmax <= 65504.0f)) atomicOr(status, (uint32_t)DFOL_RANGE_X_OVERFLOW);
}

int wd_cus() {
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

}  // namespace

// The shapes the persistent wide kernel takes AND pays for: four column blocks (X would be split three or four times by the tiled
// kernel), at least four k-steps, and enough 128-row blocks to give three quarters of the CUs one (fewer: the tiled kernel's 64-row blocks
// fill the chip better).  Both kernels return the same bits, so the choice may depend on M.  DFOL_DENSE_WIDE=0 switches it off (A/B runs).
extern "C" int dfol_linear_wide_supported(int64_t M, int32_t N, int32_t K) {
    static const int on = getenv("DFOL_DENSE_WIDE") ? atoi(getenv("DFOL_DENSE_WIDE")) : 1;
    // (N <= 384: the kernel would spend a full second half's MFMAs on a few columns - the 256 -> 300 attribute layer took 34.8 us here
    // against 28 us tiled - so it leaves three-block outputs to the tiled kernel unless forced)
    if (!on || N <= 256 || N > WD_NMAX || K < 4 * WD_BK || K % 4 != 0 || M >= (1ll << 31) - 256) return 0;
    if (on != 2 && N <= 384) return 0;
    // (a block per CU and round: 25600 rows are 200 blocks = 0.78 of one round of 256 CUs; 40000 rows would be 313 blocks = 0.61 of two
    // rounds, and there the tiled kernel's 1252 smaller tiles balance better: 307 us against 339)
    const int64_t blocks = (M + WD_BM - 1) / WD_BM, rounds = (blocks + wd_cus() - 1) / wd_cus();
    return on == 2 || 4 * blocks >= 3 * rounds * wd_cus();
}

extern "C" int dfol_linear_wide_h2_f32(const float* X, int64_t ldx, const void* W_split, const float* bias, float* Y, int64_t ldy, int32_t M, int32_t N,
                                       int32_t K, int32_t act, void* stream) {
    DFOL_REQUIRE(M > 0 && N > 256 && N <= WD_NMAX && K >= 4 * WD_BK && K % 4 == 0 && ldx % 2 == 0 && ldx >= K && ldy >= N,
                 "linear_wide_h2: bad sizes M=%d N=%d K=%d (256 < N <= 512, K >= 128, K %% 4, ldx %% 2)", M, N, K);
    DFOL_REQUIRE(X && W_split && Y, "linear_wide_h2: null pointer");
    DFOL_REQUIRE(((uintptr_t)X % 8 == 0) && ((uintptr_t)W_split % 16 == 0), "linear_wide_h2: X must be 8-byte and W_split 16-byte aligned");
    const bool x16 = (uintptr_t)X % 16 == 0 && ldx % 4 == 0;
    const int ksteps = dfol_cdiv(K, WD_BK), nbn = dfol_cdiv(N, 128);
    const int grid = std::min(dfol_cdiv(M, WD_BM), wd_cus());
    uint32_t* status = dfol_range_status_ptr();
    hipStream_t st = (hipStream_t)stream;
#define DFOL_WD_K(A, XVV)                                                                                                                   \
    {                                                                                                                                      \
        static const hipError_t ok = hipFuncSetAttribute((const void*)wide_h2_kernel<A, XVV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WD_LDS); \
        DFOL_REQUIRE(ok == hipSuccess, "linear_wide_h2: cannot reserve %zu bytes of LDS (%s)", WD_LDS, hipGetErrorString(ok));              \
        hipLaunchKernelGGL((wide_h2_kernel<A, XVV>), dim3(grid), dim3(512), WD_LDS, st, X, ldx, (const u32x4*)W_split, bias, Y, ldy, M, N, K, ksteps, \
                           nbn, status);                                                                                                   \
    }
#define DFOL_WD(A)                                                                                                                          \
    if (x16) DFOL_WD_K(A, 4) else DFOL_WD_K(A, 2)
    switch (act) {
        case DFOL_ACT_NONE: DFOL_WD(DFOL_ACT_NONE); break;
        case DFOL_ACT_SIGMOID: DFOL_WD(DFOL_ACT_SIGMOID); break;
        case DFOL_ACT_ELU: DFOL_WD(DFOL_ACT_ELU); break;
        case DFOL_ACT_LOGSIGMOID: DFOL_WD(DFOL_ACT_LOGSIGMOID); break;
        default: DFOL_REQUIRE(false, "linear_wide_h2: unknown activation %d", act);
    }
#undef DFOL_WD
#undef DFOL_WD_K
    DFOL_LAUNCH_CHECK("linear_wide_h2");
    return 0;
}
