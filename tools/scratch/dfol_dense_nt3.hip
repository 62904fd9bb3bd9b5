// Y = act(X W^T + b) with fp32 results from the bf16 matrix pipe, registers only: no LDS, no barriers.
//
// Same arithmetic as csrc/dfol_dense_split.hip (exact three-way operand split x = h + m + l, six of the nine piece products,
// fp32 accumulation) on v_mfma_f32_32x32x16_bf16, in the shape csrc/dfol_dense_wgrad.hip found for the weight gradient: ONE
// wavefront per SIMD owning 32 TM rows x 128 columns of Y (TM x 4 accumulator tiles of 32 x 32: 256 accumulator registers at
// TM = 4), operands straight from global memory into the MFMA operand registers:
//   A (X rows): the MFMA wants, per lane, eight consecutive k of one row (row = lane & 31, k = 8 (lane >> 5) + 0..7): two 16-byte
//     loads of the lane's own row, split into the three bf16 pieces by ~5.5 VALU instructions per element, in registers;
//   B (weights): split and packed once per weight version in exactly the operand order ([column block][k block of 16][piece]
//     [column tile][lane] x 16 bytes), so a fragment is one coalesced 1 KB load per wavefront - from L2 (the image is < 1 MB) -
//     and needs no arithmetic.
// A step of 16 k: split the TM row tiles that arrived (VALU, 704 cycles at TM = 4), request the next step's X rows and weight
// fragments, then 24 TM MFMAs back to back (3072 cycles at TM = 4) under which they arrive.  Addresses: buffer descriptors rebuilt
// per step from wave-uniform values, constant per-lane offsets; rows past M read as zero and are not stored.
// The four wavefronts of a workgroup take four consecutive tasks - the column blocks of one row block (TM = 4: X is fetched from HBM
// once and hits in L1 for the other column blocks) or, for the short-row variants, four row blocks of one column block (they share
// the weight fragments instead).  TM = 4, 2, 1 give bit-identical results (same k order per output element), so the launcher picks
// TM by M without breaking run-to-run or world-size-to-world-size equality.
#include "dfol_common.h"

#include <algorithm>
#include <type_traits>

// -DDFOL_NT3_TRACE: s_memtime stamps of the first tasks of one wavefront (tools/scratch/nt3_lab.sh)
#ifdef DFOL_NT3_TRACE
__device__ long long dfol_nt3_trace_buf[64];
__device__ int dfol_nt3_trace_pos;
#define NTRACE() do { if (blockIdx.x == 7 && threadIdx.x == 0 && dfol_nt3_trace_pos < 64) dfol_nt3_trace_buf[dfol_nt3_trace_pos++] = __builtin_readcyclecounter(); } while (0)
#else
#define NTRACE()
#endif

namespace {

typedef float n3_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 n3_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t n3_u32x4 __attribute__((ext_vector_type(4)));

constexpr int N3_BN = 128;                        // columns of a column block (four tiles of 32)
constexpr int N3_FRAGS = 3 * 4 * 64;              // 16-byte fragments of one (column block, k block): [piece][column tile][lane]

__device__ __forceinline__ void n3_split(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = __float_as_uint(x);
    const float r = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r);
    l = __float_as_uint(r - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ uint32_t n3_pack(uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }
__device__ __forceinline__ void n3_split8(const float (&v)[8], n3_u32x4& h, n3_u32x4& m, n3_u32x4& l) {
    uint32_t ph[8], pm[8], pl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) n3_split(v[j], ph[j], pm[j], pl[j]);
    h = n3_u32x4{n3_pack(ph[0], ph[1]), n3_pack(ph[2], ph[3]), n3_pack(ph[4], ph[5]), n3_pack(ph[6], ph[7])};
    m = n3_u32x4{n3_pack(pm[0], pm[1]), n3_pack(pm[2], pm[3]), n3_pack(pm[4], pm[5]), n3_pack(pm[6], pm[7])};
    l = n3_u32x4{n3_pack(pl[0], pl[1]), n3_pack(pl[2], pl[3]), n3_pack(pl[4], pl[5]), n3_pack(pl[6], pl[7])};
}

// One thread per 16-byte fragment.  transpose: W is [K, N] (the operand of the backward product dz @ W).
__global__ void nt3_pack_kernel(const float* __restrict__ W, int64_t ldw, int N, int K, int kblocks, int ncb, int transpose,
                                n3_u32x4* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)ncb * kblocks * N3_FRAGS) return;
    const int lane = (int)(idx & 63), u = (int)((idx >> 6) & 3);
    int64_t rest = idx >> 8;
    const int p = (int)(rest % 3);
    rest /= 3;
    const int kb = (int)(rest % kblocks), cb = (int)(rest / kblocks);
    const int n = cb * N3_BN + 32 * u + (lane & 31), k0 = kb * 16 + 8 * (lane >> 5);
    uint32_t piece[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        const float w = (n < N && k < K) ? (transpose ? W[(int64_t)k * ldw + n] : W[(int64_t)n * ldw + k]) : 0.f;
        uint32_t h, m, l;
        n3_split(w, h, m, l);
        piece[e] = p == 0 ? h : (p == 1 ? m : l);
    }
    out[idx] = n3_u32x4{n3_pack(piece[0], piece[1]), n3_pack(piece[2], piece[3]), n3_pack(piece[4], piece[5]), n3_pack(piece[6], piece[7])};
}

// (every input through readfirstlane: hipcc wraps each buffer access in a waterfall loop unless the descriptor is provably uniform)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t n3_descriptor(const void* base, int64_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes < 0 ? 0 : (bytes < 0x7fffffff ? bytes : 0x7fffffff)));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

template <int ACT>
__device__ __forceinline__ float n3_act(float x) {
    // branch-free forms on the hardware exp / log / rcp (as csrc/dfol_dense_split.hip)
    if (ACT == DFOL_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + dfol_exp(-x));
    if (ACT == DFOL_ACT_ELU) return fmaxf(x, dfol_exp(fminf(x, 0.f)) - 1.0f);
    if (ACT == DFOL_ACT_LOGSIGMOID) return fminf(x, 0.f) - dfol_log(1.0f + dfol_exp(-fabsf(x)));
    return x;
}

// TM row tiles x UC column tiles of one task: rows m0 .., columns n0 ..
template <int TM, int UC>
__device__ __forceinline__ void nt3_task(const float* __restrict__ X, int64_t ldx, const n3_u32x4* __restrict__ wp, const float* __restrict__ bias,
                                         float* __restrict__ Y, int64_t ldy, int M, int N, int K, int kblocks, int m0, int n0, int act, int lane) {
    const int col = lane & 31, half = lane >> 5;
    n3_f32x16 acc[TM][UC];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < UC; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
    int va[TM];                                                                    // byte offset of this lane's eight k within a step, per row tile
#pragma unroll
    for (int t = 0; t < TM; ++t) va[t] = (int)(((32 * t + col) * ldx + 8 * half) * 4);
    const float* xrow = X + (int64_t)m0 * ldx;                                     // wave-uniform
    const int64_t xbytes = ((int64_t)(M - m0 - 1) * ldx + K) * 4;                  // from xrow to the end of the last row's K columns

    n3_u32x4 ra[TM][2];                                                            // the step's X: dead once split, then refilled with the next step's
    n3_u32x4 bp[2][UC][3];                                                         // weight fragments: this step's and the next one's
    auto load_a = [&](int kb) __attribute__((always_inline)) {
        const auto d = n3_descriptor(xrow + kb * 16, xbytes - (int64_t)kb * 64);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            ra[t][0] = __builtin_amdgcn_raw_buffer_load_b128(d, va[t], 0, 0);
            ra[t][1] = __builtin_amdgcn_raw_buffer_load_b128(d, va[t] + 16, 0, 0);
        }
    };
    auto load_b = [&](int kb, auto set_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        const n3_u32x4* w = wp + (int64_t)kb * N3_FRAGS + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int u = 0; u < UC; ++u) bp[S][u][p] = w[(p * 4 + u) * 64];
    };
    constexpr int PA6[6] = {2, 0, 1, 1, 0, 0}, PB6[6] = {0, 2, 1, 0, 1, 0};       // (A piece, B piece): l*h, h*l, m*m, m*h, h*m, h*h
    // (has_next is a compile-time flag: a conditional load in the steady-state loop would make the compiler's wait counts conservative)
    auto step = [&](int kb, auto set_tag, auto next_tag, bool tail, auto has_next) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        n3_u32x4 ap[TM][3];
        const int kleft = K - kb * 16 - 8 * half;                                  // k of this lane's eight that exist (last k block only)
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = __uint_as_float(ra[t][e >> 2][e & 3]);
                if (tail) v[e] = e < kleft ? v[e] : 0.f;                           // (the k beyond K of a row are the next row's first columns)
            }
            n3_split8(v, ap[t][0], ap[t][1], ap[t][2]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (decltype(has_next)::value) {
            load_b(kb + 1, next_tag);
            load_a(kb + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UC; ++u)
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int x = 0; x < 6; ++x)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(n3_bf16x8, ap[t][PA6[x]]),
                                                                        __builtin_bit_cast(n3_bf16x8, bp[S][u][PB6[x]]), acc[t][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    const std::integral_constant<int, 0> S0;
    const std::integral_constant<int, 1> S1;
    NTRACE();
    load_b(0, S0);
    load_a(0);
    __builtin_amdgcn_sched_barrier(0);
    NTRACE();
    const bool ragged_k = (K & 15) != 0;
    int kb = 0;
    const std::true_type yes;
    const std::false_type no;
    for (; kb + 2 < kblocks; kb += 2) {                                            // (the last one or two k blocks are peeled: tail masking)
        step(kb, S0, S1, false, yes);
        step(kb + 1, S1, S0, false, yes);
    }
    NTRACE();
    if (kb + 2 == kblocks) {
        step(kb, S0, S1, false, yes);
        step(kb + 1, S1, S0, ragged_k, no);
    } else {
        step(kb, S0, S1, ragged_k, no);
    }

    NTRACE();
    // D tile: column j = lane & 31, row i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): a register of tile (t, u) is 32 consecutive floats
    // of two rows of Y.  Rows past M are dropped by the descriptor's bounds check, columns past N by the lane mask.  Offset = a scalar row
    // part + one per-lane part per column tile.
    const auto dy = n3_descriptor(Y + (int64_t)m0 * ldy, ((int64_t)(M - m0 - 1) * ldy + N) * 4);
    const int ldy4 = (int)ldy * 4;
    auto store = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int A = decltype(act_tag)::value;
#pragma unroll
        for (int u = 0; u < UC; ++u) {
            const int n = n0 + 32 * u + col;
            if (n >= N) continue;
            const float b = bias ? bias[n] : 0.f;
            const int lane_off = 4 * half * ldy4 + 4 * n;
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rowc = 32 * t + (i & 3) + 8 * (i >> 2);              // + 4 half: this lane's row of the tile
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(n3_act<A>(acc[t][u][i] + b)), dy, rowc * ldy4 + lane_off, 0, 0);
                }
        }
    };
    if (act == DFOL_ACT_SIGMOID) store(std::integral_constant<int, DFOL_ACT_SIGMOID>());
    else if (act == DFOL_ACT_ELU) store(std::integral_constant<int, DFOL_ACT_ELU>());
    else if (act == DFOL_ACT_LOGSIGMOID) store(std::integral_constant<int, DFOL_ACT_LOGSIGMOID>());
    else store(std::integral_constant<int, DFOL_ACT_NONE>());
    NTRACE();
}

// grid: ceil(tasks / 4) workgroups of four wavefronts; task -> (row block, column block), see the header.  (A persistent variant - one
// workgroup per CU walking through its tasks - was measured 10-15 % slower on the pair layer's products: its wavefronts stay in
// lockstep, so all of them load, and later all of them store, at the same time.)
template <int TM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void linear_act_nt3_kernel(
    const float* __restrict__ X, int64_t ldx, const n3_u32x4* __restrict__ Wp, const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy,
    int M, int N, int K, int kblocks, int ncb, int nrb, int act) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= (int64_t)nrb * ncb) return;
    int rb, cb;
    if (TM == 4) rb = (int)(task / ncb), cb = (int)(task - (int64_t)rb * ncb);     // a workgroup: the column blocks of a row block
    else cb = (int)(task / nrb), rb = (int)(task - (int64_t)cb * nrb);             // a workgroup: four row blocks of a column block
    const int m0 = rb * 32 * TM, n0 = cb * N3_BN;
    const n3_u32x4* wp = Wp + (int64_t)cb * kblocks * N3_FRAGS;
    if (N - n0 > 64) nt3_task<TM, 4>(X, ldx, wp, bias, Y, ldy, M, N, K, kblocks, m0, n0, act, lane);
    else nt3_task<TM, 2>(X, ldx, wp, bias, Y, ldy, M, N, K, kblocks, m0, n0, act, lane);
}

}  // namespace

#ifdef DFOL_NT3_TRACE
extern "C" int dfol_nt3_trace_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dfol_nt3_trace_buf), sizeof(dfol_nt3_trace_buf)); }
#endif

extern "C" int64_t dfol_linear_pack_w_nt3_bytes(int32_t N, int32_t K) {
    if (N <= 0 || K <= 0) return 0;
    return (int64_t)dfol_cdiv(N, N3_BN) * dfol_cdiv(K, 16) * N3_FRAGS * 16;
}

extern "C" int dfol_linear_pack_w_nt3(const float* W, int64_t ldw, int32_t N, int32_t K, int32_t transpose, void* W_packed, void* stream) {
    DFOL_REQUIRE(W && W_packed && N > 0 && K > 0 && ldw >= (transpose ? N : K), "linear_pack_w_nt3: bad arguments N=%d K=%d", N, K);
    DFOL_REQUIRE((uintptr_t)W_packed % 16 == 0, "linear_pack_w_nt3: output must be 16-byte aligned");
    const int kblocks = dfol_cdiv(K, 16), ncb = dfol_cdiv(N, N3_BN);
    const int64_t total = (int64_t)ncb * kblocks * N3_FRAGS;
    hipLaunchKernelGGL(nt3_pack_kernel, dim3((unsigned)dfol_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, kblocks, ncb,
                       transpose, (n3_u32x4*)W_packed);
    DFOL_LAUNCH_CHECK("linear_pack_w_nt3");
    return 0;
}

extern "C" int dfol_linear_act_nt3_f32(const float* X, int64_t ldx, const void* W_packed, const float* bias, float* Y, int64_t ldy, int32_t M,
                                       int32_t N, int32_t K, int32_t act, void* stream) {
    DFOL_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K && ldy >= N, "linear_act_nt3: bad sizes M=%d N=%d K=%d", M, N, K);
    DFOL_REQUIRE(act >= DFOL_ACT_NONE && act <= DFOL_ACT_LOGSIGMOID, "linear_act_nt3: unknown activation %d", act);
    if (M == 0) return 0;
    DFOL_REQUIRE(X && W_packed && Y, "linear_act_nt3: null pointer");
    DFOL_REQUIRE((uintptr_t)W_packed % 16 == 0 && (uintptr_t)X % 4 == 0, "linear_act_nt3: W_packed must be 16-byte aligned");
    DFOL_REQUIRE(128 * std::max(ldx, ldy) * 4 < (1ll << 31), "linear_act_nt3: row stride too large (%lld)", (long long)std::max(ldx, ldy));
    const int kblocks = dfol_cdiv(K, 16), ncb = dfol_cdiv(N, N3_BN);
    // rows per task: the largest of 128, 64, 32 that still gives every SIMD of the chip (1024) most of a task
    const int tm = (int64_t)dfol_cdiv(M, 128) * ncb >= 768 ? 4 : ((int64_t)dfol_cdiv(M, 64) * ncb >= 768 ? 2 : 1);
    const int nrb = dfol_cdiv(M, 32 * tm);
    const int64_t tasks = (int64_t)nrb * ncb;
    DFOL_REQUIRE(tasks < ((int64_t)1 << 32), "linear_act_nt3: too many tasks");
    const dim3 grid((unsigned)dfol_cdiv(tasks, 4));
    hipStream_t st = (hipStream_t)stream;
    if (tm == 4)
        hipLaunchKernelGGL((linear_act_nt3_kernel<4>), grid, dim3(256), 0, st, X, ldx, (const n3_u32x4*)W_packed, bias, Y, ldy, M, N, K, kblocks, ncb,
                           nrb, act);
    else if (tm == 2)
        hipLaunchKernelGGL((linear_act_nt3_kernel<2>), grid, dim3(256), 0, st, X, ldx, (const n3_u32x4*)W_packed, bias, Y, ldy, M, N, K, kblocks, ncb,
                           nrb, act);
    else
        hipLaunchKernelGGL((linear_act_nt3_kernel<1>), grid, dim3(256), 0, st, X, ldx, (const n3_u32x4*)W_packed, bias, Y, ldy, M, N, K, kblocks, ncb,
                           nrb, act);
    DFOL_LAUNCH_CHECK("linear_act_nt3");
    return 0;
}
