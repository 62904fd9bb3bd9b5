// Fused pair MLP with the second layer on the bf16 matrix pipes at fp32 accuracy (gfx950).
//
// The fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate (v_mfma_f32_16x16x32_bf16), and the pair MLP's
// second layer ([pairs, HID1] x [HID1, HID2], 0.4 TFLOP per 256 questions of 100 objects) is what bounds the full-size
// forward.  This kernel keeps fp32 RESULTS but feeds the bf16 pipes: every fp32 operand x is cut EXACTLY into three bf16
// pieces x = h + m + l (h = top 16 bits of x, m = top 16 bits of x - h, l = x - h - m: 8 + 8 + 8 mantissa bits, all
// subtractions exact), and a product a * w is accumulated (in fp32, by the MFMA) as the six piece products of order
// <= 2^-16:  al*wh + ah*wl + am*wm + am*wh + ah*wm + ah*wh.  The three dropped ones (am*wl, al*wm, al*wl) are below
// 2^-23 |a w| - the size of the rounding error one fp32 FMA makes on the same product - so the result is as close to
// the exact dot product as the fp32 kernel's (tests/test_kernels_gpu.py compares both with a float64 evaluation).
// Six bf16 MFMAs replace sixteen fp32 ones' worth of matrix-pipe time: 6/16 of the cost.
//
// W2 comes pre-split and packed (dfol_pair_pack_w2_bf16x3): per 32 k (one MFMA's depth) a 60 KB chunk of 20 column tiles x 3 pieces,
// rows >= HID2 zero, the four 8-element k-groups of row r stored at group kq ^ swz[(r >> 2) & 3] (64-byte rows: the same bank
// geometry as the fp32 image, so every ds_read_b128 of a B fragment is conflict-free; SQ_LDS_BANK_CONFLICT = 0).  A chunk is
// copied to LDS verbatim by LDS-DMA.  Every wavefront owns 32 pair slots and all 19 column tiles (152 accumulator registers).
// The schedule (ping-pong between the two wavefronts of a SIMD, see pair_ll32s_kernel) came out of clock64 traces
// (tools/scratch/trace_pair.py, -DDFOL_PAIR_TRACE).
#include "dfol_common.h"

#include <stdlib.h>

#include <type_traits>

#ifndef DFOL_SP_NT
#define DFOL_SP_NT 1
#endif
// W2 chunks requested in the prologue (1 or 2).  Chunk 1 is first read in tick 3: requested by Y in its build tick 1 - the steady-state
// protocol one chunk early - it leaves the prologue waiting for 60 KB of DMA instead of 120.
#ifndef DFOL_PAIR_PRE
#define DFOL_PAIR_PRE 1
#endif


// -DDFOL_PAIR_TRACE: clock64 stamps of one wavefront per half in a few workgroups (tools/scratch/trace_pair.py reads them)
#ifdef DFOL_PAIR_TRACE
__device__ long long dfol_trace_buf[8 * 8 * 64];
#define TRACE(slot)                                                                                                  \
    do {                                                                                                             \
        if (trace_on && lane == 0) dfol_trace_buf[(trace_blk * 8 + wave) * 64 + (slot)] = clock64();                 \
    } while (0)
#else
#define TRACE(slot)
#endif

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr float SP_NL2E = -1.44269504088896340736f;          // -log2(e)
constexpr int SP_CH = 32;                                   // K per chunk = one v_mfma_f32_16x16x32_bf16

__device__ __forceinline__ int sp_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }      // {0,3,2,1}[(row>>2)&3]

// x = h + m + l exactly, each the fp32 whose low 16 bits are (or can be taken as) zero: the bf16 piece is the top half.
__device__ __forceinline__ void sp_split(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = __float_as_uint(x);
    const float r = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r);
    l = __float_as_uint(r - __uint_as_float(m & 0xffff0000u));
}
// {top half of x0, top half of x1} as one register (element 0 in the low half)
__device__ __forceinline__ uint32_t sp_pack(uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// Packed image geometry: a chunk always holds 20 column tiles (rows >= HID2 zero) in two REGIONS - tiles 0..7 and 8..19 - each
// stored [piece][row][4 k-groups] x 16 bytes, so that a region is one contiguous run of 1536 / 2304 DMA pieces (6 / 9 passes of a
// 256-thread workgroup: no guards, every wait count is static).
constexpr int SP_T0 = 8, SP_TILES = 20;
constexpr int SP_R0_PIECES = 3 * SP_T0 * 16 * 4, SP_R1_PIECES = 3 * (SP_TILES - SP_T0) * 16 * 4, SP_PIECES = SP_R0_PIECES + SP_R1_PIECES;

// One thread per 16-byte piece of the packed image.
__global__ void pair_pack_w2_split_kernel(const float* __restrict__ W2, int64_t ld_w2, int HID2, int HID1, u32x4* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (HID1 / SP_CH) * SP_PIECES) return;
    const int c = idx / SP_PIECES;
    int rem = idx - c * SP_PIECES;
    const int region = rem >= SP_R0_PIECES;
    rem -= region ? SP_R0_PIECES : 0;
    const int rows_r = (region ? SP_TILES - SP_T0 : SP_T0) * 16;
    const int p = rem / (rows_r * 4), rr = rem - p * rows_r * 4, r = (region ? SP_T0 * 16 : 0) + (rr >> 2), slot = rr & 3;
    const int kq = slot ^ sp_swz(r);
    uint32_t piece[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float w = r < HID2 ? W2[(int64_t)r * ld_w2 + c * SP_CH + kq * 8 + j] : 0.f;
        uint32_t h, m, l;
        sp_split(w, h, m, l);
        piece[j] = p == 0 ? h : (p == 1 ? m : l);
    }
    out[idx] = u32x4{sp_pack(piece[0], piece[1]), sp_pack(piece[2], piece[3]), sp_pack(piece[4], piece[5]), sp_pack(piece[6], piece[7])};
}

// Every wavefront owns 32 pair slots (two 16-slot tiles) and all column tiles.  Two schedules:
//   PP = false: 4-wavefront workgroups, two of them per CU, one chunk buffer refilled region by region.  The two wavefronts of a
//               SIMD belong to different workgroups and nothing orders their phases: measured with clock64 stamps they drift into
//               doing the same thing at the same time (both building A pieces, then both queueing for the matrix pipe).
//   PP = true:  ONE 8-wavefront workgroup per CU and a strict ping-pong between its halves (wavefronts w and w + 4 share a SIMD):
//               in every "tick" one half runs the 228 MFMAs of a chunk while the other half loads U/V rows, requests the next W2
//               chunk and builds its A pieces; a workgroup barrier ends the tick and the roles swap.  The matrix pipe of a SIMD
//               always has exactly one wavefront feeding it.  The second half lags the first by one tick, so a chunk is live for
//               two ticks: two chunk buffers (120 KB of LDS).
// (A persistent variant - each half running through its own task sequence half a task apart, so that one half's epilogue and
// next-task setup overlap the other's main loop - was built and measured: correct, but 1.85 ms against 1.78 ms.  Its build ticks
// became longer than the multiply ticks (U/V row latency no longer hidden by a fresh workgroup's prologue), and prefetching those
// rows under the MFMAs needs 16-32 registers that the 152 accumulators do not leave: any spill reload inside the multiply waits,
// in order, for the prefetch itself.)
template <int NB16, bool TBF16, bool PP>
__global__ __launch_bounds__(PP ? 512 : 256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pair_ll32s_kernel(
    const float* __restrict__ UV, int64_t ld_uv, int HID1, const float* __restrict__ pos, int64_t ld_pos,
    const float* __restrict__ Wg, const u32x4* __restrict__ W2s, const float* __restrict__ b2, int HID2,
    const float* __restrict__ E, int64_t ld_e, const float* __restrict__ be, const int32_t* __restrict__ n_obj,
    const int32_t* __restrict__ obj_off, int Q, int tiles_per_image, const int32_t* __restrict__ req_col,
    const int32_t* __restrict__ req_tile, const uint8_t* __restrict__ req_orient, int K, int NS, float dflt,
    void* __restrict__ tiles_v) {
    constexpr int MT = 2, WAVES = PP ? 8 : 4;
    constexpr int ROWS = NB16 * 16, T = WAVES * 64, SLOTS = MT * 16 * WAVES;
    static_assert(NB16 > SP_T0 && NB16 <= SP_TILES, "geometry");
    __shared__ __attribute__((aligned(16))) u32x4 Bs[(PP ? 2 : 1) * SP_PIECES];     // W2 chunk(s), all three pieces (60 KB each)
    __shared__ __attribute__((aligned(16))) float Wgs[256 * 4];
    constexpr int PP_STAGE_FLOATS = PP ? 8192 : 4;                  // PP: the epilogue's bias / embedding rows have their own 32 KB
    __shared__ __attribute__((aligned(16))) float stage_pp[PP_STAGE_FLOATS];
    // (an XCD-aware order - each XCD a contiguous run of tiles, so that an image's U / V rows sit in one L2 instead of eight - was measured:
    // 1645 us against 1646; the kernel does not wait for those rows)
    const int q = blockIdx.x / tiles_per_image, tb = blockIdx.x - q * tiles_per_image;
    const int n = n_obj[q], npairs = n * (n - 1);          // slots enumerate the ORDERED PAIRS s != o (row-major in s, util.py:87-103): the
    if (tb * SLOTS >= npairs) return;                       // diagonal is never computed (round 3: one workgroup in 40 at N = 100, one in 6 at N = 36)
    bool any = false;
    for (int k = 0; k < K; ++k) any |= req_col[(int64_t)k * Q + q] >= 0;
    if (!any) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kh = lane >> 4, r16 = lane & 15;
#ifdef DFOL_PAIR_TRACE
    const int trace_blk = ((int)blockIdx.x - 3000) / 500;
    const bool trace_on = blockIdx.x >= 3000 && (blockIdx.x - 3000) % 500 == 0 && trace_blk < 8;
#endif
    TRACE(0);
    if constexpr (PP) {                                      // chunk 0 (DFOL_PAIR_PRE chunks) requested before anything else: it lands under
        const int nck = HID1 / SP_CH;                        // the geometry arithmetic below
#pragma unroll
        for (int cb = 0; cb < DFOL_PAIR_PRE; ++cb)
#pragma unroll
            for (int i = 0; i < (SP_PIECES + T - 1) / T; ++i)
                if (cb < nck && T * i + wave * 64 < SP_PIECES)
                    __builtin_amdgcn_global_load_lds(W2s + (int64_t)cb * SP_PIECES + T * i + tid,
                                                     (__attribute__((address_space(3))) void*)&Bs[cb * SP_PIECES + T * i + wave * 64], 16, 0, 0);
    }
    const int first = obj_off[q];
    float geo[MT][4];
    const float* Urow[MT];
    const float* Vrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int e_slot = tb * SLOTS + wave * (MT * 16) + m * 16 + r16;
        const bool valid = e_slot < npairs;
        const int s = valid ? e_slot / (n - 1) : 0, oo_ = valid ? e_slot - s * (n - 1) : 0, o = oo_ + (oo_ >= s);      // (n >= 2 here)
        const float* ps = pos + (int64_t)(first + s) * ld_pos;
        const float* po = pos + (int64_t)(first + o) * ld_pos;
        const float x1 = ps[0], y1 = ps[1], w1 = ps[2], h1 = ps[3], x2 = po[0], y2 = po[1], w2 = po[2], h2 = po[3];
        const float dx = x1 + w1 / 2.0f - x2 - w2 / 2.0f, dy = y1 + h1 / 2.0f - y2 - h2 / 2.0f;
        const float dist = sqrtf(dx * dx + dy * dy);
        geo[m][0] = dist;
        geo[m][1] = asinf(dy / fmaxf(dist, 1e-10f));
        geo[m][2] = (x2 - x1 > 0.f) ? 1.f : ((x2 - x1 < 0.f) ? -1.f : 0.f);
        geo[m][3] = (y2 - y1 > 0.f) ? 1.f : ((y2 - y1 < 0.f) ? -1.f : 0.f);
        Urow[m] = UV + (int64_t)(first + s) * ld_uv + 8 * kh;
        Vrow[m] = UV + (int64_t)(first + o) * ld_uv + HID1 + 8 * kh;
    }
    for (int i = tid; i < HID1; i += T) {                    // geometry weights, transposed to [feature][k]
        const float4 g = *reinterpret_cast<const float4*>(Wg + i * 4);
        Wgs[i] = g.x, Wgs[256 + i] = g.y, Wgs[512 + i] = g.z, Wgs[768 + i] = g.w;
    }

    if constexpr (PP) {                                      // the epilogue's rows are staged up front (the first barrier publishes them)
        constexpr int SR = PP_STAGE_FLOATS / ROWS - 1;
        for (int i = tid; i < ROWS; i += T) stage_pp[i] = SP_NL2E * (i < HID2 ? b2[i] : -1.0e30f);        // Sigmoid(x + b) = 1 / (1 + 2^(-L2E x - L2E b))
        for (int k = 0; k < (K < SR ? K : SR); ++k) {
            const int col = req_col[(int64_t)k * Q + q];
            for (int i = tid; i < ROWS; i += T) stage_pp[ROWS * (1 + k) + i] = (col >= 0 && i < HID2) ? E[(int64_t)col * ld_e + i] : 0.f;
        }
    }

    floatx4 acc[MT][NB16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < NB16; ++i) acc[m][i] = floatx4{0.f, 0.f, 0.f, 0.f};

    // Region `region` of W2 chunk c -> Bs: a wavefront's 64 pieces of a pass are 1 KiB of contiguous LDS
    auto dma_w2 = [&](int c, int region) __attribute__((always_inline)) {
        const int base = region ? SP_R0_PIECES : 0, pieces = region ? SP_R1_PIECES : SP_R0_PIECES, passes = (pieces + T - 1) / T;
#pragma unroll
        for (int i = 0; i < passes; ++i) {
            if (pieces % T == 0 || T * i + wave * 64 < pieces) {    // (a multiple of 64: the guard is wave-uniform)
                const u32x4* src = W2s + (int64_t)c * SP_PIECES + base + T * i + tid;
                u32x4* dst = &Bs[base + T * i + wave * 64];         // wave-uniform base; the hardware adds lane * 16 bytes
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };
    // PP: the whole chunk c -> chunk buffer `buf`, requested by the NW wavefronts first_wave .. first_wave + NW - 1
    auto dma_chunk = [&](int c, int buf, int first_wave, auto nw_tag) __attribute__((always_inline)) {
        constexpr int TT = decltype(nw_tag)::value * 64, passes = (SP_PIECES + TT - 1) / TT;
        const int w = wave - first_wave, t = tid - first_wave * 64;
#pragma unroll
        for (int i = 0; i < passes; ++i) {
            if (SP_PIECES % TT == 0 || TT * i + w * 64 < SP_PIECES) {
                const u32x4* src = W2s + (int64_t)c * SP_PIECES + TT * i + t;
                u32x4* dst = &Bs[buf * SP_PIECES + TT * i + w * 64];
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };
    // A pieces of a chunk for the lane's slots: k = 32 c + 8 kh + 0..7.  Two ADJACENT k of one slot form every packed-math pair
    // (U, V and the transposed geometry weights are contiguous in k): no register shuffles.
    typedef float f2 __attribute__((ext_vector_type(2)));
    float4 ru[MT][2], rv[MT][2];                                    // [slot][half]: the lane's 8 first-layer terms of a chunk
    auto load_uv = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                ru[m][half] = *reinterpret_cast<const float4*>(Urow[m] + SP_CH * c + 4 * half);
                rv[m][half] = *reinterpret_cast<const float4*>(Vrow[m] + SP_CH * c + 4 * half);
            }
    };
    u32x4 ap[MT][3];                                                // [slot][piece h, m, l]
    auto make_a = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                f2 g[4];                                            // geometry weights of k, k+1 for the four geometry features
#pragma unroll
                for (int d = 0; d < 4; ++d) g[d] = *reinterpret_cast<const f2*>(&Wgs[d * 256 + SP_CH * c + 8 * kh + 4 * half + 2 * jp]);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f2 uu = jp == 0 ? (f2){ru[m][half].x, ru[m][half].y} : (f2){ru[m][half].z, ru[m][half].w};
                    const f2 vv = jp == 0 ? (f2){rv[m][half].x, rv[m][half].y} : (f2){rv[m][half].z, rv[m][half].w};
                    f2 z = uu + vv;
#pragma unroll
                    for (int d = 0; d < 4; ++d) z = __builtin_elementwise_fma(g[d], (f2){geo[m][d], geo[m][d]}, z);
                    const float a0 = z.x > 0.f ? z.x : dfol_exp(z.x) - 1.0f;   // nn.ELU
                    const float a1 = z.y > 0.f ? z.y : dfol_exp(z.y) - 1.0f;
                    uint32_t h0, m0, l0, h1, m1, l1;
                    sp_split(a0, h0, m0, l0);
                    sp_split(a1, h1, m1, l1);
                    ap[m][0][2 * half + jp] = sp_pack(h0, h1);
                    ap[m][1][2 * half + jp] = sp_pack(m0, m1);
                    ap[m][2][2 * half + jp] = sp_pack(l0, l1);
                }
            }
    };
    const int boff = r16 * 4 + (kh ^ sp_swz(r16));                  // the lane's 16-byte piece inside a 16-row block
    // The MFMAs of column tiles i .. i+NT-1: six piece products for each slot tile, smallest terms first.  Consecutive MFMAs go to
    // different accumulators (NT * MT of them in rotation): an MFMA that accumulates onto the result of the one just issued waits for
    // its full latency, about twice its issue time.
    int bbase = boff;                                               // + the chunk buffer's offset (PP)
    auto load_b = [&](int i, bf16x8 (&b)[3]) {
        const int region = i >= SP_T0, rows_r = (region ? SP_TILES - SP_T0 : SP_T0) * 16;
        const int at = (region ? SP_R0_PIECES : 0) + (i - (region ? SP_T0 : 0)) * 64 + bbase;
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, Bs[at + p * rows_r * 4]);
    };
    auto tiles_mfma = [&](int i, auto nt_tag) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_tag)::value;
        bf16x8 b[NT][3];
#pragma unroll
        for (int t = 0; t < NT; ++t) load_b(i + t, b[t]);
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m][i + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[m][PA[x]]), b[t][PB[x]], acc[m][i + t], 0, 0, 0);
    };
    constexpr int NT = DFOL_SP_NT;
    constexpr int PA6[6] = {2, 0, 1, 1, 0, 0}, PB6[6] = {0, 2, 1, 0, 1, 0};
    auto region_mfma = [&](int t0, int t1) __attribute__((always_inline)) {
        if constexpr (PP) {
            // hand-pipelined: the B fragments of tile i+1 are requested before the MFMAs of tile i, and the scheduler may not move
            // anything across tiles (left alone it hoists the reads of all 19 tiles to the top and spills)
            bf16x8 bq[2][3];
            load_b(t0, bq[0]);
#pragma unroll
            for (int i = t0; i < t1; ++i) {
                if (i + 1 < t1) load_b(i + 1, bq[(i + 1 - t0) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                // the six products of one accumulator back to back: a dependent MFMA takes its C operand from the previous result
                // without a register-file read and issues faster than one on a fresh accumulator (17.5 vs 20 cycles measured)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int x = 0; x < 6; ++x)
                        acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[m][PA6[x]]), bq[(i - t0) & 1][PB6[x]], acc[m][i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int i = t0; i < t1; i += NT) {
                if (i + NT <= t1) tiles_mfma(i, std::integral_constant<int, NT>());
                else {
#pragma unroll
                    for (int j = i; j < t1; ++j) tiles_mfma(j, std::integral_constant<int, 1>());
                }
            }
        }
    };
    // LDS-DMA completion is tracked by vmcnt only; a workgroup barrier does not wait for it by itself
    auto dma_barrier = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0)
        __syncthreads();
    };

    const int nchunk = HID1 / SP_CH, lastc = nchunk - 1;
    if constexpr (PP) {
        // Tick tau: half X (wavefronts 0-3) builds chunk tau/2 on even ticks and multiplies it on the next; half Y (4-7) does the
        // same one tick later.  Chunk c is read in ticks 2c+1 (X) and 2c+2 (Y); its buffer is refilled with chunk c+2 by Y during
        // its build tick 2c+3 (U/V rows requested first: vmcnt retires in order) and Y drains that DMA at the end of its multiply
        // tick 2c+4, one tick before X needs it.
        const int g = wave >> 2;
        // (X's first U/V rows requested HERE, in flight over the barrier, were measured: the 32 registers alive across it spill, 1.68 -> 1.85 ms)
        dma_barrier();                                      // Wgs, the staged epilogue rows and the first chunk(s) visible
        // Each half runs its own copy of the loop (plain straight-line bodies for the register allocator); the barriers pair up by
        // count: X executes 2 per chunk, Y one idle tick first and none after its last multiply.
        auto run_half = [&](auto y_tag) __attribute__((always_inline)) {
            constexpr bool Y = decltype(y_tag)::value;
            TRACE(1);
            if (Y) __syncthreads();                         // tick 0: X builds chunk 0.  (Y building ITS chunk 0 here too, so that X's first multiply is
                                                            // not slowed by Y's longer first build: 36 bytes of scratch per lane and 1.69 -> 1.73 ms)
            TRACE(2);
            for (int c = 0; c < nchunk; ++c) {
                load_uv(c);
                if (Y && c >= DFOL_PAIR_PRE - 1 && c < lastc) dma_chunk(c + 1, (c + 1) & 1, 4, std::integral_constant<int, 4>());
                make_a(c);
                __builtin_amdgcn_sched_barrier(0);          // the A pieces are pure register arithmetic: without this fence the compiler
                TRACE(3 + 4 * c);                           // sinks them below the barrier, in front of the MFMAs of the multiply tick
                __syncthreads();                            // end of the build tick
                __builtin_amdgcn_sched_barrier(0);
                TRACE(4 + 4 * c);
                bbase = boff + (c & 1) * SP_PIECES;
                region_mfma(0, NB16);                       // (s_setprio 1..3 around the MFMAs: no effect, 1.75-1.79 ms)
                if (Y) __builtin_amdgcn_s_waitcnt(0x0F70);  // the chunk requested in the build tick has landed
                TRACE(5 + 4 * c);
                if (!Y || c < lastc) __syncthreads();       // end of the multiply tick (Y's last one has no partner)
                TRACE(6 + 4 * c);
            }
        };
        if (g == 0) run_half(std::false_type());
        else run_half(std::true_type());
    } else {
    // One chunk buffer, refilled region by region behind the wavefronts: region 0 of chunk c+1 is requested when everyone has left
    // region 0 of chunk c (barrier "B") and lands under the MFMAs of region 1; region 1 of chunk c+1 is requested at barrier "A" and
    // lands under the A building and the region-0 MFMAs of chunk c+1.  vmcnt retires in order: the U/V rows of the next chunk are
    // requested just before that DMA, so that building the A pieces waits for them only.
    load_uv(0);
    dma_w2(0, 0);
    dma_w2(0, 1);
    dma_barrier();                                          // Wgs and chunk 0 visible
    for (int c = 0; c < nchunk; ++c) {
        make_a(c);
        region_mfma(0, SP_T0);
        dma_barrier();                                      // B: region 0 of chunk c fully read; region 1 landed
        if (c < lastc) dma_w2(c + 1, 0);
        region_mfma(SP_T0, NB16);
        dma_barrier();                                      // A: region 1 of chunk c fully read; region 0 of chunk c+1 landed
        if (c < lastc) {
            load_uv(c + 1);
            dma_w2(c + 1, 1);
        }
    }
    }

    // The W2 chunk is free now: stage the hidden bias and the requested embedding rows in it.  Padding columns get bias
    // -1e30, whose Sigmoid is exactly 0.
    constexpr int STAGE_ROWS = (PP ? PP_STAGE_FLOATS : SP_PIECES * 4) / ROWS - 1;      // embedding rows that fit beside the bias
    float* stage = PP ? stage_pp : reinterpret_cast<float*>(&Bs[0]);
    const int Kc = K < STAGE_ROWS ? K : STAGE_ROWS;
    if constexpr (!PP) {
        for (int i = tid; i < ROWS; i += T) stage[i] = SP_NL2E * (i < HID2 ? b2[i] : -1.0e30f);
        for (int k = 0; k < Kc; ++k) {
            const int col = req_col[(int64_t)k * Q + q];
            for (int i = tid; i < ROWS; i += T) stage[ROWS * (1 + k) + i] = (col >= 0 && i < HID2) ? E[(int64_t)col * ld_e + i] : 0.f;
        }
        __syncthreads();
    }
    const int64_t tile_sz = (int64_t)NS * NS;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int i = 0; i < NB16; ++i) {
            const float bv = stage[i * 16 + r16];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[m][i][e] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(acc[m][i][e], SP_NL2E, bv)));
        }
        for (int k = 0; k < K; ++k) {
            const int col = req_col[(int64_t)k * Q + q];
            if (col < 0) continue;
            float part[4] = {0.f, 0.f, 0.f, 0.f};
            if (k < Kc) {
                const float* erow = stage + ROWS * (1 + k) + r16;
#pragma unroll
                for (int i = 0; i < NB16; ++i) {
                    const float ev = erow[i * 16];
#pragma unroll
                    for (int e = 0; e < 4; ++e) part[e] = fmaf(acc[m][i][e], ev, part[e]);
                }
            } else {
                const float* erow = E + (int64_t)col * ld_e;
#pragma unroll
                for (int i = 0; i < NB16; ++i) {
                    const float ev = erow[min(i * 16 + r16, HID2 - 1)];     // padding columns: activation is exactly 0
#pragma unroll
                    for (int e = 0; e < 4; ++e) part[e] = fmaf(acc[m][i][e], ev, part[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) part[e] = dfol_group_sum<16>(part[e]);
            if (r16 < 4) {
                const float v = r16 == 0 ? part[0] : (r16 == 1 ? part[1] : (r16 == 2 ? part[2] : part[3]));
                const int ee = tb * SLOTS + wave * (MT * 16) + m * 16 + 4 * kh + r16;
                if (ee < npairs) {
                    // ee / (n - 1) without the integer-division sequence: (ee + 0.5) / (n - 1) is at least 0.5 / (n - 1) away from an integer
                    const int ss = (int)(((float)ee + 0.5f) * __builtin_amdgcn_rcpf((float)(n - 1))), op = ee - ss * (n - 1), oo = op + (op >= ss);
                    const float x = v + (be ? be[col] : 0.f);
                    const float val = fminf(x, 0.f) - dfol_log(1.0f + dfol_exp(-fabsf(x)));        // nn.LogSigmoid (the diagonal keeps the caller's fill)
                    const int64_t at = (int64_t)req_tile[(int64_t)k * Q + q] * tile_sz +
                                       ((req_orient && req_orient[(int64_t)k * Q + q]) ? (int64_t)oo * NS + ss : (int64_t)ss * NS + oo);
                    if (TBF16) {
                        uint32_t u = __float_as_uint(val);
                        u += 0x7fffu + ((u >> 16) & 1u);                 // round to nearest even
                        reinterpret_cast<uint16_t*>(tiles_v)[at] = (uint16_t)(u >> 16);
                    } else {
                        reinterpret_cast<float*>(tiles_v)[at] = val;
                    }
                }
            }
        }
    }
    TRACE(60);
}


}  // namespace

#ifdef DFOL_PAIR_TRACE
extern "C" int dfol_pair_trace_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dfol_trace_buf), sizeof(dfol_trace_buf)); }
#endif

extern "C" int dfol_pair_pack_w2_bf16x3(const float* W2, int64_t ld_w2, int32_t HID2, int32_t HID1, void* W2_split, void* stream) {
    DFOL_REQUIRE(HID1 > 0 && HID1 <= 256 && HID1 % SP_CH == 0, "pair_pack_w2_bf16x3: HID1=%d must be a multiple of %d, <= 256", HID1, SP_CH);
    DFOL_REQUIRE(HID2 > 256 && HID2 <= 320, "pair_pack_w2_bf16x3: HID2=%d must be in (256, 320]", HID2);
    DFOL_REQUIRE(W2 && W2_split && ld_w2 >= HID1, "pair_pack_w2_bf16x3: null pointer or ld_w2 < HID1");
    DFOL_REQUIRE((uintptr_t)W2_split % 16 == 0, "pair_pack_w2_bf16x3: output must be 16-byte aligned");
    const int total = (HID1 / SP_CH) * SP_PIECES;
    hipLaunchKernelGGL(pair_pack_w2_split_kernel, dim3(dfol_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, W2, ld_w2, HID2, HID1,
                       (u32x4*)W2_split);
    DFOL_LAUNCH_CHECK("pair_pack_w2_bf16x3");
    return 0;
}

extern "C" int dfol_pair_ll_split_f32(const float* UV, int64_t ld_uv, int32_t HID1, const float* pos, int64_t ld_pos, const float* Wg,
                                      const void* W2_split, const float* b2, int32_t HID2, const float* E, int64_t ld_e,
                                      const float* be, const int32_t* n_obj, const int32_t* obj_off, int32_t Q, int32_t max_n,
                                      const int32_t* req_col, const int32_t* req_tile, const uint8_t* req_orient, int32_t K,
                                      int32_t NS, float default_ll, int32_t tile_dtype, void* tiles_v, void* stream) {
    DFOL_REQUIRE(tile_dtype == DFOL_TILE_F32 || (tile_dtype == DFOL_TILE_BF16 && NS % 8 == 0), "pair_ll_split: tile_dtype=%d (bf16 tiles need NS %% 8 == 0)", tile_dtype);
    DFOL_REQUIRE(Q >= 0 && K >= 0 && NS > 0 && NS % 4 == 0 && max_n >= 0 && max_n <= NS, "pair_ll_split: bad sizes Q=%d K=%d NS=%d max_n=%d", Q, K, NS, max_n);
    DFOL_REQUIRE(HID1 > 0 && HID1 <= 256 && HID1 % SP_CH == 0 && ld_uv % 4 == 0, "pair_ll_split: HID1=%d must be a multiple of %d, <= 256, UV rows 16-byte aligned", HID1, SP_CH);
    DFOL_REQUIRE(HID2 > 256 && HID2 <= 320, "pair_ll_split: HID2=%d must be in (256, 320]", HID2);
    if (Q == 0 || K == 0 || max_n < 2) return 0;
    DFOL_REQUIRE(UV && pos && Wg && W2_split && b2 && E && n_obj && obj_off && req_col && req_tile && tiles_v, "pair_ll_split: null pointer");
    DFOL_REQUIRE(((uintptr_t)UV % 16 == 0) && ((uintptr_t)W2_split % 16 == 0) && ((uintptr_t)Wg % 16 == 0), "pair_ll_split: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    static const int pp = getenv("DFOL_PAIR_SPLIT_PP") ? atoi(getenv("DFOL_PAIR_SPLIT_PP")) : 1;
    // pp = 1 (default): one ping-pong workgroup per 256 slots; 0: two 4-wavefront workgroups per CU.  (The persistent variant with the
    // epilogue fused into the next task's first multiply tick - measured slower, DESIGN.md 3.3 - is parked in tools/scratch/.)
    const int tpi = dfol_cdiv((int64_t)max_n * (max_n - 1), pp ? 256 : 128);
    DFOL_REQUIRE((int64_t)Q * tpi < ((int64_t)1 << 31), "pair_ll_split: too many tiles");
    const dim3 grid((unsigned)Q * tpi);
#define DFOL_PAIR32S(NBV, BF)                                                                                                         \
    if (pp)                                                                                                                           \
        hipLaunchKernelGGL((pair_ll32s_kernel<NBV, BF, true>), grid, dim3(512), 0, st, UV, ld_uv, HID1, pos, ld_pos, Wg, (const u32x4*)W2_split, b2, HID2, \
                           E, ld_e, be, n_obj, obj_off, Q, tpi, req_col, req_tile, req_orient, K, NS, default_ll, tiles_v);          \
    else                                                                                                                              \
        hipLaunchKernelGGL((pair_ll32s_kernel<NBV, BF, false>), grid, dim3(256), 0, st, UV, ld_uv, HID1, pos, ld_pos, Wg, (const u32x4*)W2_split, b2, HID2, \
                           E, ld_e, be, n_obj, obj_off, Q, tpi, req_col, req_tile, req_orient, K, NS, default_ll, tiles_v)
    if (HID2 <= 272) { if (tile_dtype == DFOL_TILE_BF16) DFOL_PAIR32S(17, true); else DFOL_PAIR32S(17, false); }
    else if (HID2 <= 288) { if (tile_dtype == DFOL_TILE_BF16) DFOL_PAIR32S(18, true); else DFOL_PAIR32S(18, false); }
    else if (HID2 <= 304) { if (tile_dtype == DFOL_TILE_BF16) DFOL_PAIR32S(19, true); else DFOL_PAIR32S(19, false); }
    else { if (tile_dtype == DFOL_TILE_BF16) DFOL_PAIR32S(20, true); else DFOL_PAIR32S(20, false); }
#undef DFOL_PAIR32S
    DFOL_LAUNCH_CHECK("pair_ll_split");
    return 0;
}
