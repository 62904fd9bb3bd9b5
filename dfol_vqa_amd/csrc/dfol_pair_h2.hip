// Fused pair MLP with the second layer on the fp16 matrix pipe at fp32 accuracy, THREE products per fp32 product (gfx950).
//
// Round 3's kernel (csrc/dfol_pair_split.hip) cuts every fp32 operand into three bf16 pieces and issues six piece products per fp32
// product: 0.55 of the bf16 pipe executed, 0.09 in algorithmic flops, and three rounds of schedule work moved it by < 3 %.  The lever
// is the arithmetic: fp16 carries 11 significand bits against bf16's 8, so TWO pieces x = h + l (h = fp16(x), l = fp16(x - h), both
// rounded to nearest even; x - h is exact in fp32) hold 22 - 23 bits, and the three products  al*wh + ah*wl + ah*wh  on
// v_mfma_f32_16x16x32_f16 (fp32 accumulation, the same 2.5 PFLOP/s as bf16) leave out only al*wl <= 2^-22 |a w|.
//
// fp16 has a narrow exponent range, and this is where the accuracy is decided (measured, tools/lab/split_accuracy.hip ->
// profiles/r04_split_accuracy_lab.txt, K = 256 dot products of ELU activations with U(-1/16, 1/16) weights, error against float64):
//   * the low piece of a value below 2^-3 is SUBNORMAL in fp16.  The matrix pipe keeps subnormal operands (probe in the same lab: a
//     2^-20 input survives); a pipe that flushed them would be off by 1e-4.  A subnormal low piece still has an absolute error of
//     2^-25, so an operand's error is max(2^-22 |x|, 2^-25): harmless for activations of order 1, but weights of order 1/16 would
//     carry 2^-21 relative.  unscaled weights: mean error 3.1e-7 (the fp32 FMA chain: 1.5e-7, bf16x3: 1.25e-7);
//   * so every ROW of W2 is scaled by a power of two 2^e_r that puts its largest magnitude into [2^13, 2^14) (exact; chosen by the pack
//     kernel on the device, no host round trip, overflow impossible for finite weights), and the epilogue folds 2^-e_r into the
//     multiplier of the Sigmoid's exponent (the product is there anyway).  scaled weights: mean error 1.0e-7 - BELOW the fp32 FMA chain
//     and the bf16x3 kernel, because the MFMA rounds once per 32 products;
//   * activations are ELU outputs in (-1, inf): they are split unscaled and saturate at 6e4 (fp16's largest finite value is 65504).
//     The first layer is fed by Sigmoid outputs and box geometry, |z| <= sum |W1| ~ 1036 |w|: an activation of 6e4 needs weights of
//     magnitude 58.  The clamp costs nothing (it is the third operand of the v_med3 that implements the ELU's select).
//
// W2 image (dfol_pair_pack_w2_f16x2): per 32 k (one MFMA's depth) a 40 KB chunk [2 pieces][320 rows][4 k-groups] x 16 bytes, rows >=
// HID2 zero, the four 8-element k-groups of row r stored at group kq ^ swz[(r >> 2) & 3] (64-byte rows: every ds_read_b128 of a B
// fragment is bank-conflict-free), copied to LDS verbatim by LDS-DMA; after the chunks 320 floats -log2(e) 2^-e_r (the Sigmoid's
// per-column multiplier) and the 320 exponents.
//
// Schedule: the ping-pong of the bf16x3 kernel (one 8-wavefront workgroup per CU; in every tick one half runs a chunk's MFMAs - 114 now,
// not 228 - while the other half loads U / V rows, requests the next W2 chunk and builds its A pieces).
#include "dfol_common.h"

#include <stdlib.h>

#include <type_traits>

// -DDFOL_PAIR_TRACE: clock64 stamps of one wavefront per half in a few workgroups (tools/lab/trace_pair.py reads them)
#ifdef DFOL_PAIR_TRACE
__device__ long long dfol_h2_trace_buf[8 * 8 * 64];
#define TRACE(slot)                                                                                                  \
    do {                                                                                                             \
        if (trace_on && lane == 0) dfol_h2_trace_buf[(trace_blk * 8 + wave) * 64 + (slot)] = clock64();              \
    } while (0)
#else
#define TRACE(slot)
#endif

// U / V rows of chunk c + 1 requested at the top of the multiply tick of chunk c (1) or at the top of their own build tick (0)
#ifndef DFOL_H2_PREFETCH
#define DFOL_H2_PREFETCH 1
#endif
// tiles of B fragments requested ahead of the MFMAs that use them (a tile's six MFMAs are 96 cycles of the pipe: less than an LDS round trip
// under load, so one tile ahead - the bf16x3 kernel's distance, with twelve MFMAs per tile - leaves the reads exposed)
#ifndef DFOL_H2_BDEPTH
#define DFOL_H2_BDEPTH 2
#endif
// Y's chunk requests as inline asm (1): in flight across the build tick's closing barrier
#ifndef DFOL_H2_DMA_ASM
#define DFOL_H2_DMA_ASM 1
#endif
// lab switch for the TRAIN variant (tools/lab/time_train_fwd.py): bit 0 = no Z stores, bit 1 = no pre2 stores (what each costs)
#ifndef DFOL_H2T_SKIP
#define DFOL_H2T_SKIP 0
#endif
// Y's request for the next W2 chunk: 0 = at the top of its build tick, 1 = after its A pieces are built
#ifndef DFOL_H2_DMA_LATE
#define DFOL_H2_DMA_LATE 0
#endif


namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr float H2_NL2E = -1.44269504088896340736f;         // -log2(e)
constexpr int H2_CH = 32;                                   // K per chunk = one v_mfma_f32_16x16x32_f16
constexpr int H2_ROWS = 320, H2_TILES = 20;                 // rows (hidden columns) of the packed image
constexpr int H2_PIECES = 2 * H2_ROWS * 4;                  // 16-byte pieces per chunk: 2560 = 40 KB
constexpr float H2_AMAX = 60000.0f;                         // activations saturate here (fp16 max 65504); in units of 1 / ln 2 (see make_a): ELU outputs of 41 589
constexpr float H2_L2E = 1.44269504088896340736f;           // log2(e) = 1 / ln 2
constexpr float H2_LN2 = 0.69314718055994530942f;

__device__ __forceinline__ int h2_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }      // {0,3,2,1}[(row>>2)&3]

// (x0, x1) = (h0 + l0, h1 + l1) up to 2^-22 |x| (2^-25 absolute below 2^-3): v_cvt_pk_f16_f32, two v_cvt_f32_f16, v_pk_add_f32, v_cvt_pk_f16_f32
__device__ __forceinline__ void h2_split2(float x0, float x1, uint32_t& h, uint32_t& l) {
    const f32x2 x = {x0, x1};
    const f16x2 hh = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(hh, f32x2);
    const f16x2 ll = __builtin_convertvector(r, f16x2);
    h = __builtin_bit_cast(uint32_t, hh);
    l = __builtin_bit_cast(uint32_t, ll);
}

// The ELU's negative branch in the kernel's units: z' = z / ln 2 comes in (UV and the geometry weights are pre-multiplied by log2(e)), and
// (e^min(z, 0) - 1) / ln 2 = (2^min(z', 0) - 1) / ln 2 goes out - the hardware exponential with the CLAMP output modifier (2^z' clamped to
// [0, 1]: exactly 1 for z' >= 0, +inf included) and one fused multiply-add.  TWO instructions per element where round 4 had four (v_med3 for
// min(z, 0), v_mul by log2(e), v_exp, v_add -1): a build tick is paced by its instruction COUNT - one issue slot per MFMA of the SIMD's
// other wavefront - so the sixteen elements of a chunk cost 32 slots less.  The factor ln 2 is folded into W2 by the pack kernel.
__device__ __forceinline__ float h2_elu_neg(float zs) {
    const float p = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(zs), 0.0f, 1.0f);        // (folds into v_exp_f32 ... clamp)
    return fmaf(p, H2_L2E, -H2_L2E);
}

// The k of a 32-chunk that k-group kq (a lane's kh) holds at position e = 0..7 of its MFMA operand register: 16 (e >> 2) + 4 kq + (e & 3).
// The order of k inside a chunk is free as long as A and B agree; this one makes the A operand coincide with the C / D layout of an MFMA
// whose ROWS are k: a lane's four accumulator elements of the two geometry MFMAs of a chunk (rows 4 kh + e of k-tile t) are exactly its
// eight A elements, so the first layer's geometry term comes out of the matrix pipe in place (see make_a).
__host__ __device__ __forceinline__ constexpr int h2_kperm(int kq, int e) { return 16 * (e >> 2) + 4 * kq + (e & 3); }

// x - float(h.lo) / x - float(h.hi) in one instruction (v_fma_mix_f32: fma with per-source fp16 / fp32 selection): the residual of the split
template <bool HI>
__device__ __forceinline__ float h2_resid(float x, uint32_t hpair) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x));
    else asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x));
    return r;
}

// One wavefront per row of W2: e_r puts the row's largest magnitude into [2^13, 2^14); tail[r] = -log2(e) 2^-e_r, tail[320 + r] = e_r
__global__ void h2_row_scale_kernel(const float* __restrict__ W2, int64_t ld_w2, int HID2, int HID1, float* __restrict__ tail) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= H2_ROWS) return;
    float m = 0.f;
    if (r < HID2)
        for (int k = lane; k < HID1; k += 64) m = fmaxf(m, fabsf(W2[(int64_t)r * ld_w2 + k] * H2_LN2));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    int e = 0;
    if (m > 0.f && m < 3.0e38f) {
        int x;
        (void)frexpf(m, &x);                                // m = f 2^x, f in [0.5, 1)
        e = 14 - x;
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
    }
    if (lane == 0) {
        tail[r] = ldexpf(H2_NL2E, -e);
        reinterpret_cast<int32_t*>(tail)[H2_ROWS + r] = e;
    }
}

// One thread per 16-byte piece of the packed image.
__global__ void h2_pack_w2_kernel(const float* __restrict__ W2, int64_t ld_w2, int HID2, int HID1, const float* __restrict__ tail,
                                  u32x4* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (HID1 / H2_CH) * H2_PIECES) return;
    const int c = idx / H2_PIECES, rem = idx - c * H2_PIECES;
    const int p = rem / (H2_ROWS * 4), rr = rem - p * H2_ROWS * 4, r = rr >> 2, slot = rr & 3;
    const int kq = slot ^ h2_swz(r);
    const int e = reinterpret_cast<const int32_t*>(tail)[H2_ROWS + r];
    uint32_t piece[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float w0 = 0.f, w1 = 0.f;
        if (r < HID2) {
            // (times ln 2: the kernel feeds the second layer ELU outputs in units of 1 / ln 2 - see make_a; one rounding per weight)
            w0 = ldexpf(W2[(int64_t)r * ld_w2 + c * H2_CH + h2_kperm(kq, 2 * j)] * H2_LN2, e);
            w1 = ldexpf(W2[(int64_t)r * ld_w2 + c * H2_CH + h2_kperm(kq, 2 * j + 1)] * H2_LN2, e);
        }
        uint32_t h, l;
        h2_split2(w0, w1, h, l);
        piece[j] = p == 0 ? h : l;
    }
    out[idx] = u32x4{piece[0], piece[1], piece[2], piece[3]};
}

// TRAIN (round 6): the same kernel as the FORWARD of a train step's pair MLP (trainer.py:429-442 over classifier_oracle.py:145-156) - what the
// backward needs leaves the registers on the way: the first hidden layer Z (multiplied back from the kernel's units of 1 / ln 2: one v_mul per
// element of the build), the pair geometry, the second layer's pre-activations pre2 and the first reader's raw logits; no relation tile is written.  Replaces dfol_pair_hidden1_fwd_f32 + the tall product over Z (Z written, then read again: 5.2 GB).
struct H2Train {
    float* Z;                       // [pairs, HID1]
    float* pre2;                    // [pairs, ld_pre2]
    int64_t ld_pre2;
    float* geo;                     // [pairs, 4]
    float* x;                       // [K, ld_x]: x[k][row] = Sigmoid(pre2[row]) . E[req_col[k][image(row)]]  (no bias)
    int64_t ld_x;
    const int64_t* pair_off;        // [Q]: first pair row of every image
};

// One 8-wavefront workgroup per CU owns 256 ordered pairs (s != o, row-major in s: util.py:87-103) of one image; a wavefront owns 32
// of them (two 16-slot tiles) and all NB16 column tiles (2 x NB16 accumulator tiles: 152 registers at NB16 = 19).
template <int NB16, bool TBF16, bool TRAIN = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void pair_ll32h_kernel(
    const float* UV /* not __restrict__: see load_uv */, int64_t ld_uv, int HID1, const float* __restrict__ pos, int64_t ld_pos,
    const float* __restrict__ Wg, const u32x4* __restrict__ W2h, const float* __restrict__ b2, int HID2,
    const float* __restrict__ E, int64_t ld_e, const float* __restrict__ be, const int32_t* __restrict__ n_obj,
    const int32_t* __restrict__ obj_off, int Q, int tiles_per_image, const int32_t* __restrict__ req_col,
    const int32_t* __restrict__ req_tile, const uint8_t* __restrict__ req_orient, int K, int NS, float dflt,
    void* __restrict__ tiles_v, H2Train tr) {
    constexpr int MT = 2, WAVES = 8;
    constexpr int ROWS = NB16 * 16, T = WAVES * 64, SLOTS = MT * 16 * WAVES;
    static_assert(NB16 > 16 && NB16 <= H2_TILES, "geometry");
    __shared__ __attribute__((aligned(16))) u32x4 Bs[2 * H2_PIECES];          // two W2 chunks, both pieces (40 KB each)
    __shared__ __attribute__((aligned(16))) u32x4 WgA[(256 / H2_CH) * 2 * 64];      // the geometry weights as MFMA A fragments: [chunk][k-tile][lane], 16 KB
    constexpr int STAGE_FLOATS = 8192;                          // the epilogue's bias / multiplier / embedding rows (32 KB)
    __shared__ __attribute__((aligned(16))) float stage[STAGE_FLOATS];
    const int q = blockIdx.x / tiles_per_image, tb = blockIdx.x - q * tiles_per_image;
    const int n = n_obj[q], npairs = n * (n - 1);
    if (tb * SLOTS >= npairs) return;
    bool any = TRAIN;                                           // (training keeps every pair's activations, requested or not)
    for (int k = 0; k < K; ++k) any |= req_col[(int64_t)k * Q + q] >= 0;
    if (!any) return;
    const int64_t prow0 = TRAIN ? tr.pair_off[q] : 0;           // the image's first pair row

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kh = lane >> 4, r16 = lane & 15;
#ifdef DFOL_PAIR_TRACE
    const int trace_blk = ((int)blockIdx.x - 3000) / 500;
    const bool trace_on = blockIdx.x >= 3000 && (blockIdx.x - 3000) % 500 == 0 && trace_blk < 8;
#endif
    TRACE(0);
    const int nchunk = HID1 / H2_CH, lastc = nchunk - 1;
    // chunk 0 is requested before anything else: it lands under the geometry arithmetic below
#pragma unroll
    for (int i = 0; i < H2_PIECES / T; ++i)
        __builtin_amdgcn_global_load_lds(W2h + T * i + tid, (__attribute__((address_space(3))) void*)&Bs[T * i + wave * 64], 16, 0, 0);
    const float* cf = reinterpret_cast<const float*>(W2h + (int64_t)nchunk * H2_PIECES);      // -log2(e) 2^-e_r per hidden column
    const int first = obj_off[q];
    // The pair geometry {distance, angle, sign dx, sign dy} (batch_gqa_boxfeatures_pipeline.py:263-279) as the B fragment of the geometry
    // MFMAs (make_a): 32 contraction slots, twelve in use - [0..3] geo_h, [4..7] geo_l (meeting wg_h), [8..11] geo_h (meeting wg_l)
    u32x4 geoB[MT];
    // U / V rows as 32-bit byte offsets from the image's first row (a scalar base: the loads take the saddr + voffset form; 64-bit per-lane
    // pointers cost 16 registers that the prefetched rows need)
    const char* img_uv = reinterpret_cast<const char*>(UV + (int64_t)first * ld_uv);
    uint32_t uoff[MT], voff[MT];
    uint32_t zoff[MT];                                          // TRAIN: byte offset of the slot's Z row piece from the image's first row (~0: no such pair)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int e_slot = tb * SLOTS + wave * (MT * 16) + m * 16 + r16;
        const bool valid = e_slot < npairs;
        zoff[m] = (TRAIN && valid) ? (uint32_t)(e_slot * HID1 + 4 * kh) * 4u : 0xffffffffu;
        const int s = valid ? e_slot / (n - 1) : 0, oo_ = valid ? e_slot - s * (n - 1) : 0, o = oo_ + (oo_ >= s);      // (n >= 2 here)
        const float* ps = pos + (int64_t)(first + s) * ld_pos;
        const float* po = pos + (int64_t)(first + o) * ld_pos;
        const float x1 = ps[0], y1 = ps[1], w1 = ps[2], h1 = ps[3], x2 = po[0], y2 = po[1], w2 = po[2], h2 = po[3];
        const float dx = x1 + w1 / 2.0f - x2 - w2 / 2.0f, dy = y1 + h1 / 2.0f - y2 - h2 / 2.0f;
        const float dist = sqrtf(dx * dx + dy * dy);
        uint32_t gh01, gl01, gh23, gl23;
        const float ang = asinf(dy / fmaxf(dist, 1e-10f));
        const float sgx = (x2 - x1 > 0.f) ? 1.f : ((x2 - x1 < 0.f) ? -1.f : 0.f), sgy = (y2 - y1 > 0.f) ? 1.f : ((y2 - y1 < 0.f) ? -1.f : 0.f);
        h2_split2(dist, ang, gh01, gl01);
        h2_split2(sgx, sgy, gh23, gl23);
        if (TRAIN && valid && kh == 0) *reinterpret_cast<floatx4*>(tr.geo + (prow0 + e_slot) * 4) = floatx4{dist, ang, sgx, sgy};      // (dfol_pair_train.hip: pair_geometry)
        geoB[m] = kh == 0 ? u32x4{gh01, gh23, gl01, gl23} : (kh == 1 ? u32x4{gh01, gh23, 0u, 0u} : u32x4{0u, 0u, 0u, 0u});
        uoff[m] = (uint32_t)(s * (int)ld_uv + 4 * kh) * 4u;          // the lane's k of a chunk: 16 t + 4 kh + 0..3, t = 0, 1 (h2_kperm)
        voff[m] = (uint32_t)(o * (int)ld_uv + HID1 + 4 * kh) * 4u;
    }
    for (int i = tid; i < (HID1 / 16) * 64; i += T) {       // A fragment of k-tile i / 64 (16 consecutive k): lane (row k' = r16, khh) holds slots 8 khh ..
        const int kt = i >> 6, ln = i & 63, khh = ln >> 4;
        u32x4 frag = {0u, 0u, 0u, 0u};
        if (khh < 2) {
            const float4 g = *reinterpret_cast<const float4*>(Wg + (kt * 16 + (ln & 15)) * 4);
            uint32_t h01, l01, h23, l23;                        // (times log2(e): the first layer's sums come out in units of ln 2, as UV holds them)
            h2_split2(g.x * H2_L2E, g.y * H2_L2E, h01, l01);
            h2_split2(g.z * H2_L2E, g.w * H2_L2E, h23, l23);
            frag = khh == 0 ? u32x4{h01, h23, h01, h23} : u32x4{l01, l23, 0u, 0u};      // [0..3] wg_h, [4..7] wg_h, [8..11] wg_l
        }
        WgA[i] = frag;
    }
    // the epilogue's rows are staged up front (the first barrier publishes them): row 0 the hidden bias times -log2(e), row 1 the
    // per-column multiplier -log2(e) 2^-e_r, then the requested embedding rows.  Padding columns get bias -1e30: Sigmoid exactly 0.
    constexpr int SR = STAGE_FLOATS / ROWS - 2;
    for (int i = tid; i < ROWS; i += T) {
        stage[i] = H2_NL2E * (i < HID2 ? b2[i] : -1.0e30f);     // Sigmoid(x + b) = 1 / (1 + 2^(-L2E x - L2E b))
        stage[ROWS + i] = i < HID2 ? cf[i] : 0.f;
    }
    const int Kc = K < SR ? K : SR;
    for (int k = 0; k < Kc; ++k) {
        const int col = req_col[(int64_t)k * Q + q];
        for (int i = tid; i < ROWS; i += T) stage[ROWS * (2 + k) + i] = (col >= 0 && i < HID2) ? E[(int64_t)col * ld_e + i] : 0.f;
    }

    floatx4 acc[MT][NB16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < NB16; ++i) acc[m][i] = floatx4{0.f, 0.f, 0.f, 0.f};

    // The whole chunk c -> chunk buffer `buf`, requested by the 4 wavefronts of half Y (10 passes of 256 pieces; a wavefront's 64 pieces of a
    // pass are 1 KiB of contiguous LDS: wave-uniform base, the hardware adds lane * 16 bytes)
    // (a wavefront copies a CONTIGUOUS 10 KiB quarter of the chunk, 1 KiB per request, and the requests use the instruction's immediate
    // offset - it advances the global and the LDS address alike - so one pointer / M0 setting serves four requests: three settings per chunk
    // instead of ten.  Y's build tick is paced by its instruction count, and every request used to come with five address instructions.)
    auto dma_chunk = [&](int c, int buf) __attribute__((always_inline)) {
        const int w = wave - 4;
        constexpr int PER_WAVE = H2_PIECES / 4;                                 // 16-byte pieces of a wavefront's quarter (640 = ten requests)
        const u32x4* src = W2h + (int64_t)c * H2_PIECES + w * PER_WAVE + lane;
        u32x4* dst = &Bs[buf * H2_PIECES + w * PER_WAVE];
#pragma unroll
        for (int g = 0; g < PER_WAVE / 64; g += 4) {
#if DFOL_H2_DMA_ASM
            // (inline asm: the compiler does not know these requests, so the barrier that closes the build tick does not drain them - they land
            // under Y's multiply tick, whose closing s_waitcnt vmcnt(0) is theirs)
            const uint32_t d = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(dst + 64 * g);
            const u32x4* sp = src + 64 * g;
            // (ONE asm block per group, M0 in its clobber list: the compiler may write M0 itself - the builtin form at the kernel's top, readlane /
            // movrel - and must neither place such a write between the group's requests nor assume M0 keeps its earlier value: ADVICE r5)
            // (the target keeps M0 reserved - the compiler sets it right before each of its own uses - so clang warns about the clobber; it is still
            // recorded as a definition of M0, which is what the pass that merges identical M0 settings looks at)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(d);
            const int rem = PER_WAVE / 64 - g;                      // requests of this group (compile-time after unrolling): 4, 4, 2
            if (rem >= 4)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072" ::"v"(sp), "s"(m0v) : "memory", "m0");
            else if (rem == 3)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, off offset:2048" ::"v"(sp), "s"(m0v) : "memory", "m0");
            else if (rem == 2)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024" ::"v"(sp), "s"(m0v) : "memory", "m0");
            else
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(sp), "s"(m0v) : "memory", "m0");
#pragma clang diagnostic pop
#else
            auto d = (__attribute__((address_space(3))) void*)(dst + 64 * g);
            __builtin_amdgcn_global_load_lds(src + 64 * g, d, 16, 0, 0);
            if (g + 1 < PER_WAVE / 64) __builtin_amdgcn_global_load_lds(src + 64 * g, d, 16, 1024, 0);
            if (g + 2 < PER_WAVE / 64) __builtin_amdgcn_global_load_lds(src + 64 * g, d, 16, 2048, 0);
            if (g + 3 < PER_WAVE / 64) __builtin_amdgcn_global_load_lds(src + 64 * g, d, 16, 3072, 0);
#endif
        }
    };
    // A pieces of a chunk for the lane's slots: k = 32 c + 8 kh + 0..7.  Two ADJACENT k of one slot form every packed-math pair
    // (U, V and the transposed geometry weights are contiguous in k): no register shuffles.
    floatx4 ru[MT][2], rv[MT][2];                                    // [slot][half]: the lane's 8 first-layer terms of a chunk
    auto load_uv = [&](int c) __attribute__((always_inline)) {
        // (UV is deliberately not a __restrict__ pointer: loads through a noalias readonly pointer are free to move, and the compiler sinks
        // the rows requested at the top of a multiply tick below the tick's closing barrier, to their first use - the prefetch then
        // prefetches nothing; loads that may alias the kernel's stores stay on their side of the barrier's release fence)
        // the chunk's offset stays in a scalar register (readfirstlane: loop strength reduction otherwise turns the four row addresses into
        // 64-bit per-lane induction variables - 8 registers that spill); the pointer keeps its global address space (a pointer rebuilt
        // from an integer is a FLAT one: flat loads also count in lgkmcnt, and the multiply tick's waits for B fragments would wait for them)
        const char* base = img_uv + (uint32_t)__builtin_amdgcn_readfirstlane(H2_CH * 4 * c);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int half = 0; half < 2; ++half) {                  // (k-tile t = half: 16 k = 64 bytes further)
                ru[m][half] = *reinterpret_cast<const floatx4*>(base + uoff[m] + 64 * half);
                rv[m][half] = *reinterpret_cast<const floatx4*>(base + voff[m] + 64 * half);
            }
    };
    u32x4 ap[MT][2];                                                // [slot][piece h, l]
    // A pieces of chunk c.  The first layer's sums z = U[s] + V[o] + Wg geo(s, o) come out of the matrix pipe: per k-tile t (16 k) one MFMA
    // per slot tile with the geometry weights as A (rows = k), the pair geometry as B (columns = slots) and U + V as the C operand - twelve
    // contraction slots hold the three piece products wg_h geo_h + wg_h geo_l + wg_l geo_h.  With the k order of h2_kperm its result
    // registers ARE the lane's A elements of the main product (slot r16, k = 16 t + 4 kh + e): four MFMAs replace 32 packed FMAs and eight LDS
    // reads per chunk.  Then nn.ELU with the saturation - med3(z, e^min(z, 0) - 1, AMAX) = z in (0, AMAX], e^z - 1 for z <= 0 (e^z - 1 >= z),
    // AMAX beyond - and the split.
    auto make_a = [&](int c) __attribute__((always_inline)) {
        floatx4 z[MT][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f16x8 wa = __builtin_bit_cast(f16x8, WgA[(2 * c + t) * 64 + lane]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                z[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, __builtin_bit_cast(f16x8, geoB[m]), ru[m][t] + rv[m][t], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const float z0 = z[m][t][2 * jp], z1 = z[m][t][2 * jp + 1];
                    const float a0 = __builtin_amdgcn_fmed3f(z0, h2_elu_neg(z0), H2_AMAX);
                    const float a1 = __builtin_amdgcn_fmed3f(z1, h2_elu_neg(z1), H2_AMAX);
                    const uint32_t hh = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a0, a1}, f16x2));
                    const uint32_t ll = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){h2_resid<false>(a0, hh), h2_resid<true>(a1, hh)}, f16x2));
                    ap[m][0][2 * t + jp] = hh;
                    ap[m][1][2 * t + jp] = ll;
                    if (TRAIN) z[m][t][2 * jp] = a0 * H2_LN2, z[m][t][2 * jp + 1] = a1 * H2_LN2;       // the activations themselves, back in plain units
                }
        if (TRAIN && !(DFOL_H2T_SKIP & 1)) {
            // the lane's four consecutive k of k-tile t of chunk c (h2_kperm: 32 c + 16 t + 4 kh + 0..3): one 16-byte store; the sixteen lanes
            // kh = 0..3 x t = 0, 1 of a slot cover 128 contiguous bytes of its row per chunk
            char* zimg = reinterpret_cast<char*>(tr.Z + prow0 * HID1) + (uint32_t)__builtin_amdgcn_readfirstlane(H2_CH * 4 * c);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (zoff[m] != 0xffffffffu) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) *reinterpret_cast<floatx4*>(zimg + zoff[m] + 64 * t) = z[m][t];
                }
        }
    };
    const int boff = r16 * 4 + (kh ^ h2_swz(r16));                  // the lane's 16-byte piece inside a 16-row block
    int bbase = boff;                                               // + the chunk buffer's offset
    auto load_b = [&](int i, f16x8 (&b)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p) b[p] = __builtin_bit_cast(f16x8, Bs[bbase + i * 64 + p * H2_ROWS * 4]);
    };
    // The MFMAs of a chunk: the B fragments of tile i+1 are requested before the MFMAs of tile i, and the scheduler may not move anything
    // across tiles (left alone it hoists the reads of all tiles to the top and spills).  The three products of one accumulator are issued
    // back to back, smallest first: a dependent MFMA takes its C operand from the previous result without a register-file read.
    constexpr int PA3[3] = {1, 0, 0}, PB3[3] = {0, 1, 0};           // al wh, ah wl, ah wh
    auto chunk_mfma = [&]() __attribute__((always_inline)) {
        constexpr int D = DFOL_H2_BDEPTH;                           // tiles of B fragments in flight ahead of the MFMAs
        f16x8 bq[D + 1][2];
#pragma unroll
        for (int d = 0; d < D; ++d) load_b(d, bq[d]);
#pragma unroll
        for (int i = 0; i < NB16; ++i) {
            if (i + D < NB16) load_b(i + D, bq[(i + D) % (D + 1)]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int x = 0; x < 3; ++x)
                    acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ap[m][PA3[x]]), bq[i % (D + 1)][PB3[x]], acc[m][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // LDS-DMA completion is tracked by vmcnt of the ISSUING wavefront only; a workgroup barrier does not wait for it by itself
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
    __syncthreads();                                        // WgA, the staged epilogue rows and chunk 0 visible

    // Tick tau: half X (wavefronts 0-3) builds chunk tau/2 on even ticks and multiplies it on the next; half Y (4-7) does the same one
    // tick later.  Chunk c is read in ticks 2c+1 (X) and 2c+2 (Y); its buffer is refilled with chunk c+2 by Y during its build tick 2c+3
    // and Y drains that DMA at the end of its multiply tick 2c+4, one tick before X needs it.  Chunk 1 goes into the free second buffer
    // in Y's idle tick 0.  Each half runs its own copy of the loop (plain straight-line bodies for the register allocator); the barriers
    // pair up by count: X executes 2 per chunk, Y one idle tick first and none after its last multiply.
    auto run_half = [&](auto y_tag) __attribute__((always_inline)) {
        constexpr bool Y = decltype(y_tag)::value;
        TRACE(1);
        if (Y) {
            if (nchunk > 1) dma_chunk(1, 1);
            if (DFOL_H2_PREFETCH) load_uv(0);
            __syncthreads();                                // tick 0: X builds chunk 0
        } else if (DFOL_H2_PREFETCH) {
            load_uv(0);
        }
        TRACE(2);
        for (int c = 0; c < nchunk; ++c) {
            if (!DFOL_H2_PREFETCH) load_uv(c);
            // The U / V rows of this chunk have landed - spelled out, and TIED to the registers: an s_waitcnt whose in-out operands are the
            // rows, so that nothing that reads them can be scheduled above it.  (The compiler's own wait in front of their first use went
            // missing on some paths once the explicit drain at the end of Y's multiply tick covered the loop's back edge, and a bare
            // s_waitcnt builtin does not order the VALU instructions around it: cold launches then built A pieces from registers the loads
            // had not reached - tools/lab/stress_pair.py, tools/lab/diag_pair.py.)  Nothing else is in flight here: it costs nothing.
            if (DFOL_H2_PREFETCH) {
                floatx4 t0 = ru[0][0], t1 = ru[0][1], t2 = ru[1][0], t3 = ru[1][1], t4 = rv[0][0], t5 = rv[0][1], t6 = rv[1][0], t7 = rv[1][1];
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7));
                ru[0][0] = t0, ru[0][1] = t1, ru[1][0] = t2, ru[1][1] = t3, rv[0][0] = t4, rv[0][1] = t5, rv[1][0] = t6, rv[1][1] = t7;
            }
            TRACE(40 + c);
            if (Y && !DFOL_H2_DMA_LATE && c >= 1 && c < lastc) dma_chunk(c + 1, (c + 1) & 1);
            TRACE(50 + c);
            make_a(c);
            // the A pieces are pure register arithmetic: without these fences the compiler sinks them below the barrier, in front of the
            // MFMAs of the multiply tick - the IR-level sinking into the block that uses them (the empty asm pins the values here), and
            // the machine scheduler (sched_barrier)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p) asm volatile("" : "+v"(ap[m][p]));
            __builtin_amdgcn_sched_barrier(0);
            if (Y && DFOL_H2_DMA_LATE && c >= 1 && c < lastc) dma_chunk(c + 1, (c + 1) & 1);
            TRACE(3 + 4 * c);
            // end of the build tick.  (__syncthreads() carries a release fence, for which the compiler drains the chunk request Y has just
            // issued - vmcnt(0) in front of the barrier.  Two ways around that wait were built and measured slower: the bare s_barrier
            // instruction for Y, the DMA landing under its multiply tick (1.27 ms against 1.19: its LDS writes compete with the multiply tick's
            // fragment reads), and THREE chunk buffers with the request two chunks ahead, drained at the top of Y's next build tick
            // (1.28 ms against 1.16: a chunk takes more than a tick - ~3.5 k cycles - to arrive from L2 when every CU streams the 320 KB
            // image, so the wait only moves).)
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            TRACE(4 + 4 * c);
            bbase = boff + (c & 1) * H2_PIECES;
            if (DFOL_H2_PREFETCH && c < lastc) load_uv(c + 1);      // lands under the MFMAs
            chunk_mfma();
            if (Y) __builtin_amdgcn_s_waitcnt(0x0F70);      // the chunk requested in the build tick has landed
            TRACE(5 + 4 * c);
            if (!Y || c < lastc) __syncthreads();           // end of the multiply tick (Y's last one has no partner)
            TRACE(6 + 4 * c);
        }
    };
    // (Round 5 also measured the form WITHOUT ping-pong - all eight wavefronts build a chunk, barrier, all eight multiply it: 1167 - 1184 us
    // against 1144 - 1150; the SIMD's two wavefronts then multiply one after the other - the older one is served first - at 20.6 cycles per
    // MFMA, profiles/r05_pair_h2_trace.txt.)
    if (wave < 4) run_half(std::false_type());
    else run_half(std::true_type());

    // Epilogue: Sigmoid of the hidden layer (the row scale of W2 folded into the exponent's multiplier), dot products with the requested
    // embedding rows (16-lane DPP reduction), LogSigmoid, straight into the [s][o] (or [o][s]) tile the Relate kernel reads.
    const int64_t tile_sz = (int64_t)NS * NS;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < NB16; ++i) {
            const float bv = stage[i * 16 + r16], cm = stage[ROWS + i * 16 + r16];
            if (TRAIN && !(DFOL_H2T_SKIP & 2)) {
                // pre2 = acc 2^-e_r + b2 = -ln 2 (acc cm + bv): column 16 i + r16 of the rows 4 kh + e of slot tile m (sixteen lanes = 64 contiguous
                // bytes of a row; the next column tile's store continues them).  (Staged through the free chunk buffers for 16-byte stores along the
                // rows, two forms: 2.28 and 2.06 ms per launch against 1.97 for these dword stores - tools/lab/time_train_fwd.py.)
                const int colp = i * 16 + r16;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ee = tb * SLOTS + wave * (MT * 16) + m * 16 + 4 * kh + e;
                    // (a scalar base per image + a 32-bit lane offset: the saddr form of the store, no 64-bit multiply per element)
                    if (colp < HID2 && ee < npairs)
                        *reinterpret_cast<float*>(reinterpret_cast<char*>(tr.pre2 + prow0 * tr.ld_pre2) + (uint32_t)(ee * (int)tr.ld_pre2 + colp) * 4u) =
                            -H2_LN2 * fmaf(acc[m][i][e], cm, bv);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[m][i][e] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(acc[m][i][e], cm, bv)));
        }
    for (int k = 0; k < K; ++k) {
        const int col = req_col[(int64_t)k * Q + q];
        if (col < 0) continue;
        float vm[MT];                                           // the logit of slot 4 kh + r16 of slot tile m, in the lanes r16 < 4
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float part[4] = {0.f, 0.f, 0.f, 0.f};
            if (k < Kc) {
                const float* erow = stage + ROWS * (2 + k) + r16;
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
            vm[m] = r16 == 0 ? part[0] : (r16 == 1 ? part[1] : (r16 == 2 ? part[2] : part[3]));
        }
        // One store instruction for the wavefront's 32 slots - consecutive ordered pairs, i.e. mostly consecutive floats of a tile row: the
        // lanes r16 = 4..7 take the second slot tile's logits from the lanes r16 - 4 (DPP row_shr:4), so a kh group writes slots 4 kh + 0..3
        // of both tiles.  (One instruction per slot tile wrote two misaligned 64-byte runs: 23 MB of WRITE_SIZE for 10 MB of tiles.)
        const float shifted = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(vm[1]), 0x114, 0xF, 0xF, false));
        if (r16 < 8) {
            const int hi = r16 >> 2;
            const float v = hi ? shifted : vm[0];
            const int ee = tb * SLOTS + wave * (MT * 16) + hi * 16 + 4 * kh + (r16 & 3);
            if (ee < npairs) {
                // ee / (n - 1) without the integer-division sequence: (ee + 0.5) / (n - 1) is at least 0.5 / (n - 1) away from an integer
                if (TRAIN) {                                          // the reader's raw logit of this pair row (its bias and LogSigmoid stay with the caller)
                    tr.x[(int64_t)k * tr.ld_x + prow0 + ee] = v;
                    continue;
                }
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
    TRACE(60);
}

// The range check of pair_ll32h_kernel, outside it (its build tick is paced by its instruction count: the same test inside cost 4 %).  A pair's
// ELU input z[k] = U[s][k] + V[o][k] + Wg[k] . geo(s, o) saturates above H2_AMAX (units of 1 / ln 2).  Per image and hidden unit k this kernel
// bounds it from above by  max_s U[s][k] + max_o V[o][k] + |Wg[k]| . (dmax, pi / 2, 1, 1),  dmax = the diagonal of the box that holds the
// image's centres - reached by a real pair unless the two maxima belong to the same object - and ORs DFOL_RANGE_PAIR_SATURATED into the
// caller's status word when the bound passes H2_AMAX or is NaN.  One workgroup per image; 16-byte loads, four row groups side by side and four
// rows of each in flight (the pair kernel reads the same rows next, so this pass also warms L2 / MALL for it).
__global__ __launch_bounds__(512) void h2_uv_range_kernel(const float* __restrict__ UV, int64_t ld_uv, int HID1, const float* __restrict__ pos,
                                                          int64_t ld_pos, const float* __restrict__ Wg, const int32_t* __restrict__ n_obj,
                                                          const int32_t* __restrict__ obj_off, uint32_t* __restrict__ status) {
    // thread t: four columns 4 (t % G) .. of the rows t / G, t / G + R, ... (G = 2 HID1 / 4 threads cover a 2 HID1-float row with 16-byte loads,
    // R = 512 / G row groups walk the image's rows side by side: 128 threads x 4 row groups at HID1 = 256)
    __shared__ float4 part[512];
    __shared__ float box[4][8];
    const int q = blockIdx.x, n = n_obj[q], first = obj_off[q], t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (n < 2) return;
    const int G = (2 * HID1) / 4, R = 512 / G, cg = t % G, rg = t / G;          // (HID1 a multiple of 32, <= 256: G in {16, .., 128} divides 512)
    float4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    bool nan = false;
    const float* base = UV + (int64_t)first * ld_uv + 4 * cg;
    auto take = [&](const float4 a) {
        nan |= (a.x != a.x) | (a.y != a.y) | (a.z != a.z) | (a.w != a.w);
        m.x = fmaxf(m.x, a.x), m.y = fmaxf(m.y, a.y), m.z = fmaxf(m.z, a.z), m.w = fmaxf(m.w, a.w);
    };
    int o = rg;
    for (; o + 3 * R < n; o += 4 * R) {
        const float4 a = *reinterpret_cast<const float4*>(base + (int64_t)o * ld_uv), b = *reinterpret_cast<const float4*>(base + (int64_t)(o + R) * ld_uv);
        const float4 c = *reinterpret_cast<const float4*>(base + (int64_t)(o + 2 * R) * ld_uv), d = *reinterpret_cast<const float4*>(base + (int64_t)(o + 3 * R) * ld_uv);
        take(a), take(b), take(c), take(d);
    }
    for (; o < n; o += R) take(*reinterpret_cast<const float4*>(base + (int64_t)o * ld_uv));
    if (nan) m.x = NAN;
    part[t] = m;
    float lox = INFINITY, hix = -INFINITY, loy = INFINITY, hiy = -INFINITY;
    for (int i = t; i < n; i += 512) {
        const float* p = pos + (int64_t)(first + i) * ld_pos;
        const float cx = p[0] + p[2] / 2.0f, cy = p[1] + p[3] / 2.0f;
        lox = fminf(lox, cx), hix = fmaxf(hix, cx), loy = fminf(loy, cy), hiy = fmaxf(hiy, cy);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        lox = fminf(lox, __shfl_xor(lox, s, 64)), hix = fmaxf(hix, __shfl_xor(hix, s, 64));
        loy = fminf(loy, __shfl_xor(loy, s, 64)), hiy = fmaxf(hiy, __shfl_xor(hiy, s, 64));
    }
    if (lane == 0) box[0][wave] = lox, box[1][wave] = hix, box[2][wave] = loy, box[3][wave] = hiy;
    __syncthreads();
    if (t < HID1) {                                                              // hidden unit k = t: U column k, V column HID1 + k
#pragma unroll
        for (int w = 0; w < 8; ++w) lox = fminf(lox, box[0][w]), hix = fmaxf(hix, box[1][w]), loy = fminf(loy, box[2][w]), hiy = fmaxf(hiy, box[3][w]);
        const float* pf = reinterpret_cast<const float*>(part);
        float mu = -INFINITY, mv = -INFINITY;
        bool bad = false;
        for (int r = 0; r < R; ++r) {                                            // (a NaN travels in component x of its thread's partial)
            const float u = pf[(r * G + t / 4) * 4 + (t & 3)], v = pf[(r * G + (HID1 + t) / 4) * 4 + (t & 3)];
            const float ux = pf[(r * G + t / 4) * 4], vx = pf[(r * G + (HID1 + t) / 4) * 4];
            bad |= (ux != ux) | (vx != vx);
            mu = fmaxf(mu, u), mv = fmaxf(mv, v);
        }
        const float dmax = sqrtf((hix - lox) * (hix - lox) + (hiy - loy) * (hiy - loy));
        const float4 g = *reinterpret_cast<const float4*>(Wg + t * 4);
        const float geo = H2_L2E * (fabsf(g.x) * dmax + fabsf(g.y) * 1.57079632679489661923f + fabsf(g.z) + fabsf(g.w));
        const float bound = mu + mv + geo;
        if (bad || !(bound <= H2_AMAX)) atomicOr(status, (uint32_t)DFOL_RANGE_PAIR_SATURATED);
    }
}

}  // namespace

#ifdef DFOL_PAIR_TRACE
extern "C" int dfol_pair_h2_trace_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dfol_h2_trace_buf), sizeof(dfol_h2_trace_buf)); }
#endif

extern "C" int64_t dfol_pair_w2_f16x2_bytes(int32_t HID1) { return (int64_t)(HID1 / H2_CH) * H2_PIECES * 16 + 2 * H2_ROWS * 4; }

extern "C" int dfol_pair_pack_w2_f16x2(const float* W2, int64_t ld_w2, int32_t HID2, int32_t HID1, void* W2_split, void* stream) {
    DFOL_REQUIRE(HID1 > 0 && HID1 <= 256 && HID1 % H2_CH == 0, "pair_pack_w2_f16x2: HID1=%d must be a multiple of %d, <= 256", HID1, H2_CH);
    DFOL_REQUIRE(HID2 > 256 && HID2 <= 320, "pair_pack_w2_f16x2: HID2=%d must be in (256, 320]", HID2);
    DFOL_REQUIRE(W2 && W2_split && ld_w2 >= HID1, "pair_pack_w2_f16x2: null pointer or ld_w2 < HID1");
    DFOL_REQUIRE((uintptr_t)W2_split % 16 == 0, "pair_pack_w2_f16x2: output must be 16-byte aligned");
    const int total = (HID1 / H2_CH) * H2_PIECES;
    float* tail = reinterpret_cast<float*>(reinterpret_cast<u32x4*>(W2_split) + total);
    hipLaunchKernelGGL(h2_row_scale_kernel, dim3(H2_ROWS / 4), dim3(256), 0, (hipStream_t)stream, W2, ld_w2, HID2, HID1, tail);
    hipLaunchKernelGGL(h2_pack_w2_kernel, dim3(dfol_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, W2, ld_w2, HID2, HID1, (const float*)tail,
                       (u32x4*)W2_split);
    DFOL_LAUNCH_CHECK("pair_pack_w2_f16x2");
    return 0;
}

extern "C" int dfol_pair_ll_h2_f32(const float* UV, int64_t ld_uv, int32_t HID1, const float* pos, int64_t ld_pos, const float* Wg,
                                   const void* W2_split, const float* b2, int32_t HID2, const float* E, int64_t ld_e,
                                   const float* be, const int32_t* n_obj, const int32_t* obj_off, int32_t Q, int32_t max_n,
                                   const int32_t* req_col, const int32_t* req_tile, const uint8_t* req_orient, int32_t K,
                                   int32_t NS, float default_ll, int32_t tile_dtype, void* tiles_v, void* stream) {
    DFOL_REQUIRE(tile_dtype == DFOL_TILE_F32 || (tile_dtype == DFOL_TILE_BF16 && NS % 8 == 0), "pair_ll_h2: tile_dtype=%d (bf16 tiles need NS %% 8 == 0)", tile_dtype);
    DFOL_REQUIRE(Q >= 0 && K >= 0 && NS > 0 && NS % 4 == 0 && max_n >= 0 && max_n <= NS, "pair_ll_h2: bad sizes Q=%d K=%d NS=%d max_n=%d", Q, K, NS, max_n);
    DFOL_REQUIRE(HID1 > 0 && HID1 <= 256 && HID1 % H2_CH == 0 && ld_uv % 4 == 0, "pair_ll_h2: HID1=%d must be a multiple of %d, <= 256, UV rows 16-byte aligned", HID1, H2_CH);
    DFOL_REQUIRE(HID2 > 256 && HID2 <= 320, "pair_ll_h2: HID2=%d must be in (256, 320]", HID2);
    if (Q == 0 || K == 0 || max_n < 2) return 0;
    DFOL_REQUIRE(UV && pos && Wg && W2_split && b2 && E && n_obj && obj_off && req_col && req_tile && tiles_v, "pair_ll_h2: null pointer");
    DFOL_REQUIRE(((uintptr_t)UV % 16 == 0) && ((uintptr_t)W2_split % 16 == 0) && ((uintptr_t)Wg % 16 == 0), "pair_ll_h2: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int tpi = dfol_cdiv((int64_t)max_n * (max_n - 1), 256);
    DFOL_REQUIRE((int64_t)Q * tpi < ((int64_t)1 << 31), "pair_ll_h2: too many tiles");
    DFOL_REQUIRE((int64_t)max_n * ld_uv * 4 < ((int64_t)1 << 31), "pair_ll_h2: an image's U / V rows must span less than 2 GB");
    const dim3 grid((unsigned)Q * tpi);
    if (uint32_t* status = dfol_range_status_ptr())           // (dfol_set_range_status: saturated ELU outputs are reported, not answered with)
        hipLaunchKernelGGL(h2_uv_range_kernel, dim3((unsigned)Q), dim3(512), 0, st, UV, ld_uv, HID1, pos, ld_pos, Wg, n_obj, obj_off, status);
    const H2Train none = {nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr};
#define DFOL_PAIR32H(NBV, BF)                                                                                                       \
    hipLaunchKernelGGL((pair_ll32h_kernel<NBV, BF>), grid, dim3(512), 0, st, UV, ld_uv, HID1, pos, ld_pos, Wg, (const u32x4*)W2_split, b2, HID2, \
                       E, ld_e, be, n_obj, obj_off, Q, tpi, req_col, req_tile, req_orient, K, NS, default_ll, tiles_v, none)
    if (HID2 <= 272) { if (tile_dtype == DFOL_TILE_BF16) DFOL_PAIR32H(17, true); else DFOL_PAIR32H(17, false); }
    else if (HID2 <= 288) { if (tile_dtype == DFOL_TILE_BF16) DFOL_PAIR32H(18, true); else DFOL_PAIR32H(18, false); }
    else if (HID2 <= 304) { if (tile_dtype == DFOL_TILE_BF16) DFOL_PAIR32H(19, true); else DFOL_PAIR32H(19, false); }
    else { if (tile_dtype == DFOL_TILE_BF16) DFOL_PAIR32H(20, true); else DFOL_PAIR32H(20, false); }
#undef DFOL_PAIR32H
    DFOL_LAUNCH_CHECK("pair_ll_h2");
    return 0;
}


// The forward of a train step's pair MLP in ONE launch (round 6): per ordered pair of every image, Z = ELU(U[s] + V[o] + Wg geo), pre2 = W2 Z + b2, the pair geometry and, for K reader slots, the raw logits
// x[k][row] = Sigmoid(pre2[row]) . E[req_row[k][image]] (req_row < 0: that image has no reader in slot k; its x entries are left alone).
// Same arithmetic as dfol_pair_ll_h2_f32 (two fp16 pieces per operand, three products, fp32 accumulation).
extern "C" int dfol_pair_train_fwd_h2_f32(const float* UV, int64_t ld_uv, int32_t HID1, const float* pos, int64_t ld_pos, const float* Wg,
                                          const void* W2_split, const float* b2, int32_t HID2, const float* E, int64_t ld_e, const int32_t* n_obj,
                                          const int32_t* obj_off, const int64_t* pair_off, int32_t Q, int32_t max_n, const int32_t* req_row, int32_t K,
                                          float* Z, float* pre2, int64_t ld_pre2, float* geo, float* x, int64_t ld_x, void* stream) {
    DFOL_REQUIRE(Q >= 0 && K >= 0 && max_n >= 0, "pair_train_fwd_h2: bad sizes Q=%d K=%d max_n=%d", Q, K, max_n);
    DFOL_REQUIRE(HID1 > 0 && HID1 <= 256 && HID1 % H2_CH == 0 && ld_uv % 4 == 0, "pair_train_fwd_h2: HID1=%d must be a multiple of %d, <= 256, UV rows 16-byte aligned", HID1, H2_CH);
    DFOL_REQUIRE(HID2 > 256 && HID2 <= 320 && ld_pre2 >= HID2, "pair_train_fwd_h2: HID2=%d must be in (256, 320], ld_pre2 >= HID2", HID2);
    if (Q == 0 || max_n < 2) return 0;
    DFOL_REQUIRE(UV && pos && Wg && W2_split && b2 && n_obj && obj_off && pair_off && Z && pre2 && geo && (K == 0 || (E && req_row && x)), "pair_train_fwd_h2: null pointer");
    DFOL_REQUIRE(((uintptr_t)UV % 16 == 0) && ((uintptr_t)W2_split % 16 == 0) && ((uintptr_t)Wg % 16 == 0) && ((uintptr_t)Z % 16 == 0) && ((uintptr_t)geo % 16 == 0),
                 "pair_train_fwd_h2: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int tpi = dfol_cdiv((int64_t)max_n * (max_n - 1), 256);
    DFOL_REQUIRE((int64_t)Q * tpi < ((int64_t)1 << 31), "pair_train_fwd_h2: too many tiles");
    DFOL_REQUIRE((int64_t)max_n * ld_uv * 4 < ((int64_t)1 << 31) && (int64_t)max_n * (max_n - 1) * HID1 * 4 < ((int64_t)1 << 32) - 1,
                 "pair_train_fwd_h2: an image's rows must span less than 2 GB (U | V) / 4 GB (Z)");
    if (uint32_t* status = dfol_range_status_ptr())
        hipLaunchKernelGGL(h2_uv_range_kernel, dim3((unsigned)Q), dim3(512), 0, st, UV, ld_uv, HID1, pos, ld_pos, Wg, n_obj, obj_off, status);
    const H2Train tr = {Z, pre2, ld_pre2, geo, x, ld_x, pair_off};
    const dim3 grid((unsigned)Q * tpi);
#define DFOL_PAIR32T(NBV)                                                                                                                    \
    hipLaunchKernelGGL((pair_ll32h_kernel<NBV, false, true>), grid, dim3(512), 0, st, UV, ld_uv, HID1, pos, ld_pos, Wg, (const u32x4*)W2_split, b2, HID2, \
                       E, ld_e, (const float*)nullptr, n_obj, obj_off, Q, tpi, req_row, (const int32_t*)nullptr, (const uint8_t*)nullptr, K, 4, 0.f,      \
                       (void*)nullptr, tr)
    if (HID2 <= 272) DFOL_PAIR32T(17);
    else if (HID2 <= 288) DFOL_PAIR32T(18);
    else if (HID2 <= 304) DFOL_PAIR32T(19);
    else DFOL_PAIR32T(20);
#undef DFOL_PAIR32T
    DFOL_LAUNCH_CHECK("pair_train_fwd_h2");
    return 0;
}
