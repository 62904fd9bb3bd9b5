// Training path of the pair MLP (trainer.py:429-442 through classifier_oracle.py:145-156): the element-wise and gather / scatter
// stages around the two tall GEMMs, fused (gfx950).  All four kernels are HBM-bound streams over the per-pair activations
// ([pairs, HID1] and [pairs, HID2] fp32: 2.6 GB and 3.1 GB at 256 x 100 objects); what they replace is ~25 separate passes over
// those tensors (gathers of the per-object halves, adds, ELU, Sigmoid, the embedding product and its row sums, and in the backward
// the scatter-adds of the gathers - atomics - plus the activation derivatives and reductions).
//
//   hidden1_fwd : Z[r, :]   = ELU(U[s(r), :] + V[o(r), :] + Wg geo(r))      geo from the box positions, also written out [pairs, 4]
//   hidden1_bwd : dpre      = dZ * ELU'(pre)  (= dZ * (Z > 0 ? 1 : Z + 1)),   dU[s] = sum_o dpre,  dV[o] = sum_s dpre,  dWg = sum dpre (x) geo
//                 one workgroup per image, no atomics: the sums over o are reduced across the workgroup through LDS, the sums over s
//                 live in registers, the geometry-weight gradient leaves as one partial per image
//   logit_fwd   : x[r]      = sum_j Sigmoid(P2[r, j]) E[p(r), j] + be[p(r)]     p(r): the predicate whose contiguous row range holds r
//   logit_bwd   : dP2[r, j] = dx[r] E[p, j] h (1 - h),   dE[p, j] = sum_r dx[r] h[r, j],   dbe[p] = sum_r dx[r]     (h recomputed)
//                 one workgroup per predicate, no atomics
// Rows are the reference's ordered pairs: image-major, subject-major, the diagonal left out (util.py:87-103).
#include "dfol_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// Sigmoid / ELU on the hardware exp2 / rcp (1 ulp each, absolute error < 2e-7 on these ranges - the forms the inference kernels use).
// With libm's expf / expm1f and an IEEE division these streams were VALU-bound at 2.0 - 3.6 TB/s (25 - 30 instructions per element).
__device__ __forceinline__ float pt_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x)); }
__device__ __forceinline__ float pt_elu(float x) { return x > 0.f ? x : __builtin_amdgcn_exp2f(1.44269504088896340736f * x) - 1.0f; }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
// streams that are written / read once per step and are far larger than the caches (2.6 - 3 GB): non-temporal accesses
typedef float pt_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_stream(float* p, const float4& v) {
    __builtin_nontemporal_store(pt_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<pt_f32x4*>(p));
}
__device__ __forceinline__ float4 ld4_stream(const float* p) {
    const pt_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const pt_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// bf16 storage of the per-pair activations (the bf16 mode, BASELINE configs[3]: Z, pre2 and their gradients are the 14 GB a train step
// moves): the same streams over 2-byte elements, arithmetic in fp32 registers as before.  `pt_bf16` = the bits of a bfloat16.
typedef uint16_t pt_bf16;
typedef uint32_t pt_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pt_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pt_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pt_rne2(float x0, float x1) {                 // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(pt_f32x2{x0, x1}, pt_bf16x2));
}
__device__ __forceinline__ float4 pt_widen(const pt_u32x2& v) {
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ float4 ld4(const pt_bf16* p) { return pt_widen(*reinterpret_cast<const pt_u32x2*>(p)); }
__device__ __forceinline__ float4 ld4_stream(const pt_bf16* p) { return pt_widen(__builtin_nontemporal_load(reinterpret_cast<const pt_u32x2*>(p))); }
__device__ __forceinline__ void st4_stream(pt_bf16* p, const float4& v) {
    __builtin_nontemporal_store(pt_u32x2{pt_rne2(v.x, v.y), pt_rne2(v.z, v.w)}, reinterpret_cast<pt_u32x2*>(p));
}
// the same loads as raw registers (a consumer that keeps loads in flight across a scheduling barrier widens them when it uses them)
template <typename T> struct pt_raw;
template <> struct pt_raw<float> { using type = pt_f32x4; };
template <> struct pt_raw<pt_bf16> { using type = pt_u32x2; };
__device__ __forceinline__ pt_f32x4 ld4_stream_raw(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const pt_f32x4*>(p)); }
__device__ __forceinline__ pt_u32x2 ld4_stream_raw(const pt_bf16* p) { return __builtin_nontemporal_load(reinterpret_cast<const pt_u32x2*>(p)); }
__device__ __forceinline__ float4 pt_widen(const pt_f32x4& v) { return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const pt_bf16* p) { return __uint_as_float((uint32_t)*p << 16); }

__device__ __forceinline__ float4 pair_geometry(const float* ps, const float* po) {      // batch_gqa_boxfeatures_pipeline.py:263-279
    const float x1 = ps[0], y1 = ps[1], w1 = ps[2], h1 = ps[3], x2 = po[0], y2 = po[1], w2 = po[2], h2 = po[3];
    const float dx = x1 + w1 / 2.0f - x2 - w2 / 2.0f, dy = y1 + h1 / 2.0f - y2 - h2 / 2.0f;
    const float dist = sqrtf(dx * dx + dy * dy);
    return make_float4(dist, asinf(dy / fmaxf(dist, 1e-10f)), (x2 - x1 > 0.f) ? 1.f : ((x2 - x1 < 0.f) ? -1.f : 0.f),
                       (y2 - y1 > 0.f) ? 1.f : ((y2 - y1 < 0.f) ? -1.f : 0.f));
}

// geometry of every ordered pair, one thread per pair (the per-pair arithmetic - a square root, an arc sine, two divisions - would
// otherwise be repeated by every wavefront that touches the pair's row of Z)
__global__ __launch_bounds__(256) void pair_geometry_kernel(const float* __restrict__ pos, int64_t ld_pos, const int32_t* __restrict__ obj_off,
                                                            const int64_t* __restrict__ pair_off, const int32_t* __restrict__ n_obj,
                                                            float* __restrict__ geo_out) {
    const int q = blockIdx.y, n = n_obj[q];
    const int e = blockIdx.x * 256 + (int)threadIdx.x;
    if (n < 2 || e >= n * (n - 1)) return;
    const int s = e / (n - 1), oo = e - s * (n - 1), o = oo + (oo >= s), first = obj_off[q];
    st4(geo_out + (pair_off[q] + e) * 4, pair_geometry(pos + (int64_t)(first + s) * ld_pos, pos + (int64_t)(first + o) * ld_pos));
}

// grid (tiles_per_image, Q); 256 threads = 256 / (H1 / 4) row slots of H1 / 4 lanes; a slot walks HF_ROWS consecutive pairs of one image
// (four at a time: their loads are issued together), a lane owns 4 hidden units whose geometry weights stay in registers
constexpr int HF_ROWS = 16;

template <typename TZ>
__global__ __launch_bounds__(256) void pair_hidden1_fwd_kernel(const float* __restrict__ U, int64_t ld_u, const float* __restrict__ V,
                                                                int64_t ld_v, const float* __restrict__ Wg, const int32_t* __restrict__ obj_off,
                                                                const int64_t* __restrict__ pair_off, const int32_t* __restrict__ n_obj,
                                                                int H1, TZ* __restrict__ Z, const float* __restrict__ geo) {
    const int q = blockIdx.y, n = n_obj[q], lpr = H1 >> 2, slots = 256 / lpr;
    const int slot = (int)threadIdx.x / lpr, k = ((int)threadIdx.x % lpr) * 4;
    const int rows = n * (n - 1);
    const int e0 = (blockIdx.x * slots + slot) * HF_ROWS;
    if (n < 2 || e0 >= rows) return;
    const int first = obj_off[q];
    const int64_t base = pair_off[q];
    float4 w[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) w[t] = ld4(Wg + (int64_t)(k + t) * 4);
    // the slot's HF_ROWS consecutive pairs have at most two subjects when an image has more than HF_ROWS objects: their U rows are loaded
    // once per slot instead of once per pair (a third of the kernel's L2 reads, which - not the 2.6 GB it writes - is what bounds it)
    const bool two = n - 1 >= HF_ROWS;
    const int sA = e0 / (n - 1);
    float4 uA, uB;
    if (two) {
        uA = ld4(U + (int64_t)(first + sA) * ld_u + k);
        uB = ld4(U + (int64_t)(first + min(sA + 1, n - 1)) * ld_u + k);
    }
    for (int r0 = 0; r0 < HF_ROWS; r0 += 4) {
        float4 g[4], u[4], v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                            // all loads of four rows first (rows beyond the image are clamped)
            const int e = min(e0 + r0 + i, rows - 1);
            const int s = e / (n - 1), oo = e - s * (n - 1), o = oo + (oo >= s);
            g[i] = ld4(geo + (base + e) * 4);
            if (two) u[i] = s == sA ? uA : uB;
            else u[i] = ld4(U + (int64_t)(first + s) * ld_u + k);
            v[i] = ld4(V + (int64_t)(first + o) * ld_v + k);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + r0 + i;
            if (e >= rows) break;
            const float z[4] = {u[i].x + v[i].x, u[i].y + v[i].y, u[i].z + v[i].z, u[i].w + v[i].w};
            float out[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) out[t] = pt_elu(z[t] + (w[t].x * g[i].x + w[t].y * g[i].y + w[t].z * g[i].z + w[t].w * g[i].w));      // nn.ELU
            st4_stream(Z + (base + e) * H1 + k, make_float4(out[0], out[1], out[2], out[3]));
        }
    }
}

template <int N, int I = 0, typename F>
__device__ __forceinline__ void hb_unroll(F&& f) {              // f(integral_constant<0>) ... f(integral_constant<N - 1>)
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        hb_unroll<N, I + 1>(f);
    }
}

constexpr int HB_MAXO = 16;                                   // objects per lane group in hidden1_bwd

// One workgroup (THREADS = G groups of H1 / 4 lanes) per image; a group's lanes walk the objects o = g, g + G, ... (MAXO slots) of every
// subject s.  THREADS = 1024, BATCH = 1: a slot's loads sit inside the branch that tests the slot, i.e. every slot is a full memory round
// trip (s_waitcnt vmcnt(0) in each: seven in a row per subject at 100 objects) and only the 16 wavefronts of the CU overlap them; 128
// registers per lane leave no room for more (batching the loads there spills in the loop: 1.95 ms against 0.8).  THREADS = 512 (images of up
// to MAXO * G = 128 objects at HID1 = 256): 256 registers per lane, the loads of BATCH slots are issued together without branches (rows that do
// not exist are clamped to one that does and masked out of the sums).  Recomputing z from U[s], V[o] and the geometry instead of reading it
// (half the HBM bytes) did not pay while every batch was a full round trip (round 3: no faster in fp32 storage, 0.15 ms slower per step in
// bf16 storage); with the loads pipelined the fp32 pass sits at 5.9 - 6.0 TB/s and the rebuilt form (RECOMP below) takes 664 against 864 us
// at 256 x 100 objects, HID1 = 256 - bound by its VALU work (about 24 lane operations per element), like the bf16-storage pass (658 us).
// RECOMP (fp32 storage, BATCH > 1): Z is not read.  z = ELU(U[s] + V[o] + Wg geo) is rebuilt with the forward kernel's own expression from the
// image's V rows (staged in LDS once: 128 KB at most), the subject's U row, the lane's four rows of Wg and the geometry the pass reads
// anyway: the pass reads dZ alone, half the bytes.
struct HbRecompute {
    const float* U; int64_t ld_u;
    const float* V; int64_t ld_v;
    const float* Wg;
};
template <typename TZ, int THREADS, int MAXO, int BATCH, bool RECOMP = false>
__global__ __launch_bounds__(THREADS) void pair_hidden1_bwd_kernel(const TZ* __restrict__ dZ, const TZ* __restrict__ Z,
                                                                    const float* __restrict__ geo, const int32_t* __restrict__ obj_off,
                                                                    const int64_t* __restrict__ pair_off, const int32_t* __restrict__ n_obj,
                                                                    int H1, float* __restrict__ dU, int64_t ld_du, float* __restrict__ dV,
                                                                    int64_t ld_dv, float* __restrict__ dWg_partial, HbRecompute rc) {
    static_assert(!RECOMP || (BATCH > 1 && sizeof(TZ) == 4), "hidden1_bwd: z is rebuilt in the batched fp32 form only");
    extern __shared__ __attribute__((aligned(16))) float red[];            // [2][G][H1] floats (+ RECOMP: [n][H1] rows of V)
    const int q = blockIdx.x, n = n_obj[q], lpr = H1 >> 2, G = THREADS / lpr;
    const int g = (int)threadIdx.x / lpr, k = ((int)threadIdx.x % lpr) * 4, first = obj_off[q];
    const int64_t base = pair_off[q];
    float4 dv[MAXO];
    float dwg[4][4];
#pragma unroll
    for (int i = 0; i < MAXO; ++i) dv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int d = 0; d < 4; ++d) dwg[t][d] = 0.f;
    auto add = [&](int i, const float4& dz, const float4& z, const float4& ge, float4& du) __attribute__((always_inline)) {
        const float4 dp = make_float4(dz.x * (z.x > 0.f ? 1.f : z.x + 1.f), dz.y * (z.y > 0.f ? 1.f : z.y + 1.f),
                                      dz.z * (z.z > 0.f ? 1.f : z.z + 1.f), dz.w * (z.w > 0.f ? 1.f : z.w + 1.f));
        du.x += dp.x, du.y += dp.y, du.z += dp.z, du.w += dp.w;
        dv[i].x += dp.x, dv[i].y += dp.y, dv[i].z += dp.z, dv[i].w += dp.w;
        const float gd[4] = {ge.x, ge.y, ge.z, ge.w}, dpv[4] = {dp.x, dp.y, dp.z, dp.w};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int d = 0; d < 4; ++d) dwg[t][d] = fmaf(dpv[t], gd[d], dwg[t][d]);
    };
    // the subject's sum over the groups (= over o) in a fixed order; `red` is double-buffered over s: one barrier per subject
    auto reduce_du = [&](int s, const float4& du) __attribute__((always_inline)) {
        float* r = red + (s & 1) * G * H1;
        st4(&r[g * H1 + k], du);
        __syncthreads();
        if (g == 0) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j = 0; j < G; ++j) {
                const float4 p = ld4(&r[j * H1 + k]);
                acc.x += p.x, acc.y += p.y, acc.z += p.z, acc.w += p.w;
            }
            st4(dU + (int64_t)(first + s) * ld_du + k, acc);
        }
    };
    if constexpr (BATCH == 1) {
        for (int s = 0; s < n; ++s) {
            float4 du = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < MAXO; ++i) {
                const int o = g + i * G;
                if (o < n && o != s) {
                    const int64_t row = base + (int64_t)s * (n - 1) + (o - (o > s));
                    add(i, ld4_stream(dZ + row * H1 + k), ld4_stream(Z + row * H1 + k), ld4(geo + row * 4), du);
                }
            }
            reduce_du(s, du);
        }
    } else {
        // Two register sets of BATCH slots: while one batch of (subject, slots) is summed, the loads of the NEXT batch are in flight and
        // the loads of the one after are issued right behind the sums - the walk over (s, batch) is one sequence whose position parity
        // names the set, so everything stays in registers (NBC = the batches that hold an object of this image; an odd NBC walks two
        // subjects per pass).  Before: issue, wait, sum per batch - the loaded latency of HBM 2 n times in a row per image.  Measured,
        // 256 images, HID1 = 256 (tools/lab/time_hidden1.py): 36 objects 168 -> 138 us fp32, 142 -> 109 us bf16 storage; 100 objects
        // 889 -> 889 us fp32 (5.9 TB/s read either way: the stream is at the chip's bandwidth there), 685 -> 696 us bf16.
        typename pt_raw<TZ>::type dzr[2][BATCH], zr[RECOMP ? 1 : 2][RECOMP ? 1 : BATCH];
        float4 ger[2][BATCH], ur[2], wg4[4];
        float* vs = red + 2 * G * H1;
        if constexpr (RECOMP) {
#pragma unroll
            for (int t = 0; t < 4; ++t) wg4[t] = ld4(rc.Wg + (k + t) * 4);
            for (int o = g; o < n; o += G) st4(&vs[o * H1 + k], ld4(rc.V + (int64_t)(first + o) * rc.ld_v + k));
            __syncthreads();
        }
        auto issue = [&](auto SET, auto BI, int s) __attribute__((always_inline)) {
            if constexpr (RECOMP) ur[decltype(SET)::value] = ld4(rc.U + (int64_t)(first + s) * rc.ld_u + k);
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int slot = decltype(BI)::value * BATCH + j;       // (static once unrolled)
                if (slot * G < n) {                               // (workgroup-uniform: the slot holds an object of this image)
                    const int o = g + slot * G;
                    // (a lane without a pair in this slot reads the image's first pair row; n >= 2 here: slot 0 of n == 1 has s == o)
                    const int64_t row = (o < n && o != s) ? base + (int64_t)s * (n - 1) + (o - (o > s)) : (n >= 2 ? base : 0);
                    dzr[decltype(SET)::value][j] = ld4_stream_raw(dZ + row * H1 + k);
                    if constexpr (!RECOMP) zr[decltype(SET)::value][j] = ld4_stream_raw(Z + row * H1 + k);
                    ger[decltype(SET)::value][j] = ld4(geo + row * 4);
                }
            }
        };
        auto consume = [&](auto SET, auto BI, int s, float4& du) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int slot = decltype(BI)::value * BATCH + j;       // (static once unrolled)
                if (slot * G < n) {
                    const int o = g + slot * G;
                    const float m = (o < n && o != s) ? 1.f : 0.f;                  // (a lane without a pair adds exact zeros)
                    const float4 d = pt_widen(dzr[decltype(SET)::value][j]), ge = ger[decltype(SET)::value][j];
                    float4 z;
                    if constexpr (RECOMP) {                      // (pair_hidden1_fwd_kernel's expression, term for term)
                        const float4 u = ur[decltype(SET)::value], v = ld4(&vs[min(o, n - 1) * H1 + k]);
                        const float uv[4] = {u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w};
                        float out[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            out[t] = pt_elu(uv[t] + (wg4[t].x * ge.x + wg4[t].y * ge.y + wg4[t].z * ge.z + wg4[t].w * ge.w));
                        z = make_float4(out[0], out[1], out[2], out[3]);
                    } else {
                        z = pt_widen(zr[decltype(SET)::value][j]);
                    }
                    add(slot, make_float4(m * d.x, m * d.y, m * d.z, m * d.w), z, ge, du);
                }
            }
        };
        auto walk = [&](auto NBC_) __attribute__((always_inline)) {
            constexpr int NBC = decltype(NBC_)::value, SU = (NBC & 1) ? 2 : 1, LEN = SU * NBC;
            static_assert(LEN % 2 == 0 && NBC * BATCH <= MAXO, "hidden1_bwd: the walk alternates two register sets");
            if (n == 1) {                                         // (no pair rows at all: nothing to read)
                reduce_du(0, make_float4(0.f, 0.f, 0.f, 0.f));
                return;
            }
            issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
            if (1 / NBC < n) issue(std::integral_constant<int, 1>{}, std::integral_constant<int, 1 % NBC>{}, 1 / NBC);
            float4 du = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s0 = 0; s0 < n; s0 += SU) {
                hb_unroll<LEN>([&](auto P) __attribute__((always_inline)) {
                    constexpr int p = decltype(P)::value, bi = p % NBC, p2 = p + 2;
                    const int s = s0 + p / NBC, s2 = s0 + p2 / NBC;
                    if (s < n) consume(std::integral_constant<int, p & 1>{}, std::integral_constant<int, bi>{}, s, du);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s2 < n) issue(std::integral_constant<int, p & 1>{}, std::integral_constant<int, p2 % NBC>{}, s2);
                    __builtin_amdgcn_sched_barrier(0);
                    if (bi == NBC - 1 && s < n) {
                        reduce_du(s, du);
                        du = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                });
            }
        };
        const int nbc = (n + G * BATCH - 1) / (G * BATCH);
        static_assert(MAXO / BATCH == 4, "hidden1_bwd: four batches of slots");
        if (nbc <= 1) walk(std::integral_constant<int, 1>{});
        else if (nbc == 2) walk(std::integral_constant<int, 2>{});
        else if (nbc == 3) walk(std::integral_constant<int, 3>{});
        else walk(std::integral_constant<int, 4>{});
    }
    __syncthreads();                                             // (the last subject's sum is read before `red` is reused below)
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
        const int o = g + i * G;
        if (o < n) st4(dV + (int64_t)(first + o) * ld_dv + k, dv[i]);
    }
    // geometry-weight gradient of this image: [H1, 4], reduced over the groups through the same LDS array, four columns at a time
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        st4(&red[g * H1 + k], make_float4(dwg[0][d], dwg[1][d], dwg[2][d], dwg[3][d]));
        __syncthreads();
        if (g == 0) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j = 0; j < G; ++j) {
                const float4 p = ld4(&red[j * H1 + k]);
                acc.x += p.x, acc.y += p.y, acc.z += p.z, acc.w += p.w;
            }
            float* out = dWg_partial + (int64_t)q * H1 * 4;
            out[(k + 0) * 4 + d] = acc.x, out[(k + 1) * 4 + d] = acc.y, out[(k + 2) * 4 + d] = acc.z, out[(k + 3) * 4 + d] = acc.w;
        }
        __syncthreads();
    }
}

#ifndef LGB_RF16
#define LGB_RF16 8
#endif
#ifndef LGB_RF32
#define LGB_RF32 8
#endif
constexpr int LG_T = 8;                                       // hidden units per lane: H2 <= 512

// grid (row tiles, P): a workgroup covers 16 consecutive rows of one predicate, a wavefront 4 of them (four independent load streams)
template <typename TP>
__global__ __launch_bounds__(256) void pair_logit_fwd_kernel(const TP* __restrict__ P2, int64_t ld_p2, int H2,
                                                              const float* __restrict__ E, int64_t ld_e, const float* __restrict__ be,
                                                              const int64_t* __restrict__ pred_off, float* __restrict__ x) {
    const int p = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r1 = pred_off[p + 1], row0 = pred_off[p] + (int64_t)blockIdx.x * 16 + wave * 4;
    if (row0 >= r1) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < LG_T; ++t) {
        const int j = lane + 64 * t;
        if (j < H2) {
            const float ev = E[(int64_t)p * ld_e + j];
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ld1(P2 + min(row0 + i, r1 - 1) * ld_p2 + j);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(pt_sigmoid(v[i]), ev, acc[i]);
        }
    }
    const float b = be ? be[p] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float sum = dfol_wave_sum(acc[i]);
        if (lane == 0 && row0 + i < r1) x[row0 + i] = sum + b;
    }
}

// The same with four consecutive hidden units per lane and load (H2 % 4 == 0; the bf16 storage's form: 8-byte loads)
template <typename TP>
__global__ __launch_bounds__(256) void pair_logit_fwd4_kernel(const TP* __restrict__ P2, int64_t ld_p2, int H2,
                                                               const float* __restrict__ E, int64_t ld_e, const float* __restrict__ be,
                                                               const int64_t* __restrict__ pred_off, float* __restrict__ x) {
    const int p = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, F4 = H2 >> 2;
    const int64_t r1 = pred_off[p + 1], row0 = pred_off[p] + (int64_t)blockIdx.x * 16 + wave * 4;
    if (row0 >= r1) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < LG_T / 4; ++t) {
        const int j4 = lane + 64 * t;
        if (j4 < F4) {
            const float4 ev = ld4(E + (int64_t)p * ld_e + 4 * j4);
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ld4_stream(P2 + min(row0 + i, r1 - 1) * ld_p2 + 4 * j4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = fmaf(pt_sigmoid(v[i].x), ev.x, fmaf(pt_sigmoid(v[i].y), ev.y, fmaf(pt_sigmoid(v[i].z), ev.z, fmaf(pt_sigmoid(v[i].w), ev.w, acc[i]))));
        }
    }
    const float b = be ? be[p] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float sum = dfol_wave_sum(acc[i]);
        if (lane == 0 && row0 + i < r1) x[row0 + i] = sum + b;
    }
}

// One workgroup (1024 threads) per predicate: rows pred_off[p] .. pred_off[p+1].  H2 % 4 == 0: the predicate's rows are one contiguous
// array of float4 (F4 = H2 / 4 per row); RG = 1024 / F4 rows are covered per trip by RG * F4 active threads, thread t always on
// float4 t % F4 of its row, so its share of dE stays in four registers and every load / store of the workgroup is one contiguous
// block (975 x 16 bytes for H2 = 300).  Two trips are in flight.  The RG row groups are combined through LDS in a fixed order.
template <typename TP>
__global__ __launch_bounds__(1024) void pair_logit_bwd4_kernel(const float* __restrict__ dx, const TP* __restrict__ P2, int64_t ld_p2,
                                                                int H2, const float* __restrict__ E, int64_t ld_e,
                                                                const int64_t* __restrict__ pred_off, TP* __restrict__ dP2,
                                                                int64_t ld_dp2, float* __restrict__ dE, int64_t ld_de,
                                                                float* __restrict__ dbe, float* __restrict__ dB2, int64_t ld_db2) {
    extern __shared__ __attribute__((aligned(16))) float red4[];           // [RG][H2] floats (+ RG for the bias)
    const int p = blockIdx.x, F4 = H2 >> 2, RG = 1024 / F4, tid = threadIdx.x;
    const int rg = tid / F4, j4 = tid - rg * F4;
    const bool act = rg < RG;
    const int64_t r0 = pred_off[p], r1 = pred_off[p + 1];
    const float4 ev = act ? ld4(E + (int64_t)p * ld_e + 4 * j4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 de = make_float4(0.f, 0.f, 0.f, 0.f), d2 = make_float4(0.f, 0.f, 0.f, 0.f);
    float db = 0.f;
    auto one = [&](int64_t row, const float4& v, float g) __attribute__((always_inline)) {
        const float h[4] = {pt_sigmoid(v.x), pt_sigmoid(v.y), pt_sigmoid(v.z), pt_sigmoid(v.w)};
        const float4 dp = make_float4(g * ev.x * h[0] * (1.0f - h[0]), g * ev.y * h[1] * (1.0f - h[1]), g * ev.z * h[2] * (1.0f - h[2]),
                                      g * ev.w * h[3] * (1.0f - h[3]));
        if (dP2) st4_stream(dP2 + row * ld_dp2 + 4 * j4, dp);          // (null: dpre2 is produced inside the two products that read it)
        if (dB2) d2.x += dp.x, d2.y += dp.y, d2.z += dp.z, d2.w += dp.w;          // this predicate's part of the second layer's bias gradient
        de.x = fmaf(g, h[0], de.x), de.y = fmaf(g, h[1], de.y), de.z = fmaf(g, h[2], de.z), de.w = fmaf(g, h[3], de.w);
        if (j4 == 0) db += g;
    };
    if (act) {
        constexpr int RF = sizeof(TP) == 2 ? LGB_RF16 : LGB_RF32;       // rows in flight per thread (half the bytes per row in bf16: twice the rows)
        for (int64_t row = r0 + rg; row < r1; row += RF * RG) {          // (rows past the end: clamped loads)
            float4 v[RF];
            float g[RF];
#pragma unroll
            for (int i = 0; i < RF; ++i) {
                const int64_t r = min(row + (int64_t)i * RG, r1 - 1);
                v[i] = ld4_stream(P2 + r * ld_p2 + 4 * j4);
                g[i] = dx[r];
            }
#pragma unroll
            for (int i = 0; i < RF; ++i)
                if (row + (int64_t)i * RG < r1) one(row + (int64_t)i * RG, v[i], g[i]);
        }
        st4(&red4[rg * H2 + 4 * j4], de);
        if (j4 == 0) red4[RG * H2 + rg] = db;
    }
    __syncthreads();
    for (int j = tid; j < H2; j += 1024) {
        float acc = 0.f;
        for (int w = 0; w < RG; ++w) acc += red4[w * H2 + j];
        dE[(int64_t)p * ld_de + j] = acc;
    }
    if (tid == 0 && dbe) {
        float acc = 0.f;
        for (int w = 0; w < RG; ++w) acc += red4[RG * H2 + w];
        dbe[p] = acc;
    }
    if (dB2) {                                                         // the same fixed-order reduction for the column sums of dpre2
        __syncthreads();
        if (act) st4(&red4[rg * H2 + 4 * j4], d2);
        __syncthreads();
        for (int j = tid; j < H2; j += 1024) {
            float acc = 0.f;
            for (int w = 0; w < RG; ++w) acc += red4[w * H2 + j];
            dB2[(int64_t)p * ld_db2 + j] = acc;
        }
    }
}

// (any H2 <= 512: one float per lane and load)
__global__ __launch_bounds__(1024) void pair_logit_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ P2, int64_t ld_p2,
                                                               int H2, const float* __restrict__ E, int64_t ld_e,
                                                               const int64_t* __restrict__ pred_off, float* __restrict__ dP2,
                                                               int64_t ld_dp2, float* __restrict__ dE, int64_t ld_de,
                                                               float* __restrict__ dbe) {
    __shared__ float red[16][64 * LG_T + 1];
    const int p = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = pred_off[p], r1 = pred_off[p + 1];
    float ev[LG_T], de[LG_T], db = 0.f;
#pragma unroll
    for (int t = 0; t < LG_T; ++t) {
        const int j = lane + 64 * t;
        ev[t] = j < H2 ? E[(int64_t)p * ld_e + j] : 0.f;
        de[t] = 0.f;
    }
    for (int64_t row = r0 + wave; row < r1; row += 32) {     // two rows per trip: twice the loads in flight
        const int64_t row_b = row + 16;
        const bool has_b = row_b < r1;
        const float ga = dx[row], gb = has_b ? dx[row_b] : 0.f;
        db += ga + gb;
        float va[LG_T], vb[LG_T];
#pragma unroll
        for (int t = 0; t < LG_T; ++t) {
            const int j = lane + 64 * t;
            va[t] = j < H2 ? P2[row * ld_p2 + j] : 0.f;
            vb[t] = (j < H2 && has_b) ? P2[row_b * ld_p2 + j] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < LG_T; ++t) {
            const int j = lane + 64 * t;
            if (j < H2) {
                const float ha = pt_sigmoid(va[t]), hb = pt_sigmoid(vb[t]);
                if (dP2) dP2[row * ld_dp2 + j] = ga * ev[t] * ha * (1.0f - ha);
                if (dP2 && has_b) dP2[row_b * ld_dp2 + j] = gb * ev[t] * hb * (1.0f - hb);
                de[t] = fmaf(ga, ha, fmaf(gb, hb, de[t]));
            }
        }
    }
#pragma unroll
    for (int t = 0; t < LG_T; ++t) red[wave][lane + 64 * t] = de[t];
    if (lane == 0) red[wave][64 * LG_T] = db;
    __syncthreads();
    for (int j = threadIdx.x; j < H2; j += 1024) {
        float acc = 0.f;
        for (int w = 0; w < 16; ++w) acc += red[w][j];
        dE[(int64_t)p * ld_de + j] = acc;
    }
    if (threadIdx.x == 0 && dbe) {
        float acc = 0.f;
        for (int w = 0; w < 16; ++w) acc += red[w][64 * LG_T];
        dbe[p] = acc;
    }
}

bool hidden_width_ok(int H1) { return H1 >= 16 && H1 <= 1024 && H1 % 4 == 0 && 1024 % (H1 / 4) == 0 && 256 % (H1 / 4) == 0; }

}  // namespace

template <typename TZ>
static int hidden1_fwd_launch(const float* U, int64_t ld_u, const float* V, int64_t ld_v, const float* pos, int64_t ld_pos, const float* Wg,
                              const int32_t* obj_off, const int64_t* pair_off, const int32_t* n_obj, int32_t Q, int32_t max_n, int32_t H1, TZ* Z,
                              float* geo, void* stream) {
    DFOL_REQUIRE(hidden_width_ok(H1), "pair_hidden1_fwd: HID1=%d must be 16..1024 with HID1/4 a power of two <= 256", H1);
    DFOL_REQUIRE(ld_u % 4 == 0 && ld_v % 4 == 0 && Q >= 0 && max_n >= 0, "pair_hidden1_fwd: rows of U and V must be 16-byte aligned");
    if (Q == 0 || max_n < 2) return 0;
    DFOL_REQUIRE(U && V && pos && Wg && obj_off && pair_off && n_obj && Z && geo, "pair_hidden1_fwd: null pointer");
    const int rows_per_block = 256 / (H1 / 4);
    hipLaunchKernelGGL(pair_geometry_kernel, dim3(dfol_cdiv((int64_t)max_n * (max_n - 1), 256), Q), dim3(256), 0, (hipStream_t)stream, pos, ld_pos,
                       obj_off, pair_off, n_obj, geo);
    DFOL_LAUNCH_CHECK("pair_hidden1_fwd (geometry)");
    const dim3 grid(dfol_cdiv((int64_t)max_n * (max_n - 1), rows_per_block * HF_ROWS), Q);
    hipLaunchKernelGGL(pair_hidden1_fwd_kernel<TZ>, grid, dim3(256), 0, (hipStream_t)stream, U, ld_u, V, ld_v, Wg, obj_off, pair_off, n_obj, H1, Z,
                       geo);
    DFOL_LAUNCH_CHECK("pair_hidden1_fwd");
    return 0;
}

extern "C" int dfol_pair_hidden1_fwd_f32(const float* U, int64_t ld_u, const float* V, int64_t ld_v, const float* pos, int64_t ld_pos,
                                         const float* Wg, const int32_t* obj_off, const int64_t* pair_off, const int32_t* n_obj, int32_t Q,
                                         int32_t max_n, int32_t H1, float* Z, float* geo, void* stream) {
    return hidden1_fwd_launch<float>(U, ld_u, V, ld_v, pos, ld_pos, Wg, obj_off, pair_off, n_obj, Q, max_n, H1, Z, geo, stream);
}
extern "C" int dfol_pair_hidden1_fwd_bf16(const float* U, int64_t ld_u, const float* V, int64_t ld_v, const float* pos, int64_t ld_pos,
                                          const float* Wg, const int32_t* obj_off, const int64_t* pair_off, const int32_t* n_obj, int32_t Q,
                                          int32_t max_n, int32_t H1, void* Z_bf16, float* geo, void* stream) {
    return hidden1_fwd_launch<pt_bf16>(U, ld_u, V, ld_v, pos, ld_pos, Wg, obj_off, pair_off, n_obj, Q, max_n, H1, (pt_bf16*)Z_bf16, geo, stream);
}

template <typename TZ>
static int hidden1_bwd_launch(const TZ* dZ, const TZ* Z, const float* geo, const int32_t* obj_off, const int64_t* pair_off, const int32_t* n_obj,
                              int32_t Q, int32_t max_n, int32_t H1, float* dU, int64_t ld_du, float* dV, int64_t ld_dv, float* dWg_partial,
                              void* stream) {
    DFOL_REQUIRE(hidden_width_ok(H1), "pair_hidden1_bwd: HID1=%d must be 16..1024 with HID1/4 a power of two <= 256", H1);
    const int G = 1024 / (H1 / 4);
    DFOL_REQUIRE(max_n <= HB_MAXO * G, "pair_hidden1_bwd: max_n=%d exceeds %d objects per image at HID1=%d", max_n, HB_MAXO * G, H1);
    DFOL_REQUIRE(ld_du % 4 == 0 && ld_dv % 4 == 0, "pair_hidden1_bwd: rows of dU and dV must be 16-byte aligned");
    if (Q == 0) return 0;
    DFOL_REQUIRE(dZ && Z && geo && obj_off && pair_off && n_obj && dU && dV && dWg_partial, "pair_hidden1_bwd: null pointer");
    // the 512-thread form when the image fits its object slots and the width gives it whole wavefront groups (DFOL_H1B_THREADS=1024 forces the other)
    static const int force = getenv("DFOL_H1B_THREADS") ? atoi(getenv("DFOL_H1B_THREADS")) : 0;
    const int G5 = 512 / (H1 / 4);
    if (force != 1024 && H1 / 4 <= 512 && G5 >= 1 && max_n <= HB_MAXO * G5)
        hipLaunchKernelGGL((pair_hidden1_bwd_kernel<TZ, 512, HB_MAXO, 4>), dim3(Q), dim3(512), (size_t)2 * G5 * H1 * sizeof(float), (hipStream_t)stream, dZ, Z,
                           geo, obj_off, pair_off, n_obj, H1, dU, ld_du, dV, ld_dv, dWg_partial, HbRecompute{});
    else
        hipLaunchKernelGGL((pair_hidden1_bwd_kernel<TZ, 1024, HB_MAXO, 1>), dim3(Q), dim3(1024), (size_t)2 * G * H1 * sizeof(float), (hipStream_t)stream, dZ, Z,
                           geo, obj_off, pair_off, n_obj, H1, dU, ld_du, dV, ld_dv, dWg_partial, HbRecompute{});
    DFOL_LAUNCH_CHECK("pair_hidden1_bwd");
    return 0;
}

// The same sums WITHOUT reading Z (fp32 storage): z is rebuilt from U, V, Wg and the geometry - see HbRecompute.  Images of up to
// 16 * (512 / (HID1 / 4)) objects (the 512-thread form: dfol_pair_hidden1_bwd_recompute_supported).
extern "C" int dfol_pair_hidden1_bwd_recompute_supported(int32_t max_n, int32_t H1) {
    return hidden_width_ok(H1) && H1 / 4 <= 512 && max_n <= HB_MAXO * (512 / (H1 / 4));
}
extern "C" int dfol_pair_hidden1_bwd_recompute_f32(const float* dZ, const float* U, int64_t ld_u, const float* V, int64_t ld_v, const float* Wg,
                                                   const float* geo, const int32_t* obj_off, const int64_t* pair_off, const int32_t* n_obj,
                                                   int32_t Q, int32_t max_n, int32_t H1, float* dU, int64_t ld_du, float* dV, int64_t ld_dv,
                                                   float* dWg_partial, void* stream) {
    DFOL_REQUIRE(dfol_pair_hidden1_bwd_recompute_supported(max_n, H1), "pair_hidden1_bwd_recompute: HID1=%d, max_n=%d not supported", H1, max_n);
    DFOL_REQUIRE(ld_du % 4 == 0 && ld_dv % 4 == 0 && ld_u % 4 == 0 && ld_v % 4 == 0, "pair_hidden1_bwd_recompute: rows of U, V, dU and dV must be 16-byte aligned");
    if (Q == 0) return 0;
    DFOL_REQUIRE(dZ && U && V && Wg && geo && obj_off && pair_off && n_obj && dU && dV && dWg_partial, "pair_hidden1_bwd_recompute: null pointer");
    DFOL_REQUIRE(((uintptr_t)U % 16 == 0) && ((uintptr_t)V % 16 == 0) && ((uintptr_t)Wg % 16 == 0), "pair_hidden1_bwd_recompute: U, V and Wg must be 16-byte aligned");
    const int G5 = 512 / (H1 / 4);
    const size_t lds = ((size_t)2 * G5 + (size_t)max_n) * H1 * sizeof(float);              // <= 16 KB + 128 KB
    auto kern = pair_hidden1_bwd_kernel<float, 512, HB_MAXO, 4, true>;
    static const hipError_t lds_ok = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    DFOL_REQUIRE(lds_ok == hipSuccess && lds <= 144 * 1024, "pair_hidden1_bwd_recompute: cannot reserve %zu bytes of LDS", lds);
    hipLaunchKernelGGL(kern, dim3(Q), dim3(512), lds, (hipStream_t)stream, dZ, (const float*)nullptr, geo, obj_off, pair_off, n_obj, H1, dU, ld_du, dV,
                       ld_dv, dWg_partial, HbRecompute{U, ld_u, V, ld_v, Wg});
    DFOL_LAUNCH_CHECK("pair_hidden1_bwd_recompute");
    return 0;
}

extern "C" int dfol_pair_hidden1_bwd_f32(const float* dZ, const float* Z, const float* geo, const int32_t* obj_off, const int64_t* pair_off,
                                         const int32_t* n_obj, int32_t Q, int32_t max_n, int32_t H1, float* dU, int64_t ld_du, float* dV,
                                         int64_t ld_dv, float* dWg_partial, void* stream) {
    return hidden1_bwd_launch<float>(dZ, Z, geo, obj_off, pair_off, n_obj, Q, max_n, H1, dU, ld_du, dV, ld_dv, dWg_partial, stream);
}
extern "C" int dfol_pair_hidden1_bwd_bf16(const void* dZ_bf16, const void* Z_bf16, const float* geo, const int32_t* obj_off,
                                          const int64_t* pair_off, const int32_t* n_obj, int32_t Q, int32_t max_n, int32_t H1, float* dU,
                                          int64_t ld_du, float* dV, int64_t ld_dv, float* dWg_partial, void* stream) {
    return hidden1_bwd_launch<pt_bf16>((const pt_bf16*)dZ_bf16, (const pt_bf16*)Z_bf16, geo, obj_off, pair_off, n_obj, Q, max_n, H1, dU, ld_du, dV,
                                       ld_dv, dWg_partial, stream);
}

extern "C" int dfol_pair_logit_fwd_f32(const float* P2, int64_t ld_p2, int32_t H2, const float* E, int64_t ld_e, const float* be,
                                       const int64_t* pred_off, int32_t P, int64_t rows, int64_t max_rows, float* x, void* stream) {
    DFOL_REQUIRE(H2 > 0 && H2 <= 64 * LG_T && P >= 0 && rows >= 0, "pair_logit_fwd: HID2=%d must be <= %d", H2, 64 * LG_T);
    if (rows == 0 || P == 0) return 0;
    DFOL_REQUIRE(P2 && E && pred_off && x, "pair_logit_fwd: null pointer");
    DFOL_REQUIRE(max_rows > 0 && max_rows <= rows && (max_rows + 15) / 16 < ((int64_t)1 << 31) && P < 65536,
                 "pair_logit_fwd: max_rows=%lld must be the largest row count of a predicate (P < 65536)", (long long)max_rows);
    hipLaunchKernelGGL(pair_logit_fwd_kernel<float>, dim3((unsigned)dfol_cdiv(max_rows, 16), P), dim3(256), 0, (hipStream_t)stream, P2, ld_p2, H2, E,
                       ld_e, be, pred_off, x);
    DFOL_LAUNCH_CHECK("pair_logit_fwd");
    return 0;
}

// bf16 storage of pre2 / dpre2: HID2 % 4 == 0, rows 8-byte aligned (ld % 4 == 0), E rows 16-byte aligned
static bool logit_bf16_ok(const void* P2, int64_t ld_p2, int32_t H2, const float* E, int64_t ld_e) {
    return H2 % 4 == 0 && H2 >= 16 && H2 <= 64 * LG_T && ld_p2 % 4 == 0 && ld_e % 4 == 0 && ((uintptr_t)P2 % 8 == 0) && ((uintptr_t)E % 16 == 0);
}

extern "C" int dfol_pair_logit_fwd_bf16(const void* P2_bf16, int64_t ld_p2, int32_t H2, const float* E, int64_t ld_e, const float* be,
                                        const int64_t* pred_off, int32_t P, int64_t rows, int64_t max_rows, float* x, void* stream) {
    DFOL_REQUIRE(P >= 0 && rows >= 0, "pair_logit_fwd_bf16: bad sizes");
    if (rows == 0 || P == 0) return 0;
    DFOL_REQUIRE(P2_bf16 && E && pred_off && x, "pair_logit_fwd_bf16: null pointer");
    DFOL_REQUIRE(logit_bf16_ok(P2_bf16, ld_p2, H2, E, ld_e), "pair_logit_fwd_bf16: HID2=%d must be a multiple of 4 in 16..%d, rows 8-byte aligned", H2, 64 * LG_T);
    DFOL_REQUIRE(max_rows > 0 && max_rows <= rows && (max_rows + 15) / 16 < ((int64_t)1 << 31) && P < 65536,
                 "pair_logit_fwd_bf16: max_rows=%lld must be the largest row count of a predicate (P < 65536)", (long long)max_rows);
    hipLaunchKernelGGL(pair_logit_fwd4_kernel<pt_bf16>, dim3((unsigned)dfol_cdiv(max_rows, 16), P), dim3(256), 0, (hipStream_t)stream,
                       (const pt_bf16*)P2_bf16, ld_p2, H2, E, ld_e, be, pred_off, x);
    DFOL_LAUNCH_CHECK("pair_logit_fwd_bf16");
    return 0;
}

extern "C" int dfol_pair_logit_bwd_f32(const float* dx, const float* P2, int64_t ld_p2, int32_t H2, const float* E, int64_t ld_e,
                                       const int64_t* pred_off, int32_t P, float* dP2, int64_t ld_dp2, float* dE, int64_t ld_de, float* dbe,
                                       void* stream) {
    DFOL_REQUIRE(H2 > 0 && H2 <= 64 * LG_T && P >= 0, "pair_logit_bwd: HID2=%d must be <= %d", H2, 64 * LG_T);
    if (P == 0) return 0;
    DFOL_REQUIRE(dx && P2 && E && pred_off && dE, "pair_logit_bwd: null pointer");      // (dP2 null: dE and dbe only)
    if (H2 % 4 == 0 && H2 >= 16 && ld_p2 % 4 == 0 && ld_dp2 % 4 == 0 && ld_e % 4 == 0 && ((uintptr_t)P2 % 16 == 0) && ((uintptr_t)dP2 % 16 == 0) &&
        ((uintptr_t)E % 16 == 0)) {
        const int RG = 1024 / (H2 / 4);
        hipLaunchKernelGGL(pair_logit_bwd4_kernel<float>, dim3(P), dim3(1024), (size_t)(RG * H2 + RG) * sizeof(float), (hipStream_t)stream, dx, P2, ld_p2, H2,
                           E, ld_e, pred_off, dP2, ld_dp2, dE, ld_de, dbe, (float*)nullptr, (int64_t)0);
    } else {
        hipLaunchKernelGGL(pair_logit_bwd_kernel, dim3(P), dim3(1024), 0, (hipStream_t)stream, dx, P2, ld_p2, H2, E, ld_e, pred_off, dP2, ld_dp2,
                           dE, ld_de, dbe);
    }
    DFOL_LAUNCH_CHECK("pair_logit_bwd");
    return 0;
}

extern "C" int dfol_pair_logit_bwd_bf16(const float* dx, const void* P2_bf16, int64_t ld_p2, int32_t H2, const float* E, int64_t ld_e,
                                        const int64_t* pred_off, int32_t P, void* dP2_bf16, int64_t ld_dp2, float* dE, int64_t ld_de, float* dbe,
                                        void* stream) {
    DFOL_REQUIRE(P >= 0, "pair_logit_bwd_bf16: bad sizes");
    if (P == 0) return 0;
    DFOL_REQUIRE(dx && P2_bf16 && E && pred_off && dE, "pair_logit_bwd_bf16: null pointer");      // (dP2 null: dE and dbe only)
    DFOL_REQUIRE(logit_bf16_ok(P2_bf16, ld_p2, H2, E, ld_e) && ld_dp2 % 4 == 0 && ((uintptr_t)dP2_bf16 % 8 == 0),
                 "pair_logit_bwd_bf16: HID2=%d must be a multiple of 4 in 16..%d, rows 8-byte aligned", H2, 64 * LG_T);
    const int RG = 1024 / (H2 / 4);
    hipLaunchKernelGGL(pair_logit_bwd4_kernel<pt_bf16>, dim3(P), dim3(1024), (size_t)(RG * H2 + RG) * sizeof(float), (hipStream_t)stream, dx,
                       (const pt_bf16*)P2_bf16, ld_p2, H2, E, ld_e, pred_off, (pt_bf16*)dP2_bf16, ld_dp2, dE, ld_de, dbe, (float*)nullptr, (int64_t)0);
    DFOL_LAUNCH_CHECK("pair_logit_bwd_bf16");
    return 0;
}

// The sums of the logit layer's backward WITHOUT dpre2 (which dfol_pair_dz_fused_f32 and dfol_pair_wgrad_fused_f32 produce on the fly):
// dE [P, H2], dbe [P] (or NULL) and dB2 [P, H2] = every predicate's column sums of dpre2 (the caller adds the P rows: the second
// layer's bias gradient).  H2 % 4 == 0, 16-byte aligned rows.
extern "C" int dfol_pair_logit_bwd_sums_f32(const float* dx, const float* P2, int64_t ld_p2, int32_t H2, const float* E, int64_t ld_e,
                                            const int64_t* pred_off, int32_t P, float* dE, int64_t ld_de, float* dbe, float* dB2, int64_t ld_db2,
                                            void* stream) {
    DFOL_REQUIRE(H2 >= 16 && H2 <= 64 * LG_T && H2 % 4 == 0 && P >= 0, "pair_logit_bwd_sums: HID2=%d must be a multiple of 4 in 16..%d", H2, 64 * LG_T);
    if (P == 0) return 0;
    DFOL_REQUIRE(dx && P2 && E && pred_off && dE && dB2, "pair_logit_bwd_sums: null pointer");
    DFOL_REQUIRE(ld_p2 % 4 == 0 && ld_e % 4 == 0 && ((uintptr_t)P2 % 16 == 0) && ((uintptr_t)E % 16 == 0), "pair_logit_bwd_sums: rows of pre2 and E must be 16-byte aligned");
    const int RG = 1024 / (H2 / 4);
    hipLaunchKernelGGL(pair_logit_bwd4_kernel<float>, dim3(P), dim3(1024), (size_t)(RG * H2 + RG) * sizeof(float), (hipStream_t)stream, dx, P2, ld_p2, H2, E,
                       ld_e, pred_off, (float*)nullptr, (int64_t)0, dE, ld_de, dbe, dB2, ld_db2);
    DFOL_LAUNCH_CHECK("pair_logit_bwd_sums");
    return 0;
}
