// Shared device helpers and ABI plumbing for the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dfol_vqa.h"

#define DFOL_EPS 1e-20f          // util.py:25
#define DFOL_WAVE 64

void dfol_set_error(const char* fmt, ...);
uint32_t* dfol_range_status_ptr();      // the calling thread's status word for the fp16-range kernels (dfol_set_range_status), or NULL

#define DFOL_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            dfol_set_error(__VA_ARGS__);        \
            return 1;                           \
        }                                       \
    } while (0)

#define DFOL_LAUNCH_CHECK(name)                                                          \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) {                                                          \
            dfol_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));        \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

// ---- transcendental helpers -----------------------------------------------------------------------
// The logic kernels evaluate exp and log once or twice per byte they stream, so they use the hardware
// v_exp_f32 / v_log_f32 directly (1 ulp each; arguments are log-probabilities <= 0 and probabilities in
// [1e-20, 2], so neither overflow nor denormal-input handling is needed).  -DDFOL_PRECISE_MATH swaps in
// the libm versions for A/B parity runs.
__device__ __forceinline__ float dfol_exp(float x) {
#ifdef DFOL_PRECISE_MATH
    return expf(x);
#else
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
#endif
}

__device__ __forceinline__ float dfol_log(float x) {
#ifdef DFOL_PRECISE_MATH
    return logf(x);
#else
    return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
#endif
}

// util.py:22-25
__device__ __forceinline__ float dfol_slog(float x) { return dfol_log(fmaxf(x, DFOL_EPS)); }
// util.py:35-36
__device__ __forceinline__ float dfol_lnot(float x) { return dfol_slog(1.0f - dfol_exp(x)); }
// util.py:46-47 with beta = 1:  log(max(alpha + (1 - 2 alpha) e^x, eps)); c = 1 - 2 alpha
__device__ __forceinline__ float dfol_pnot(float x, float alpha, float c) { return dfol_slog(alpha + c * dfol_exp(x)); }

// ---- EXISTS aggregation without cancellation ----------------------------------------------------------
// The reference aggregates an EXISTS variable as  log_not(sum_i log_not(u_i)) = log(1 - prod_i (1 - y_i)),  y_i = e^{u_i}
// (util.py:35-36, batch_base_ops.py:102-133, batch_base_types.py:115-123).  Evaluated as written in fp32, every factor 1 - y_i is
// rounded at 2^-25 ABSOLUTE, and the final 1 - e^S amplifies that by 1 / (1 - e^S): the reference's own fp32 run carries 1e-5 .. 1e-4 of
// noise on a log-probability of -4.  The kernels keep the COMPLEMENT  q = 1 - prod (1 - y_i)  instead, by the recurrence of the
// probabilistic OR:   q <- q + y - q y   (= 1 - (1 - q)(1 - y)).
// q is a sum of non-negative terms, so every step is accurate RELATIVE to q (2^-24 per step, no cancellation anywhere), and the
// aggregate is simply log(max(q, eps)): the float64 value of the reference's formula to ~1e-6 in the log domain, whatever the size of q.
// The reference's per-term clamp max(1 - y, eps) only matters when y = 1, where both forms give q = 1 (log 1 = 0); its outer clamp
// max(1 - e^S, eps) is the max(q, eps) here.  y > 1 (a prior above log 1) is not handled by this form: callers detect it and fall back
// to the general, clamping code.
__device__ __forceinline__ float dfol_or(float q, float y) { return fmaf(-q, y, q + y); }

// Sum over all 64 lanes; every lane gets the total.
__device__ __forceinline__ float dfol_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ---- DPP reductions ---------------------------------------------------------------------------------
// x + (x of the lane selected by the DPP control word); lanes whose source is masked off add 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dfol_dpp_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}

// Sum over aligned groups of G consecutive lanes (G = 1..64, power of two) without touching LDS: quad permutes and row
// mirrors inside each 16-lane row, then row_bcast15 / row_bcast31 across rows.  The total is valid in the LAST lane of
// each group (for G <= 16 in every lane of the group).
template <int G>
__device__ __forceinline__ float dfol_group_sum(float x) {
    if (G >= 2) x = dfol_dpp_add<0xB1, 0xF>(x);       // quad_perm [1,0,3,2]
    if (G >= 4) x = dfol_dpp_add<0x4E, 0xF>(x);       // quad_perm [2,3,0,1]
    if (G >= 8) x = dfol_dpp_add<0x141, 0xF>(x);      // row_half_mirror
    if (G >= 16) x = dfol_dpp_add<0x140, 0xF>(x);     // row_mirror
    if (G >= 32) x = dfol_dpp_add<0x142, 0xA>(x);     // row_bcast15 into rows 1 and 3
    if (G >= 64) x = dfol_dpp_add<0x143, 0xC>(x);     // row_bcast31 into rows 2 and 3
    return x;
}

// The lane selected by the DPP control word hands over its value; lanes whose source is masked off get 0 (the neutral element of OR).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dfol_dpp_or(float q) {
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), CTRL, ROW_MASK, 0xF, false));
    return fmaf(o, 1.0f - q, q);                         // q + o (1 - q): the DPP move folds into the multiply-add
}

// Probabilistic OR over aligned groups of G consecutive lanes, same lane schedule as dfol_group_sum (valid in the LAST lane of each
// group; for G <= 16 in every lane).  1 - q is rounded once per step, but it enters as a FACTOR of the incoming term, so the step
// stays accurate relative to q.
template <int G>
__device__ __forceinline__ float dfol_group_or(float q) {
    if (G >= 2) q = dfol_dpp_or<0xB1, 0xF>(q);
    if (G >= 4) q = dfol_dpp_or<0x4E, 0xF>(q);
    if (G >= 8) q = dfol_dpp_or<0x141, 0xF>(q);
    if (G >= 16) q = dfol_dpp_or<0x140, 0xF>(q);
    if (G >= 32) q = dfol_dpp_or<0x142, 0xA>(q);
    if (G >= 64) q = dfol_dpp_or<0x143, 0xC>(q);
    return q;
}

static inline int dfol_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
