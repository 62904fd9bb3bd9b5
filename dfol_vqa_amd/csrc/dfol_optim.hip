// The tail of a train step (trainer.py:439-441: nn.utils.clip_grad_norm_ -> optimizer.step(), torch.optim.Adam) over the flat gradient bucket
// as two launches (three for a capturable optimizer, whose step counters live on the device) instead of seventeen (nine of clip_grad_norm_'s
// foreach form, eight of Adam's): every gradient of the model is a view into
// one contiguous fp32 buffer (parallel.GradBucket), so the norm is one reduction over it and the update one pass over a chunk table.
//
//   dfol_grad_sqnorm_f32     per-workgroup partial sums of g^2 over the flat buffer, fixed grid, fixed order: deterministic
//   dfol_clip_adam_f32       every workgroup adds the partials up in the same order -> total norm -> clip coefficient
//                            min(1, max_norm / (norm + 1e-6)) (clip_grad_norm_'s formula); then for its chunk: g <- g coef (clip_grad_norm_
//                            scales the gradients in place), Adam's moments and the update with torch's formulas (lerp of exp_avg, mul + addcmul
//                            of exp_avg_sq, bias corrections from the step count, eps added after the division by sqrt(bias_correction2))
// No atomics.  Not bit-identical to torch's kernels (the norm is accumulated in another order, fused multiply-adds where torch rounds twice):
// equal to a few ulp per step, tests/test_backward_gpu.py::test_fused_clip_adam_equals_torch.
#include "dfol_common.h"

namespace {

constexpr int OP_PARTS = 1024;              // workgroups of the norm kernel = partial sums
constexpr int OP_CHUNK = 4096;              // elements of a chunk of the update kernel (256 threads x 4 x 4)

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partials) {
    __shared__ float wsum[4];
    float s = 0.f;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = g4[i];
        s = fmaf(v.x, v.x, s), s = fmaf(v.y, v.y, s), s = fmaf(v.z, v.z, s), s = fmaf(v.w, v.w, s);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {                    // the tail of a buffer whose length is not a multiple of 4
        const float v = g[(n4 << 2) + threadIdx.x];
        s = fmaf(v, v, s);
    }
    s = dfol_wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

struct AdamHyper {
    float lr, beta1, beta2, eps, weight_decay, max_norm;
};

// chunk c covers elements [chunk_start[c], chunk_start[c] + OP_CHUNK) of tensor chunk_tensor[c] (clipped to its numel)
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ g, const float* __restrict__ partials, int nparts, const int64_t* __restrict__ param,
                                                        const int64_t* __restrict__ exp_avg, const int64_t* __restrict__ exp_avg_sq,
                                                        const int64_t* __restrict__ goff, const int64_t* __restrict__ numel,
                                                        const int32_t* __restrict__ chunk_tensor, const int64_t* __restrict__ chunk_start,
                                                        const int64_t* __restrict__ step_ptr, float step_host, AdamHyper h,
                                                        float* __restrict__ norm_out) {
    __shared__ float wsum[4];
    // the total: every workgroup adds the same partials in the same order
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partials[i];
    s = dfol_wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    const float total = sqrtf((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
    // torch's clip_grad_norm_: clamp(max_norm / (total + 1e-6), max = 1) - a NaN norm gives a NaN coefficient that poisons every gradient
    // (fminf would drop the NaN and step on), max_norm = 0 zeroes the gradients; max_norm < 0 = no clipping at all
    const float coef = h.max_norm >= 0.f ? ((total != total) ? total : fminf(h.max_norm / (total + 1e-6f), 1.0f)) : 1.0f;
    const int t = chunk_tensor[blockIdx.x];
    // the step count of this update: a device counter per tensor (capturable optimizers) or a host value
    float step = step_host;
    float* sp = step_ptr ? reinterpret_cast<float*>(step_ptr[t]) : nullptr;
    if (sp) step = sp[0] + 1.0f;
    const float bc1 = 1.0f - powf(h.beta1, step), bc2 = 1.0f - powf(h.beta2, step);
    const float step_size = h.lr / bc1, rs2 = 1.0f / sqrtf(bc2);
    const int64_t start = chunk_start[blockIdx.x], n = numel[t];
    float* __restrict__ p = reinterpret_cast<float*>(param[t]);
    float* __restrict__ m = reinterpret_cast<float*>(exp_avg[t]);
    float* __restrict__ v = reinterpret_cast<float*>(exp_avg_sq[t]);
    float* __restrict__ gt = g + goff[t];
#pragma unroll
    for (int k = 0; k < OP_CHUNK / 256; ++k) {
        const int64_t i = start + k * 256 + threadIdx.x;
        if (i < n) {
            float gi = gt[i] * coef;
            gt[i] = gi;                                                // (clip_grad_norm_ leaves the scaled gradients behind)
            const float pi = p[i];
            if (h.weight_decay != 0.f) gi = fmaf(h.weight_decay, pi, gi);
            const float mi = fmaf(gi - m[i], 1.0f - h.beta1, m[i]);
            const float vi = fmaf((1.0f - h.beta2) * gi, gi, v[i] * h.beta2);
            m[i] = mi, v[i] = vi;
            p[i] = pi - step_size * (mi / (sqrtf(vi) * rs2 + h.eps));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = total;
    // (the step counters are advanced by a one-workgroup launch BEHIND this one: other chunks of a tensor may still be reading theirs)
}

__global__ void adam_advance_steps_kernel(const int64_t* __restrict__ step_ptr, int T) {
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        float* sp = reinterpret_cast<float*>(step_ptr[t]);
        sp[0] = sp[0] + 1.0f;
    }
}

}  // namespace

extern "C" int32_t dfol_grad_sqnorm_parts() { return OP_PARTS; }
extern "C" int32_t dfol_clip_adam_chunk() { return OP_CHUNK; }

extern "C" int dfol_grad_sqnorm_f32(const float* g, int64_t n, float* partials, void* stream) {
    DFOL_REQUIRE(n >= 0 && partials && (g || n == 0), "grad_sqnorm: null pointer");
    DFOL_REQUIRE((uintptr_t)g % 16 == 0, "grad_sqnorm: the gradient buffer must be 16-byte aligned");
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(OP_PARTS), dim3(256), 0, (hipStream_t)stream, g, n, partials);
    DFOL_LAUNCH_CHECK("grad_sqnorm");
    return 0;
}

extern "C" int dfol_clip_adam_f32(float* g, const float* partials, const int64_t* param, const int64_t* exp_avg, const int64_t* exp_avg_sq,
                                  const int64_t* goff, const int64_t* numel, int32_t n_tensors, const int32_t* chunk_tensor, const int64_t* chunk_start,
                                  int32_t n_chunks, const int64_t* step_ptr, float step_host, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, float max_norm, float* norm_out, void* stream) {
    DFOL_REQUIRE(n_tensors >= 0 && n_chunks >= 0, "clip_adam: negative counts");
    if (n_chunks == 0) return 0;
    DFOL_REQUIRE(g && partials && param && exp_avg && exp_avg_sq && goff && numel && chunk_tensor && chunk_start, "clip_adam: null pointer");
    DFOL_REQUIRE(step_ptr || step_host >= 1.0f, "clip_adam: the step count of the update (device counters or a host value >= 1)");
    const AdamHyper h = {lr, beta1, beta2, eps, weight_decay, max_norm};
    hipLaunchKernelGGL(clip_adam_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, g, partials, OP_PARTS, param, exp_avg, exp_avg_sq, goff, numel,
                       chunk_tensor, chunk_start, step_ptr, step_host, h, norm_out);
    if (step_ptr) hipLaunchKernelGGL(adam_advance_steps_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, step_ptr, n_tensors);
    DFOL_LAUNCH_CHECK("clip_adam");
    return 0;
}
