// microbenchmark: what the bf16 matrix pipe sustains over the whole chip (power-limited clock), for the two MFMA shapes the bf16x3
// kernels use, D independent accumulators in rotation, W wavefronts per SIMD, long enough (~20 ms) for the clock to settle
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int D>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    floatx4 acc[D];
    for (int d = 0; d < D; ++d) acc[d] = floatx4{0, 0, 0, 0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)((threadIdx.x + j) & 7); b[j] = (__bf16)(float)((threadIdx.x * 3 + j) & 7); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[d], 0, 0, 0);
    }
    float s = 0;
    for (int d = 0; d < D; ++d) s += acc[d][0] + acc[d][1] + acc[d][2] + acc[d][3];
    if (s == 12345.f) out[threadIdx.x] = s;
}
template <int D>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    floatx16 acc[D];
    for (int d = 0; d < D; ++d) for (int i = 0; i < 16; ++i) acc[d][i] = 0;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)((threadIdx.x + j) & 7); b[j] = (__bf16)(float)((threadIdx.x * 3 + j) & 7); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[d], 0, 0, 0);
    }
    float s = 0;
    for (int d = 0; d < D; ++d) s += acc[d][0] + acc[d][5];
    if (s == 12345.f) out[threadIdx.x] = s;
}
template <typename F>
void run(const char* name, F launch, double mfma_per_wave_iter, double flops_per_mfma, int blocks, float* dout) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 20000;
    launch(blocks, iters / 10);
    hipEventRecord(e0);
    launch(blocks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)blocks * 4 * iters * mfma_per_wave_iter;
    const double per_simd = mf / 1024.0;                       // MFMAs each SIMD executed
    printf("%-28s blocks=%4d: %7.3f ms  %7.1f TFLOP/s (%.0f %% of 2.5 PF)  %.1f ns per MFMA per SIMD\n", name, blocks, ms, mf * flops_per_mfma / ms / 1e9,
           mf * flops_per_mfma / ms / 1e9 / 2500 * 100, ms * 1e6 / per_simd);
}
int main() {
    float* dout; hipMalloc(&dout, 4096);
#define L16(D) [&](int b, int it) { hipLaunchKernelGGL(k16<D>, dim3(b), dim3(256), 0, 0, dout, it); }
#define L32(D) [&](int b, int it) { hipLaunchKernelGGL(k32<D>, dim3(b), dim3(256), 0, 0, dout, it); }
    run("16x16x32 D=4  1 wave/SIMD", L16(4), 32, 16384, 256, dout);
    run("16x16x32 D=8  1 wave/SIMD", L16(8), 64, 16384, 256, dout);
    run("16x16x32 D=4  2 waves/SIMD", L16(4), 32, 16384, 512, dout);
    run("16x16x32 D=8  2 waves/SIMD", L16(8), 64, 16384, 512, dout);
    run("32x32x16 D=1  1 wave/SIMD", L32(1), 4, 32768, 256, dout);
    run("32x32x16 D=2  1 wave/SIMD", L32(2), 8, 32768, 256, dout);
    run("32x32x16 D=4  1 wave/SIMD", L32(4), 16, 32768, 256, dout);
    run("32x32x16 D=8  1 wave/SIMD", L32(8), 32, 32768, 256, dout);
    run("32x32x16 D=4  2 waves/SIMD", L32(4), 16, 32768, 512, dout);
    return 0;
}
