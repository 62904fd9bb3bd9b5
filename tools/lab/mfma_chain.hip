// microbenchmark: issue rate of v_mfma_f32_16x16x32_bf16 with D accumulators in rotation (dependency distance D)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int D>
__global__ __launch_bounds__(256) void chain(float* out, int iters) {
    floatx4 acc[D];
    for (int d = 0; d < D; ++d) acc[d] = floatx4{0, 0, 0, 0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(threadIdx.x * 3 + j); }
    long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[d], 0, 0, 0);
    }
    long t1 = clock64();
    float s = 0;
    for (int d = 0; d < D; ++d) s += acc[d][0] + acc[d][1] + acc[d][2] + acc[d][3];
    if (s == 12345.f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 8.0f * D);
}
template <int D>
void run(float* dout, int blocks, int threads) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    hipLaunchKernelGGL(chain<D>, dim3(blocks), dim3(threads), 0, 0, dout, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<D>, dim3(blocks), dim3(threads), 0, 0, dout, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float ticks; hipMemcpy(&ticks, dout, 4, hipMemcpyDeviceToHost);
    double mf = (double)blocks * (threads / 64) * iters * 8.0 * D;
    printf("D=%d blocks=%d threads=%d: %.3f ms, %.2f clock64 ticks/MFMA/wave, %.1f TFLOP/s\n", D, blocks, threads, ms, ticks, mf * 16384 / ms / 1e9);
}
int main() {
    float* dout; hipMalloc(&dout, 4096);
    run<1>(dout, 256, 256); run<2>(dout, 256, 256); run<3>(dout, 256, 256); run<4>(dout, 256, 256); run<8>(dout, 256, 256);
    run<1>(dout, 512, 256); run<2>(dout, 512, 256); run<4>(dout, 512, 256);
    run<1>(dout, 1024, 256); run<2>(dout, 1024, 256);
    return 0;
}
