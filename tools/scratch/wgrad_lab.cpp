// Scratch harness (not part of the product): times dfol_linear_wgrad_f32 from a variant of csrc/dfol_dense_wgrad.hip linked into this binary.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
extern "C" int dfol_linear_wgrad_slabs(int64_t, int32_t, int32_t);
extern "C" int dfol_linear_wgrad_f32(const float*, int64_t, const float*, int64_t, int64_t, int32_t, int32_t, float*, float*, void*);
__global__ void fill(float* p, int64_t n, uint32_t seed) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(int)(h & 0xffff) / 65536.f - 0.5f;
    }
}
int main(int argc, char** argv) {
    int64_t M = argc > 1 ? atoll(argv[1]) : 2534400; int N = argc > 2 ? atoi(argv[2]) : 300, K = argc > 3 ? atoi(argv[3]) : 256;
    float *dy, *x, *ws, *dw;
    hipMalloc(&dy, M * N * 4); hipMalloc(&x, M * K * 4);
    int slabs = dfol_linear_wgrad_slabs(M, N, K);
    hipMalloc(&ws, (size_t)slabs * ((N * K + 3) / 4 * 4) * 4); hipMalloc(&dw, N * K * 4);
    fill<<<4096, 256>>>(dy, M * N, 1); fill<<<4096, 256>>>(x, M * K, 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) dfol_linear_wgrad_f32(dy, N, x, K, M, N, K, ws, dw, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) dfol_linear_wgrad_f32(dy, N, x, K, M, N, K, ws, dw, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[4]; hipMemcpy(h, dw, 16, hipMemcpyDeviceToHost);
    printf("%s M=%lld N=%d K=%d: %.3f ms  (dw[0]=%g)\n", argv[0], (long long)M, N, K, ms / 10, h[0]);
    return 0;
}
