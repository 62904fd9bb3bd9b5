// Scratch harness (not part of the product): times dfol_linear_act_split_f32 from a variant of csrc/dfol_dense_split.hip linked into this binary.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
extern "C" int dfol_linear_pack_w_bf16x3(const float*, int64_t, int32_t, int32_t, void*, void*);
extern "C" int dfol_linear_act_split_f32(const float*, int64_t, const void*, const float*, float*, int64_t, int32_t, int32_t, int32_t, int32_t, void*);
__global__ void fill(float* p, int64_t n, uint32_t seed) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(int)(h & 0xffff) / 65536.f - 0.5f;
    }
}
int main(int argc, char** argv) {
    int64_t M = argc > 1 ? atoll(argv[1]) : 2534400; int N = argc > 2 ? atoi(argv[2]) : 300, K = argc > 3 ? atoi(argv[3]) : 256, act = argc > 4 ? atoi(argv[4]) : 1;
    float *x, *w, *b, *y; void* wp;
    hipMalloc(&x, M * K * 4); hipMalloc(&y, M * N * 4); hipMalloc(&w, N * K * 4); hipMalloc(&b, N * 4);
    hipMalloc(&wp, (size_t)((N + 127) / 128) * ((K + 31) / 32) * 24576);
    fill<<<4096, 256>>>(x, M * K, 1); fill<<<64, 256>>>(w, N * K, 2); fill<<<1, 256>>>(b, N, 3);
    dfol_linear_pack_w_bf16x3(w, K, N, K, wp, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) dfol_linear_act_split_f32(x, K, wp, b, y, N, (int)M, N, K, act, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) dfol_linear_act_split_f32(x, K, wp, b, y, N, (int)M, N, K, act, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[4]; hipMemcpy(h, y, 16, hipMemcpyDeviceToHost);
    printf("%s M=%lld N=%d K=%d act=%d: %.3f ms  (y[0]=%g)\n", argv[0], (long long)M, N, K, act, ms / 10, h[0]);
    return 0;
}
