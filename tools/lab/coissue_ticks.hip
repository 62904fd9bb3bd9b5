// Lab: the ping-pong tick in miniature.  Per CU one 512-thread workgroup; in every tick wavefronts 0-3 issue NM MFMAs (chains of 3) and
// wavefronts 4-7 a burst of NV VALU instructions (the ELU mix), then everyone meets at a barrier.  How long is a tick - max or sum?
//   hipcc --offload-arch=gfx950 -O3 tools/lab/coissue_ticks.hip -o build/coissue_ticks && build/coissue_ticks
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool MF, bool VA, int PRIO, bool LDS>
__global__ __launch_bounds__(512) void k(long long* out, int ticks, float seed) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ __attribute__((aligned(16))) f16x8 sh[2560];
    for (int i = threadIdx.x; i < 2560; i += 512)
        for (int j = 0; j < 8; ++j) sh[i][j] = (_Float16)(seed + (i & 7) * 0.125f);
    __syncthreads();
    long long t0 = clock64(), tw = 0;
    float sink = 0.f;
    if (wave < 4) {
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) a[j] = (_Float16)(seed + j), b[j] = (_Float16)(seed * 0.5f + lane);
        floatx4 acc[38];
        for (int i = 0; i < 38; ++i) acc[i] = floatx4{0, 0, 0, 0};
        for (int t = 0; t < ticks; ++t) {
            if (MF) {
#pragma unroll
                for (int i = 0; i < 38; ++i) {
                    if (LDS && (i & 1) == 0) b = sh[(i * 32 + lane) % 2560];      // a B fragment per column tile, as the kernel reads them
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                }
            }
            long long w0 = clock64();
            __syncthreads();
            tw += clock64() - w0;
        }
        for (int i = 0; i < 38; ++i) sink += acc[i][0];
    } else {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = seed + i + lane;
        for (int t = 0; t < ticks; ++t) {
            if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
            if (VA) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {              // 32 elements x 5 = 160 VALU instructions, 32 of them transcendental
                        float z = v[i];
                        float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(z, 0.f, -3e38f) * 1.4426950408889634f) - 1.0f;
                        v[i] = __builtin_amdgcn_fmed3f(z, e, 60000.f) + seed;
                    }
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            long long w0 = clock64();
            __syncthreads();
            tw += clock64() - w0;
        }
        for (int i = 0; i < 16; ++i) sink += v[i];
    }
    long long t1 = clock64();
    if (lane == 0) {
        out[(blockIdx.x * 8 + wave) * 3] = t1 - t0;
        out[(blockIdx.x * 8 + wave) * 3 + 1] = tw;
        out[(blockIdx.x * 8 + wave) * 3 + 2] = (long long)sink;
    }
}

template <bool MF, bool VA, int PRIO, bool LDS>
void run(const char* name) {
    long long* d;
    (void)hipMalloc(&d, 256 * 8 * 3 * 8);
    const int ticks = 400;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MF, VA, PRIO, LDS>), dim3(256), dim3(512), 0, 0, d, ticks, 1.0001f);
    (void)hipDeviceSynchronize();
    long long h[8 * 3];
    (void)hipMemcpy(h, d + 100 * 24, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s tick %6.0f cycles | MFMA wave waits %6.0f at the barrier | VALU wave waits %6.0f\n", name, (double)h[0] / ticks, (double)h[1] / ticks,
           (double)h[12 + 1] / ticks);
    (void)hipFree(d);
}

int main() {
    run<true, false, 0, false>("114 MFMAs alone");
    run<false, true, 0, false>("160-instruction VALU burst alone");
    run<true, true, 0, false>("both, equal priority");
    run<true, true, 3, false>("both, VALU burst at priority 3");
    run<true, true, 1, false>("both, VALU burst at priority 1");
    run<true, false, 0, true>("114 MFMAs with LDS fragment reads alone");
    run<true, true, 0, true>("both, LDS fragment reads, equal priority");
    run<true, true, 3, true>("both, LDS fragment reads, VALU burst at priority 3");
    return 0;
}
