// Lab: how fast does a wavefront's VALU stream issue while the OTHER wavefront of its SIMD issues MFMAs (the ping-pong schedule of the
// pair kernels relies on this)?  One 512-thread workgroup per CU: wavefronts 0-3 run NM MFMAs (v_mfma_f32_16x16x32_f16; chains of CH
// dependent MFMAs per accumulator), wavefronts 4-7 run NV VALU instructions of a kind; each side's own duration by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/coissue.hip -o build/coissue && build/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 0: v_fma_f32 (independent x8), 1: v_pk_fma_f32, 2: v_exp_f32, 3: v_cvt_pk_f16_f32, 4: the build's mix (med3, mul, exp, add, med3, cvt_pk ...)
template <int KIND, int CH, bool MF, bool VA, int PRIO = 0, int MPRIO = 0>
__global__ __launch_bounds__(512) void k(long long* out, int reps, float seed) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ float sh[64];
    if (threadIdx.x < 64) sh[threadIdx.x] = seed;
    __syncthreads();
    long long t0 = 0, t1 = 0;
    float sink = 0.f;
    if (wave < 4) {
        if (MF) {
            f16x8 a, b;
            for (int j = 0; j < 8; ++j) a[j] = (_Float16)(seed + j), b[j] = (_Float16)(seed * 0.5f + lane);
            floatx4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = floatx4{0, 0, 0, 0};
            if (MPRIO) __builtin_amdgcn_s_setprio(MPRIO);
            t0 = clock64();
            for (int r = 0; r < reps; ++r) {
                if (CH == 32) {                               // two accumulators alternating, three products each (chains interleaved)
#pragma unroll
                    for (int i = 0; i < 8; i += 2)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                            acc[i + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i + 1], 0, 0, 0);
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int c = 0; c < CH; ++c) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                }
            }
            for (int i = 0; i < 8; ++i) sink += acc[i][0];
            t1 = clock64();
        }
    } else if (VA) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed + i + lane;
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        t0 = clock64();
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (KIND == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], seed, 0.5f);
                } else if (KIND == 1) {
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        f32x2 x = {v[i], v[i + 1]};
                        x = __builtin_elementwise_fma(x, (f32x2){seed, seed}, (f32x2){0.5f, 0.25f});
                        x = __builtin_elementwise_fma(x, (f32x2){seed, seed}, (f32x2){0.5f, 0.25f});
                        v[i] = x.x, v[i + 1] = x.y;
                    }
                } else if (KIND == 5) {                       // three VGPR sources per instruction
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], v[(i + 3) & 7], v[(i + 5) & 7]);
                } else if (KIND == 6) {                       // packed, three VGPR-pair sources
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        f32x2 x = {v[i], v[i + 1]}, y = {v[(i + 2) & 7], v[(i + 3) & 7]}, w = {v[(i + 4) & 7], v[(i + 5) & 7]};
                        x = __builtin_elementwise_fma(x, y, w);
                        y = __builtin_elementwise_fma(y, w, x);
                        v[i] = x.x, v[i + 1] = x.y, v[(i + 2) & 7] = y.x, v[(i + 3) & 7] = y.y;
                    }
                } else if (KIND == 2) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
                } else if (KIND == 3) {
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        h2 h = __builtin_convertvector((f32x2){v[i], v[i + 1]}, h2);
                        h2 g = __builtin_convertvector((f32x2){v[i + 1], v[i]}, h2);
                        v[i] = __builtin_bit_cast(float, h);
                        v[i + 1] = __builtin_bit_cast(float, g);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float z = v[i];
                        float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(z, 0.f, -3e38f) * 1.4426950408889634f) - 1.0f;
                        v[i] = __builtin_amdgcn_fmed3f(z, e, 60000.f) + seed;
                    }
                }
            }
        }
        for (int i = 0; i < 8; ++i) sink += v[i];
        t1 = clock64();
    }
    if (lane == 0) {
        out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        out[(blockIdx.x * 8 + wave) * 2 + 1] = (long long)sink;
    }
}

template <int KIND, int CH, bool MF, bool VA, int PRIO = 0, int MPRIO = 0>
void run(const char* name, int valu_per_rep, int mfma_per_rep) {
    long long* d;
    hipMalloc(&d, 256 * 8 * 2 * 8);
    const int reps = 2000;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<KIND, CH, MF, VA, PRIO, MPRIO>), dim3(256), dim3(512), 0, 0, d, reps, 1.0001f);
    hipDeviceSynchronize();
    long long h[8 * 2];
    hipMemcpy(h, d + 100 * 16, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-46s MFMA wave: %7.1f cycles per MFMA | VALU wave: %7.1f cycles per VALU instruction\n", name,
           MF ? (double)h[0] / reps / mfma_per_rep : 0.0, VA ? (double)h[8] / reps / valu_per_rep : 0.0);
    hipFree(d);
}

int main() {
    run<0, 3, true, false>("MFMA chains of 3 alone", 32, 24);
    run<0, 1, true, false>("MFMA independent alone", 32, 8);
    run<0, 3, false, true>("v_fma_f32 alone", 32, 24);
    run<1, 3, false, true>("v_pk_fma_f32 alone", 32, 24);
    run<2, 3, false, true>("v_exp_f32 alone", 32, 24);
    run<3, 3, false, true>("v_cvt_pk_f16_f32 alone", 32, 24);
    run<4, 3, false, true>("ELU mix (6 instr / element) alone", 32 * 6, 24);
    run<0, 3, true, true>("MFMA chains of 3 + v_fma_f32", 32, 24);
    run<0, 1, true, true>("MFMA independent + v_fma_f32", 32, 8);
    run<1, 3, true, true>("MFMA chains of 3 + v_pk_fma_f32", 32, 24);
    run<2, 3, true, true>("MFMA chains of 3 + v_exp_f32", 32, 24);
    run<3, 3, true, true>("MFMA chains of 3 + v_cvt_pk_f16_f32", 32, 24);
    run<4, 3, true, true>("MFMA chains of 3 + ELU mix", 32 * 6, 24);
    run<4, 1, true, true>("MFMA independent + ELU mix", 32 * 6, 8);
    run<0, 32, true, false>("MFMA 2 interleaved chains of 3 alone", 32, 24);
    run<0, 32, true, true>("MFMA 2 interleaved chains of 3 + v_fma_f32", 32, 24);
    run<4, 32, true, true>("MFMA 2 interleaved chains of 3 + ELU mix", 32 * 6, 24);
    run<0, 3, true, true, 3>("MFMA chains of 3 + v_fma_f32, VALU wave prio 3", 32, 24);
    run<1, 3, true, true, 3>("MFMA chains of 3 + v_pk_fma_f32, VALU wave prio 3", 32, 24);
    run<2, 3, true, true, 3>("MFMA chains of 3 + v_exp_f32, VALU wave prio 3", 32, 24);
    run<3, 3, true, true, 3>("MFMA chains of 3 + v_cvt_pk, VALU wave prio 3", 32, 24);
    run<4, 3, true, true, 3>("MFMA chains of 3 + ELU mix, VALU wave prio 3", 32 * 6, 24);
    run<0, 3, true, true, 0, 3>("MFMA chains of 3 (prio 3) + v_fma_f32", 32, 24);
    run<5, 3, false, true>("v_fma_f32 3 VGPR sources alone", 32, 24);
    run<5, 3, true, true>("MFMA chains of 3 + v_fma_f32 3 VGPR", 32, 24);
    run<5, 3, true, true, 3>("MFMA chains of 3 + v_fma_f32 3 VGPR, VALU prio 3", 32, 24);
    run<6, 3, false, true>("v_pk_fma_f32 3 VGPR pairs alone", 32, 24);
    run<6, 3, true, true>("MFMA chains of 3 + v_pk_fma_f32 3 VGPR pairs", 32, 24);
    run<6, 3, true, true, 3>("MFMA chains of 3 + v_pk_fma_f32 3 VGPR pairs, VALU prio 3", 32, 24);
    return 0;
}
