// Lab: what the fp16 two-piece split (three products) costs in accuracy against the bf16 three-piece split (six products) and a plain
// fp32 FMA chain, on the pair layer's shapes; and whether the matrix pipe keeps fp16 SUBNORMAL operands (the low piece of a value
// below 0.125 is subnormal in fp16).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/lab/split_accuracy.hip -o build/split_accuracy && build/split_accuracy
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// C[16][16] = A[16][K] * B[16][K]^T, one wavefront.  mode 0: fp16 x 2 pieces, 3 products; 1: bf16 x 3 pieces, 6 products;
// 2: fp16 x 2 with the low pieces flushed when subnormal (what a flushing pipe would compute); 3: fp16 x 2, 4 products
__global__ void dot_kernel(const float* A, const float* B, int K, int mode, float* C) {
    const int lane = threadIdx.x, kh = lane >> 4, r16 = lane & 15;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        float a[8], b[8];
        for (int j = 0; j < 8; ++j) a[j] = A[r16 * K + k0 + 8 * kh + j], b[j] = B[r16 * K + k0 + 8 * kh + j];
        if (mode == 1) {
            bf16x8 ap[3], bp[3];
            for (int j = 0; j < 8; ++j) {
                float x = a[j];
                for (int p = 0; p < 3; ++p) {
                    uint32_t u = __float_as_uint(x) & 0xffff0000u;
                    ap[p][j] = __builtin_bit_cast(__bf16, (uint16_t)(u >> 16));
                    x -= __uint_as_float(u);
                }
                x = b[j];
                for (int p = 0; p < 3; ++p) {
                    uint32_t u = __float_as_uint(x) & 0xffff0000u;
                    bp[p][j] = __builtin_bit_cast(__bf16, (uint16_t)(u >> 16));
                    x -= __uint_as_float(u);
                }
            }
            const int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            for (int x = 0; x < 6; ++x) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[PA[x]], bp[PB[x]], acc, 0, 0, 0);
        } else {
            f16x8 ah, al, bh, bl;
            for (int j = 0; j < 8; ++j) {
                ah[j] = (_Float16)a[j];
                al[j] = (_Float16)(a[j] - (float)ah[j]);
                bh[j] = (_Float16)b[j];
                bl[j] = (_Float16)(b[j] - (float)bh[j]);
                if (mode == 2) {
                    if (fabsf((float)al[j]) < 6.103515625e-5f) al[j] = (_Float16)0.f;
                    if (fabsf((float)bl[j]) < 6.103515625e-5f) bl[j] = (_Float16)0.f;
                }
            }
            if (mode == 3) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
        }
    }
    for (int e = 0; e < 4; ++e) C[(4 * kh + e) * 16 + r16] = acc[e];
}

// subnormal probe: A[m][0] = 2^-20 (fp16 subnormal), B[n][0] = 1024: product 2^-10 unless the pipe flushes
__global__ void subnormal_kernel(float* C) {
    const int lane = threadIdx.x, kh = lane >> 4;
    f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (kh == 0) a[0] = (_Float16)9.5367431640625e-7f, b[0] = (_Float16)1024.f;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) C[0] = acc[0], C[1] = (float)a[0];
}

static double frand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }

int main() {
    float* dC;
    hipMalloc(&dC, 1024);
    hipLaunchKernelGGL(subnormal_kernel, dim3(1), dim3(64), 0, 0, dC);
    float hc[2];
    hipMemcpy(hc, dC, 8, hipMemcpyDeviceToHost);
    printf("subnormal probe: 2^-20 (fp16 subnormal, converted value %.9g) x 1024 through v_mfma_f32_16x16x32_f16 = %.9g (kept: %.9g, flushed: 0)\n", hc[1], hc[0],
           9.5367431640625e-7 * 1024);
    const int K = 256, TR = 200;
    float *hA = (float*)malloc(16 * K * 4), *hB = (float*)malloc(16 * K * 4), *dA, *dB;
    hipMalloc(&dA, 16 * K * 4);
    hipMalloc(&dB, 16 * K * 4);
    const char* names[4] = {"fp16 x 2 pieces, 3 products", "bf16 x 3 pieces, 6 products", "fp16 x 2, subnormal low pieces flushed", "fp16 x 2 pieces, 4 products"};
    for (int wscale_log2 = 0; wscale_log2 <= 8; wscale_log2 += 8) {
        double err[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0}, ref_mag = 0;
        long cnt = 0;
        srand(1);
        for (int t = 0; t < TR; ++t) {
            for (int i = 0; i < 16 * K; ++i) {
                double z = (frand() * 2 - 1) * 3.0;                       // first-layer sums
                hA[i] = (float)(z > 0 ? z : exp(z) - 1);                  // ELU
                hB[i] = (float)((frand() * 2 - 1) / 16.0) * (float)(1 << wscale_log2);
            }
            hipMemcpy(dA, hA, 16 * K * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, hB, 16 * K * 4, hipMemcpyHostToDevice);
            double ref[256];
            float f32[256];
            for (int m = 0; m < 16; ++m)
                for (int n = 0; n < 16; ++n) {
                    double s = 0;
                    float f = 0;
                    for (int k = 0; k < K; ++k) s += (double)hA[m * K + k] * hB[n * K + k], f = fmaf(hA[m * K + k], hB[n * K + k], f);
                    ref[m * 16 + n] = s, f32[m * 16 + n] = f;
                }
            for (int mode = 0; mode < 4; ++mode) {
                float c[256];
                hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, mode, dC);
                hipMemcpy(c, dC, 1024, hipMemcpyDeviceToHost);
                for (int i = 0; i < 256; ++i) {
                    double e = fabs(c[i] - ref[i]) / (1 << wscale_log2);
                    err[mode] += e, mx[mode] = e > mx[mode] ? e : mx[mode];
                }
            }
            for (int i = 0; i < 256; ++i) {
                double e = fabs(f32[i] - ref[i]) / (1 << wscale_log2);
                err[4] += e, mx[4] = e > mx[4] ? e : mx[4];
                ref_mag += fabs(ref[i]) / (1 << wscale_log2);
            }
            cnt += 256;
        }
        printf("K = %d dot products, A = ELU(U(-3,3)), W = U(-1/16,1/16) x 2^%d; mean |result| %.3g\n", K, wscale_log2, ref_mag / cnt);
        for (int mode = 0; mode < 4; ++mode) printf("  %-44s mean abs err %.3e  max %.3e\n", names[mode], err[mode] / cnt, mx[mode]);
        printf("  %-44s mean abs err %.3e  max %.3e\n", "fp32 FMA chain (host)", err[4] / cnt, mx[4]);
    }
    return 0;
}
