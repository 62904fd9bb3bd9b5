// microbenchmark: a "multiply tick" of the ping-pong pair kernel in isolation.  Per CU one workgroup of 8 wavefronts: wavefronts 0-3
// (one per SIMD) run MFMAs fed by ds_read_b128 fragments from a 60 KB LDS chunk, wavefronts 4-7 (their SIMD partners) run a VALU loop
// shaped like the A-piece build (adds, fmas, an exp, the three-way split).  Which MFMA shape keeps the pipe busier under that load?
//   shape 0: v_mfma_f32_16x16x32_bf16, 2 row tiles x 19 column tiles, 3 fragments and 12 MFMAs per column tile  (today's kernel)
//   shape 1: v_mfma_f32_32x32x16_bf16, 1 row tile x 10 column tiles x 2 k blocks, 3 fragments and 6 MFMAs per (tile, k block)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool PARTNER>
__global__ __launch_bounds__(512) void tick(float* out, int ticks) {
    __shared__ __attribute__((aligned(16))) u32x4 chunk[3840];          // 60 KB
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 3840; i += 512) chunk[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f003f00u, 0x3f803f80u};
    __syncthreads();
    float sink = 0.f;
    if (wave < 4) {
        bf16x8 a[2][3];
        for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) for (int j = 0; j < 8; ++j) a[i][p][j] = (__bf16)(float)(((lane + i + p + j) & 7) * 0.25f);
        if (SHAPE == 0) {
            floatx4 acc[2][19];
            for (int i = 0; i < 2; ++i) for (int t = 0; t < 19; ++t) acc[i][t] = floatx4{0, 0, 0, 0};
            for (int it = 0; it < ticks; ++it) {
#pragma unroll
                for (int t = 0; t < 19; ++t) {
                    bf16x8 b[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, chunk[(p * 20 + t) * 64 + lane]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int x = 0; x < 6; ++x) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][x % 3], b[(x + 1) % 3], acc[i][t], 0, 0, 0);
                }
                __syncthreads();
            }
            for (int i = 0; i < 2; ++i) for (int t = 0; t < 19; ++t) sink += acc[i][t][0];
        } else {
            floatx16 acc[10];
            for (int t = 0; t < 10; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0;
            for (int it = 0; it < ticks; ++it) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int t = 0; t < 10; ++t) {
                        bf16x8 b[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, chunk[((kb * 3 + p) * 10 + t) * 64 + lane]);
#pragma unroll
                        for (int x = 0; x < 6; ++x) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][x % 3], b[(x + 1) % 3], acc[t], 0, 0, 0);
                    }
                __syncthreads();
            }
            for (int t = 0; t < 10; ++t) sink += acc[t][0] + acc[t][7];
        }
    } else {
        // partner: ~600 VALU + 16 transcendental instructions per tick, like building 16 A elements per lane
        float v[16];
        for (int j = 0; j < 16; ++j) v[j] = 0.001f * (lane + j);
        for (int it = 0; it < ticks; ++it) {
            if (PARTNER) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float x = v[j];
#pragma unroll
                    for (int r = 0; r < 8; ++r) x = __builtin_fmaf(x, 1.0001f, 0.37f * r);
                    x = fmaxf(x, __builtin_amdgcn_exp2f(fminf(x, 0.f) * 1.44f) - 1.f);
                    unsigned h = __float_as_uint(x) & 0xffff0000u;
                    float rr = x - __uint_as_float(h);
                    unsigned m = __float_as_uint(rr) & 0xffff0000u;
                    float l = rr - __uint_as_float(m);
                    v[j] = __uint_as_float(h) * 0.5f + __uint_as_float(m) + l;
#pragma unroll
                    for (int r = 0; r < 20; ++r) v[j] = __builtin_fmaf(v[j], 0.999f, 0.001f);
                }
            }
            __syncthreads();
        }
        for (int j = 0; j < 16; ++j) sink += v[j];
    }
    if (sink == 12345.678f) out[threadIdx.x] = sink;
}

// Schedule "merged": no ping-pong.  All 8 wavefronts run the same loop - per column tile 3 fragment reads, 12 MFMAs and 1/19 of the build
// work of the NEXT chunk (VALU of the same wavefront, issued while its MFMAs execute) - so both wavefronts of a SIMD issue MFMAs.
// One barrier per chunk.  A "tick" here serves 2 x 228 MFMAs per SIMD: compare with TWO ping-pong ticks.
__global__ __launch_bounds__(512) void tick_merged(float* out, int ticks) {
    __shared__ __attribute__((aligned(16))) u32x4 chunk[3840];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 3840; i += 512) chunk[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f003f00u, 0x3f803f80u};
    __syncthreads();
    bf16x8 a[2][3];
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) for (int j = 0; j < 8; ++j) a[i][p][j] = (__bf16)(float)(((lane + i + p + j) & 7) * 0.25f);
    floatx4 acc[2][19];
    for (int i = 0; i < 2; ++i) for (int t = 0; t < 19; ++t) acc[i][t] = floatx4{0, 0, 0, 0};
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = 0.001f * (lane + j);
    for (int it = 0; it < ticks; ++it) {
#pragma unroll
        for (int t = 0; t < 19; ++t) {
            bf16x8 b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, chunk[(p * 20 + t) * 64 + lane]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int x = 0; x < 6; ++x) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][x % 3], b[(x + 1) % 3], acc[i][t], 0, 0, 0);
            if (t < 16) {                                   // one element of the build per tile (16 elements per chunk)
                float x = v[t];
#pragma unroll
                for (int r = 0; r < 8; ++r) x = __builtin_fmaf(x, 1.0001f, 0.37f * r);
                x = fmaxf(x, __builtin_amdgcn_exp2f(fminf(x, 0.f) * 1.44f) - 1.f);
                unsigned h = __float_as_uint(x) & 0xffff0000u;
                float rr = x - __uint_as_float(h);
                unsigned m = __float_as_uint(rr) & 0xffff0000u;
                float l = rr - __uint_as_float(m);
                v[t] = __uint_as_float(h) * 0.5f + __uint_as_float(m) + l;
#pragma unroll
                for (int r = 0; r < 20; ++r) v[t] = __builtin_fmaf(v[t], 0.999f, 0.001f);
            }
        }
        __syncthreads();
    }
    float sink = 0.f;
    for (int i = 0; i < 2; ++i) for (int t = 0; t < 19; ++t) sink += acc[i][t][0];
    for (int j = 0; j < 16; ++j) sink += v[j];
    if (sink == 12345.678f) out[threadIdx.x] = sink;
}

template <int SHAPE, bool PARTNER>
void run(const char* name, float* dout) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ticks = 4000;
    hipLaunchKernelGGL((tick<SHAPE, PARTNER>), dim3(256), dim3(512), 0, 0, dout, 200);
    hipEventRecord(e0);
    hipLaunchKernelGGL((tick<SHAPE, PARTNER>), dim3(256), dim3(512), 0, 0, dout, ticks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_cycles = SHAPE == 0 ? 228 * 16.0 : 120 * 32.0;   // pipe cycles of one tick
    printf("%-52s %.3f ms: %.2f us per tick; pipe-busy cycles per tick %.0f -> busy %.0f %% at 2.1 GHz\n", name, ms, ms * 1e3 / ticks, mfma_cycles,
           mfma_cycles / (ms * 1e-3 / ticks * 2.1e9) * 100);
}
int main() {
    float* dout; hipMalloc(&dout, 4096);
    run<0, false>("16x16x32, partner idle", dout);
    run<0, true>("16x16x32, partner building", dout);
    run<1, false>("32x32x16, partner idle", dout);
    run<1, true>("32x32x16, partner building", dout);
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int ticks = 2000;
        hipLaunchKernelGGL(tick_merged, dim3(256), dim3(512), 0, 0, dout, 200);
        hipEventRecord(e0);
        hipLaunchKernelGGL(tick_merged, dim3(256), dim3(512), 0, 0, dout, ticks);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %.3f ms: %.2f us per chunk of BOTH wavefronts of a SIMD (= two ping-pong ticks); busy %.0f %% at 2.1 GHz\n",
               "16x16x32, merged (every wavefront builds and multiplies)", ms, ms * 1e3 / ticks, 2 * 3648.0 / (ms * 1e-3 / ticks * 2.1e9) * 100);
    }
    return 0;
}
