// Lab: can ONE wavefront hide the pair kernel's build arithmetic (the ELU + fp16 split of the next chunk's A pieces) and its B-fragment LDS
// reads in the gaps of its OWN MFMA stream?  (The ping-pong of two wavefronts per SIMD cannot: tools/lab/coissue.hip - a VALU stream beside the
// other wavefront's MFMA chains gets one issue slot per MFMA, and priorities / MFMA order change nothing inside the kernel,
// profiles/r05_pair_ab_order_prio.txt.)  Stream per "column tile": TS slot tiles x 3 dependent MFMAs (v_mfma_f32_16x16x32_f16, the
// kernel's chains), F filler VALU instructions of the build's mix, R ds_read_b128 - interleaved by sched_group_barrier.  One wavefront per
// SIMD (256-thread workgroups, 512 registers) or two (512 threads).
//   hipcc --offload-arch=gfx950 -O3 tools/lab/interleave.hip -o build/interleave && build/interleave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// one element of the build: ELU with saturation, then the two fp16 pieces of a PAIR of elements (7 VALU per element as the kernel counts them)
__device__ __forceinline__ void build_pair(float z0, float z1, uint32_t& hh, uint32_t& ll) {
    const float e0 = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(z0, 0.f, -3e38f) * 1.4426950408889634f) - 1.0f;
    const float e1 = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(z1, 0.f, -3e38f) * 1.4426950408889634f) - 1.0f;
    const float a0 = __builtin_amdgcn_fmed3f(z0, e0, 60000.f), a1 = __builtin_amdgcn_fmed3f(z1, e1, 60000.f);
    const f16x2 h = __builtin_convertvector((f32x2){a0, a1}, f16x2);
    const f32x2 r = (f32x2){a0, a1} - __builtin_convertvector(h, f32x2);
    hh = __builtin_bit_cast(uint32_t, h);
    ll = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}

// the interleave pattern of one chunk: per MFMA gap its share of the build's VALU instructions, per column tile its two B-fragment reads
template <int G, int M, int V, int TS3, bool RD>
__device__ __forceinline__ void emit_groups() {
    if constexpr (G < M) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (RD && G % TS3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        constexpr int v = (V * (G + 1)) / M - (V * G) / M;
        if constexpr (v > 0) __builtin_amdgcn_sched_group_barrier(0x002, v, 0);
        emit_groups<G + 1, M, V, TS3, RD>();
    }
}

// TS slot tiles, NT column tiles per "chunk", PAIRS element pairs built per chunk, READS ds_read_b128 per column tile, GROUPS: use sched_group_barrier
template <int TS, int NT, int PAIRS, int READS, bool GROUPS, int THREADS>
__global__ __launch_bounds__(THREADS) void k(long long* out, int reps, float seed, float* sinkbuf) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ __attribute__((aligned(16))) u32x4 Bs[2560];
    for (int i = threadIdx.x; i < 2560; i += THREADS) Bs[i] = u32x4{(uint32_t)i, 1u, 2u, 3u};
    __syncthreads();
    floatx4 acc[TS][NT];
#pragma unroll
    for (int m = 0; m < TS; ++m)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[m][i] = floatx4{0, 0, 0, 0};
    u32x4 ap[TS][2];
#pragma unroll
    for (int m = 0; m < TS; ++m) ap[m][0] = u32x4{1u, 2u, 3u, 4u}, ap[m][1] = u32x4{5u, 6u, 7u, 8u};
    float z[2 * PAIRS > 0 ? 2 * PAIRS : 1];
#pragma unroll
    for (int j = 0; j < 2 * PAIRS; ++j) z[j] = seed * (j + 1) + lane * 1e-3f;
    uint32_t nh[PAIRS > 0 ? PAIRS : 1], nl[PAIRS > 0 ? PAIRS : 1];
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        // the build of the NEXT chunk's pieces: independent of this chunk's MFMAs (the scheduler may place it anywhere in the region)
#pragma unroll
        for (int j = 0; j < PAIRS; ++j) build_pair(z[2 * j], z[2 * j + 1], nh[j], nl[j]);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            f16x8 b[2];
            if (READS > 0) {
                b[0] = __builtin_bit_cast(f16x8, Bs[(i * 64 + lane) % 2560]);
                b[1] = __builtin_bit_cast(f16x8, Bs[(i * 64 + lane + 1280) % 2560]);
            } else {
                b[0] = __builtin_bit_cast(f16x8, ap[0][0]);
                b[1] = __builtin_bit_cast(f16x8, ap[0][1]);
            }
#pragma unroll
            for (int m = 0; m < TS; ++m) {
                acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ap[m][1]), b[0], acc[m][i], 0, 0, 0);
                acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ap[m][0]), b[1], acc[m][i], 0, 0, 0);
                acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ap[m][0]), b[0], acc[m][i], 0, 0, 0);
            }
        }
        if constexpr (GROUPS) emit_groups<0, TS * NT * 3, PAIRS * 14, TS * 3, (READS > 0)>();
        // the built pieces become the next chunk's A operands (keeps the build alive and creates the real dependency)
#pragma unroll
        for (int m = 0; m < TS; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (PAIRS > 0) ap[m][0][e] ^= nh[(m * 4 + e) % PAIRS] & 1u, ap[m][1][e] ^= nl[(m * 4 + e) % PAIRS] & 1u;
#pragma unroll
        for (int j = 0; j < 2 * PAIRS; ++j) z[j] = z[j] * 0.999f + 1e-4f;
    }
    const long long t1 = clock64();
    float sink = 0.f;
#pragma unroll
    for (int m = 0; m < TS; ++m)
#pragma unroll
        for (int i = 0; i < NT; ++i) sink += acc[m][i][0] + acc[m][i][3];
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (sink == 12345.f) sinkbuf[0] = sink;
}

template <int TS, int NT, int PAIRS, int READS, bool GROUPS, int THREADS>
void run(const char* name) {
    long long* d;
    float* sb;
    hipMalloc(&d, 256 * 8 * 8);
    hipMalloc(&sb, 64);
    const int reps = 400;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<TS, NT, PAIRS, READS, GROUPS, THREADS>), dim3(256), dim3(THREADS), 0, 0, d, reps, 0.37f, sb);
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, d + 100 * 8, sizeof(h), hipMemcpyDeviceToHost);
    const int mf = TS * NT * 3;
    printf("%-78s %6.1f cycles per MFMA (%d MFMAs, %d build VALU, %d ds_read_b128 per chunk; %.2f fillers per gap)\n", name, (double)h[0] / reps / mf, mf,
           PAIRS * 14, READS ? 2 * NT : 0, (PAIRS * 14 + (READS ? 2 * NT : 0)) / (double)mf);
    hipFree(d);
    hipFree(sb);
}

int main() {
    // the kernel's chunk: 19 column tiles; a wavefront's slot tiles: 2 (today), 3 or 4; the build: 8 element pairs per slot tile and chunk
    run<2, 19, 0, 0, false, 256>("1 wave/SIMD, 2 slot tiles, MFMAs only");
    run<3, 19, 0, 0, false, 256>("1 wave/SIMD, 3 slot tiles, MFMAs only");
    run<3, 19, 0, 1, false, 256>("1 wave/SIMD, 3 slot tiles, + B reads, compiler order");
    run<3, 19, 0, 1, true, 256>("1 wave/SIMD, 3 slot tiles, + B reads, grouped");
    run<3, 19, 24, 1, false, 256>("1 wave/SIMD, 3 slot tiles, + B reads + build (24 pairs), compiler order");
    run<3, 19, 24, 1, true, 256>("1 wave/SIMD, 3 slot tiles, + B reads + build (24 pairs), grouped");
    run<3, 19, 48, 1, true, 256>("1 wave/SIMD, 3 slot tiles, + B reads + 2x build (48 pairs), grouped");
    run<4, 19, 32, 1, true, 256>("1 wave/SIMD, 4 slot tiles, + B reads + build (32 pairs), grouped");
    run<4, 19, 32, 1, false, 256>("1 wave/SIMD, 4 slot tiles, + B reads + build (32 pairs), compiler order");
    run<2, 19, 16, 1, true, 512>("2 waves/SIMD, 2 slot tiles, + B reads + build (16 pairs), grouped");
    run<2, 19, 16, 1, false, 512>("2 waves/SIMD, 2 slot tiles, + B reads + build (16 pairs), compiler order");
    run<2, 19, 0, 1, true, 512>("2 waves/SIMD, 2 slot tiles, + B reads, grouped");
    return 0;
}
