// Device pieces of the attention-calibration passes shared by their stand-alone kernels (dfol_logic.hip) and the walk kernel that runs a whole
// run of them in one launch (dfol_program.hip: calib_walk_kernel).  Reference: batch_base_interpreter.py:87-140, batch_base_ops.py:265-286.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The whole cell in one launch.  A workgroup (512 threads) takes LC_ROWS = 16 rows x LC_UNITS = 8 hidden units (their 32 gate columns i, f, g, o) and
// stages the rows' x and h in LDS (row-interleaved: four 16-byte LDS broadcasts hand a weight's sixteen inputs over); thread (column, K slice) walks one
// of 16 slices of K = KX + H with all of its weight loads in flight at once (the weights come TRANSPOSED, [K, 4H]); the slices meet in LDS, then the
// pointwise stage.  What bounds a cell is the bytes one CU can pull through its L1 under L2 latency (~25 B per clock measured here): the first two
// versions gave a workgroup 4 rows and ALL 200 gate columns, i.e. the whole 294 KB of weights per CU - 16 us per cell at 256 rows x (318 + 50) -> 200
// whichever way the loads were scheduled (one K slice with 8 loads in flight, or two with 16 + 16 prefetched); this tiling pulls 24 KB of inputs and
// 47 KB of weights per workgroup over 16 x 7 workgroups.  Eight cells per calibrated forward.
// LcTokens: the rows of x built in the staging loop from an operator's tokens (what dfol_calib_features_f32 would write: [head | table[idx[row]]], or
// zeros for a no-op token) instead of read - x == nullptr.
constexpr int LC_ROWS = 16, LC_UNITS = 8, LC_COLS = 4 * LC_UNITS, LC_THREADS = 512, LC_SLICES = LC_THREADS / LC_COLS, LC_MAXK = 24, LC_PS = LC_COLS + 1;
struct LcTokens {
    const float* head; int n_head;
    const float* table; int E;
    const int32_t* idx;
};
struct LcCell {
    const float* x; int64_t ld_x; int KX;                      // x == nullptr: rows from `tk`
    const float* h; int64_t ld_h;
    const float* c;
    const float* Wih; int64_t ld_wih;
    const float* Whh; int64_t ld_whh;
    const float* bih; const float* bhh;
    int rows, H;
    float* hy; float* cy; float* gates_out;
    LcTokens tk;
};
__host__ __device__ inline size_t lc_lds_floats(int KX, int H) { return (size_t)LC_ROWS * ((size_t)KX + H + (size_t)LC_SLICES * LC_PS); }

// Stage the inputs of rows r0 .. r0 + LC_ROWS - 1 as in_s[k][row] (no barrier).  A lane keeps its row and walks k (four consecutive k per wavefront:
// 16-byte pieces of sixteen rows, LDS stores without bank conflicts - lanes along k, the first version, put 32 lanes on one bank).
__device__ __forceinline__ void lc_stage(const LcCell& p, int r0, float* __restrict__ in_s) {
    static_assert(LC_THREADS % LC_ROWS == 0, "lstm_cell: staging map");
    const int tid = threadIdx.x, K = p.KX + p.H;
    const int r = tid % LC_ROWS, row = min(r0 + r, p.rows - 1);
    const int t = p.x ? 0 : p.tk.idx[row];
    for (int k = tid / LC_ROWS; k < K; k += LC_THREADS / LC_ROWS) {
        float v;
        if (k >= p.KX) v = p.h[(int64_t)row * p.ld_h + (k - p.KX)];
        else if (p.x) v = p.x[(int64_t)row * p.ld_x + k];
        else v = t < 0 ? 0.f : (k < p.tk.n_head ? p.tk.head[k] : p.tk.table[(int64_t)t * p.tk.E + (k - p.tk.n_head)]);
        in_s[k * LC_ROWS + r] = v;
    }
}

// Hidden units j0 .. j0 + LC_UNITS - 1 of the staged rows: gate products over 16 K slices, the slices' sums, the pointwise stage.  Two barriers: the
// first stands between the staging (or the previous call's reads of part_s) and the products, the second between the partial sums and their readers.
__device__ __forceinline__ void lc_units(const LcCell& p, int r0, int j0, const float* __restrict__ in_s, float* __restrict__ part_s) {
    const int tid = threadIdx.x, K = p.KX + p.H, H = p.H, KX = p.KX;
    const int col = tid % LC_COLS, ks = tid / LC_COLS;             // column = (gate q, unit jj)
    const int q = col / LC_UNITS, j = min(j0 + col % LC_UNITS, H - 1), g = q * H + j;
    float acc[LC_ROWS];
#pragma unroll
    for (int r = 0; r < LC_ROWS; ++r) acc[r] = 0.f;
    // the slice's weights: its first LC_MAXK loads are issued before the barrier on the staged inputs (they do not depend on them)
    const int kb0 = (int)((int64_t)K * ks / LC_SLICES), ke = (int)((int64_t)K * (ks + 1) / LC_SLICES);
    float w[LC_MAXK];
    auto load = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < LC_MAXK; ++u) {
            const int k = min(kb + u, K - 1);
            w[u] = k < KX ? p.Wih[(int64_t)k * p.ld_wih + g] : p.Whh[(int64_t)(k - KX) * p.ld_whh + g];
        }
    };
    load(kb0);
    __syncthreads();
    for (int kb = kb0; kb < ke; kb += LC_MAXK) {
        if (kb != kb0) load(kb);
#pragma unroll
        for (int u = 0; u < LC_MAXK; ++u) {
            if (kb + u < ke) {
                const float4* v = reinterpret_cast<const float4*>(in_s + LC_ROWS * (kb + u));
#pragma unroll
                for (int r4 = 0; r4 < LC_ROWS / 4; ++r4) {
                    const float4 t = v[r4];
                    acc[4 * r4] = fmaf(w[u], t.x, acc[4 * r4]), acc[4 * r4 + 1] = fmaf(w[u], t.y, acc[4 * r4 + 1]);
                    acc[4 * r4 + 2] = fmaf(w[u], t.z, acc[4 * r4 + 2]), acc[4 * r4 + 3] = fmaf(w[u], t.w, acc[4 * r4 + 3]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < LC_ROWS; ++r) part_s[(ks * LC_ROWS + r) * LC_PS + col] = acc[r];
    __syncthreads();
    if (tid < LC_ROWS * LC_UNITS) {
        const int r = tid / LC_UNITS, jj = tid % LC_UNITS, row = r0 + r, ju = j0 + jj;
        if (row < p.rows && ju < H) {
            float gv[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                // (the slices' sums in two halves, each in slice order - the order lc_wide's two lanes of a column produce)
                float lo = 0.f, hi = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < LC_SLICES / 2; ++s2) {
                    lo += part_s[(s2 * LC_ROWS + r) * LC_PS + qq * LC_UNITS + jj];
                    hi += part_s[((s2 + LC_SLICES / 2) * LC_ROWS + r) * LC_PS + qq * LC_UNITS + jj];
                }
                const int gg = qq * H + ju;
                gv[qq] = (lo + hi) + ((p.bih ? p.bih[gg] : 0.f) + (p.bhh ? p.bhh[gg] : 0.f));
            }
            const float si = 1.0f / (1.0f + expf(-gv[0])), sf = 1.0f / (1.0f + expf(-gv[1])), so = 1.0f / (1.0f + expf(-gv[3]));
            const float tg = tanhf(gv[2]);
            const float cn = sf * p.c[(int64_t)row * H + ju] + si * tg;
            p.cy[(int64_t)row * H + ju] = cn;
            p.hy[(int64_t)row * H + ju] = so * tanhf(cn);
            if (p.gates_out) {                               // training: the activated gates (i, f, g, o) for dfol_lstm_cell_bwd_f32
                float* go_ = p.gates_out + (int64_t)row * 4 * H;
                go_[ju] = si, go_[H + ju] = sf, go_[2 * H + ju] = tg, go_[3 * H + ju] = so;
            }
        }
    }
}

// The same cell for ALL hidden units of the staged rows in one pass (the walk kernel: one workgroup per row block, nobody to share the columns with):
// thread (gate column g, half) walks eight of the sixteen K slices one after the other - each slice its own fmaf chain from zero, added to the running
// sum in slice order - so that a gate is (slices 0..7) + (slices 8..15), the bits of lc_units.  part_s: [2][LC_ROWS][4H + 1] floats.  One barrier
// inside (the caller puts one between the staging and this call).
__host__ __device__ inline size_t lc_wide_lds_floats(int KX, int H) { return (size_t)LC_ROWS * ((size_t)KX + H + 2 * ((size_t)4 * H + 1)); }
__device__ __forceinline__ void lc_wide(const LcCell& p, int r0, const float* __restrict__ in_s, float* __restrict__ part_s) {
    static_assert(LC_SLICES % 2 == 0 && LC_THREADS == 512, "lstm_cell: two halves of the K slices");
    const int tid = threadIdx.x, K = p.KX + p.H, H = p.H, KX = p.KX, G = 4 * H, PS = G + 1;
    const int half = tid / 256;
    for (int g = tid % 256; g < G; g += 256) {
        float run[LC_ROWS];
#pragma unroll
        for (int r = 0; r < LC_ROWS; ++r) run[r] = 0.f;
        for (int sl = half * (LC_SLICES / 2); sl < (half + 1) * (LC_SLICES / 2); ++sl) {
            const int kb0 = (int)((int64_t)K * sl / LC_SLICES), ke = (int)((int64_t)K * (sl + 1) / LC_SLICES);
            float acc[LC_ROWS];
#pragma unroll
            for (int r = 0; r < LC_ROWS; ++r) acc[r] = 0.f;
            for (int kb = kb0; kb < ke; kb += LC_MAXK) {
                float w[LC_MAXK];
#pragma unroll
                for (int u = 0; u < LC_MAXK; ++u) {
                    const int k = min(kb + u, K - 1);
                    w[u] = k < KX ? p.Wih[(int64_t)k * p.ld_wih + g] : p.Whh[(int64_t)(k - KX) * p.ld_whh + g];
                }
#pragma unroll
                for (int u = 0; u < LC_MAXK; ++u) {
                    if (kb + u < ke) {
                        const float4* v = reinterpret_cast<const float4*>(in_s + LC_ROWS * (kb + u));
#pragma unroll
                        for (int r4 = 0; r4 < LC_ROWS / 4; ++r4) {
                            const float4 t = v[r4];
                            acc[4 * r4] = fmaf(w[u], t.x, acc[4 * r4]), acc[4 * r4 + 1] = fmaf(w[u], t.y, acc[4 * r4 + 1]);
                            acc[4 * r4 + 2] = fmaf(w[u], t.z, acc[4 * r4 + 2]), acc[4 * r4 + 3] = fmaf(w[u], t.w, acc[4 * r4 + 3]);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < LC_ROWS; ++r) run[r] += acc[r];
        }
#pragma unroll
        for (int r = 0; r < LC_ROWS; ++r) part_s[(half * LC_ROWS + r) * PS + g] = run[r];
    }
    __syncthreads();
    for (int i = tid; i < LC_ROWS * H; i += LC_THREADS) {
        const int r = i / H, ju = i - r * H, row = r0 + r;
        if (row >= p.rows) continue;
        float gv[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int gg = qq * H + ju;
            gv[qq] = (part_s[r * PS + gg] + part_s[(LC_ROWS + r) * PS + gg]) + ((p.bih ? p.bih[gg] : 0.f) + (p.bhh ? p.bhh[gg] : 0.f));
        }
        const float si = 1.0f / (1.0f + expf(-gv[0])), sf = 1.0f / (1.0f + expf(-gv[1])), so = 1.0f / (1.0f + expf(-gv[3]));
        const float tg = tanhf(gv[2]);
        const float cn = sf * p.c[(int64_t)row * H + ju] + si * tg;
        p.cy[(int64_t)row * H + ju] = cn;
        p.hy[(int64_t)row * H + ju] = so * tanhf(cn);
    }
}

// The attention-output network on 16 rows (p0 ..): sixteen lanes per row, 256 threads (whole wavefronts: the lanes meet in a butterfly).  See
// attention_modulations_kernel in dfol_logic.hip.
constexpr int AM_N = 8;
__device__ __forceinline__ void am_rows(const float* __restrict__ fs, const float* __restrict__ bs, const float* __restrict__ W, int64_t ld_w,
                                        const float* __restrict__ b, int P, int S, int N, float* __restrict__ out, int p0, int tid) {
    const int l = tid & 15, p = p0 + (tid >> 4);
    const int pc = min(p, P - 1);
    for (int j0 = 0; j0 < N; j0 += AM_N) {
        float acc[AM_N];
#pragma unroll
        for (int j = 0; j < AM_N; ++j) acc[j] = 0.f;
        for (int half = 0; half < 2; ++half) {
            const float* st = half ? bs : fs;
            if (!st) continue;
            for (int k = l; k < S; k += 16) {
                const float v = st[(int64_t)pc * S + k];
#pragma unroll
                for (int j = 0; j < AM_N; ++j)
                    if (j0 + j < N) acc[j] = fmaf(W[(int64_t)(j0 + j) * ld_w + half * S + k], v, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < AM_N; ++j) {
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) acc[j] += __shfl_xor(acc[j], m, 16);
        }
        if (l == 0 && p < P) {
#pragma unroll
            for (int j = 0; j < AM_N; ++j)
                if (j0 + j < N) out[(int64_t)p * N + j0 + j] = 1.0f / (1.0f + expf(-(acc[j] + (b ? b[j0 + j] : 0.f))));
        }
    }
}
