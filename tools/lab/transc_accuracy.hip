// Accuracy of the hardware transcendentals the logic kernels use, on the argument ranges that matter (round 3):
//   v_log_f32 near 1 (the log of a product of (1 - p) factors), v_exp_f32 of x * log2(e) for log-probabilities x in [-30, 0].
// Build: hipcc --offload-arch=gfx950 -O3 tools/scratch/transc_accuracy.hip -o gpurun_out/transc_accuracy ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>

#include "../../dfol_vqa_amd/csrc/dfol_common.h"

void dfol_set_error(const char*, ...) {}

__global__ void k_eval(const float* x, float* o_hwlog, float* o_fixlog, float* o_hwexp, float* o_fixexp, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_hwlog[i] = __builtin_amdgcn_logf(x[i]) * 0.69314718055994530942f;
    o_fixlog[i] = dfol_log(x[i]);
    o_hwexp[i] = __builtin_amdgcn_exp2f(x[i] * 1.44269504088896340736f);
    o_fixexp[i] = dfol_exp(x[i]);
}

static void report(const char* what, const std::vector<float>& x, const std::vector<float>& got, bool is_log) {
    double max_abs = 0, max_rel = 0, sum_rel = 0;
    for (size_t i = 0; i < x.size(); ++i) {
        double ref = is_log ? log((double)x[i]) : exp((double)x[i]);
        double e = fabs((double)got[i] - ref);
        double r = ref != 0 ? e / fabs(ref) : 0;
        if (e > max_abs) max_abs = e;
        if (r > max_rel) max_rel = r;
        sum_rel += r;
    }
    printf("%-44s max abs %.3e  max rel %.3e  mean rel %.3e\n", what, max_abs, max_rel, sum_rel / x.size());
}

int main() {
    const int n = 1 << 20;
    struct Range { const char* name; double lo, hi; bool is_log; };
    Range ranges[] = {{"log x, x in [0.999, 1]", 0.999, 1.0, true},  {"log x, x in [0.99, 1]", 0.99, 1.0, true},
                      {"log x, x in [0.9, 1]", 0.9, 1.0, true},      {"log x, x in [0.5, 1]", 0.5, 1.0, true},
                      {"log x, x in [1, 2]", 1.0, 2.0, true},        {"log x, x in [1e-6, 0.5]", 1e-6, 0.5, true},
                      {"exp x, x in [-1, 0]", -1.0, 0.0, false},     {"exp x, x in [-8, -1]", -8.0, -1.0, false},
                      {"exp x, x in [-30, -8]", -30.0, -8.0, false}};
    float *dx, *d0, *d1, *d2, *d3;
    hipMalloc(&dx, n * 4), hipMalloc(&d0, n * 4), hipMalloc(&d1, n * 4), hipMalloc(&d2, n * 4), hipMalloc(&d3, n * 4);
    for (auto& r : ranges) {
        std::vector<float> x(n), a(n), b(n), c(n), d(n);
        unsigned long long s = 88172645463325252ull;
        for (int i = 0; i < n; ++i) {
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            x[i] = (float)(r.lo + (r.hi - r.lo) * ((s >> 11) * (1.0 / 9007199254740992.0)));
        }
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_eval, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
        hipMemcpy(a.data(), d0, n * 4, hipMemcpyDeviceToHost), hipMemcpy(b.data(), d1, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), d2, n * 4, hipMemcpyDeviceToHost), hipMemcpy(d.data(), d3, n * 4, hipMemcpyDeviceToHost);
        printf("== %s\n", r.name);
        if (r.is_log) {
            report("  v_log_f32 * ln2", x, a, true);
            report("  dfol_log", x, b, true);
        } else {
            report("  v_exp_f32(x * log2e)", x, c, false);
            report("  dfol_exp", x, d, false);
        }
    }
    return 0;
}
