// jx_fastmath.hpp against the device library and against long double on the host: accuracy in ulp over the ranges the per-walker
// kernel meets (and far beyond), special values, and the cost of a chain shaped like the kernel's grid pass.
// build: hipcc -O3 --offload-arch=gfx950 -I../../joxsz_amd/csrc -o explog explog.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "jx_tables.hpp"
#include "jx_fastmath.hpp"
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_eval(const double* tab, const double* x, double* oe, double* ol, double* le, double* ll, int n) {
    __shared__ double st[JX_FM_TABLE_DOUBLES];
    for (int i = threadIdx.x; i < JX_FM_TABLE_DOUBLES; i += blockDim.x) st[i] = tab[i];
    __syncthreads();
    JxFm t{st, st + JX_FM_EXP_N};
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    oe[i] = jx_fm_exp(t, x[i]); le[i] = exp(x[i]);
    ol[i] = jx_fm_log(t, x[i]); ll[i] = log(x[i]);
}

// the shape of the grid pass: per "radius" four exp and three log in two dependent groups, two radii per lane in flight
template <int FAST>
__global__ void __launch_bounds__(256) k_chain(const double* tab, double* out, int trips, double seed) {
    __shared__ double st[JX_FM_TABLE_DOUBLES];
    for (int i = threadIdx.x; i < JX_FM_TABLE_DOUBLES; i += blockDim.x) st[i] = tab[i];
    __syncthreads();
    JxFm t{st, st + JX_FM_EXP_N};
    double acc = 0.0;
    for (int s = 0; s < trips; ++s) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double lx = seed + 1e-3 * threadIdx.x + 0.37 * s + 0.11 * h;
            const double xa = FAST ? jx_fm_exp(t, 1.05 * lx) : exp(1.05 * lx);
            const double l1 = FAST ? jx_fm_log(t, 1.0 + xa) : log(1.0 + xa);
            const double p = FAST ? jx_fm_exp(t, -(0.3 * lx + 4.0 * l1)) : exp(-(0.3 * lx + 4.0 * l1));
            const double u = FAST ? jx_fm_exp(t, 3.0 * (lx - 1.0)) : exp(3.0 * (lx - 1.0));
            const double l2 = FAST ? jx_fm_log(t, 1.0 + lx * lx) : log(1.0 + lx * lx);
            const double l3 = FAST ? jx_fm_log(t, 1.0 + u) : log(1.0 + u);
            const double ne = FAST ? jx_fm_exp(t, -(0.5 * lx + 1.2 * l2 + 0.9 * l3)) : exp(-(0.5 * lx + 1.2 * l2 + 0.9 * l3));
            acc += p / ne;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static double ulps(double got, long double want) {
    if (std::isnan(got) && std::isnan((double)want)) return 0.0;
    if (std::isinf(got) || std::isinf((double)want) || want == 0.0L) return (got == (double)want) ? 0.0 : 1e30;
    int e; frexpl(want, &e);
    const long double ulp = ldexpl(1.0L, e - 53);
    return (double)(fabsl((long double)got - want) / ulp);
}

int main() {
    std::vector<double> tab;
    jxt::fastmath_tables(tab);
    double* d_tab; CHK(hipMalloc(&d_tab, sizeof(double) * tab.size()));
    CHK(hipMemcpy(d_tab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
    std::mt19937_64 rng(1);
    struct Range { const char* name; double lo, hi; int logspace; };
    const Range ranges[] = {{"exp: [-745, 710]", -745.0, 710.0, 0}, {"exp: [-40, 40]", -40.0, 40.0, 0}, {"exp: [-1e-3, 1e-3]", -1e-3, 1e-3, 0},
                            {"log: [1e-300, 1e300]", 1e-300, 1e300, 1}, {"log: [0.5, 2]", 0.5, 2.0, 0}, {"log: [1 - 1e-6, 1 + 1e-6]", 1.0 - 1e-6, 1.0 + 1e-6, 0},
                            {"log: [1, 1e8] (1 + x^a)", 1.0, 1e8, 1}};
    const int n = 1 << 20;
    double *dx, *doe, *dol, *dle, *dll;
    CHK(hipMalloc(&dx, 8 * n)); CHK(hipMalloc(&doe, 8 * n)); CHK(hipMalloc(&dol, 8 * n)); CHK(hipMalloc(&dle, 8 * n)); CHK(hipMalloc(&dll, 8 * n));
    std::vector<double> x(n), oe(n), ol(n), le(n), ll(n);
    for (const Range& r : ranges) {
        std::uniform_real_distribution<double> U(0.0, 1.0);
        for (int i = 0; i < n; ++i) x[i] = r.logspace ? std::exp(std::log(r.lo) + U(rng) * (std::log(r.hi) - std::log(r.lo))) : r.lo + U(rng) * (r.hi - r.lo);
        CHK(hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_eval, dim3(n / 256), dim3(256), 0, 0, d_tab, dx, doe, dol, dle, dll, n);
        CHK(hipMemcpy(oe.data(), doe, 8 * n, hipMemcpyDeviceToHost)); CHK(hipMemcpy(ol.data(), dol, 8 * n, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(le.data(), dle, 8 * n, hipMemcpyDeviceToHost)); CHK(hipMemcpy(ll.data(), dll, 8 * n, hipMemcpyDeviceToHost));
        const bool is_exp = r.name[0] == 'e';
        double worst = 0.0, worst_lib = 0.0, sum = 0.0; double at = 0.0;
        for (int i = 0; i < n; ++i) {
            const long double want = is_exp ? expl((long double)x[i]) : logl((long double)x[i]);
            const double u = ulps(is_exp ? oe[i] : ol[i], want), ul = ulps(is_exp ? le[i] : ll[i], want);
            if (u > worst) { worst = u; at = x[i]; }
            worst_lib = std::max(worst_lib, ul); sum += u;
        }
        printf("%-32s table-driven: max %.3f ulp (at %.17g), mean %.3f | device library: max %.3f ulp\n", r.name, worst, at, sum / n, worst_lib);
    }
    // special values
    const double sp[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, NAN, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 709.782712893384, 709.79, -745.13, -745.14, -800.0, 800.0, 1e-320};
    const int ns = sizeof(sp) / sizeof(sp[0]);
    CHK(hipMemcpy(dx, sp, 8 * ns, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_eval, dim3(1), dim3(256), 0, 0, d_tab, dx, doe, dol, dle, dll, ns);
    CHK(hipMemcpy(oe.data(), doe, 8 * ns, hipMemcpyDeviceToHost)); CHK(hipMemcpy(ol.data(), dol, 8 * ns, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(le.data(), dle, 8 * ns, hipMemcpyDeviceToHost)); CHK(hipMemcpy(ll.data(), dll, 8 * ns, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < ns; ++i) {
        const bool eq_e = (oe[i] == le[i]) || (std::isnan(oe[i]) && std::isnan(le[i])) || std::fabs(oe[i] - le[i]) <= 4e-16 * std::fabs(le[i]);
        const bool eq_l = (ol[i] == ll[i]) || (std::isnan(ol[i]) && std::isnan(ll[i])) || std::fabs(ol[i] - ll[i]) <= 4e-16 * std::fabs(ll[i]);
        if (!eq_e || !eq_l) ++bad;
        printf("x = %-24.17g exp: %-24.17g (library %-24.17g) log: %-24.17g (library %-24.17g)%s\n", sp[i], oe[i], le[i], ol[i], ll[i], (eq_e && eq_l) ? "" : "   <-- differs");
    }
    printf("special values that differ from the device library: %d\n", bad);
    // cost
    double* dout; CHK(hipMalloc(&dout, 8 * 1024 * 256));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int fast = 0; fast < 2; ++fast) {
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0, 0));
            if (fast) hipLaunchKernelGGL(k_chain<1>, dim3(1024), dim3(256), 0, 0, d_tab, dout, 64, 0.5);
            else hipLaunchKernelGGL(k_chain<0>, dim3(1024), dim3(256), 0, 0, d_tab, dout, 64, 0.5);
            CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        }
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: 1024 blocks x 256 threads x 128 radii (4 exp + 3 log each): %.1f us\n", fast ? "table-driven " : "device library", ms * 1e3);
    }
    return bad ? 2 : 0;
}
