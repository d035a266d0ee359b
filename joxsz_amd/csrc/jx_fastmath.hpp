// exp and log in fp64 for the per-walker kernel's profile chains: table-driven (Tang 1989/1990; the layout of the log follows
// the ARM optimized routines: 128 intervals of the mantissa range, z * (1/c) - 1 by one fused multiply-add), about 16 and 22
// vector instructions against 38 and 95 of the device library's -- those carry their results in double-double to stay under
// one ulp; these are held to 2 ulp against long double (scripts/ubench/explog.hip, tests/test_gpu_fastmath.py), which the
// profiles do not notice: they enter exponents additively or are exponentiated once.  Tables: 64 + 2 x 128 doubles, built on the
// host in long double (jxt::fastmath_tables), staged in LDS by the kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define JX_FM_EXP_N 64
#define JX_FM_LOG_N 128
#define JX_FM_TABLE_DOUBLES (JX_FM_EXP_N + 2 * JX_FM_LOG_N)
#define JX_FM_LOG_OFF_HI 0x3FE5F000            // high word of the lower end of the reduced range: 1.0 is the centre of interval 80

struct JxFm {
    const double* et;          // [64] 2^(j/64)
    const double* lt;          // [128][2] (1/c_i, log c_i)
};

__device__ __forceinline__ double jx_fm_exp(const JxFm& t, double x) {
    // exp(x) = 2^m 2^(j/64) exp(r), k = 64 m + j = rint(64 x / ln 2), |r| <= ln 2 / 128
    x = (x > 800.0) ? 800.0 : x;                                  // (overflow and underflow happen in the ldexp; a NaN passes through)
    x = (x < -800.0) ? -800.0 : x;
    const double kd = rint(x * 92.332482616893657);               // 64 / ln 2
    double r = fma(kd, -1.08304246932675596e-02, x);              // ln 2 / 64, high part (its last 21 bits are zero: kd * it is exact)
    r = fma(kd, -2.98158582698529328e-12, r);                     // low part
    const int k = (int)kd;
    double p = fma(r, 1.0 / 720.0, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = p * r;                                                    // exp(r) - 1
    const double s = t.et[k & (JX_FM_EXP_N - 1)];
    return ldexp(fma(s, p, s), k >> 6);
}

__device__ __forceinline__ double jx_fm_log(const JxFm& t, double x) {
    const int hi = __double2hiint(x);
    if ((unsigned)(hi - 0x00100000) >= (unsigned)(0x7FF00000 - 0x00100000)) return log(x);    // zero, subnormal, negative, inf, NaN: the library's
    // x = 2^k z, z in [OFF, 2 OFF); interval i of 128; log x = k ln 2 + log c_i + log1p(z / c_i - 1)
    const int tmp = hi - JX_FM_LOG_OFF_HI;
    const int i = (tmp >> 13) & (JX_FM_LOG_N - 1);
    const int k = tmp >> 20;
    const double z = __hiloint2double(hi - (k << 20), __double2loint(x));
    const double invc = t.lt[2 * i], logc = t.lt[2 * i + 1];
    const double r = fma(z, invc, -1.0);                          // |r| < 2^-7.9
    const double kd = (double)k;
    const double w = fma(kd, 0.69314718055989033, logc);          // ln 2, high part (last 11 bits zero) + log c
    double q = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    q = fma(q, r, 1.0 / 5.0);
    q = fma(q, r, -1.0 / 4.0);
    q = fma(q, r, 1.0 / 3.0);
    q = fma(q, r, -0.5);
    const double r2 = r * r;
    const double hi2 = w + r;
    const double lo = (w - hi2) + r + kd * 5.49792301870837116e-14;   // ln 2, low part
    return fma(q, r2, lo) + hi2;
}
