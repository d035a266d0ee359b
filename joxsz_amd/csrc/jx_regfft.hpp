// Compile-time-sized complex FFTs held entirely in registers (fully unrolled, constant twiddles),
// the building block of the two-level LDS FFTs in jx_conv.hpp.  Plain C++17: the same header is
// compiled for the device by hipcc and for the host by g++ (tests/test_host_tables.py checks every
// supported length against numpy).
#pragma once

#if defined(__HIPCC__)
#define JX_HD __host__ __device__ __forceinline__
#else
#define JX_HD inline
#endif

// complex number over float or double (the fp32 variant of the row transforms uses the same code)
template <typename T> struct jx_cT { T x, y; };
typedef jx_cT<double> jx_c;

template <typename T> JX_HD jx_cT<T> jxcT(T a, T b) { jx_cT<T> r; r.x = a; r.y = b; return r; }
JX_HD jx_c jxc(double a, double b) { return jxcT<double>(a, b); }
template <typename T> JX_HD jx_cT<T> jxc_add(jx_cT<T> a, jx_cT<T> b) { return jxcT<T>(a.x + b.x, a.y + b.y); }
template <typename T> JX_HD jx_cT<T> jxc_sub(jx_cT<T> a, jx_cT<T> b) { return jxcT<T>(a.x - b.x, a.y - b.y); }
template <typename T> JX_HD jx_cT<T> jxc_mul(jx_cT<T> a, jx_cT<T> b) { return jxcT<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// ---- constexpr cos/sin of 2 pi k / n (octant reduction on the integers, Taylor series on [0, pi/4]) ----
constexpr double jx_cx_pi = 3.14159265358979323846264338327950288;

constexpr double jx_cx_sin_small(double x) {
    double x2 = x * x, term = x, sum = x;
    for (int i = 1; i < 14; ++i) { term *= -x2 / ((2.0 * i) * (2.0 * i + 1.0)); sum += term; }
    return sum;
}
constexpr double jx_cx_cos_small(double x) {
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int i = 1; i < 14; ++i) { term *= -x2 / ((2.0 * i - 1.0) * (2.0 * i)); sum += term; }
    return sum;
}
// cos(2 pi k / n) and sin(2 pi k / n), exact symmetries, any integer k, n > 0
constexpr double jx_cx_cos2pi(long long k, long long n) {
    k %= n; if (k < 0) k += n;
    if (2 * k > n) k = n - k;                       // cos(2 pi - t) = cos t          -> t in [0, pi]
    long long a = k, b = n;                         // angle = 2 pi a / b
    double sign = 1.0;
    if (4 * a > b) { a = b - 2 * a; b = 2 * b; sign = -1.0; }   // cos t = -cos(pi - t) -> [0, pi/2]
    if (8 * a <= b) return sign * jx_cx_cos_small(2.0 * jx_cx_pi * (double)a / (double)b);
    return sign * jx_cx_sin_small(2.0 * jx_cx_pi * (double)(b - 4 * a) / (double)(4 * b));       // cos t = sin(pi/2 - t)
}
constexpr double jx_cx_sin2pi(long long k, long long n) {
    // sin(2 pi k/n) = cos(2 pi k/n - pi/2) = cos(2 pi (4k - n) / (4n))
    return jx_cx_cos2pi(4 * k - n, 4 * n);
}

constexpr int jx_rf_radix(int n) { return (n % 4 == 0) ? 4 : ((n % 2 == 0) ? 2 : 3); }

template <int R, bool INV> struct jx_bfly;
template <bool INV> struct jx_bfly<2, INV> {
    template <typename T> static JX_HD void run(jx_cT<T>* u) { const jx_cT<T> a = u[0], b = u[1]; u[0] = jxc_add(a, b); u[1] = jxc_sub(a, b); }
};
template <bool INV> struct jx_bfly<4, INV> {
    template <typename T> static JX_HD void run(jx_cT<T>* u) {
        const jx_cT<T> s0 = jxc_add(u[0], u[2]), d0 = jxc_sub(u[0], u[2]);
        const jx_cT<T> s1 = jxc_add(u[1], u[3]), e = jxc_sub(u[1], u[3]);
        const jx_cT<T> d1 = INV ? jxcT<T>(-e.y, e.x) : jxcT<T>(e.y, -e.x);              // -+ i (u1 - u3)
        u[0] = jxc_add(s0, s1); u[2] = jxc_sub(s0, s1);
        u[1] = jxc_add(d0, d1); u[3] = jxc_sub(d0, d1);
    }
};
template <bool INV> struct jx_bfly<3, INV> {
    template <typename T> static JX_HD void run(jx_cT<T>* u) {
        const T s = (T)(INV ? 0.86602540378443864676 : -0.86602540378443864676);
        const jx_cT<T> t = jxc_add(u[1], u[2]), d = jxc_sub(u[1], u[2]);
        const jx_cT<T> m = jxcT<T>(u[0].x - (T)0.5 * t.x, u[0].y - (T)0.5 * t.y);
        const jx_cT<T> q = jxcT<T>(-s * d.y, s * d.x);
        u[0] = jxc_add(u[0], t);
        u[1] = jxc_add(m, q);
        u[2] = jxc_sub(m, q);
    }
};

// multiply by W_n^{k} (forward: e^{-2 pi i k/n}; inverse: conjugate), constants folded at compile time
template <int K, int N, bool INV, typename T>
JX_HD jx_cT<T> jx_twmul(jx_cT<T> a) {
    constexpr int k = ((K % N) + N) % N;
    if constexpr (k == 0) return a;
    else if constexpr (4 * k == N) return INV ? jxcT<T>(-a.y, a.x) : jxcT<T>(a.y, -a.x);
    else if constexpr (2 * k == N) return jxcT<T>(-a.x, -a.y);
    else if constexpr (4 * k == 3 * N) return INV ? jxcT<T>(a.y, -a.x) : jxcT<T>(-a.y, a.x);
    else {
        constexpr T c = (T)jx_cx_cos2pi(k, N);
        constexpr T s = (T)(INV ? jx_cx_sin2pi(k, N) : -jx_cx_sin2pi(k, N));
        return jxcT<T>(a.x * c - a.y * s, a.x * s + a.y * c);
    }
}

// In-place, natural-order FFT of x[0..N-1] (N = 2^a 3^b), decimation in time, fully unrolled.
template <int N, bool INV> struct jx_regfft {
    template <int R, int M, int KK, int RR, typename T>
    static JX_HD void tw_row(jx_cT<T> (&sub)[R][M], jx_cT<T>* u) {
        if constexpr (RR < R) {
            u[RR] = jx_twmul<RR * KK, N, INV, T>(sub[RR][KK]);
            tw_row<R, M, KK, RR + 1, T>(sub, u);
        }
    }
    template <int R, int M, int KK, typename T>
    static JX_HD void combine(jx_cT<T> (&sub)[R][M], jx_cT<T>* x) {
        if constexpr (KK < M) {
            jx_cT<T> u[R];
            tw_row<R, M, KK, 0, T>(sub, u);
            jx_bfly<R, INV>::run(u);
#pragma unroll
            for (int q = 0; q < R; ++q) x[KK + M * q] = u[q];
            combine<R, M, KK + 1, T>(sub, x);
        }
    }
    template <typename T>
    static JX_HD void run(jx_cT<T>* x) {
        constexpr int R = jx_rf_radix(N);
        constexpr int M = N / R;
        if constexpr (M == 1) {
            jx_bfly<R, INV>::run(x);
        } else {
            jx_cT<T> sub[R][M];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int m = 0; m < M; ++m) sub[r][m] = x[r + R * m];
#pragma unroll
            for (int r = 0; r < R; ++r) jx_regfft<M, INV>::run(sub[r]);
            combine<R, M, 0, T>(sub, x);
        }
    }
};
template <bool INV> struct jx_regfft<1, INV> { template <typename T> static JX_HD void run(jx_cT<T>*) {} };

// factorisation L = L1 * L2 used by the two-level LDS FFT (both factors small enough for registers)
template <int L> struct jx_plan2;
#define JX_PLAN2(L, A, B) template <> struct jx_plan2<L> { static constexpr int L1 = A, L2 = B; };
JX_PLAN2(16, 4, 4) JX_PLAN2(18, 3, 6) JX_PLAN2(24, 4, 6) JX_PLAN2(32, 4, 8) JX_PLAN2(48, 6, 8) JX_PLAN2(64, 8, 8)
JX_PLAN2(72, 8, 9) JX_PLAN2(96, 8, 12) JX_PLAN2(128, 8, 16) JX_PLAN2(144, 12, 12) JX_PLAN2(256, 16, 16) JX_PLAN2(288, 16, 18)
JX_PLAN2(512, 16, 32) JX_PLAN2(576, 24, 24)
#undef JX_PLAN2
