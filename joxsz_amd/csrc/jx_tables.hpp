// Walker-independent tables, built once per context on the host (setup, not hot path).
//
//  * abel_matrix          : the forward Abel transform the reference obtains from
//                           PyAbel's direct_transform(..., backend='Python')
//                           (joxsz_funcs.py:457) as explicit upper-triangular weights.
//  * mirrored_spline_op   : second-derivative ("moment") operator of the not-a-knot
//                           cubic spline through the mirrored samples (-r, y), (r, y)
//                           that interp1d(..., 'cubic') builds at joxsz_funcs.py:460, 470.
//  * nak_eval_matrix      : evaluation matrix of the plain not-a-knot cubic spline with
//                           extrapolation (joxsz_funcs.py:476) at fixed abscissae.
//  * beam_spectrum        : half-spectrum of the zero-padded, centre-shifted beam image
//                           for the FFT convolution of joxsz_funcs.py:464 (rocFFT sequence).
//  * tf_row_table         : transfer-function weights that collapse the inverse FFT of
//                           joxsz_funcs.py:467 to the single row joxsz_funcs.py:472 reads (rocFFT sequence).
//  * tf_hy_table, lowrank_factor(_qr), beam_separable_terms, mix_* : operators of the contracted route (jx_mix.hpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

// [0, n) in contiguous pieces on up to 16 host threads (table builds only; each piece writes its own rows)
template <typename F>
static inline void jx_parallel_for(int n, F&& body) {
    int nt = (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(std::min(nt, 16), n));
    if (nt == 1) { body(0, n); return; }
    std::vector<std::thread> th;
    const int per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int a = t * per, b = std::min(n, a + per);
        if (a < b) th.emplace_back([&body, a, b]() { body(a, b); });
    }
    for (auto& t : th) t.join();
}

namespace jxt {

static const double kPi = 3.14159265358979323846264338327950288;

// ---------------------------------------------------------------------------------------
// Banded Gaussian elimination with partial pivoting on dense row-major storage.
// A is n x n with lower bandwidth kl and upper bandwidth ku; Bm is n x m (row-major).
// On return Bm holds A^-1 Bm.  Returns false on a zero pivot.
// ---------------------------------------------------------------------------------------
inline bool solve_banded(std::vector<double>& A, int n, int kl, int ku, std::vector<double>& Bm, int m) {
    const int kue = ku + kl;                       // fill-in from row swaps
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = std::fabs(A[(size_t)k * n + k]);
        const int rmax = std::min(n - 1, k + kl);
        for (int r = k + 1; r <= rmax; ++r) {
            double v = std::fabs(A[(size_t)r * n + k]);
            if (v > best) { best = v; piv = r; }
        }
        if (best == 0.0) return false;
        const int cmax = std::min(n - 1, k + kue);
        if (piv != k) {
            for (int c = k; c <= cmax; ++c) std::swap(A[(size_t)k * n + c], A[(size_t)piv * n + c]);
            for (int c = 0; c < m; ++c) std::swap(Bm[(size_t)k * m + c], Bm[(size_t)piv * m + c]);
        }
        const double inv = 1.0 / A[(size_t)k * n + k];
        for (int r = k + 1; r <= rmax; ++r) {
            const double f = A[(size_t)r * n + k] * inv;
            if (f == 0.0) continue;
            A[(size_t)r * n + k] = 0.0;
            for (int c = k + 1; c <= cmax; ++c) A[(size_t)r * n + c] -= f * A[(size_t)k * n + c];
            double* br = &Bm[(size_t)r * m];
            const double* bk = &Bm[(size_t)k * m];
            for (int c = 0; c < m; ++c) br[c] -= f * bk[c];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        const int cmax = std::min(n - 1, k + kue);
        double* bk = &Bm[(size_t)k * m];
        for (int c = k + 1; c <= cmax; ++c) {
            const double a = A[(size_t)k * n + c];
            if (a == 0.0) continue;
            const double* bc = &Bm[(size_t)c * m];
            for (int j = 0; j < m; ++j) bk[j] -= a * bc[j];
        }
        const double inv = 1.0 / A[(size_t)k * n + k];
        for (int j = 0; j < m; ++j) bk[j] *= inv;
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// Forward Abel transform weights.  ab = A pp, A[i][j] = 0 for j < i.
// Follows the published algorithm of PyAbel's _pyabel_direct_integral with
// correction=1 and int_func=np.trapz (restated in oracle/pyabel_direct.py):
// trapezoid rule over F_j/sqrt(r_j^2-r_i^2), j>i, with F_j = 2 r_j pp_j; minus half the
// trapezoid integral of the first cell's lone sample; plus the analytic integral of the
// singular cell for F linear on [r_i, r_{i+1}].
// ---------------------------------------------------------------------------------------
inline bool grid_is_uniform(const std::vector<double>& r) {
    // PyAbel's is_uniform_sampling: all second differences within 1e-13 of zero
    for (size_t i = 0; i + 2 < r.size(); ++i) {
        const double dd = (r[i + 2] - r[i + 1]) - (r[i + 1] - r[i]);
        if (!(std::fabs(dd) <= 1e-13)) return false;
    }
    return true;
}

inline void abel_matrix(const std::vector<double>& r, std::vector<double>& A) {
    const int n = (int)r.size();
    A.assign((size_t)n * n, 0.0);
    const bool uni = grid_is_uniform(r);
    const double dx = std::fabs(r[1] - r[0]);
    // trapezoid weight of sample j over the whole grid
    std::vector<double> wt(n);
    for (int j = 0; j < n; ++j) {
        if (uni) wt[j] = (j == 0 || j == n - 1) ? 0.5 * dx : dx;
        else {
            const double dl = (j > 0) ? r[j] - r[j - 1] : 0.0;
            const double dr = (j < n - 1) ? r[j + 1] - r[j] : 0.0;
            wt[j] = 0.5 * (dl + dr);
        }
    }
    for (int i = 0; i < n - 1; ++i) {
        double* row = &A[(size_t)i * n];
        for (int j = i + 1; j < n; ++j) {
            const double isq = 1.0 / std::sqrt(r[j] * r[j] - r[i] * r[i]);
            row[j] = wt[j] * isq;                           // coefficient of F_j
        }
        row[i + 1] *= 0.5;                                   // "extra triangle" correction
        const double d = r[i + 1] - r[i];
        const double s = std::sqrt(r[i + 1] * r[i + 1] - r[i] * r[i]);
        double acr;
        if (i == 0 && r[0] < r[1] * 1e-8) acr = std::acosh(std::cosh(1.0));
        else acr = std::acosh(r[i + 1] / r[i]);
        // s*F' + acr*(F_i - F' r_i),  F' = (F_{i+1}-F_i)/d
        row[i]     += -s / d + acr * (1.0 + r[i] / d);
        row[i + 1] +=  s / d - acr * r[i] / d;
        for (int j = i; j < n; ++j) row[j] *= 2.0 * r[j];    // F_j = 2 r_j pp_j
    }
}

// The same weights in the form the kernel regenerates on the fly:
//   A[i][j] = cj[j] / sqrt(r_j^2 - r_i^2)  for j >= i+2,   A[i][i] = dg[i],   A[i][i+1] = sp[i].
inline void abel_onfly_tables(const std::vector<double>& r, std::vector<double>& cj, std::vector<double>& dg,
                              std::vector<double>& sp) {
    const int n = (int)r.size();
    std::vector<double> A;
    abel_matrix(r, A);
    cj.assign(n, 0.0); dg.assign(n, 0.0); sp.assign(n, 0.0);
    const bool uni = grid_is_uniform(r);
    const double dx = std::fabs(r[1] - r[0]);
    for (int j = 0; j < n; ++j) {
        double wt;
        if (uni) wt = (j == 0 || j == n - 1) ? 0.5 * dx : dx;
        else wt = 0.5 * (((j > 0) ? r[j] - r[j - 1] : 0.0) + ((j < n - 1) ? r[j + 1] - r[j] : 0.0));
        cj[j] = 2.0 * r[j] * wt;
    }
    for (int i = 0; i < n; ++i) {
        dg[i] = A[(size_t)i * n + i];
        if (i + 1 < n) sp[i] = A[(size_t)i * n + i + 1];
    }
}

// ---------------------------------------------------------------------------------------
// Moment operator of the mirrored not-a-knot spline: M = G y, with y the samples at the
// positive knots r_0 < ... < r_{n-1}, the knot set being {-r_{n-1},..,-r_0, r_0,..,r_{n-1}}
// (no knot at 0: the centre interval is [-r_0, r_0]).  By symmetry the spline is even, so
// M(-r_0) = M(r_0) and the centre-interval polynomial is y_0 + M_0 (x^2 - r_0^2)/2.
// Not-a-knot at the outer end: third derivative continuous at r_{n-2}.
// G is returned dense (n x n, row-major).
// ---------------------------------------------------------------------------------------
inline bool mirrored_spline_op(const std::vector<double>& r, std::vector<double>& G) {
    const int n = (int)r.size();
    if (n < 3) return false;
    std::vector<double> A((size_t)n * n, 0.0);
    G.assign((size_t)n * n, 0.0);                            // starts as D (rhs operator)
    const double L = 2.0 * r[0];
    {
        const double h0 = r[1] - r[0];
        A[0] = L / 2.0 + h0 / 3.0;
        A[1] = h0 / 6.0;
        G[0] = -1.0 / h0;
        G[1] = 1.0 / h0;
    }
    for (int i = 1; i < n - 1; ++i) {
        const double hl = r[i] - r[i - 1], hr = r[i + 1] - r[i];
        A[(size_t)i * n + i - 1] = hl / 6.0;
        A[(size_t)i * n + i] = (hl + hr) / 3.0;
        A[(size_t)i * n + i + 1] = hr / 6.0;
        G[(size_t)i * n + i - 1] = 1.0 / hl;
        G[(size_t)i * n + i] = -1.0 / hl - 1.0 / hr;
        G[(size_t)i * n + i + 1] = 1.0 / hr;
    }
    {
        const double ha = r[n - 2] - r[n - 3], hb = r[n - 1] - r[n - 2];
        A[(size_t)(n - 1) * n + n - 3] = hb;
        A[(size_t)(n - 1) * n + n - 2] = -(ha + hb);
        A[(size_t)(n - 1) * n + n - 1] = ha;
    }
    return solve_banded(A, n, 2, 1, G, n);
}

// Plain not-a-knot cubic spline (both ends), moment operator M = G y on knots x.
inline bool nak_spline_op(const std::vector<double>& x, std::vector<double>& G) {
    const int n = (int)x.size();
    if (n < 4) return false;
    std::vector<double> A((size_t)n * n, 0.0);
    G.assign((size_t)n * n, 0.0);
    {
        const double h0 = x[1] - x[0], h1 = x[2] - x[1];
        A[0] = h1; A[1] = -(h0 + h1); A[2] = h0;
    }
    for (int i = 1; i < n - 1; ++i) {
        const double hl = x[i] - x[i - 1], hr = x[i + 1] - x[i];
        A[(size_t)i * n + i - 1] = hl / 6.0;
        A[(size_t)i * n + i] = (hl + hr) / 3.0;
        A[(size_t)i * n + i + 1] = hr / 6.0;
        G[(size_t)i * n + i - 1] = 1.0 / hl;
        G[(size_t)i * n + i] = -1.0 / hl - 1.0 / hr;
        G[(size_t)i * n + i + 1] = 1.0 / hr;
    }
    {
        const double ha = x[n - 2] - x[n - 3], hb = x[n - 1] - x[n - 2];
        A[(size_t)(n - 1) * n + n - 3] = hb;
        A[(size_t)(n - 1) * n + n - 2] = -(ha + hb);
        A[(size_t)(n - 1) * n + n - 1] = ha;
    }
    return solve_banded(A, n, 2, 2, G, n);
}

// E[d][k]: value at q[d] of the plain not-a-knot spline through (x_k, y_k) is sum_k E[d][k] y_k.
// Outside [x_0, x_{n-1}] the end polynomial is continued (interp1d fill_value='extrapolate').
// A NaN abscissa gives a NaN row.
inline bool nak_eval_matrix(const std::vector<double>& x, const std::vector<double>& q, std::vector<double>& E) {
    const int n = (int)x.size(), nq = (int)q.size();
    std::vector<double> G;
    if (!nak_spline_op(x, G)) return false;
    E.assign((size_t)nq * n, 0.0);
    for (int d = 0; d < nq; ++d) {
        double* e = &E[(size_t)d * n];
        if (std::isnan(q[d])) { for (int k = 0; k < n; ++k) e[k] = NAN; continue; }
        int k = (int)(std::upper_bound(x.begin(), x.end(), q[d]) - x.begin()) - 1;
        k = std::max(0, std::min(n - 2, k));
        const double h = x[k + 1] - x[k], t = q[d] - x[k];
        const double wa = 1.0 - t / h, wb = t / h;
        const double wm0 = -t * h / 3.0 + t * t / 2.0 - t * t * t / (6.0 * h);
        const double wm1 = -t * h / 6.0 + t * t * t / (6.0 * h);
        const double* g0 = &G[(size_t)k * n];
        const double* g1 = &G[(size_t)(k + 1) * n];
        for (int j = 0; j < n; ++j) e[j] = wm0 * g0[j] + wm1 * g1[j];
        e[k] += wa;
        e[k + 1] += wb;
    }
    return true;
}

// Half-bandwidth beyond which every |G[i][j]| is below eps * max|G|.
inline int band_halfwidth(const std::vector<double>& G, int n, double eps) {
    double mx = 0.0;
    for (double v : G) mx = std::max(mx, std::fabs(v));
    const double thr = eps * mx;
    int K = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (std::fabs(G[(size_t)i * n + j]) > thr) K = std::max(K, std::abs(i - j));
    return K;
}

// ---------------------------------------------------------------------------------------
// Beam half-spectrum for the 'same' FFT convolution of joxsz_funcs.py:464.
// The B x B beam is placed in a P x P zero image shifted by -(B-1)/2 in both axes
// (circularly), so that the circular convolution restricted to [0,S)^2 equals
// fftconvolve(y_2d, beam, 'same') whenever P >= S + (B-1)/2.  Output: [P][P/2+1] complex
// (re, im interleaved), multiplied by `scale` (= step^2 / P^2 for the unnormalised FFT pair).
// Only B rows/columns are non-zero, so the DFT is done directly in O(P^2 B).
// ---------------------------------------------------------------------------------------
inline void beam_spectrum(const std::vector<double>& beam, int B, int P, double scale, std::vector<double>& out) {
    const int o = (B - 1) / 2, Ph = P / 2 + 1;
    std::vector<double> cs(P), sn(P);
    for (int m = 0; m < P; ++m) { cs[m] = std::cos(2.0 * kPi * m / P); sn[m] = std::sin(2.0 * kPi * m / P); }
    // stage 1: along x for each beam row u: T[u][kx] = sum_v beam[u][v] e^{-2 pi i kx (v-o)/P}
    std::vector<double> Tr((size_t)B * Ph), Ti((size_t)B * Ph);
    for (int u = 0; u < B; ++u)
        for (int kx = 0; kx < Ph; ++kx) {
            double ar = 0.0, ai = 0.0;
            for (int v = 0; v < B; ++v) {
                long long ph = ((long long)kx * (v - o)) % P; if (ph < 0) ph += P;
                ar += beam[(size_t)u * B + v] * cs[ph];
                ai -= beam[(size_t)u * B + v] * sn[ph];
            }
            Tr[(size_t)u * Ph + kx] = ar; Ti[(size_t)u * Ph + kx] = ai;
        }
    out.assign((size_t)P * Ph * 2, 0.0);
    for (int ky = 0; ky < P; ++ky)
        for (int kx = 0; kx < Ph; ++kx) {
            double ar = 0.0, ai = 0.0;
            for (int u = 0; u < B; ++u) {
                long long ph = ((long long)ky * (u - o)) % P; if (ph < 0) ph += P;
                const double c = cs[ph], s = -sn[ph];
                const double tr = Tr[(size_t)u * Ph + kx], ti = Ti[(size_t)u * Ph + kx];
                ar += tr * c - ti * s;
                ai += tr * s + ti * c;
            }
            out[((size_t)ky * Ph + kx) * 2] = ar * scale;
            out[((size_t)ky * Ph + kx) * 2 + 1] = ai * scale;
        }
}

// ---------------------------------------------------------------------------------------
// Transfer-function row table.  The reference computes real(ifft2(fft2(conv) * filtering))
// (joxsz_funcs.py:466-467) and keeps only row r0 = S//2 from column S//2 on
// (joxsz_funcs.py:472).  With X = rfft2(conv) (unnormalised, [S][S/2+1]):
//     Z[kc]  = sum_kr X[kr][kc] * H[kr][kc]
//     row[c] = sum_kc Re( Z[kc] e^{+2 pi i kc c / S} )
// where H[kr][kc] = Fs[kr][kc] * e^{+2 pi i kr r0 / S} * w_kc / S^2, Fs is the
// Hermitian-symmetrised filter (Fs[k] = (F[k]+F[-k])/2: taking the real part of the
// inverse transform of a real-input spectrum times a real filter is exactly that), and
// w_kc = 1 for kc = 0 and for kc = S/2 when S is even, else 2.
// ---------------------------------------------------------------------------------------
inline void tf_row_table(const std::vector<double>& filt, int S, std::vector<double>& H) {
    const int Sh = S / 2 + 1, r0 = S / 2;
    H.assign((size_t)S * Sh * 2, 0.0);
    const double inv = 1.0 / ((double)S * (double)S);
    for (int kr = 0; kr < S; ++kr) {
        const long long ph = ((long long)kr * r0) % S;
        const double c = std::cos(2.0 * kPi * ph / S), s = std::sin(2.0 * kPi * ph / S);
        const int mr = (S - kr) % S;
        for (int kc = 0; kc < Sh; ++kc) {
            const int mc = (S - kc) % S;
            const double fs = 0.5 * (filt[(size_t)kr * S + kc] + filt[(size_t)mr * S + mc]);
            const double w = (kc == 0 || (S % 2 == 0 && kc == S / 2)) ? 1.0 : 2.0;
            H[((size_t)kr * Sh + kc) * 2] = fs * c * w * inv;
            H[((size_t)kr * Sh + kc) * 2 + 1] = fs * s * w * inv;
        }
    }
}


// ---------------------------------------------------------------------------------------
// Radices of a transform length for the LDS transforms of the literal route (jx_fft.hpp): the split into the fewest
// passes, among those the one with the least work per lane (butterflies of a sequence per lane x size of the butterfly,
// 64 lanes per sequence); largest radix first (the first pass has no twiddles).  Radices 10 9 8 6 5 4 3 2, and 16 12 for
// lengths beyond 640 (the kernels that may hold 256 registers).  Returns the number of passes; 0: the length has a prime
// factor beyond 5, or needs more than maxpass passes.
// ---------------------------------------------------------------------------------------
inline int fft_radices(int n, int* radix, int maxpass) {
    static const int kR[] = {16, 12, 10, 9, 8, 6, 5, 4, 3, 2};
    const int nR = (int)(sizeof(kR) / sizeof(kR[0])), tpc = 64;
    const int first = n > 640 ? 0 : 2;
    if (n < 2) return 0;
    std::vector<int> best, cur;
    double best_cost = 0.0;
    struct Rec {
        static void go(int n, int rem, int at, double cost, int first, int nR, int tpc, int maxpass, std::vector<int>& cur, std::vector<int>& best, double& best_cost) {
            if (rem == 1) {
                if (best.empty() || cur.size() < best.size() || (cur.size() == best.size() && cost < best_cost)) { best = cur; best_cost = cost; }
                return;
            }
            if ((int)cur.size() >= maxpass || (!best.empty() && cur.size() >= best.size())) return;
            for (int i = std::max(at, first); i < nR; ++i) {
                const int r = kR[i];
                if (rem % r) continue;
                cur.push_back(r);
                go(n, rem / r, i, cost + (double)((n / r + tpc - 1) / tpc) * r * (std::log2((double)r) + 1.0), first, nR, tpc, maxpass, cur, best, best_cost);
                cur.pop_back();
            }
        }
    };
    Rec::go(n, n, 0, 0.0, first, nR, tpc, maxpass, cur, best, best_cost);
    for (size_t i = 0; i < best.size(); ++i) radix[i] = best[i];
    return (int)best.size();
}

// ---------------------------------------------------------------------------------------
// Small host FFT (recursive Cooley-Tukey on the smallest prime factor, naive DFT for
// primes); sign = -1 forward, +1 inverse, unnormalised.  Used only to build tables.
// ---------------------------------------------------------------------------------------
inline void host_fft(std::vector<double>& re, std::vector<double>& im, int sign) {
    const int n = (int)re.size();
    if (n <= 1) return;
    int p = 0;
    for (int f = 2; f * f <= n; ++f) if (n % f == 0) { p = f; break; }
    if (p == 0) {                                            // prime length: direct
        std::vector<double> orr(n), oi(n);
        for (int k = 0; k < n; ++k) {
            double ar = 0, ai = 0;
            for (int j = 0; j < n; ++j) {
                const long long ph = ((long long)j * k) % n;
                const double c = std::cos(2.0 * kPi * ph / n), s2 = sign * std::sin(2.0 * kPi * ph / n);
                ar += re[j] * c - im[j] * s2; ai += re[j] * s2 + im[j] * c;
            }
            orr[k] = ar; oi[k] = ai;
        }
        re.swap(orr); im.swap(oi);
        return;
    }
    const int m = n / p;
    std::vector<std::vector<double>> sr(p, std::vector<double>(m)), si(p, std::vector<double>(m));
    for (int q = 0; q < p; ++q)
        for (int j = 0; j < m; ++j) { sr[q][j] = re[j * p + q]; si[q][j] = im[j * p + q]; }
    for (int q = 0; q < p; ++q) host_fft(sr[q], si[q], sign);
    for (int k = 0; k < n; ++k) {
        double ar = 0, ai = 0;
        for (int q = 0; q < p; ++q) {
            const long long ph = ((long long)q * k) % n;
            const double c = std::cos(2.0 * kPi * ph / n), s2 = sign * std::sin(2.0 * kPi * ph / n);
            const double xr = sr[q][k % m], xi = si[q][k % m];
            ar += xr * c - xi * s2; ai += xr * s2 + xi * c;
        }
        re[k] = ar; im[k] = ai;
    }
}

// Hy[r][kc] = w_kc / S^2 * sum_kr Fs[kr][kc] e^{2 pi i kr (c - r)/S}: the symmetrised transfer
// function taken back to real space along y, at the row offset of the extracted row c = S//2.
inline void tf_hy_table(const std::vector<double>& filt, int S, std::vector<double>& Hy) {
    const int Sh = S / 2 + 1, c = S / 2;
    Hy.assign((size_t)S * Sh * 2, 0.0);
    const double inv = 1.0 / ((double)S * (double)S);
    std::vector<double> re(S), im(S);
    for (int kc = 0; kc < Sh; ++kc) {
        const int mc = (S - kc) % S;
        for (int kr = 0; kr < S; ++kr) {
            const int mr = (S - kr) % S;
            re[kr] = 0.5 * (filt[(size_t)kr * S + kc] + filt[(size_t)mr * S + mc]);
            im[kr] = 0.0;
        }
        host_fft(re, im, +1);                                 // g[n] = sum_kr Fs e^{+2 pi i kr n/S}
        const double w = (kc == 0 || (S % 2 == 0 && kc == S / 2)) ? 1.0 : 2.0;
        for (int r = 0; r < S; ++r) {
            const int n = ((c - r) % S + S) % S;
            Hy[((size_t)r * Sh + kc) * 2] = re[n] * w * inv;
            Hy[((size_t)r * Sh + kc) * 2 + 1] = im[n] * w * inv;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Truncated singular value decomposition  A[m][n] ~ sum_{rho < r} L[rho][i] Rt[rho][j]  by one-sided
// (Hestenes) Jacobi rotations on the columns of A: afterwards the columns are orthogonal, their norms
// are the singular values and the accumulated rotations are the right singular vectors.  L carries
// the singular value (L = sigma u), Rt has unit rows.  Terms are sorted by singular value; r counts
// those above tol * sigma_max.  The transfer-function weights Hy[q][kx] of a smooth transfer function have a numerical
// rank of a few dozen, which lets stage 1 of the contracted route keep r numbers per map column instead of one per row.
// Returns r; sigma_out (optional) receives all n singular values, sorted.
// ---------------------------------------------------------------------------------------
inline int lowrank_factor(const double* A, int m, int n, double tol, std::vector<double>& L, std::vector<double>& Rt,
                          std::vector<double>* sigma_out = nullptr) {
    std::vector<double> G((size_t)n * m), V((size_t)n * n, 0.0);            // column-major: G[j*m + i], V[j*n + k]
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) G[(size_t)j * m + i] = A[(size_t)i * n + j];
    for (int j = 0; j < n; ++j) V[(size_t)j * n + j] = 1.0;
    std::vector<double> nrm(n);
    double big = 0.0;
    for (int j = 0; j < n; ++j) {
        double a = 0.0;
        for (int i = 0; i < m; ++i) a += G[(size_t)j * m + i] * G[(size_t)j * m + i];
        nrm[j] = a; big = std::max(big, a);
    }
    const double negl = big * 1e-36;                                          // columns this small never matter at double precision
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotations = 0;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                const double alpha = nrm[p], beta = nrm[q];
                if (alpha <= negl && beta <= negl) continue;
                double* gp = &G[(size_t)p * m];
                double* gq = &G[(size_t)q * m];
                double gamma = 0.0;
                for (int i = 0; i < m; ++i) gamma += gp[i] * gq[i];
                if (std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta) || gamma == 0.0) continue;
                ++rotations;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                double a2 = 0.0, b2 = 0.0;
                for (int i = 0; i < m; ++i) {
                    const double x = gp[i], y = gq[i];
                    const double xn = cs * x - sn * y, yn = sn * x + cs * y;
                    gp[i] = xn; gq[i] = yn; a2 += xn * xn; b2 += yn * yn;
                }
                nrm[p] = a2; nrm[q] = b2;
                double* vp = &V[(size_t)p * n];
                double* vq = &V[(size_t)q * n];
                for (int k = 0; k < n; ++k) {
                    const double x = vp[k], y = vq[k];
                    vp[k] = cs * x - sn * y; vq[k] = sn * x + cs * y;
                }
            }
        }
        if (!rotations) break;
    }
    std::vector<int> order(n);
    for (int j = 0; j < n; ++j) order[j] = j;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return nrm[a] > nrm[b]; });
    const double smax = std::sqrt(nrm[order[0]]);
    int r = 0;
    while (r < n && std::sqrt(nrm[order[r]]) > tol * smax) ++r;
    L.assign((size_t)r * m, 0.0);
    Rt.assign((size_t)r * n, 0.0);
    for (int rho = 0; rho < r; ++rho) {
        const int j = order[rho];
        for (int i = 0; i < m; ++i) L[(size_t)rho * m + i] = G[(size_t)j * m + i];
        for (int k = 0; k < n; ++k) Rt[(size_t)rho * n + k] = V[(size_t)j * n + k];
    }
    if (sigma_out) {
        sigma_out->resize(n);
        for (int j = 0; j < n; ++j) (*sigma_out)[j] = std::sqrt(nrm[order[j]]);
    }
    return r;
}

// ---------------------------------------------------------------------------------------
// One sample of the mirrored cubic spline (interp1d(..., 'cubic') of joxsz_funcs.py:460 evaluated at radius x, funcs:462):
//     f(x) = A y_k + B y_{k+1} + C M_k + D M_{k+1}
// on [r_k, r_{k+1}]: A = 1 - t/h, B = t/h, C = (A^3 - A) h^2/6, D = (B^3 - B) h^2/6; centre interval |x| < r_0 of the mirrored
// knots: y_0 + M_0 (x^2 - r_0^2)/2; fill value 0 outside r_{N-1}; a NaN radius gives NaN.  k16 = 16 k.
// ---------------------------------------------------------------------------------------
inline void spline_sample_weights(const std::vector<double>& r, double x, int* k16, double* w4) {
    const int N = (int)r.size();
    *k16 = 0; w4[0] = w4[1] = w4[2] = w4[3] = 0.0;
    if (x != x) { w4[0] = w4[1] = w4[2] = w4[3] = x; return; }
    if (!(x <= r[N - 1])) return;                                     // fill value
    if (x < r[0]) { w4[0] = 1.0; w4[2] = 0.5 * (x * x - r[0] * r[0]); return; }
    int k = (int)(std::upper_bound(r.begin(), r.end(), x) - r.begin()) - 1;
    k = std::max(0, std::min(N - 2, k));
    const double h = r[k + 1] - r[k], B = (x - r[k]) / h, A = 1.0 - B;
    *k16 = 16 * k;
    w4[0] = A; w4[1] = B; w4[2] = (A * A * A - A) * h * h / 6.0; w4[3] = (B * B * B - B) * h * h / 6.0;
}

// ---------------------------------------------------------------------------------------
// Operator of jx_abel_gemm_kernel: the spline ordinates and moments of a walker are linear in its pressure profile,
//     y_k = sum_j Tm[j][2k] pp_j       Tm[j][2k]   = y_scale A[k][j]                          (joxsz_funcs.py:457-459)
//     M_k = sum_j Tm[j][2k+1] pp_j     Tm[j][2k+1] = sum_{|i-k|<=K} G[k][i] y_scale A[i][j]   (joxsz_funcs.py:460)
// A = abel_matrix(r) (upper triangular), G = mirrored_spline_op(r) restricted to its band of half-width K (what the Abel
// kernel's own moment sums use).  The second product is accumulated in long double.  Row-major [rows][ld], zero padded.
// ---------------------------------------------------------------------------------------
inline void abel_spline_operator(const std::vector<double>& r, const std::vector<double>& G /*[N][N]*/, int K, double y_scale,
                                 int rows, int ld, std::vector<double>& out) {
    const int N = (int)r.size();
    std::vector<double> A;
    abel_matrix(r, A);
    out.assign((size_t)rows * ld, 0.0);
    for (int j = 0; j < N; ++j) {
        double* row = &out[(size_t)j * ld];
        for (int k = 0; k <= j; ++k) row[2 * k] = y_scale * A[(size_t)k * N + j];
        for (int k = 0; k < N; ++k) {
            long double m = 0.0L;
            const int i0 = std::max(0, k - K), i1 = std::min(std::min(N - 1, k + K), j);
            for (int i = i0; i <= i1; ++i) m += (long double)G[(size_t)k * N + i] * (long double)row[2 * i];
            row[2 * k + 1] = (double)m;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Rank-revealing form of lowrank_factor for matrices of low numerical rank (the transfer-function weights: ~60 of 257 at
// 512^2 above rounding): Householder QR with column pivoting, stopped when every remaining column is below 1e-17 of the
// largest, then the one-sided Jacobi SVD of the small triangular factor.  Same output convention as lowrank_factor
// (A ~ sum_rho L[rho][i] Rt[rho][j], L = sigma u, unit rows in Rt, sorted, r = terms above tol * sigma_max); the cost falls
// from m n^2 per Jacobi sweep to m n k + n k^2 (k = numerical rank): 3.3 s -> 0.05 s for the 1024^2 table.
// ---------------------------------------------------------------------------------------
inline int lowrank_factor_qr(const double* A, int m, int n, double tol, std::vector<double>& L, std::vector<double>& Rt,
                             std::vector<double>* sigma_out = nullptr) {
    std::vector<double> W((size_t)n * m);                                      // column-major working copy: W[j*m + i]
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) W[(size_t)j * m + i] = A[(size_t)i * n + j];
    std::vector<int> perm(n);
    std::vector<double> cn(n), beta;
    for (int j = 0; j < n; ++j) {
        perm[j] = j;
        double a = 0.0;
        for (int i = 0; i < m; ++i) a += W[(size_t)j * m + i] * W[(size_t)j * m + i];
        cn[j] = a;
    }
    double big = 0.0;
    for (double v : cn) big = std::max(big, v);
    const int kmax = std::min(m, n);
    int k = 0;
    for (; k < kmax; ++k) {
        int piv = k;
        // exact remaining norms (cheap at these sizes; avoids the downdating cancellation)
        for (int j = k; j < n; ++j) {
            double a = 0.0;
            for (int i = k; i < m; ++i) a += W[(size_t)j * m + i] * W[(size_t)j * m + i];
            cn[j] = a;
            if (a > cn[piv]) piv = j;
        }
        if (!(cn[piv] > big * 1e-34)) break;
        if (piv != k) {
            for (int i = 0; i < m; ++i) std::swap(W[(size_t)k * m + i], W[(size_t)piv * m + i]);
            std::swap(perm[k], perm[piv]); std::swap(cn[k], cn[piv]);
        }
        double* x = &W[(size_t)k * m];
        const double nrm = std::sqrt(cn[k]);
        const double alpha = (x[k] > 0.0) ? -nrm : nrm;
        const double v0 = x[k] - alpha;
        // v = (v0, x[k+1..]); H = I - b v v^T, b = 2 / (v.v)
        double vv = v0 * v0;
        for (int i = k + 1; i < m; ++i) vv += x[i] * x[i];
        const double b = (vv > 0.0) ? 2.0 / vv : 0.0;
        for (int j = k + 1; j < n; ++j) {
            double* y = &W[(size_t)j * m];
            double d = v0 * y[k];
            for (int i = k + 1; i < m; ++i) d += x[i] * y[i];
            d *= b;
            y[k] -= d * v0;
            for (int i = k + 1; i < m; ++i) y[i] -= d * x[i];
        }
        x[k] = alpha;                                                           // R[k][k]; the reflector stays below it, v0 apart
        beta.push_back(b);
        cn[k] = v0;                                                             // (re-used as storage of v0)
    }
    if (k == 0) { L.clear(); Rt.clear(); if (sigma_out) sigma_out->assign(n, 0.0); return 0; }
    // R^T as an n x k matrix (row j = permuted column j of R)
    std::vector<double> RT((size_t)n * k, 0.0);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i <= std::min(j, k - 1); ++i) RT[(size_t)j * k + i] = W[(size_t)j * m + i];
    std::vector<double> L2, R2, sig;
    const int r = lowrank_factor(RT.data(), n, k, tol, L2, R2, &sig);         // RT ~ sum L2[rho][j] R2[rho][i]
    L.assign((size_t)r * m, 0.0);
    Rt.assign((size_t)r * n, 0.0);
    std::vector<double> q(m);
    for (int rho = 0; rho < r; ++rho) {
        const double s = sig[rho];
        for (int j = 0; j < n; ++j) Rt[(size_t)rho * n + perm[j]] = L2[(size_t)rho * n + j] / s;
        // u = Q (R2[rho], 0): reflectors applied last to first
        std::fill(q.begin(), q.end(), 0.0);
        for (int i = 0; i < k; ++i) q[i] = R2[(size_t)rho * k + i];
        for (int h = k - 1; h >= 0; --h) {
            const double* x = &W[(size_t)h * m];
            const double v0 = cn[h];
            double d = v0 * q[h];
            for (int i = h + 1; i < m; ++i) d += x[i] * q[i];
            d *= beta[h];
            q[h] -= d * v0;
            for (int i = h + 1; i < m; ++i) q[i] -= d * x[i];
        }
        for (int i = 0; i < m; ++i) L[(size_t)rho * m + i] = s * q[i];
    }
    if (sigma_out) { sigma_out->assign(n, 0.0); for (int j = 0; j < k && j < n; ++j) (*sigma_out)[j] = sig[j]; }
    return r;
}

// ---------------------------------------------------------------------------------------
// Tables of the contracted route (jx_mix.hpp).  c = S/2, umap[m] = |m - c|, NU = distinct rows = distinct columns.
// ---------------------------------------------------------------------------------------
struct MixColumns {
    // a column x' is walked in `usplit` pieces (rows u in [ubeg, ubeg + ucnt) each): per piece v = x' * usplit + h
    std::vector<int> seg0, nseg, seg;      // first knot interval, segments, samples per segment [NU * usplit][segld]
    std::vector<int> urange;               // [NU * usplit] ubeg | ucnt << 16
    std::vector<double> w4;                // [NU][wld][4]
    int segld = 0, wld = 0, maxk = 0, usplit = 1;
};

// Column x' of the quadrant, rows u = 0..NU-1: radius Q[u][x'].  Returns false when the interval index decreases along a
// column (a d_mat that is not a centred distance matrix): the route is then not taken.
#ifndef MIX_SEG_COST
#define MIX_SEG_COST 0.35
#endif
inline bool mix_column_tables(const std::vector<double>& Qtab /*[qn][qn]: (|iy-c|, |ix-c|)*/, int qn, int NU,
                              const std::vector<double>& r, MixColumns& t, int usplit = 1) {
    const int N = (int)r.size();
    usplit = std::max(1, std::min(usplit, NU));
    t.usplit = usplit;
    t.wld = (NU + 3) & ~3;
    t.w4.assign((size_t)NU * t.wld * 4, 0.0);
    std::vector<int> kof((size_t)NU * NU);
    int maxseg = 1;
    t.seg0.assign((size_t)NU * usplit, 0); t.nseg.assign((size_t)NU * usplit, 0); t.urange.assign((size_t)NU * usplit, 0);
    for (int a = 0; a < NU; ++a) {
        int kprev = 0;
        for (int u = 0; u < NU; ++u) {
            int k16; double w[4];
            const double x = Qtab[(size_t)u * qn + a];
            spline_sample_weights(r, x, &k16, w);
            int k = k16 / 16;
            const bool dead = (x != x) || !(x <= r[N - 1]);                   // NaN radius / fill value: the interval does not matter
            if (dead) k = (u == 0) ? std::max(0, N - 2) : kprev;
            if (u > 0 && k < kprev) return false;
            kprev = k;
            kof[(size_t)a * NU + u] = k;
            for (int j = 0; j < 4; ++j) t.w4[((size_t)a * t.wld + u) * 4 + j] = w[j];
        }
        // the pieces of a column are cut at equal COST, not at equal length: a sample costs its 4 + R multiply-adds, a knot interval
        // entered (with or without a sample in it) a knot request and a turn of the ring -- about a third of a sample (measured:
        // profiles/r04_subsample_scan.log); far from the axis a column crosses several intervals per sample, near it several samples share one
        std::vector<double> cum(NU + 1, 0.0);
        for (int u = 0; u < NU; ++u) cum[u + 1] = cum[u] + 1.0 + MIX_SEG_COST * (u == 0 ? 1 : kof[(size_t)a * NU + u] - kof[(size_t)a * NU + u - 1]);
        std::vector<int> cut(usplit + 1, 0);
        cut[usplit] = NU;
        for (int h = 1; h < usplit; ++h) {
            const double want = cum[NU] * h / usplit;
            int u = cut[h - 1] + 1;
            while (u < NU - (usplit - h) && cum[u] < want) ++u;
            cut[h] = u;
        }
        for (int h = 0; h < usplit; ++h) {
            const int ub = cut[h], ue = cut[h + 1], v = a * usplit + h;
            t.urange[v] = ub | ((ue - ub) << 16);
            t.seg0[v] = kof[(size_t)a * NU + ub];
            t.nseg[v] = kof[(size_t)a * NU + ue - 1] - t.seg0[v] + 1;
            maxseg = std::max(maxseg, t.nseg[v]);
        }
        t.maxk = std::max(t.maxk, kof[(size_t)a * NU + NU - 1]);
    }
    t.segld = (maxseg + 23) & ~7;                                             // zero padded: the kernels read whole groups of 8, one group ahead
    t.seg.assign((size_t)NU * usplit * t.segld, 0);
    for (int a = 0; a < NU; ++a)
        for (int h = 0; h < usplit; ++h) {
            const int v = a * usplit + h, ub = t.urange[v] & 0xffff, ue = ub + (t.urange[v] >> 16);
            for (int u = ub; u < ue; ++u) t.seg[(size_t)v * t.segld + (kof[(size_t)a * NU + u] - t.seg0[v])] += 1;
        }
    return true;
}

// ---------------------------------------------------------------------------------------
// Tables of jx_fastmath.hpp: [64] 2^(j/64), then [128][2] (1/c_i, log c_i) for the 128 intervals of z in [OFF, 2 OFF) that the
// high mantissa bits select (OFF = the double with high word 0x3FE5F000: 1.0 is the centre of interval 80, which gets c = 1
// exactly); c_i = 1 / (the double nearest to the reciprocal of the interval's centre), so that z / c_i - 1 is one fused
// multiply-add.  Built in long double.
// ---------------------------------------------------------------------------------------
inline void fastmath_tables(std::vector<double>& t) {
    t.assign(64 + 2 * 128, 0.0);
    for (int j = 0; j < 64; ++j) t[j] = (double)exp2l((long double)j / 64.0L);
    auto from_hi = [](uint32_t hi) { const uint64_t b = (uint64_t)hi << 32; double d; memcpy(&d, &b, 8); return d; };
    for (int i = 0; i < 128; ++i) {
        const double z0 = from_hi(0x3FE5F000u + ((uint32_t)i << 13)), z1 = from_hi(0x3FE5F000u + ((uint32_t)(i + 1) << 13));
        double invc = (double)(1.0L / (((long double)z0 + (long double)z1) / 2.0L)), logc = (double)(-logl((long double)invc));
        if (i == 80) { invc = 1.0; logc = 0.0; }
        t[64 + 2 * i] = invc; t[64 + 2 * i + 1] = logc;
    }
}

// ---------------------------------------------------------------------------------------
// Rows (= columns) of the quadrant that stage 1 evaluates.  Away from the cluster core the Compton-y map varies on the scale
// of the radius, far above the pixel: the quadrant is recoverable from a subset of its rows and columns by local polynomial
// interpolation, Q ~ L Q_sub L^T, and the contraction needs only the transformed operators C_sub = L^T C (stage 1) and
// G_sub = G (L x I) (stage 2).  Kept: every index below u0, every second up to u1, every fourth up to 2 u1, every eighth beyond
// (each coarser stride only where sixteen of its steps fit before the edge), and the last.
// ---------------------------------------------------------------------------------------
inline void mix_row_subset(int NU, int u0, int u1, std::vector<int>& sub) {
    sub.clear();
    u0 = std::min(u0, NU); u1 = std::max(u0, std::min(u1, NU));
    // a coarser stride is entered only where at least sixteen of its steps fit before the last row: the stencils at the outer
    // edge are one-sided, and a short run of wide steps behind narrow ones is where a high-order one-sided stencil goes wrong
    // (measured: every fourth row over the last 16 of 65 rows costs 2e-8 of the row, a 16-point stencil there 5e-5)
    const int start[3] = {u0, u1, 2 * u1};
    int stride = 1, u = 0;
    while (u < NU) {
        sub.push_back(u);
        for (int t = 0; t < 3; ++t)
            if (u >= start[t] && stride < (2 << t) && NU - 1 - u >= 16 * (2 << t)) stride = 2 << t;
        u += stride;
    }
    if (sub.back() != NU - 1) sub.push_back(NU - 1);
}

// L [NU][ns], row-major: the value at index u from the npts kept indices nearest to it (Lagrange form; the kept set is mirrored
// about 0 -- the quadrant is even in u -- so that stencils near the axis stay centred).  Kept indices get a unit row.
inline void mix_interp_matrix(int NU, const std::vector<int>& sub, int npts, std::vector<double>& L) {
    const int ns = (int)sub.size();
    npts = std::max(2, std::min(npts, ns));
    std::vector<long double> ext;
    std::vector<int> emap;
    for (int i = std::min(npts, ns - 1); i >= 1; --i) { ext.push_back(-(long double)sub[i]); emap.push_back(i); }
    for (int i = 0; i < ns; ++i) { ext.push_back((long double)sub[i]); emap.push_back(i); }
    const int ne = (int)ext.size();
    L.assign((size_t)NU * ns, 0.0);
    for (int u = 0; u < NU; ++u) {
        int j = 0;
        while (j < ne && ext[j] < (long double)u) ++j;                              // first node >= u
        if (j < ne && ext[j] == (long double)u) { L[(size_t)u * ns + emap[j]] = 1.0; continue; }
        int lo = std::max(0, std::min(ne - npts, j - npts / 2));
        for (int a = 0; a < npts; ++a) {
            long double w = 1.0L;
            for (int b = 0; b < npts; ++b)
                if (b != a) w *= ((long double)u - ext[lo + b]) / (ext[lo + a] - ext[lo + b]);
            L[(size_t)u * ns + emap[lo + a]] += (double)w;
        }
    }
}

// Separable terms of the beam image, step^2 beam[a][b] ~ sum_s by[s][a] bx[s][b] (terms above tol of the largest).
inline int beam_separable_terms(const std::vector<double>& beam, int B, double scale, double tol, std::vector<double>& by,
                                std::vector<double>& bx, std::vector<double>* sigma = nullptr) {
    std::vector<double> A((size_t)B * B);
    for (size_t e = 0; e < A.size(); ++e) A[e] = beam[e] * scale;
    return lowrank_factor(A.data(), B, B, tol, by, bx, sigma);
}

// Stage-1 operator: C[u][j], j = rho * ns + s:  C = sum_q U[rho][q] sum_{m: umap[m] = u, |q - m| <= o} by_s[q - m + o]
// (the beam along y, 'same' crop and zero padding of joxsz_funcs.py:464, folded onto the distinct rows).  Row stride cld.
inline void mix_stage1_operator(const std::vector<double>& U /*[r][S]*/, int r, const std::vector<double>& by /*[ns][B]*/, int ns,
                                int S, int B, int NU, int rows_ld, int cld, std::vector<double>& Cm) {
    const int c = S / 2, o = (B - 1) / 2;
    Cm.assign((size_t)rows_ld * cld, 0.0);
    std::vector<double> T((size_t)S * NU);
    for (int s = 0; s < ns; ++s) {
        std::fill(T.begin(), T.end(), 0.0);
        for (int q = 0; q < S; ++q)
            for (int m = std::max(0, q - o); m <= std::min(S - 1, q + o); ++m) T[(size_t)q * NU + std::abs(m - c)] += by[(size_t)s * B + (q - m + o)];
        for (int rho = 0; rho < r; ++rho) {
            const int j = rho * ns + s;
            for (int q = 0; q < S; ++q) {
                const double uq = U[(size_t)rho * S + q];
                if (uq == 0.0) continue;
                const double* tq = &T[(size_t)q * NU];
                for (int u = 0; u < NU; ++u) if (tq[u] != 0.0) Cm[(size_t)u * cld + j] += uq * tq[u];
            }
        }
    }
}

// Stage-2 operator in the matrix-core layout of jx_opgemm_kernel:  Op[(kappa * 16 + (x & 15)) * ntile + (x >> 4)],
// kappa = x' * R + j, j = rho * ns + s:
//   G[x][kappa] = sum_{n: |n - c| = x'} sum_{x'' in [0,S), |x'' - n| <= o} k_rho[(c + x - x'') mod S] bx_s[x'' - n + o],
//   k_rho[d] = sum_kc V[rho][kc] cos(2 pi kc d / S)
// (beam along x with its crop and zero padding, circular kernel of term rho along the row, extraction of the columns
// c .. S-1; joxsz_funcs.py:464-467, 472).  Rows kappa >= NU R (padding up to krows) stay zero.
inline void mix_stage2_operator(const std::vector<double>& V /*[r][Sh]*/, int r, const std::vector<double>& bx /*[ns][B]*/, int ns,
                                int S, int B, int NU, size_t krows, int ntile, std::vector<double>& Op) {
    const int c = S / 2, Sh = S / 2 + 1, o = (B - 1) / 2, nrow = S - c, R = r * ns;
    Op.assign(krows * 16 * (size_t)ntile, 0.0);
    std::vector<double> cs(S), kr(S), g(S);
    for (int d = 0; d < S; ++d) cs[d] = std::cos(2.0 * kPi * d / S);
    for (int rho = 0; rho < r; ++rho) {
        for (int d = 0; d < S; ++d) {
            double a = 0.0;
            for (int kc = 0; kc < Sh; ++kc) a += V[(size_t)rho * Sh + kc] * cs[(int)(((long long)kc * d) % S)];
            kr[d] = a;
        }
        for (int s = 0; s < ns; ++s) {
            const double* b = &bx[(size_t)s * B];
            // interior columns: g[d] = sum_t k[(d - t) mod S] b[t + o]
            for (int d = 0; d < S; ++d) {
                double a = 0.0;
                for (int t = -o; t <= o; ++t) a += kr[((d - t) % S + S) % S] * b[t + o];
                g[d] = a;
            }
            const int j = rho * ns + s;
            for (int x = 0; x < nrow; ++x)
                for (int n = 0; n < S; ++n) {
                    double v;
                    if (n >= o && n + o < S) v = g[((c + x - n) % S + S) % S];
                    else {
                        v = 0.0;
                        for (int t = std::max(-o, -n); t <= std::min(o, S - 1 - n); ++t) v += kr[((c + x - n - t) % S + S) % S] * b[t + o];
                    }
                    const size_t kappa = (size_t)std::abs(n - c) * R + j;
                    Op[(kappa * 16 + (x & 15)) * ntile + (x >> 4)] += v;
                }
        }
    }
}

// Is the quadrant of pixel radii symmetric under transposition, Q[u][x'] == Q[x'][u] bit for bit (a centred distance
// matrix on one grid for both axes)?  Then the full form keeps one operator row per unordered pair.
inline bool quadrant_is_symmetric(const std::vector<double>& Qtab, int qn, int NU) {
    for (int u = 0; u < NU; ++u)
        for (int x = u + 1; x < NU; ++x)
            if (memcmp(&Qtab[(size_t)u * qn + x], &Qtab[(size_t)x * qn + u], sizeof(double)) != 0) return false;
    return true;
}

// Full form of the contracted route: Om[x][u][x'] = d out[x] / d Q[u][x'] for ANY beam image and any real transfer-function
// weights Hy[q][kx] (no separability, no truncation):
//     out[x] = sum_{q,x''} K2[q][(c + x - x'') mod S] conv[q][x''],     K2[q][d] = sum_kx Hy[q][kx] cos(2 pi kx d / S)
//     conv[q][x''] = sum_{m,n} scale beam[q - m + o][x'' - n + o] y2d[m][n]     ('same' crop, zero padding; funcs:464)
//     y2d[m][n] = Q[|m - c|][|n - c|]
// Per distinct row u (independent: one thread each):  KqU[dx][j] = sum_{m in u} sum_q K2[q][j] beam[q - m + o][dx + o],
// T[n][d] = sum_{dx: n + dx in [0,S)} KqU[dx][(d - dx) mod S]  (one table for all interior n, 2 o boundary ones), and
// Om[x][u][x'] = sum_{n in x'} T[n][(c + x - n) mod S].
// (rows: the operator one row at a time -- sink(piece, u, x, om[NU]) receives Om[x][u][0..NU) from the thread that owns the
//  rows [u0, u1) of piece `piece`; begin(piece count) is called once before the threads start)
template <typename Begin, typename Sink>
inline void mix_full_operator_rows(const std::vector<double>& beam, int B, double scale, const std::vector<double>& Hy /*[S][Sh] real*/,
                                   int S, int NU, Begin&& begin, Sink&& sink) {
    const int c = S / 2, Sh = S / 2 + 1, o = (B - 1) / 2, nrow = S - c;
    std::vector<double> K2((size_t)S * S);
    {
        std::vector<double> cs(S);
        for (int d = 0; d < S; ++d) cs[d] = std::cos(2.0 * kPi * d / S);
        auto rows = [&](int q0, int q1) {
            for (int q = q0; q < q1; ++q)
                for (int d = 0; d < S; ++d) {
                    double a = 0.0;
                    for (int kc = 0; kc < Sh; ++kc) a += Hy[(size_t)q * Sh + kc] * cs[(int)(((long long)kc * d) % S)];
                    K2[(size_t)q * S + d] = a;
                }
        };
        jx_parallel_for(S, rows);
    }
    // boundary columns (a tap would fall outside the map) get their own table each
    std::vector<int> bidx(S, -1);
    int nbnd = 0;
    for (int n = 0; n < S; ++n) if (!(n >= o && n + o < S)) bidx[n] = nbnd++;
    auto per_u = [&](int piece, int u0, int u1) {
        std::vector<double> Kq((size_t)B * S), Tint(S), Tb((size_t)std::max(nbnd, 1) * S), omr(NU);
        for (int u = u0; u < u1; ++u) {
            std::fill(Kq.begin(), Kq.end(), 0.0);
            for (int m = 0; m < S; ++m) {
                if (std::abs(m - c) != u) continue;
                for (int q = std::max(0, m - o); q <= std::min(S - 1, m + o); ++q) {
                    const double* k2 = &K2[(size_t)q * S];
                    for (int dx = -o; dx <= o; ++dx) {
                        const double b = scale * beam[(size_t)(q - m + o) * B + dx + o];
                        if (b == 0.0) continue;
                        double* kq = &Kq[(size_t)(dx + o) * S];
                        for (int j = 0; j < S; ++j) kq[j] += b * k2[j];
                    }
                }
            }
            // interior columns n (o <= n < S - o): every dx is inside the map
            for (int d = 0; d < S; ++d) {
                double a = 0.0;
                for (int dx = -o; dx <= o; ++dx) a += Kq[(size_t)(dx + o) * S + ((d - dx) % S + S) % S];
                Tint[d] = a;
            }
            for (int n = 0; n < S; ++n) {
                if (bidx[n] < 0) continue;
                for (int d = 0; d < S; ++d) {
                    double a = 0.0;
                    for (int dx = std::max(-o, -n); dx <= std::min(o, S - 1 - n); ++dx) a += Kq[(size_t)(dx + o) * S + ((d - dx) % S + S) % S];
                    Tb[(size_t)bidx[n] * S + d] = a;
                }
            }
            for (int x = 0; x < nrow; ++x) {
                std::fill(omr.begin(), omr.end(), 0.0);
                for (int n = 0; n < S; ++n) {
                    const int d = ((c + x - n) % S + S) % S;
                    omr[std::abs(n - c)] += (bidx[n] < 0) ? Tint[d] : Tb[(size_t)bidx[n] * S + d];
                }
                sink(piece, u, x, omr.data());
            }
        }
    };
    // (the pieces of jx_parallel_for: the same split, so that a piece knows its index)
    int nt = (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(std::min(nt, 16), NU));
    const int per = (NU + nt - 1) / nt, npiece = (NU + per - 1) / per;
    begin(npiece);
    std::vector<std::thread> th;
    for (int t = 0; t < npiece; ++t) {
        const int a = t * per, b2 = std::min(NU, a + per);
        th.emplace_back([&per_u, t, a, b2]() { per_u(t, a, b2); });
    }
    for (auto& t : th) t.join();
}

inline void mix_full_operator(const std::vector<double>& beam, int B, double scale, const std::vector<double>& Hy /*[S][Sh] real*/,
                              int S, int NU, std::vector<double>& Om /*[nrow][NU][NU]*/) {
    const int nrow = S - S / 2;
    Om.assign((size_t)nrow * NU * NU, 0.0);
    mix_full_operator_rows(beam, B, scale, Hy, S, NU, [](int) {},
                           [&](int, int u, int x, const double* om) { memcpy(&Om[((size_t)x * NU + u) * NU], om, sizeof(double) * NU); });
}

// ---------------------------------------------------------------------------------------
// Exact form (round 5): the extracted row as ONE constant operator on the spline ordinates.  Every map sample is linear in the
// spline arrays (spline_sample_weights: f = A y_k + B y_k+1 + C M_k + D M_k+1, joxsz_funcs.py:460-462), the moments are linear
// in the ordinates (M = G y: mirrored_spline_op, the band of half-width K the other kernels use), and the map from the
// quadrant of distinct samples to the row is Om (mix_full_operator: any beam image, any real transfer-function weights,
// nothing truncated; joxsz_funcs.py:464-467, row of :472).  So
//     out[x] = sum_i Wy[x][i] y_i,      Wy = Om W_y + Om W_M G      (sums in long double)
// with no sample sub-grid, no singular-value cut and no low-rank / full split.  Returns the number of leading ordinates
// with a non-zero column: the radial grid beyond the map's corner plus the band of G does not reach the row.
// ---------------------------------------------------------------------------------------
inline int exact_row_operator(const std::vector<double>& beam, int B, double scale, const std::vector<double>& Hy /*[S][Sh] real*/, int S, int NU,
                              const std::vector<double>& Qtab /*[qn][qn] pixel radii of the quadrant*/, int qn, const std::vector<double>& r,
                              const std::vector<double>& G /*[N][N]*/, int K, std::vector<double>& Wy /*[nrow][N]*/) {
    const int N = (int)r.size(), nrow = S - S / 2;
    std::vector<int> sk((size_t)NU * NU);
    std::vector<double> sw((size_t)NU * NU * 4);
    for (int u = 0; u < NU; ++u)
        for (int xq = 0; xq < NU; ++xq) {
            int k16;
            spline_sample_weights(r, Qtab[(size_t)u * qn + xq], &k16, &sw[((size_t)u * NU + xq) * 4]);
            sk[(size_t)u * NU + xq] = k16 / 16;
        }
    std::vector<std::vector<long double>> acc;                  // per piece: [nrow][2][N] (ordinate weights, moment weights)
    mix_full_operator_rows(beam, B, scale, Hy, S, NU,
        [&](int npiece) { acc.assign(npiece, std::vector<long double>((size_t)nrow * 2 * N, 0.0L)); },
        [&](int piece, int u, int x, const double* om) {
            long double* ay = &acc[piece][(size_t)x * 2 * N];
            long double* am = ay + N;
            for (int xq = 0; xq < NU; ++xq) {
                const double v = om[xq];
                if (v == 0.0) continue;
                const int k = sk[(size_t)u * NU + xq];
                const double* w = &sw[((size_t)u * NU + xq) * 4];
                ay[k] += (long double)v * w[0]; am[k] += (long double)v * w[2];
                if (k + 1 < N) { ay[k + 1] += (long double)v * w[1]; am[k + 1] += (long double)v * w[3]; }
            }
        });
    Wy.assign((size_t)nrow * N, 0.0);
    auto rows = [&](int x0, int x1) {
        std::vector<long double> ay(N), am(N);
        for (int x = x0; x < x1; ++x) {
            std::fill(ay.begin(), ay.end(), 0.0L); std::fill(am.begin(), am.end(), 0.0L);
            for (const auto& a : acc)                                // (pieces in order: the sums do not depend on thread timing)
                for (int i = 0; i < N; ++i) { ay[i] += a[(size_t)x * 2 * N + i]; am[i] += a[(size_t)x * 2 * N + N + i]; }
            for (int i = 0; i < N; ++i) {
                long double s = ay[i];
                for (int k = std::max(0, i - K); k <= std::min(N - 1, i + K); ++k) s += am[k] * (long double)G[(size_t)k * N + i];
                Wy[(size_t)x * N + i] = (double)s;
            }
        }
    };
    jx_parallel_for(nrow, rows);
    int nk = 1;
    for (int x = 0; x < nrow; ++x)
        for (int i = nk; i < N; ++i) if (Wy[(size_t)x * N + i] != 0.0) nk = i + 1;
    return nk;
}

// B operand of the row product (jx_rowop_tail_kernel): macro step s (16 ordinates) x sub-step e x lane (lk, li) x the NXT output
// tiles of group g,   Opk[(((g nS + s) 4 + e) 64 + 16 lk + li) NXT + t] = Wy[16 (g NXT + t) + li][16 s + 4 lk + e]
// (zero beyond nrow and beyond the grid): a lane's NXT values are contiguous, a wave's step is one contiguous run.
inline void exact_row_layout(const std::vector<double>& Wy, int nrow, int N, int nS, int NXT, int ng, std::vector<double>& Opk) {
    Opk.assign((size_t)ng * nS * 4 * 64 * NXT, 0.0);
    for (int g = 0; g < ng; ++g)
        for (int s = 0; s < nS; ++s)
            for (int e = 0; e < 4; ++e)
                for (int lane = 0; lane < 64; ++lane)
                    for (int t = 0; t < NXT; ++t) {
                        const int x = 16 * (g * NXT + t) + (lane & 15), i = 16 * s + 4 * (lane >> 4) + e;
                        if (x < nrow && i < N) Opk[((((size_t)g * nS + s) * 4 + e) * 64 + lane) * NXT + t] = Wy[(size_t)x * N + i];
                    }
}

// The LAST ordinate tile folded into the row product (jx_ordrow_kernel with an odd number nS of ordinate tiles: its 16 ordinates
// y_i = sum_k y_scale A[i][k] pp_k, i >= 16 (nS - 1), enter the row only through Wy, so their share of the row is a constant
// operator on the profile itself,   Wf[x][k] = sum_i Wy[x][i] y_scale A[i][k],   k >= 16 (nS - 1) (A is upper triangular).
// The remaining nS - 1 tiles pair up exactly and the launch has one block fewer per walker tile: 768 instead of 832 blocks at 512^2
// / 1024 walkers, three on every compute unit.  Sums in long double; layout as exact_row_layout with the nSf = nSj - (nS - 1) macro
// steps of 16 radii in the place of the ordinate tiles:  Wfk[(((g nSf + sf) 4 + e) 64 + lane) NXT + t].
inline void exact_fold_layout(const std::vector<double>& Wy, int nrow, const std::vector<double>& r, double y_scale, int nS, int nSj, int NXT, int ng,
                              std::vector<double>& Wfk) {
    const int N = (int)r.size(), s0 = nS - 1, nSf = nSj - s0, K = 16 * nSf;
    std::vector<double> A;
    abel_matrix(r, A);
    std::vector<double> Wf((size_t)nrow * K, 0.0);
    for (int x = 0; x < nrow; ++x)
        for (int kk = 0; kk < K; ++kk) {
            const int k = 16 * s0 + kk;
            if (k >= N) continue;
            long double acc = 0.0L;
            for (int i = 16 * s0; i < std::min(N, 16 * s0 + 16); ++i)
                if (i <= k) acc += (long double)Wy[(size_t)x * N + i] * ((long double)y_scale * (long double)A[(size_t)i * N + k]);
            Wf[(size_t)x * K + kk] = (double)acc;
        }
    exact_row_layout(Wf, nrow, K, nSf, NXT, ng, Wfk);
}

// B operand of the ordinate product (jx_ordrow_kernel): y_k = sum_j y_scale A[k][j] pp_j (joxsz_funcs.py:457-459), macro step s (16
// radii) x column tile t (16 ordinates) x lane (lk, li) x sub-step e,   Typ[((s nS + t) 64 + 16 lk + li) 4 + e] = y_scale A[16 t + li][16 s + 4 lk + e]
// (zero beyond the grid; A is upper triangular, so tile t has entries from step s = t on).
inline void abel_ordinate_layout(const std::vector<double>& r, double y_scale, int nS, int nSj, std::vector<double>& Typ) {
    const int N = (int)r.size();
    std::vector<double> A;
    abel_matrix(r, A);
    Typ.assign((size_t)nSj * nS * 256, 0.0);
    for (int s = 0; s < nSj; ++s)
        for (int t = 0; t <= s && t < nS; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int k = 16 * t + (lane & 15), j = 16 * s + 4 * (lane >> 4) + e;
                    if (k < N && j < N && k <= j) Typ[(((size_t)s * nS + t) * 64 + lane) * 4 + e] = y_scale * A[(size_t)k * N + j];
                }
}

// smallest even 2^a 3^b 5^c >= n
inline int next_smooth_even(int n) {
    for (int m = std::max(2, n + (n & 1));; m += 2) {
        int v = m;
        while (v % 2 == 0) v /= 2;
        while (v % 3 == 0) v /= 3;
        while (v % 5 == 0) v /= 5;
        if (v == 1) return m;
    }
}

}  // namespace jxt
