// Host-only C entry points over jx_tables.hpp so that the table builders can be
// checked on a machine without a GPU (tests/test_host_tables.py, ctypes).
#include <cstring>
#include <vector>
#include "jx_tables.hpp"

extern "C" {

int jxt_abel_matrix(const double* r, int n, double* out /*[n*n]*/) {
    std::vector<double> rv(r, r + n), A;
    jxt::abel_matrix(rv, A);
    memcpy(out, A.data(), sizeof(double) * (size_t)n * n);
    return 0;
}

int jxt_abel_onfly(const double* r, int n, double* cj, double* dg, double* sp) {
    std::vector<double> rv(r, r + n), a, b, c;
    jxt::abel_onfly_tables(rv, a, b, c);
    memcpy(cj, a.data(), sizeof(double) * n);
    memcpy(dg, b.data(), sizeof(double) * n);
    memcpy(sp, c.data(), sizeof(double) * n);
    return 0;
}

int jxt_mirrored_spline_op(const double* r, int n, double* out /*[n*n]*/) {
    std::vector<double> rv(r, r + n), G;
    if (!jxt::mirrored_spline_op(rv, G)) return -1;
    memcpy(out, G.data(), sizeof(double) * (size_t)n * n);
    return jxt::band_halfwidth(G, n, 1e-20);
}

int jxt_nak_eval_matrix(const double* x, int n, const double* q, int nq, double* out /*[nq*n]*/) {
    std::vector<double> xv(x, x + n), qv(q, q + nq), E;
    if (!jxt::nak_eval_matrix(xv, qv, E)) return -1;
    memcpy(out, E.data(), sizeof(double) * (size_t)nq * n);
    return 0;
}

int jxt_beam_spectrum(const double* beam, int B, int P, double scale, double* out /*[P*(P/2+1)*2]*/) {
    std::vector<double> bv(beam, beam + (size_t)B * B), o;
    jxt::beam_spectrum(bv, B, P, scale, o);
    memcpy(out, o.data(), sizeof(double) * o.size());
    return 0;
}

int jxt_tf_row_table(const double* filt, int S, double* out /*[S*(S/2+1)*2]*/) {
    std::vector<double> fv(filt, filt + (size_t)S * S), H;
    jxt::tf_row_table(fv, S, H);
    memcpy(out, H.data(), sizeof(double) * H.size());
    return 0;
}

int jxt_tf_hy_table(const double* filt, int S, double* out /*[S*(S/2+1)*2]*/) {
    std::vector<double> fv(filt, filt + (size_t)S * S), H;
    jxt::tf_hy_table(fv, S, H);
    memcpy(out, H.data(), sizeof(double) * H.size());
    return 0;
}

int jxt_host_fft(double* re, double* im, int n, int sign) {
    std::vector<double> r(re, re + n), i(im, im + n);
    jxt::host_fft(r, i, sign);
    memcpy(re, r.data(), sizeof(double) * n);
    memcpy(im, i.data(), sizeof(double) * n);
    return 0;
}

int jxt_next_smooth_even(int n) { return jxt::next_smooth_even(n); }

// truncated SVD: returns the rank r; L [r][m], Rt [r][n] (caller sizes them for min(m, n) terms), sigma [n]
// qr != 0: the rank-revealing form (Householder QR with column pivoting, then Jacobi on the small factor)
int jxt_lowrank_factor(const double* A, int m, int n, double tol, int qr, double* L, double* Rt, double* sigma) {
    std::vector<double> l, rt, sg;
    const int r = qr ? jxt::lowrank_factor_qr(A, m, n, tol, l, rt, &sg) : jxt::lowrank_factor(A, m, n, tol, l, rt, &sg);
    std::copy(l.begin(), l.end(), L);
    std::copy(rt.begin(), rt.end(), Rt);
    std::copy(sg.begin(), sg.end(), sigma);
    return r;
}

// operator of jx_abel_gemm_kernel, [rows][ld]; G = the dense mirrored-spline operator, K = half-width of the band in use
int jxt_abel_spline_operator(const double* r, int n, const double* G, int K, double y_scale, int rows, int ld, double* out) {
    std::vector<double> o;
    jxt::abel_spline_operator(std::vector<double>(r, r + n), std::vector<double>(G, G + (size_t)n * n), K, y_scale, rows, ld, o);
    std::copy(o.begin(), o.end(), out);
    return 0;
}
int jxt_band_halfwidth(const double* G, int n, double tol) { return jxt::band_halfwidth(std::vector<double>(G, G + (size_t)n * n), n, tol); }

// ---- contracted route (jx_mix.hpp) ----
// weights of one spline sample: out = {k, A, B, C, D}
int jxt_spline_sample(const double* r, int n, double x, double* out) {
    int k16; double w[4];
    jxt::spline_sample_weights(std::vector<double>(r, r + n), x, &k16, w);
    out[0] = k16 / 16; out[1] = w[0]; out[2] = w[1]; out[3] = w[2]; out[4] = w[3];
    return 0;
}

// column tables of stage 1: meta = {segld, wld, maxk}; call with seg0 == nullptr for the sizes.  Returns 0 when the interval
// index decreases along a column.
// usplit: pieces a column is walked in; seg0, nseg, urange then hold NU * usplit entries (piece v = x' * usplit + h), seg
// NU * usplit rows; urange[v] = first row u | rows << 16 (may be null).
int jxt_mix_columns(const double* Qtab, int qn, int NU, const double* r, int n, int usplit, int* meta, int* seg0, int* nseg, int* seg,
                    int* urange, double* w4) {
    jxt::MixColumns t;
    if (!jxt::mix_column_tables(std::vector<double>(Qtab, Qtab + (size_t)qn * qn), qn, NU, std::vector<double>(r, r + n), t, usplit)) return 0;
    meta[0] = t.segld; meta[1] = t.wld; meta[2] = t.maxk;
    if (!seg0) return 1;
    std::copy(t.seg0.begin(), t.seg0.end(), seg0);
    std::copy(t.nseg.begin(), t.nseg.end(), nseg);
    std::copy(t.seg.begin(), t.seg.end(), seg);
    if (urange) std::copy(t.urange.begin(), t.urange.end(), urange);
    std::copy(t.w4.begin(), t.w4.end(), w4);
    return 1;
}

// Both operators of the low-rank form for a beam image and a filter: Cm [NU][R] (stage 1) and G [nrow][NU * R] (stage 2, dense,
// kappa = x' * R + j).  Returns R = r * ns (counts[0] = r, counts[1] = ns); call with Cm == nullptr for the counts only.
int jxt_mix_lowrank_operators(const double* beam, int B, double scale, const double* filt, int S, double tol, double beam_tol,
                              int* counts, double* Cm, double* G) {
    const int Sh = S / 2 + 1, c = S / 2, nrow = S - c, NU = std::max(c, S - 1 - c) + 1;
    std::vector<double> hy, U, V, by, bx;
    jxt::tf_hy_table(std::vector<double>(filt, filt + (size_t)S * S), S, hy);
    std::vector<double> A((size_t)S * Sh);
    for (size_t e = 0; e < A.size(); ++e) A[e] = hy[2 * e];
    const int r = jxt::lowrank_factor_qr(A.data(), S, Sh, tol, U, V);
    const int ns = jxt::beam_separable_terms(std::vector<double>(beam, beam + (size_t)B * B), B, scale, beam_tol, by, bx);
    counts[0] = r; counts[1] = ns;
    const int R = r * ns;
    if (!Cm) return R;
    std::vector<double> cm, op;
    jxt::mix_stage1_operator(U, r, by, ns, S, B, NU, NU, R, cm);
    std::copy(cm.begin(), cm.end(), Cm);
    const int ntile = (nrow + 15) / 16;
    const size_t K = (size_t)NU * R;
    jxt::mix_stage2_operator(V, r, bx, ns, S, B, NU, K, ntile, op);
    for (size_t k = 0; k < K; ++k)
        for (int x = 0; x < nrow; ++x) G[(size_t)x * K + k] = op[(k * 16 + (x & 15)) * ntile + (x >> 4)];
    return R;
}

// Full form: Om [nrow][NU][NU] for a beam image and a filter.  Returns 0, or -1 when the weights are not real.
int jxt_mix_full_operator(const double* beam, int B, double scale, const double* filt, int S, double* Om) {
    const int Sh = S / 2 + 1, c = S / 2, NU = std::max(c, S - 1 - c) + 1;
    std::vector<double> hy, om;
    jxt::tf_hy_table(std::vector<double>(filt, filt + (size_t)S * S), S, hy);
    std::vector<double> A((size_t)S * Sh);
    double mre = 0, mim = 0;
    for (size_t e = 0; e < A.size(); ++e) { A[e] = hy[2 * e]; mre = std::max(mre, std::fabs(hy[2 * e])); mim = std::max(mim, std::fabs(hy[2 * e + 1])); }
    if (!(mim <= 1e-15 * mre)) return -1;
    jxt::mix_full_operator(std::vector<double>(beam, beam + (size_t)B * B), B, scale, A, S, NU, om);
    std::copy(om.begin(), om.end(), Om);
    return 0;
}

// the rows (= columns) of the quadrant kept by the sub-grid (returns their number; sub may be null) and the
// interpolation matrix L [NU][ns] from them to every row
int jxt_mix_row_subset(int NU, int u0, int u1, int* sub) {
    std::vector<int> v;
    jxt::mix_row_subset(NU, u0, u1, v);
    if (sub) memcpy(sub, v.data(), sizeof(int) * v.size());
    return (int)v.size();
}

int jxt_mix_interp_matrix(int NU, const int* sub, int ns, int npts, double* L /*[NU*ns]*/) {
    std::vector<int> v(sub, sub + ns);
    std::vector<double> Lm;
    jxt::mix_interp_matrix(NU, v, npts, Lm);
    memcpy(L, Lm.data(), sizeof(double) * Lm.size());
    return 0;
}

// tables of csrc/jx_fastmath.hpp: out[0..64) = 2^(j/64), out[64..320) = (1/c_i, log c_i) of the 128 mantissa intervals
int jxt_fastmath_tables(double* out /*[320]*/) {
    std::vector<double> t;
    jxt::fastmath_tables(t);
    memcpy(out, t.data(), sizeof(double) * t.size());
    return (int)t.size();
}

// Exact form (round 5): Wy [nrow][N] with  map_out[S//2, S//2 + x] = sum_i Wy[x][i] y_i  for the ordinates y of the mirrored cubic spline
// (joxsz_funcs.py:460-467, row of :472), built from d_mat's quadrant, the beam image, the filter and the radial grid alone.
// Returns the number of leading ordinates with a non-zero column, -1 when the transfer-function weights are not real, -2 when
// d_mat lacks the mirror structure of centdistmat.
int jxt_exact_row_operator(const double* d_mat, const double* beam, int B, double scale, const double* filt, int S, const double* r, int N, double* Wy) {
    const int Sh = S / 2 + 1, c = S / 2, NU = std::max(c, S - 1 - c) + 1;
    std::vector<double> Q((size_t)NU * NU);
    for (int b = 0; b < NU; ++b)
        for (int a = 0; a < NU; ++a) {
            const int iy = (c + b < S) ? c + b : c - b, ix = (c + a < S) ? c + a : c - a;
            Q[(size_t)b * NU + a] = d_mat[(size_t)iy * S + ix];
        }
    for (int iy = 0; iy < S; ++iy)
        for (int ix = 0; ix < S; ++ix)
            if (memcmp(&Q[(size_t)std::abs(iy - c) * NU + std::abs(ix - c)], &d_mat[(size_t)iy * S + ix], sizeof(double)) != 0) return -2;
    std::vector<double> hy, wy, rv(r, r + N), G;
    jxt::tf_hy_table(std::vector<double>(filt, filt + (size_t)S * S), S, hy);
    std::vector<double> A((size_t)S * Sh);
    double mre = 0, mim = 0;
    for (size_t e = 0; e < A.size(); ++e) { A[e] = hy[2 * e]; mre = std::max(mre, std::fabs(hy[2 * e])); mim = std::max(mim, std::fabs(hy[2 * e + 1])); }
    if (!(mim <= 1e-15 * mre)) return -1;
    if (!jxt::mirrored_spline_op(rv, G)) return -3;
    const int K = jxt::band_halfwidth(G, N, 1e-20);
    const int nk = jxt::exact_row_operator(std::vector<double>(beam, beam + (size_t)B * B), B, scale, A, S, NU, Q, NU, rv, G, K, wy);
    std::copy(wy.begin(), wy.end(), Wy);
    return nk;
}

// the layouts the kernels of the exact form read: Opk (row operator) and Typ (ordinate operator)
int jxt_exact_row_layout(const double* Wy, int nrow, int N, int nS, int NXT, int ng, double* Opk) {
    std::vector<double> o;
    jxt::exact_row_layout(std::vector<double>(Wy, Wy + (size_t)nrow * N), nrow, N, nS, NXT, ng, o);
    std::copy(o.begin(), o.end(), Opk);
    return 0;
}
// the last ordinate tile's share of the row as an operator on the profile (odd nS), in the row operator's layout: out [ng][nSj - (nS - 1)][4][64][NXT]
int jxt_exact_fold_layout(const double* Wy, int nrow, const double* r, int N, double y_scale, int nS, int nSj, int NXT, int ng, double* out) {
    std::vector<double> o;
    jxt::exact_fold_layout(std::vector<double>(Wy, Wy + (size_t)nrow * N), nrow, std::vector<double>(r, r + N), y_scale, nS, nSj, NXT, ng, o);
    std::copy(o.begin(), o.end(), out);
    return 0;
}
int jxt_abel_ordinate_layout(const double* r, int n, double y_scale, int nS, int nSj, double* out) {
    std::vector<double> o;
    jxt::abel_ordinate_layout(std::vector<double>(r, r + n), y_scale, nS, nSj, o);
    std::copy(o.begin(), o.end(), out);
    return 0;
}

// radices of the LDS transforms of the literal route for length n (jx_fft.hpp): passes, or 0 when the length is not 2^a 3^b 5^c
int jxt_fft_radices(int n, int* radix /*[12]*/) { return jxt::fft_radices(n, radix, 12); }

}  // extern "C"
