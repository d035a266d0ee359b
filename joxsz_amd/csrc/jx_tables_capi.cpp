// Host-only C entry points over jx_tables.hpp so that the table builders can be
// checked on a machine without a GPU (tests/test_host_tables.py, ctypes).
#include <cstring>
#include <vector>
#include "jx_tables.hpp"
#include "jx_regfft.hpp"

// in-register FFT of every supported small length, for the host test
template <int N> static void regfft_any(double* re, double* im, int inv) {
    jx_c x[N];
    for (int i = 0; i < N; ++i) x[i] = jxc(re[i], im[i]);
    if (inv) jx_regfft<N, true>::run(x); else jx_regfft<N, false>::run(x);
    for (int i = 0; i < N; ++i) { re[i] = x[i].x; im[i] = x[i].y; }
}

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

int jxt_beam_fir_taps(const double* beam, int B, int P, double scale, double* out /*[(o+1)*(P/2+1)]*/) {
    std::vector<double> bv(beam, beam + (size_t)B * B), t;
    if (!jxt::beam_is_symmetric(bv, B)) return -1;
    jxt::beam_fir_taps(bv, B, P, scale, t);
    memcpy(out, t.data(), sizeof(double) * t.size());
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

int jxt_regfft(int n, int inv, double* re, double* im) {
    switch (n) {
#define C(N) case N: regfft_any<N>(re, im, inv); return 0;
        C(2) C(3) C(4) C(6) C(8) C(9) C(12) C(16) C(18) C(24) C(27) C(32)
#undef C
    }
    return -1;
}
double jxt_cx_cos2pi(long long k, long long n) { return jx_cx_cos2pi(k, n); }
double jxt_cx_sin2pi(long long k, long long n) { return jx_cx_sin2pi(k, n); }

int jxt_conv_row_tables(int S, int o, int mirror, int* umap /*[S]*/, int* urow /*[S]*/, int* jrow /*[S]*/,
                        int* rowjob /*[S]*/, int* seg /*[3*S]*/, int* counts /*[3]: NU, NJ, nseg*/) {
    jxt::ConvRows t;
    jxt::conv_row_tables(S, o, mirror != 0, t);
    memcpy(umap, t.umap.data(), sizeof(int) * S);
    memcpy(urow, t.urow.data(), sizeof(int) * t.NU);
    memcpy(jrow, t.jrow.data(), sizeof(int) * t.NJ);
    memcpy(rowjob, t.rowjob.data(), sizeof(int) * S);
    memcpy(seg, t.seg.data(), sizeof(int) * t.seg.size());
    counts[0] = t.NU; counts[1] = t.NJ; counts[2] = t.nseg;
    return 0;
}

// truncated SVD: returns the rank r; L [r][m], Rt [r][n] (caller sizes them for min(m, n) terms), sigma [n]
int jxt_lowrank_factor(const double* A, int m, int n, double tol, double* L, double* Rt, double* sigma) {
    std::vector<double> l, rt, sg;
    const int r = jxt::lowrank_factor(A, m, n, tol, l, rt, &sg);
    std::copy(l.begin(), l.end(), L);
    std::copy(rt.begin(), rt.end(), Rt);
    std::copy(sg.begin(), sg.end(), sigma);
    return r;
}

// fused FIR + job combination: out [nb][RP][KU] (see jx_tables.hpp fused_row_operator); mirror row structure
int jxt_fused_row_operator(const double* U, int r, int S, int o, int mirror, const double* coef, int nb, int ldc, int RP, int KU, double* out) {
    jxt::ConvRows rows;
    jxt::conv_row_tables(S, o, mirror != 0, rows);
    std::vector<double> u(U, U + (size_t)r * rows.NJ), w;
    jxt::fused_row_operator(u, r, rows, S, o, coef, nb, ldc, RP, KU, w);
    std::copy(w.begin(), w.end(), out);
    return rows.NJ;
}

int jxt_custom_conv_lp(int S, int o) { return jxt::custom_conv_lp(S, o); }

int jxt_next_smooth_even(int n) { return jxt::next_smooth_even(n); }

// tables of jx_rowdct_kernel: dk [nb][na4], dw [nb][na4][4], x0k [nb], x0w [nb][4], pk [(LP/4 + 1)][4]; meta = {gl, na4, has_x0, amax}.
// Call with dk == nullptr to get meta only (sizes).  Returns 0 when the sizes do not fit.
int jxt_dct_tables(const double* Qrad, int na, int nb, const double* r, int n, int S, int LP, int* meta, int* dk, double* dw,
                   int* x0k, double* x0w, double* pk) {
    jxt::DctTables t;
    if (!jxt::dct_tables(std::vector<double>(Qrad, Qrad + (size_t)na * nb), na, nb, std::vector<double>(r, r + n), S, LP, t)) return 0;
    meta[0] = t.gl; meta[1] = t.na4; meta[2] = t.has_x0; meta[3] = t.amax;
    if (!dk) return 1;
    std::copy(t.dk.begin(), t.dk.end(), dk);
    std::copy(t.dw.begin(), t.dw.end(), dw);
    std::copy(t.x0k.begin(), t.x0k.end(), x0k);
    std::copy(t.x0w.begin(), t.x0w.end(), x0w);
    std::copy(t.pk.begin(), t.pk.end(), pk);
    return 1;
}

// operator of jx_abel_gemm_kernel, [rows][ld]; G = the dense mirrored-spline operator, K = half-width of the band in use
int jxt_abel_spline_operator(const double* r, int n, const double* G, int K, double y_scale, int rows, int ld, double* out) {
    std::vector<double> o;
    jxt::abel_spline_operator(std::vector<double>(r, r + n), std::vector<double>(G, G + (size_t)n * n), K, y_scale, rows, ld, o);
    std::copy(o.begin(), o.end(), out);
    return 0;
}
int jxt_band_halfwidth(const double* G, int n, double tol) { return jxt::band_halfwidth(std::vector<double>(G, G + (size_t)n * n), n, tol); }

// real-space circular kernels of the odd-side route: out [nmg][r][64][KQ]; returns nmg
int jxt_odd_rowspace_operator(const double* V, int r, int S, int KQ, double* out) {
    std::vector<double> o;
    int nmg = 0;
    jxt::odd_rowspace_operator(std::vector<double>(V, V + (size_t)r * (S / 2 + 1)), r, S, KQ, o, &nmg);
    if (out) std::copy(o.begin(), o.end(), out);
    return nmg;
}
int jxt_custom_conv_lp_odd(int S, int o) { return jxt::custom_conv_lp_odd(S, o); }

}  // extern "C"
