// Transforms of the literal route (conv = rocfft: joxsz_funcs.py:464-467 executed pass by pass), hand-written for gfx950.
//
// rocFFT's own 2-D plans move each spectrum through HBM once per dimension and once more for the multiplication, fetch 2.9 x the bytes
// on the strided (column) dimension and need two kernels for the real-to-complex rows of 540 (profiles/r04_pmc_traffic.json: 53 GB per
// 1024 walkers at 512^2).  Wherever both sides are 2^a 3^b 5^c and <= 1280 the sequence runs on four kernels of this file instead
// (19 GB; JOXSZ_FFT_COLUMNS=rocfft / JOXSZ_FFT_ROWS=rocfft bring rocFFT's plans back, for all of it or for the rows):
//
//   jx_fft_rows_fwd_kernel      S rows of the padded image (the zero rows are never stored) -> hermitian row spectra; two real rows per
//                               complex transform of length P, one wave per pair.
//   jx_fft_beam_cols_kernel     per walker and group of 8 (4) adjacent columns of the row spectra -- one 128-byte line per row --: the S
//                               values of each column -> LDS, forward transform of length P, the beam spectrum multiplied in by the first
//                               pass of the inverse transform (table stored one column after the other), rows 0..S-1 (the 'same'
//                               window) written back in place.  One read and one write of the spectrum.
//   jx_fft_rows_inv_tf_kernel   inverse rows of length P, the first S entries kept (the convolved map: written to memory only for the
//                               conv_2d tap), forward rows of length S of the same pair -> row spectra of the S x S window.
//   jx_fft_tf_cols_kernel       forward transform of length S of each column; its last pass multiplies by the transfer-function table
//                               (jx_tables.hpp tf_row_table, which carries the phase of the extracted row S//2) and sums down the
//                               column: Z[kc], the spectrum of the extracted row that jx_tail_kernel turns into the row.  One read of the
//                               spectrum, nothing written but Z.
//
// The transform: Stockham autosort in LDS, in place (a lane holds the inputs of its butterflies in registers between the read and the write
// of a pass), ONE WAVE PER SEQUENCE -- passes need no block barrier, only the order of a wave's own LDS instructions --, radices 10 9 8 6 5 4 3 2
// (12 and 16 in the kernels of lengths beyond 640, which may hold 256 registers) chosen on the host for the fewest passes and the least idle
// lanes (jxt::fft_radices: 540 = 10 9 6, 512 = 8 8 8), composite radices as two base butterflies inside the registers with correctly rounded
// constant roots (jx_fft_roots.inc), the twiddles of a pass from a table of the n roots (host, long double) in LDS, j mod ns by a
// multiply-high, fp64 throughout.  The leading dimensions of the spectra are multiples of 8 complex, so that column groups start on lines.
#pragma once
#include <hip/hip_runtime.h>

#define JX_FFT_MAXPASS 12
#define JX_FFT_MAX_PER_THREAD 20          // transform length / 64 (one wave per column) at most; the kernels come in NU = 4, 8, 9, 10 (8 columns per block) and 16, 17, 20 (4)

struct JxFft {
    int n;                                // transform length
    int npass;
    int radix[JX_FFT_MAXPASS];            // 2 3 4 5 (base butterflies), 6 8 9 10 12 16 (two of them in registers)
    int ns[JX_FFT_MAXPASS];               // product of the radices before the pass
    int tstep[JX_FFT_MAXPASS];            // n / (ns radix): stride of the pass's twiddles in the table of roots; 0 in the first pass (all twiddles = root[0] = 1)
    unsigned magic[JX_FFT_MAXPASS];       // ceil(2^32 / ns): j / ns without a division
    const double2* root;                  // [n] e^{-2 pi i m / n}
};

template <int R> struct JxRoot;
#include "jx_fft_roots.inc"

__device__ __forceinline__ double2 jx_cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 jx_cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 jx_csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// a * (DIR * i): DIR = -1 forward, +1 inverse
template <int DIR> __device__ __forceinline__ double2 jx_cmuli(double2 a) { return DIR < 0 ? make_double2(a.y, -a.x) : make_double2(-a.y, a.x); }

template <int DIR, int R> __device__ __forceinline__ void jx_dft(double2 (&v)[R]);

// Cooley-Tukey inside the registers of one thread: R = R1 R2, n = R2 n1 + n2, k = k1 + R1 k2
template <int DIR, int R1, int R2> __device__ __forceinline__ void jx_dft_ct(double2 (&v)[R1 * R2]) {
    constexpr int R = R1 * R2;
    double2 a[R2][R1];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
        double2 t[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) t[n1] = v[R2 * n1 + n2];
        jx_dft<DIR, R1>(t);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            const int m = (n2 * k1) % R;
            if (m == 0) a[n2][k1] = t[k1];
            else if (4 * m == R) a[n2][k1] = jx_cmuli<DIR>(t[k1]);
            else if (2 * m == R) a[n2][k1] = make_double2(-t[k1].x, -t[k1].y);
            else if (4 * m == 3 * R) a[n2][k1] = jx_cmuli<-DIR>(t[k1]);
            else a[n2][k1] = jx_cmul(t[k1], make_double2(JxRoot<R>::re[m], DIR < 0 ? JxRoot<R>::im[m] : -JxRoot<R>::im[m]));
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        double2 t[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) t[n2] = a[n2][k1];
        jx_dft<DIR, R2>(t);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = t[k2];
    }
}

template <int DIR, int R> __device__ __forceinline__ void jx_dft(double2 (&v)[R]) {
    if constexpr (R == 2) {
        const double2 a = v[0], b = v[1];
        v[0] = jx_cadd(a, b); v[1] = jx_csub(a, b);
    } else if constexpr (R == 3) {
        const double s = 0.86602540378443864676;
        const double2 t = jx_cadd(v[1], v[2]), d = jx_csub(v[1], v[2]);
        const double2 m = make_double2(v[0].x - 0.5 * t.x, v[0].y - 0.5 * t.y);
        const double2 e = jx_cmuli<DIR>(make_double2(s * d.x, s * d.y));
        v[0] = jx_cadd(v[0], t); v[1] = jx_cadd(m, e); v[2] = jx_csub(m, e);
    } else if constexpr (R == 4) {
        const double2 a = jx_cadd(v[0], v[2]), b = jx_csub(v[0], v[2]), c = jx_cadd(v[1], v[3]), d = jx_cmuli<DIR>(jx_csub(v[1], v[3]));
        v[0] = jx_cadd(a, c); v[1] = jx_cadd(b, d); v[2] = jx_csub(a, c); v[3] = jx_csub(b, d);
    } else if constexpr (R == 5) {
        const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410, s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
        const double2 a1 = jx_cadd(v[1], v[4]), a2 = jx_cadd(v[2], v[3]), b1 = jx_csub(v[1], v[4]), b2 = jx_csub(v[2], v[3]);
        const double2 r1 = make_double2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
        const double2 r2 = make_double2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
        const double2 i1 = jx_cmuli<DIR>(make_double2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
        const double2 i2 = jx_cmuli<DIR>(make_double2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
        v[0] = make_double2(v[0].x + a1.x + a2.x, v[0].y + a1.y + a2.y);
        v[1] = jx_cadd(r1, i1); v[4] = jx_csub(r1, i1);
        v[2] = jx_cadd(r2, i2); v[3] = jx_csub(r2, i2);
    }
    else if constexpr (R == 6) jx_dft_ct<DIR, 2, 3>(v);
    else if constexpr (R == 8) jx_dft_ct<DIR, 2, 4>(v);
    else if constexpr (R == 9) jx_dft_ct<DIR, 3, 3>(v);
    else if constexpr (R == 10) jx_dft_ct<DIR, 2, 5>(v);
    else if constexpr (R == 12) jx_dft_ct<DIR, 3, 4>(v);
    else { static_assert(R == 16, "radix"); jx_dft_ct<DIR, 4, 4>(v); }
}

// The threads of a column are the lanes of one wave, and a wave's LDS instructions execute in program order: between the read
// and the write of a pass, and between passes, nothing more is needed than keeping the compiler from moving LDS accesses across the point.
#define JX_FFT_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// one pass of radix R over the column `col` (n complex in LDS) by the 64 lanes of its wave; root: the table of the n roots (LDS, or global
// memory where the columns leave no room)
// MODE 1: the inputs are multiplied by gtab[input index] first (the beam spectrum between forward and inverse transform); MODE 2: the outputs
// are not stored but multiplied by gtab[output index] and summed into acc (transfer-function table and column sum); gtab is laid out per
// column, so that the lanes of the wave read consecutive entries
template <int DIR, int R, int NU, bool TW, int MODE, typename RootPtr>
__device__ __forceinline__ void jx_fft_pass(double2* __restrict__ col, RootPtr root, int n, int ns, int tstep, unsigned magic, int jj,
                                            const double2* __restrict__ gtab, double2& acc) {
    constexpr int MAXB = (NU + R - 1) / R;                       // NU: transform length / 64, rounded up
    const int nb = n / R;
    double2 v[MAXB][R];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int j = jj + i * 64;
        if (j < nb) {
#pragma unroll
            for (int t = 0; t < R; ++t) v[i][t] = col[j + t * nb];
            if constexpr (MODE == 1) {
                double2 g[R];
#pragma unroll
                for (int t = 0; t < R; ++t) g[t] = gtab[j + t * nb];
#pragma unroll
                for (int t = 0; t < R; ++t) v[i][t] = jx_cmul(v[i][t], g[t]);
            }
            if constexpr (TW) {                                      // (the first pass has none: TW = false)
                const int ks = (j - (int)__umulhi((unsigned)j, magic) * ns) * tstep;
                double2 w[R];
#pragma unroll
                for (int t = 1; t < R; ++t) w[t] = root[t * ks];
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    if (DIR > 0) w[t].y = -w[t].y;
                    v[i][t] = jx_cmul(v[i][t], w[t]);
                }
            }
            jx_dft<DIR, R>(v[i]);
        }
    }
    if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            const int j = jj + i * 64;
            if (j < nb) {
                const int k = TW ? j - (int)__umulhi((unsigned)j, magic) * ns : 0, j0 = (j - k) * R + k;
                double2 g[R];
#pragma unroll
                for (int t = 0; t < R; ++t) g[t] = gtab[j0 + t * ns];
#pragma unroll
                for (int t = 0; t < R; ++t) { const double2 z = jx_cmul(v[i][t], g[t]); acc.x += z.x; acc.y += z.y; }
            }
        }
        return;
    }
    JX_FFT_WAVE_SYNC();
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int j = jj + i * 64;
        if (j < nb) {
            const int k = TW ? j - (int)__umulhi((unsigned)j, magic) * ns : 0, j0 = (j - k) * R + k;
#pragma unroll
            for (int t = 0; t < R; ++t) col[j0 + t * ns] = v[i][t];
        }
    }
    JX_FFT_WAVE_SYNC();
}

// the whole transform of one column, natural order in, natural order out; called by the 64 lanes of the column's wave (no block barrier
// inside: the caller puts one between its own accesses in another lane order and this).  At least two passes (the host sees to it).
// PRE: the first pass multiplies its inputs by gtab; POST: the last pass leaves sum(output * gtab) of the lane in acc instead of the outputs.
template <int DIR, int NU, bool BIG, bool PRE, bool POST, typename RootPtr>
__device__ __forceinline__ void jx_fft_lds(double2* __restrict__ col, RootPtr root, const JxFft& f, int jj, const double2* __restrict__ gtab, double2& acc) {
#define JX_FFT_SWITCH(TW, MODE, p)                                                                                      \
    switch (f.radix[p]) {                                                                                               \
        JX_FFT_CASE(2, TW, MODE, p) JX_FFT_CASE(3, TW, MODE, p) JX_FFT_CASE(4, TW, MODE, p) JX_FFT_CASE(5, TW, MODE, p) JX_FFT_CASE(6, TW, MODE, p)   \
        JX_FFT_CASE(8, TW, MODE, p) JX_FFT_CASE(9, TW, MODE, p) JX_FFT_CASE(10, TW, MODE, p)                            \
        default:                                     /* radices 12 and 16 only where a wave may hold 256 registers */  \
            if constexpr (BIG) {                                                                                        \
                if (f.radix[p] == 12) jx_fft_pass<DIR, 12, NU, TW, MODE>(col, root, f.n, f.ns[p], f.tstep[p], f.magic[p], jj, gtab, acc);   \
                else if (f.radix[p] == 16) jx_fft_pass<DIR, 16, NU, TW, MODE>(col, root, f.n, f.ns[p], f.tstep[p], f.magic[p], jj, gtab, acc);   \
            }                                                                                                           \
            break;                                                                                                      \
    }
#define JX_FFT_CASE(RR, TW, MODE, p) case RR: jx_fft_pass<DIR, RR, NU, TW, MODE>(col, root, f.n, f.ns[p], f.tstep[p], f.magic[p], jj, gtab, acc); break;
    const int last = f.npass - 1;
    JX_FFT_SWITCH(false, (PRE ? 1 : 0), 0)
    for (int p = 1; p < last; ++p) JX_FFT_SWITCH(true, 0, p)
    JX_FFT_SWITCH(true, (POST ? 2 : 0), last)
#undef JX_FFT_CASE
#undef JX_FFT_SWITCH
}

// rows ly, ly + rpi, ... < n (at most NU of them): every load is requested before the first use (one trip to HBM / L2 instead of one per row)
#define JX_FFT_BATCHED(n, LOAD, USE)                                                   \
    for (int y0_ = ly; y0_ < (n); y0_ += rpi * NU) {                           \
        double2 t_[NU];                                                        \
        _Pragma("unroll") for (int u_ = 0; u_ < NU; ++u_) { const int y = y0_ + u_ * rpi; t_[u_] = y < (n) ? LOAD : make_double2(0.0, 0.0); }   \
        _Pragma("unroll") for (int u_ = 0; u_ < NU; ++u_) { const int y = y0_ + u_ * rpi; const double2 t = t_[u_]; if (y < (n)) { USE; } }   \
    }

// lengths up to 640 (NU <= 10): the roots sit behind the columns in LDS, four waves per SIMD; beyond (up to 1280): roots read from global memory, two
#define JX_FFT_SMALL(NU) ((NU) <= 10)
#define JX_FFT_RUN(DIR, PRE, POST, gtab, acc) do {                                                                      \
        if constexpr (JX_FFT_SMALL(NU)) jx_fft_lds<DIR, NU, false, PRE, POST>(jx_fft_sm + c * L, (const double2*)(jx_fft_sm + CB * L), f, jj, gtab, acc);   \
        else jx_fft_lds<DIR, NU, true, PRE, POST>(jx_fft_sm + c * L, f.root, f, jj, gtab, acc);                        \
    } while (0)
#define JX_FFT_LDS_BYTES(n, CB, ROOTS) (((size_t)(CB) * ((size_t)(n) + 1) + ((ROOTS) ? (size_t)(n) : 0)) * sizeof(double2))      // the columns, then the roots

// spec [walker][S][ldc] (row spectra of the padded image, rows 0..S-1), bhat [ldc][P] (beam spectrum times step^2 / P^2, one column after the other).
// grid (walkers, ldc / CB), 64 CB threads (one wave per column), JX_FFT_LDS_BYTES(P, CB) of LDS (roots in LDS) or without the roots
template <int CB, int NU>
__global__ void __launch_bounds__(64 * CB) __attribute__((amdgpu_waves_per_eu(JX_FFT_SMALL(NU) ? 4 : 2, JX_FFT_SMALL(NU) ? 4 : 2)))
jx_fft_beam_cols_kernel(JxFft f, double2* __restrict__ spec, const double2* __restrict__ bhat, int S, int ldc) {
    extern __shared__ double2 jx_fft_sm[];
    const int P = f.n, L = P + 1, tid = threadIdx.x;
    constexpr int tpc = 64, rpi = 64, NT = 64 * CB;                            // lanes per column; rows per load instruction
    const int c0 = blockIdx.y * CB;                                            // (walkers fastest: the blocks in flight share a slice of bhat)
    double2* g = spec + (size_t)blockIdx.x * S * ldc + c0;
    const int lc = tid % CB, ly = tid / CB, c = tid / tpc, jj = tid % tpc;
    JX_FFT_BATCHED(P, (y < S ? g[(size_t)y * ldc + lc] : make_double2(0.0, 0.0)), jx_fft_sm[lc * L + y] = t)
    if (JX_FFT_SMALL(NU)) for (int m = tid; m < P; m += NT) jx_fft_sm[CB * L + m] = f.root[m];
    __syncthreads();
    double2 none;
    JX_FFT_RUN(-1, false, false, (const double2*)nullptr, none);
    JX_FFT_RUN(+1, true, false, bhat + (size_t)(c0 + c) * P, none);        // (the column belongs to this wave from the first pass to the last)
    __syncthreads();
    for (int y = ly; y < S; y += rpi) g[(size_t)y * ldc + lc] = jx_fft_sm[lc * L + y];
}

// tfspec [walker][S][ldt] (row spectra of the S x S window), htab [ldt][S] (tf_row_table one column after the other, zero in the padding columns),
// zout [walker][2][Sh]: re, im of Z[kc] = sum_kr X[kr][kc] H[kr][kc].   grid (walkers, ldt / CB), JX_FFT_LDS_BYTES(S, CB) of LDS
template <int CB, int NU>
__global__ void __launch_bounds__(64 * CB) __attribute__((amdgpu_waves_per_eu(JX_FFT_SMALL(NU) ? 4 : 2, JX_FFT_SMALL(NU) ? 4 : 2)))
jx_fft_tf_cols_kernel(JxFft f, const double2* __restrict__ tfspec, const double2* __restrict__ htab, int ldt, int Sh, double* __restrict__ zout,
                      int) {
    extern __shared__ double2 jx_fft_sm[];
    const int S = f.n, L = S + 1, tid = threadIdx.x;
    constexpr int tpc = 64, rpi = 64, NT = 64 * CB;
    const int c0 = blockIdx.y * CB;
    const double2* g = tfspec + (size_t)blockIdx.x * S * ldt + c0;
    const int lc = tid % CB, ly = tid / CB, c = tid / tpc, jj = tid % tpc;
    JX_FFT_BATCHED(S, g[(size_t)y * ldt + lc], jx_fft_sm[lc * L + y] = t)
    if (JX_FFT_SMALL(NU)) for (int m = tid; m < S; m += NT) jx_fft_sm[CB * L + m] = f.root[m];
    __syncthreads();
    // the last pass multiplies by the table and sums down the column instead of storing; then across the lanes of the column's wave
    double2 z = make_double2(0.0, 0.0);
    JX_FFT_RUN(-1, false, true, htab + (size_t)(c0 + c) * S, z);
#pragma unroll
    for (int o = 32; o; o >>= 1) { z.x += __shfl_xor(z.x, o); z.y += __shfl_xor(z.y, o); }
    if (jj == 0 && c0 + c < Sh) {
        zout[((size_t)blockIdx.x * 2) * Sh + c0 + c] = z.x;
        zout[((size_t)blockIdx.x * 2 + 1) * Sh + c0 + c] = z.y;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Row passes.  Two real rows ride one complex transform (z = a + i b; A[k] = (Z[k] + conj Z[n-k]) / 2, B[k] = (Z[k] - conj Z[n-k]) / 2i),
// one wave per pair of rows, its own n complex in LDS: no block barrier after the roots have been copied.  Rows are numbered through
// the whole launch (walker * S + y), so a pair may straddle two walkers.
//
//   jx_fft_rows_fwd_kernel      img rows [R][P] (first S entries non-zero) -> hermitian row spectra spec [R][ldc]      (joxsz_funcs.py:464, the
//                               forward half of fftconvolve)
//   jx_fft_rows_inv_tf_kernel   spec rows -> inverse transform of length P -> the first S entries (the 'same' window; written to conv only
//                               when the convolved map is tapped) -> forward transform of length S of the same pair -> tfspec [R][ldt]
//                               (joxsz_funcs.py:464 inverse half, :466 forward half).  The convolved map never goes through HBM.
// ---------------------------------------------------------------------------------------------------------------------------------
#define JX_FFT_ROWS_LDS_BYTES(n, WPB, NROOT) (((size_t)(WPB) * ((size_t)(n) + 1) + (size_t)(NROOT)) * sizeof(double2))

template <int WPB, int NU>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(JX_FFT_SMALL(NU) ? 4 : 2, JX_FFT_SMALL(NU) ? 4 : 2)))
jx_fft_rows_fwd_kernel(JxFft f, const double* __restrict__ img, double2* __restrict__ spec, int S, int ldc, int R) {
    extern __shared__ double2 jx_fft_sm[];
    const int P = f.n, L = P + 1, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (JX_FFT_SMALL(NU)) {
        for (int m = tid; m < P; m += 64 * WPB) jx_fft_sm[WPB * L + m] = f.root[m];
        __syncthreads();
    }
    const int r0 = 2 * (blockIdx.x * WPB + wave);
    if (r0 >= R) return;
    const bool two = r0 + 1 < R;
    double2* col = jx_fft_sm + wave * L;
    const double* a = img + (size_t)r0 * P;
    const double* b = a + P;
    {
        double2 va[NU / 2 + 1], vb[NU / 2 + 1];
#pragma unroll
        for (int u = 0; u < NU / 2 + 1; ++u) {                     // two entries of each row per lane and step
            const int x = 2 * (lane + 64 * u);
            va[u] = make_double2(0.0, 0.0); vb[u] = va[u];
            if (x + 1 < S) { va[u] = *reinterpret_cast<const double2*>(a + x); if (two) vb[u] = *reinterpret_cast<const double2*>(b + x); }
            else if (x < S) { va[u].x = a[x]; if (two) vb[u].x = b[x]; }
        }
#pragma unroll
        for (int u = 0; u < NU / 2 + 1; ++u) {
            const int x = 2 * (lane + 64 * u);
            if (x < P) col[x] = make_double2(va[u].x, vb[u].x);
            if (x + 1 < P) col[x + 1] = make_double2(va[u].y, vb[u].y);
        }
    }
    JX_FFT_WAVE_SYNC();
    double2 none;
    if constexpr (JX_FFT_SMALL(NU)) jx_fft_lds<-1, NU, false, false, false>(col, (const double2*)(jx_fft_sm + WPB * L), f, lane, (const double2*)nullptr, none);
    else jx_fft_lds<-1, NU, true, false, false>(col, f.root, f, lane, (const double2*)nullptr, none);
    double2* sa = spec + (size_t)r0 * ldc;
    double2* sb = sa + ldc;
    for (int k = lane; k <= P / 2; k += 64) {
        const double2 zk = col[k], zm = col[k ? P - k : 0];
        sa[k] = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
        if (two) sb[k] = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
    }
}

template <int WPB, int NU>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(JX_FFT_SMALL(NU) ? 4 : 2, JX_FFT_SMALL(NU) ? 4 : 2)))
jx_fft_rows_inv_tf_kernel(JxFft fP, JxFft fS, const double2* __restrict__ spec, double* __restrict__ conv, double2* __restrict__ tfspec,
                          int ldc, int ldt, int R) {
    extern __shared__ double2 jx_fft_sm[];
    const int P = fP.n, S = fS.n, L = P + 1, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (JX_FFT_SMALL(NU)) {
        for (int m = tid; m < P; m += 64 * WPB) jx_fft_sm[WPB * L + m] = fP.root[m];
        for (int m = tid; m < S; m += 64 * WPB) jx_fft_sm[WPB * L + P + m] = fS.root[m];
        __syncthreads();
    }
    const int r0 = 2 * (blockIdx.x * WPB + wave);
    if (r0 >= R) return;
    const bool two = r0 + 1 < R;
    double2* col = jx_fft_sm + wave * L;
    const double2* sa = spec + (size_t)r0 * ldc;
    const double2* sb = sa + ldc;
    {
        double2 va[NU / 2 + 1], vb[NU / 2 + 1];
#pragma unroll
        for (int u = 0; u < NU / 2 + 1; ++u) {
            const int k = lane + 64 * u;
            va[u] = make_double2(0.0, 0.0); vb[u] = va[u];
            if (k <= P / 2) { va[u] = sa[k]; if (two) vb[u] = sb[k]; }
        }
#pragma unroll
        for (int u = 0; u < NU / 2 + 1; ++u) {                     // Z[k] = A[k] + i B[k]; beyond n / 2 from the hermitian symmetry of A and B
            const int k = lane + 64 * u;
            if (k <= P / 2) {
                col[k] = make_double2(va[u].x - vb[u].y, va[u].y + vb[u].x);
                if (k && 2 * k < P) col[P - k] = make_double2(va[u].x + vb[u].y, vb[u].x - va[u].y);
            }
        }
    }
    JX_FFT_WAVE_SYNC();
    double2 none;
    if constexpr (JX_FFT_SMALL(NU)) jx_fft_lds<+1, NU, false, false, false>(col, (const double2*)(jx_fft_sm + WPB * L), fP, lane, (const double2*)nullptr, none);
    else jx_fft_lds<+1, NU, true, false, false>(col, fP.root, fP, lane, (const double2*)nullptr, none);
    if (conv) {
        double* ca = conv + (size_t)r0 * P;
        for (int x = lane; x < S; x += 64) { const double2 z = col[x]; ca[x] = z.x; if (two) ca[P + x] = z.y; }
        JX_FFT_WAVE_SYNC();
    }
    if constexpr (JX_FFT_SMALL(NU)) jx_fft_lds<-1, NU, false, false, false>(col, (const double2*)(jx_fft_sm + WPB * L + P), fS, lane, (const double2*)nullptr, none);
    else jx_fft_lds<-1, NU, true, false, false>(col, fS.root, fS, lane, (const double2*)nullptr, none);
    double2* ta = tfspec + (size_t)r0 * ldt;
    double2* tb = ta + ldt;
    for (int k = lane; k <= S / 2; k += 64) {
        const double2 zk = col[k], zm = col[k ? S - k : 0];
        ta[k] = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
        if (two) tb[k] = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
    }
}
