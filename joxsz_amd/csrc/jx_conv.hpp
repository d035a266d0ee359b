// Hand-written beam + transfer-function convolution for gfx950, fp64, in three passes
// over a mixed (row, kx) domain -- the "custom" alternative to the rocFFT sequence.
//
// The reference computes (joxsz_funcs.py:464-467, 472)
//     conv = fftconvolve(y_2d, beam_2d, 'same') * step^2
//     map  = real(ifft2(fft2(conv) * filtering));   row = map[S//2, S//2:]
// With P = 2*LP >= S + (B-1)/2 the same numbers are obtained as
//   pass 1  Y[r][kx]  = rFFT_P(zero-padded map row r)                    (x direction)
//   pass 2  C[r][kx]  = sum_u Bx[u][kx] * Y[r+o-u][kx]                   (y direction: a
//           B-tap FIR whose taps are the x-transforms of the beam rows; for a beam image
//           symmetric under both flips -- the only kind mybeam builds, joxsz_funcs.py:46-76
//           -- the taps are real and symmetric in u)
//   pass 3  conv[r][:] = irFFT_P(C[r]) cropped to S;  X[r][kc] = rFFT_S(conv[r]);
//           Zpart[kc] += X[r][kc] * Hy[r][kc]                            (Hy: the transfer
//           function transformed back to real space along y at the offset of row S//2)
//   tail    row[x] = sum_kc Re(Z[kc] e^{2 pi i kc x / S})
// Every pass keeps rows contiguous in memory ([row][kx], 16-byte complex), so all global
// accesses are coalesced 1-KiB wave transactions and no transpose is needed.
//
// FFTs are Stockham autosort passes in LDS with radices 4, 2, 3 chosen at compile time from
// the length; a real transform of length 2L is done as a complex transform of length L plus
// the usual even/odd split.
#pragma once
#include <hip/hip_runtime.h>
#include "jx_regfft.hpp"

typedef double2 cplx;

__device__ __forceinline__ cplx c_mul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx c_mulc(cplx a, cplx b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
__device__ __forceinline__ cplx c_add(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx c_sub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx c_conj(cplx a) { return make_double2(a.x, -a.y); }
// multiply by -i (forward) or +i (inverse)
template <bool INV> __device__ __forceinline__ cplx c_rot(cplx a) { return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x); }

template <int R, bool INV> struct Bfly;
template <bool INV> struct Bfly<2, INV> {
    static __device__ __forceinline__ void run(cplx* u) { const cplx a = u[0], b = u[1]; u[0] = c_add(a, b); u[1] = c_sub(a, b); }
};
template <bool INV> struct Bfly<4, INV> {
    static __device__ __forceinline__ void run(cplx* u) {
        const cplx s0 = c_add(u[0], u[2]), d0 = c_sub(u[0], u[2]);
        const cplx s1 = c_add(u[1], u[3]), d1 = c_rot<INV>(c_sub(u[1], u[3]));
        u[0] = c_add(s0, s1); u[2] = c_sub(s0, s1);
        u[1] = c_add(d0, d1); u[3] = c_sub(d0, d1);
    }
};
template <bool INV> struct Bfly<3, INV> {
    static __device__ __forceinline__ void run(cplx* u) {
        const double s = INV ? 0.86602540378443864676 : -0.86602540378443864676;      // sin(+-2 pi/3)
        const cplx t = c_add(u[1], u[2]), d = c_sub(u[1], u[2]);
        const cplx m = make_double2(u[0].x - 0.5 * t.x, u[0].y - 0.5 * t.y);
        const cplx q = make_double2(-s * d.y, s * d.x);                              // i*s*d
        u[0] = c_add(u[0], t);
        u[1] = c_add(m, q);
        u[2] = c_sub(m, q);
    }
};

constexpr int jx_radix_of(int rem) { return (rem % 4 == 0) ? 4 : ((rem % 2 == 0) ? 2 : 3); }
constexpr bool jx_len_ok(int L) {
    while (L % 2 == 0) L /= 2;
    while (L % 3 == 0) L /= 3;
    return L == 1;
}

// One Stockham pass of radix R over `nrows` rows of length L (row stride LD) from src to dst.
// tw[n] = exp(-2 pi i n / L).
template <int L, int LD, int Ns, bool INV>
__device__ __forceinline__ void jx_fft_passes(cplx*& src, cplx*& dst, int nrows, const cplx* tw) {
    if constexpr (Ns < L) {
        constexpr int R = jx_radix_of(L / Ns);
        constexpr int T = L / R;
        constexpr int TWS = L / (Ns * R);
        for (int b = threadIdx.x; b < nrows * T; b += blockDim.x) {
            const int row = b / T, j = b - row * T;
            const int k = j % Ns;
            const cplx* in = src + row * LD;
            cplx u[R];
#pragma unroll
            for (int t = 0; t < R; ++t) u[t] = in[j + t * T];
            if (Ns > 1) {
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    const cplx w = tw[t * k * TWS];
                    u[t] = INV ? c_mulc(u[t], w) : c_mul(u[t], w);
                }
            }
            Bfly<R, INV>::run(u);
            cplx* out = dst + row * LD + (j - k) * R + k;
#pragma unroll
            for (int t = 0; t < R; ++t) out[t * Ns] = u[t];
        }
        __syncthreads();
        cplx* tmp = src; src = dst; dst = tmp;
        jx_fft_passes<L, LD, Ns * R, INV>(src, dst, nrows, tw);
    }
}

// in-LDS complex FFT of `nrows` rows; on return `a` points at the result buffer
template <int L, int LD, bool INV>
__device__ __forceinline__ void jx_fft_rows(cplx*& a, cplx*& b, int nrows, const cplx* tw) {
    jx_fft_passes<L, LD, 1, INV>(a, b, nrows, tw);
}

struct JxConv {
    int S, Sh, B, o, P, Ph, LP, LS;
    int ntap;                       // o + 1
    int nblk3;                      // pass-3 blocks per walker (partials to sum in the tail)
    const cplx* tw_lp;              // [LP]   exp(-2 pi i n / LP)
    const cplx* tw_ls;              // [LS]   exp(-2 pi i n / LS)
    const cplx* tw_p;               // [LP+1] exp(-2 pi i k / P)
    const cplx* tw_s;               // [LS+1] exp(-2 pi i k / S)
    const double* taps;             // [ntap][Ph] real beam taps: tap[t][kx] multiplies rows r -+ t
    const cplx* hy;                 // [S][Sh]
};

#define JX_FIR_TILE 64              // output rows per pass-2 block
#define JX_FIR_NR 8                 // consecutive output rows per thread

// ------------------------------------------------------------------------------------
// pass 1: real rows of the map -> half spectra.  grid = (S / ROWS, walkers)
// ------------------------------------------------------------------------------------
template <int LP, int ROWS>
__global__ void __launch_bounds__(256)
jx_rowfft_kernel(JxConv c, const double* __restrict__ img, size_t img_ld, size_t img_ws, cplx* __restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    cplx* bufA = reinterpret_cast<cplx*>(sm);
    cplx* bufB = bufA + ROWS * LP;
    cplx* tw = bufB + ROWS * LP;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int r0 = blockIdx.x * ROWS, w = blockIdx.y;
    const int nrows = min(ROWS, c.S - r0);
    const int half = c.S / 2;
    for (int n = tid; n < LP; n += nth) tw[n] = c.tw_lp[n];
    for (int e = tid; e < nrows * LP; e += nth) {
        const int row = e / LP, n = e - row * LP;
        cplx z = make_double2(0.0, 0.0);
        if (n < half) z = *reinterpret_cast<const double2*>(img + (size_t)w * img_ws + (size_t)(r0 + row) * img_ld + 2 * n);
        bufA[row * LP + n] = z;
    }
    __syncthreads();
    cplx *a = bufA, *b = bufB;
    jx_fft_rows<LP, LP, false>(a, b, nrows, tw);
    // X[k] = (Z[k] + conj Z[LP-k])/2 - (i/2) e^{-2 pi i k/P} (Z[k] - conj Z[LP-k]),  k = 0..LP
    const int Ph = c.Ph;
    for (int e = tid; e < nrows * Ph; e += nth) {
        const int row = e / Ph, k = e - row * Ph;
        const cplx zk = a[row * LP + (k == LP ? 0 : k)];
        const cplx zc = c_conj(a[row * LP + (k == 0 ? 0 : LP - k)]);
        const cplx s = c_add(zk, zc), d = c_mul(c_sub(zk, zc), c.tw_p[k]);
        // -(i/2) * d = (d.y/2, -d.x/2)
        Y[((size_t)w * c.S + r0 + row) * Ph + k] = make_double2(0.5 * (s.x + d.y), 0.5 * (s.y - d.x));
    }
}

// ------------------------------------------------------------------------------------
// pass 2: FIR along rows with real, symmetric, kx-dependent taps.
// grid = (ceil(Ph/64), ceil(S/JX_FIR_TILE), walkers); block = 512 threads: lane = kx in the
// 64-wide slab, wave = group of JX_FIR_NR consecutive output rows.  The input tile
// (JX_FIR_TILE + 2o rows) and the taps sit in LDS; each thread slides over its 2o + NR input
// rows keeping a window of NR taps in registers (one new tap and one new input per step).
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
jx_beamfir_kernel(JxConv c, const cplx* __restrict__ Y, cplx* __restrict__ C) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int o = c.o, Ph = c.Ph, S = c.S;
    const int nin = JX_FIR_TILE + 2 * o;
    cplx* tile = reinterpret_cast<cplx*>(sm);                 // [nin][64]
    double* taps = sm + (size_t)2 * nin * 64;                 // [o+1][64]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwv = blockDim.x >> 6;
    const int kx0 = blockIdx.x * 64, rt0 = blockIdx.y * JX_FIR_TILE, w = blockIdx.z;
    const int kx = kx0 + lane;
    const bool kok = kx < Ph;
    const cplx* Yw = Y + (size_t)w * S * Ph;
    for (int rr = wv; rr < nin; rr += nwv) {
        const int m = rt0 - o + rr;
        cplx v = make_double2(0.0, 0.0);
        if (kok && m >= 0 && m < S) v = Yw[(size_t)m * Ph + kx];
        tile[rr * 64 + lane] = v;
    }
    for (int t = wv; t <= o; t += nwv) taps[t * 64 + lane] = kok ? c.taps[(size_t)t * Ph + kx] : 0.0;
    __syncthreads();

    for (int g = wv; g < JX_FIR_TILE / JX_FIR_NR; g += nwv) {
        const int rbase = g * JX_FIR_NR;                      // first output row of the group, tile-relative
        if (rt0 + rbase >= S) break;
        double ar[JX_FIR_NR], ai[JX_FIR_NR], tp[JX_FIR_NR];
#pragma unroll
        for (int j = 0; j < JX_FIR_NR; ++j) { ar[j] = 0.0; ai[j] = 0.0; tp[j] = 0.0; }
        // input index i runs over tile rows rbase .. rbase + 2o + NR - 1 (output j sits at tile row rbase + o + j);
        // tap for (i, j) is taps[|i - o - j|] when that is <= o.  The window tp[j] = taps[|i - o - j|] shifts
        // by one output per input step: tp_new[j] = tp_old[j-1], tp_new[0] = taps[|i - o|] (0 beyond o).
        const int nstep = 2 * o + JX_FIR_NR;
        for (int i0 = 0; i0 < nstep; i0 += JX_FIR_NR) {
#pragma unroll
            for (int s = 0; s < JX_FIR_NR; ++s) {
                const int i = i0 + s;
                if (i < nstep) {
                    const int d = abs(i - o);
                    // rotate the window: slot (s) becomes the new j = 0 entry after s shifts of the unrolled body
#pragma unroll
                    for (int j = JX_FIR_NR - 1; j > 0; --j) tp[j] = tp[j - 1];
                    tp[0] = (d <= o) ? taps[d * 64 + lane] : 0.0;
                    const cplx v = tile[(rbase + i) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < JX_FIR_NR; ++j) { ar[j] = fma(tp[j], v.x, ar[j]); ai[j] = fma(tp[j], v.y, ai[j]); }
                }
            }
        }
        if (kok) {
#pragma unroll
            for (int j = 0; j < JX_FIR_NR; ++j) {
                const int r = rt0 + rbase + j;
                if (r < S) C[((size_t)w * S + r) * Ph + kx] = make_double2(ar[j], ai[j]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// pass 3: per row inverse real FFT (P), crop to S, forward real FFT (S), weight by Hy and
// reduce over the block's rows.  grid = (S / ROWS, walkers).
//   part [walkers][nblk3][Sh] partial sums of Z;  tap_conv (optional) [walkers][S][S]
// ------------------------------------------------------------------------------------
template <int LP, int LS, int ROWS>
__global__ void __launch_bounds__(256)
jx_rowtf_kernel(JxConv c, const cplx* __restrict__ C, cplx* __restrict__ part, double* __restrict__ tap_conv) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    cplx* bufA = reinterpret_cast<cplx*>(sm);
    cplx* bufB = bufA + ROWS * LP;
    cplx* twp = bufB + ROWS * LP;          // [LP]
    cplx* tws = twp + LP;                        // [LS]
    const int tid = threadIdx.x, nth = blockDim.x;
    const int r0 = blockIdx.x * ROWS, w = blockIdx.y;
    const int S = c.S, Ph = c.Ph, Sh = c.Sh;
    const int nrows = min(ROWS, S - r0);
    for (int n = tid; n < LP; n += nth) twp[n] = c.tw_lp[n];
    for (int n = tid; n < LS; n += nth) tws[n] = c.tw_ls[n];
    // Z[k] = (X[k] + conj X[LP-k]) + i e^{+2 pi i k/P} (X[k] - conj X[LP-k]),  k = 0..LP-1
    for (int e = tid; e < nrows * LP; e += nth) {
        const int row = e / LP, k = e - row * LP;
        const cplx* Xr = C + ((size_t)w * S + r0 + row) * Ph;
        const cplx xk = Xr[k], xc = c_conj(Xr[LP - k]);
        const cplx s = c_add(xk, xc), d = c_mulc(c_sub(xk, xc), c.tw_p[k]);     // d = (xk - xc) e^{+2 pi i k/P}
        bufA[row * LP + k] = make_double2(s.x - d.y, s.y + d.x);                  // s + i d
    }
    __syncthreads();
    cplx *a = bufA, *b = bufB;
    jx_fft_rows<LP, LP, true>(a, b, nrows, twp);
    // a[row][n] = (conv[2n], conv[2n+1]); the first LS entries are the cropped row
    if (tap_conv) {
        for (int e = tid; e < nrows * LS; e += nth) {
            const int row = e / LS, n = e - row * LS;
            *reinterpret_cast<double2*>(tap_conv + ((size_t)w * S + r0 + row) * S + 2 * n) = a[row * LP + n];
        }
    }
    jx_fft_rows<LS, LP, false>(a, b, nrows, tws);
    // X'[k] = (Z[k] + conj Z[LS-k])/2 - (i/2) e^{-2 pi i k/S}(Z[k] - conj Z[LS-k]);  Zpart[k] = sum_rows X'[k] Hy[r][k]
    for (int k = tid; k < Sh; k += nth) {
        const cplx tk = c.tw_s[k];
        double zr = 0.0, zi = 0.0;
        for (int row = 0; row < nrows; ++row) {
            const cplx zk = a[row * LP + (k == LS ? 0 : k)];
            const cplx zc = c_conj(a[row * LP + (k == 0 ? 0 : LS - k)]);
            const cplx s = c_add(zk, zc), d = c_mul(c_sub(zk, zc), tk);
            const cplx x = make_double2(0.5 * (s.x + d.y), 0.5 * (s.y - d.x));
            const cplx h = c.hy[(size_t)(r0 + row) * Sh + k];
            zr += x.x * h.x - x.y * h.y;
            zi += x.x * h.y + x.y * h.x;
        }
        part[((size_t)w * c.nblk3 + blockIdx.x) * Sh + k] = make_double2(zr, zi);
    }
}


// ====================================================================================
// Two-level versions of passes 1 and 3: L = L1 * L2 with both sub-transforms done in
// registers (jx_regfft.hpp).  Step A: one thread per (row, n2) transforms the L1 samples
// n = L2 n1 + n2 and applies the twiddle W_L^{n2 k1}; step B: one thread per (row, k1)
// transforms over n2 and owns the outputs k = k1 + L1 k2.  Two LDS round trips per
// transform instead of one per radix pass, three or four workgroup barriers instead of
// six, and 16-18 independent loads in flight per thread.  LDS rows use odd strides
// (in 16-byte slots) so that neither step meets a bank conflict.
// ====================================================================================
__device__ __forceinline__ jx_c jx_ld(const cplx* p) { const cplx v = *p; return jxc(v.x, v.y); }
__device__ __forceinline__ void jx_st(cplx* p, jx_c v) { *p = make_double2(v.x, v.y); }

template <int L> struct jx_lay {
    static constexpr int L1 = jx_plan2<L>::L1, L2 = jx_plan2<L>::L2;
    static constexpr int L2P = L2 | 1;                     // padded inner stride (odd)
    static constexpr int SPAN = (L1 * L2P > L + 1 ? L1 * L2P : L + 1);
    static constexpr int RS = SPAN | 1;                    // row stride (odd), >= L + 1
    static constexpr int TMAX = L1 > L2 ? L1 : L2;
};

// step A on registers already loaded: x[n1] -> FFT over n1, twiddle, store to padded layout
template <int L, bool INV>
__device__ __forceinline__ void jx_stepA_store(jx_c* x, int n2, cplx* Mrow, const cplx* tw) {
    constexpr int L1 = jx_lay<L>::L1, L2P = jx_lay<L>::L2P;
    jx_regfft<L1, INV>::run(x);
#pragma unroll
    for (int k1 = 0; k1 < L1; ++k1) {
        jx_c v = x[k1];
        if (k1 > 0) {
            const cplx w = tw[n2 * k1];                     // e^{-2 pi i n2 k1 / L}
            v = INV ? jxc(v.x * w.x + v.y * w.y, v.y * w.x - v.x * w.y) : jxc(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
        }
        jx_st(Mrow + k1 * L2P + n2, v);
    }
}

// step B: read the L2 samples of (row, k1) from the padded layout, transform, leave in y
template <int L, bool INV>
__device__ __forceinline__ void jx_stepB_load(jx_c* y, int k1, const cplx* Mrow) {
    constexpr int L2 = jx_lay<L>::L2, L2P = jx_lay<L>::L2P;
#pragma unroll
    for (int n2 = 0; n2 < L2; ++n2) y[n2] = jx_ld(Mrow + k1 * L2P + n2);
    jx_regfft<L2, INV>::run(y);
}

// ------------------------------------------------------------------------------------
// pass 1 (two-level): grid = (ceil(S / ROWS), walkers), 256 threads, ROWS = 256 / max(L1, L2)
// ------------------------------------------------------------------------------------
template <int LP, int ROWS>
__global__ void __launch_bounds__(256)
jx_rowfft2_kernel(JxConv c, const double* __restrict__ img, size_t img_ld, size_t img_ws, cplx* __restrict__ Y) {
    typedef jx_lay<LP> Lay;
    constexpr int L1 = Lay::L1, L2 = Lay::L2, RS = Lay::RS;
    static_assert(ROWS * Lay::TMAX <= 256, "one task per thread");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    cplx* M = reinterpret_cast<cplx*>(sm);                  // [ROWS][RS]
    cplx* tw = M + ROWS * RS;                               // [LP]
    const int tid = threadIdx.x, nth = blockDim.x;
    const int r0 = blockIdx.x * ROWS, w = blockIdx.y;
    const int nrows = min(ROWS, c.S - r0), half = c.S / 2;
    for (int n = tid; n < LP; n += nth) tw[n] = c.tw_lp[n];

    const int rowA = tid / L2, n2 = tid - rowA * L2;
    const bool actA = tid < ROWS * L2 && rowA < nrows;
    jx_c x[L1];
    if (actA) {
        const double* src = img + (size_t)w * img_ws + (size_t)(r0 + rowA) * img_ld;
#pragma unroll
        for (int n1 = 0; n1 < L1; ++n1) {
            const int n = n1 * L2 + n2;
            x[n1] = jxc(0.0, 0.0);
            if (n < half) { const double2 v = *reinterpret_cast<const double2*>(src + 2 * n); x[n1] = jxc(v.x, v.y); }
        }
    }
    __syncthreads();
    if (actA) jx_stepA_store<LP, false>(x, n2, M + rowA * RS, tw);
    __syncthreads();
    const int rowB = tid / L1, k1 = tid - rowB * L1;
    const bool actB = tid < ROWS * L1 && rowB < nrows;
    jx_c y[L2];
    if (actB) jx_stepB_load<LP, false>(y, k1, M + rowB * RS);
    __syncthreads();
    if (actB) {
#pragma unroll
        for (int k2 = 0; k2 < L2; ++k2) jx_st(M + rowB * RS + k1 + L1 * k2, y[k2]);
    }
    __syncthreads();
    const int Ph = c.Ph;
    for (int e = tid; e < nrows * Ph; e += nth) {
        const int row = e / Ph, k = e - row * Ph;
        const cplx zk = M[row * RS + (k == LP ? 0 : k)];
        const cplx zc = c_conj(M[row * RS + (k == 0 ? 0 : LP - k)]);
        const cplx s = c_add(zk, zc), d = c_mul(c_sub(zk, zc), c.tw_p[k]);
        Y[((size_t)w * c.S + r0 + row) * Ph + k] = make_double2(0.5 * (s.x + d.y), 0.5 * (s.y - d.x));
    }
}

// ------------------------------------------------------------------------------------
// pass 3 (two-level): grid = (ceil(S / ROWS), walkers), 256 threads
// ------------------------------------------------------------------------------------
template <int LP, int LS, int ROWS>
__global__ void __launch_bounds__(256)
jx_rowtf2_kernel(JxConv c, const cplx* __restrict__ C, cplx* __restrict__ part, double* __restrict__ tap_conv) {
    typedef jx_lay<LP> LayP;
    typedef jx_lay<LS> LayS;
    constexpr int P1 = LayP::L1, P2 = LayP::L2, P2P = LayP::L2P;
    constexpr int S1 = LayS::L1, S2 = LayS::L2, S2P = LayS::L2P;
    constexpr int RS = LayP::RS > LayS::RS ? LayP::RS : LayS::RS;
    static_assert(ROWS * LayP::TMAX <= 256 && ROWS * LayS::TMAX <= 256, "one task per thread");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    cplx* M = reinterpret_cast<cplx*>(sm);                  // [ROWS][RS]
    cplx* twp = M + ROWS * RS;                              // [LP]
    cplx* tws = twp + LP;                                   // [LS]
    const int tid = threadIdx.x, nth = blockDim.x;
    const int r0 = blockIdx.x * ROWS, w = blockIdx.y;
    const int S = c.S, Ph = c.Ph, Sh = c.Sh;
    const int nrows = min(ROWS, S - r0);
    for (int n = tid; n < LP; n += nth) twp[n] = c.tw_lp[n];
    for (int n = tid; n < LS; n += nth) tws[n] = c.tw_ls[n];
    // Z[k] = (X[k] + conj X[LP-k]) + i e^{+2 pi i k/P} (X[k] - conj X[LP-k]) into the padded layout of n = k
    for (int e = tid; e < nrows * LP; e += nth) {
        const int row = e / LP, k = e - row * LP;
        const cplx* Xr = C + ((size_t)w * S + r0 + row) * Ph;
        const cplx xk = Xr[k], xc = c_conj(Xr[LP - k]);
        const cplx s = c_add(xk, xc), d = c_mulc(c_sub(xk, xc), c.tw_p[k]);
        const int n1 = k / P2, n2 = k - n1 * P2;
        M[row * RS + n1 * P2P + n2] = make_double2(s.x - d.y, s.y + d.x);
    }
    __syncthreads();
    {   // inverse transform of length LP
        const int rowA = tid / P2, n2 = tid - rowA * P2;
        const bool actA = tid < ROWS * P2 && rowA < nrows;
        jx_c x[P1];
        if (actA) {
#pragma unroll
            for (int n1 = 0; n1 < P1; ++n1) x[n1] = jx_ld(M + rowA * RS + n1 * P2P + n2);
            jx_stepA_store<LP, true>(x, n2, M + rowA * RS, twp);       // same column: in place
        }
        __syncthreads();
        const int rowB = tid / P1, k1 = tid - rowB * P1;
        const bool actB = tid < ROWS * P1 && rowB < nrows;
        jx_c y[P2];
        if (actB) jx_stepB_load<LP, true>(y, k1, M + rowB * RS);
        __syncthreads();
        if (actB) {
            // z[n] = (conv[2n], conv[2n+1]), n = k1 + P1 k2; the first LS of them are the cropped row
#pragma unroll
            for (int k2 = 0; k2 < P2; ++k2) {
                const int n = k1 + P1 * k2;
                if (n < LS) {
                    const int a = n / S2, b = n - a * S2;
                    jx_st(M + rowB * RS + a * S2P + b, y[k2]);
                    if (tap_conv)
                        *reinterpret_cast<double2*>(tap_conv + ((size_t)w * S + r0 + rowB) * S + 2 * n) = make_double2(y[k2].x, y[k2].y);
                }
            }
        }
        __syncthreads();
    }
    {   // forward transform of length LS
        const int rowA = tid / S2, n2 = tid - rowA * S2;
        const bool actA = tid < ROWS * S2 && rowA < nrows;
        jx_c x[S1];
        if (actA) {
#pragma unroll
            for (int n1 = 0; n1 < S1; ++n1) x[n1] = jx_ld(M + rowA * RS + n1 * S2P + n2);
            jx_stepA_store<LS, false>(x, n2, M + rowA * RS, tws);
        }
        __syncthreads();
        const int rowB = tid / S1, k1 = tid - rowB * S1;
        const bool actB = tid < ROWS * S1 && rowB < nrows;
        jx_c y[S2];
        if (actB) jx_stepB_load<LS, false>(y, k1, M + rowB * RS);
        __syncthreads();
        if (actB) {
#pragma unroll
            for (int k2 = 0; k2 < S2; ++k2) jx_st(M + rowB * RS + k1 + S1 * k2, y[k2]);
        }
        __syncthreads();
    }
    for (int k = tid; k < Sh; k += nth) {
        const cplx tk = c.tw_s[k];
        double zr = 0.0, zi = 0.0;
        for (int row = 0; row < nrows; ++row) {
            const cplx zk = M[row * RS + (k == LS ? 0 : k)];
            const cplx zc = c_conj(M[row * RS + (k == 0 ? 0 : LS - k)]);
            const cplx s = c_add(zk, zc), d = c_mul(c_sub(zk, zc), tk);
            const cplx x = make_double2(0.5 * (s.x + d.y), 0.5 * (s.y - d.x));
            const cplx h = c.hy[(size_t)(r0 + row) * Sh + k];
            zr += x.x * h.x - x.y * h.y;
            zi += x.x * h.y + x.y * h.x;
        }
        part[((size_t)w * c.nblk3 + blockIdx.x) * Sh + k] = make_double2(zr, zi);
    }
}

// ------------------------------------------------------------------------------------
// pass 2, second form: 32-kx slabs (two blocks per CU), tap window kept in registers by
// rotating the register NAMES (slot of logical tap j at step i is (j - i) mod NR), so the
// unrolled body contains no moves.  grid = (ceil(Ph/32), ceil(S/64), walkers), 256 threads:
// lane & 31 = kx in the slab, (wave, lane >> 5) = group of 8 consecutive output rows.
// ------------------------------------------------------------------------------------
#define JX_FIR2_KX 32
__global__ void __launch_bounds__(256)
jx_beamfir2_kernel(JxConv c, const cplx* __restrict__ Y, cplx* __restrict__ C) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int o = c.o, Ph = c.Ph, S = c.S;
    const int nin = JX_FIR_TILE + 2 * o;
    cplx* tile = reinterpret_cast<cplx*>(sm);                         // [nin][32]
    double* taps = sm + (size_t)2 * nin * JX_FIR2_KX;                 // [o+1][32]
    const int tid = threadIdx.x;
    const int lx = tid & 31, grp = tid >> 5;                          // 8 groups of 8 rows
    const int kx0 = blockIdx.x * JX_FIR2_KX, rt0 = blockIdx.y * JX_FIR_TILE, w = blockIdx.z;
    const int kx = kx0 + lx;
    const bool kok = kx < Ph;
    const cplx* Yw = Y + (size_t)w * S * Ph;
    for (int rr = grp; rr < nin; rr += 8) {
        const int m = rt0 - o + rr;
        cplx v = make_double2(0.0, 0.0);
        if (kok && m >= 0 && m < S) v = Yw[(size_t)m * Ph + kx];
        tile[rr * JX_FIR2_KX + lx] = v;
    }
    for (int t = grp; t <= o; t += 8) taps[t * JX_FIR2_KX + lx] = kok ? c.taps[(size_t)t * Ph + kx] : 0.0;
    __syncthreads();

    const int rbase = grp * JX_FIR_NR;
    if (rt0 + rbase >= S) return;
    double ar[JX_FIR_NR], ai[JX_FIR_NR], tp[JX_FIR_NR];
#pragma unroll
    for (int j = 0; j < JX_FIR_NR; ++j) { ar[j] = 0.0; ai[j] = 0.0; tp[j] = 0.0; }
    const int nstep = 2 * o + JX_FIR_NR;
    for (int i0 = 0; i0 < nstep; i0 += JX_FIR_NR) {
#pragma unroll
        for (int s = 0; s < JX_FIR_NR; ++s) {
            const int i = i0 + s;
            if (i < nstep) {
                const int d = abs(i - o);
                tp[(JX_FIR_NR - s) % JX_FIR_NR] = (d <= o) ? taps[d * JX_FIR2_KX + lx] : 0.0;   // logical j = 0 at step i
                const cplx v = tile[(rbase + i) * JX_FIR2_KX + lx];
#pragma unroll
                for (int j = 0; j < JX_FIR_NR; ++j) {
                    const double t = tp[(j - s + JX_FIR_NR) % JX_FIR_NR];
                    ar[j] = fma(t, v.x, ar[j]);
                    ai[j] = fma(t, v.y, ai[j]);
                }
            }
        }
    }
    if (kok) {
#pragma unroll
        for (int j = 0; j < JX_FIR_NR; ++j) {
            const int r = rt0 + rbase + j;
            if (r < S) C[((size_t)w * S + r) * Ph + kx] = make_double2(ar[j], ai[j]);
        }
    }
}
