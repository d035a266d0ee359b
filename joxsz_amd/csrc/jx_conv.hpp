// Hand-written beam + transfer-function convolution for gfx950, fp64, in three passes
// over a mixed (row, kx) domain -- the "custom" alternative to the rocFFT sequence.
//
// The reference computes (joxsz_funcs.py:464-467, 472)
//     conv = fftconvolve(y_2d, beam_2d, 'same') * step^2
//     map  = real(ifft2(fft2(conv) * filtering));   row = map[S//2, S//2:]
// With P = 2*LP >= S + (B-1)/2 the same numbers are obtained as
//   pass 1  Y[u][kx]  = rFFT_P(zero-padded map row urow[u])              (x direction)
//   pass 2  C[q][kx]  = sum_m tap[|r_q - m|][kx] * Y[umap[m]][kx]        (y direction: a
//           B-tap FIR whose taps are the x-transforms of the beam rows; for a beam image
//           symmetric under both flips -- the only kind mybeam builds, joxsz_funcs.py:46-76
//           -- the taps are real and symmetric in the row offset)
//   pass 3  conv[r_q][:] = irFFT_P(C[q]) cropped to S;  X[q][kc] = rFFT_S(conv[r_q]);
//           Zpart[kc] += X[q][kc] * Hyc[q][kc]                           (Hy: the transfer
//           function transformed back to real space along y at the offset of row S//2)
//   tail    row[x] = sum_kc Re(Z[kc] e^{2 pi i kc x / S})
// Row bookkeeping (built on the host, jx_finalize): when the map has the mirror structure
// y_2d[c+b] == y_2d[c-b] (d_mat from centdistmat) only the NU distinct map rows are
// transformed (urow/umap), and conv rows that are provably identical (same multiset of
// (source row, tap) pairs) are computed once ("jobs", r_q) with their Hy weights summed.
// Without the structure the tables are the identity and every row is its own job.
// x symmetry (xsym): a map row that is mirror-symmetric about column c = S/2, with its one unpaired
// column 0 (even S) set aside, has
//     Y(kx) = e^{-i phi} R(kx),  phi = 2 pi kx c / P,  R real.
// The FIR taps are real, so C(kx) = e^{-i phi} Rc(kx) with Rc = FIR(R) real: pass 1 stores R (and x0, the
// column-0 value of each distinct row), pass 2 filters the real array R (half the bytes and half the
// FMAs), pass 3 rebuilds Z from Rc with one precomputed complex factor per term.  What column 0 adds to
// the convolved row is put back in real space after the inverse transform: it only reaches output
// columns 0..o,  conv[r][x] += step^2 sum_t beam[o+t][o+x] (x0[r-t] + x0[r+t]).
// Default route: the weights Hyc[q][kc] factor as sum_rho U[rho][q] v_rho[kc] (truncated SVD built at
// jx_finalize), so the jobs are combined before pass 3, and pass 2 together with that combination is one real matrix per
// column kx applied to the walkers as columns: jx_rowfft2_kernel (tmode: walker-minor rows) -> jx_lowrank_kernel (fp64
// MFMA, batched over kx) -> jx_rowtf2_kernel over the r combined rows -> jx_tail_fft_kernel.  The kernels of the
// step-by-step description above remain as the route of the beam-convolved-map tap and of the fallbacks.
// In the step-by-step routes every pass keeps rows contiguous in memory ([row][kx]), so all global accesses are
// coalesced wave transactions; the default route trades that for walker-minor / rho-minor arrays where a GEMM wants them.
//
// FFTs: a real transform of length 2L is a complex transform of length L plus the usual
// even/odd split; the complex transform is two-level, L = L1 * L2, with both
// sub-transforms done in registers (jx_regfft.hpp).
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

struct JxConv {
    int S, Sh, B, o, P, Ph, LP, LS;
    int ntap;                       // o + 1
    int NU, NJ, nseg;               // distinct map rows, conv jobs, contiguous job segments
    int CROWS;                      // rows per walker in C: NJ + 1 (one spare row for store overshoot)
    int mirror;                     // 1: umap[m] = |m - S/2| (mirror structure), 0: umap[m] = m
    int xsym;                       // 1: map rows are also mirror-symmetric in x: Y and C hold ONE real array per row
    int fir_ld;                     // doubles per row of Y / C: Ph rounded up to 16 (xsym: rows start on cache lines) or 2 Ph
    // fused FIR + job combination (JxFused): pass 1 writes its rows walker-minor, pass 3 reads rho-minor
    int tmode;                      // 1: pass 1 -> Rt[k][tKU][tW] (block = distinct row x walker group); pass 3 <- Ct[w][Ph][64], Ct0[w][JX_CT0_X][64]
    int tW, tKU, tn;                // walker stride (multiple of 16), padded distinct-row count, walkers in this launch
    int kact;                       // columns kx < kact are kept: beyond, every beam tap is below 0.03 of the singular-value cut
    const double* ct0;              // [tW][JX_CT0_X][64] column-0 terms of the combined rows
    int quad;                       // 1: the map arrives as its quadrant [S/2+1][img_ld] of distinct pixels (|iy-c|, |ix-c|)
    const cplx* zab;                // [LP][2] pass-3 pre-process factors: Z[k] = zab[k][0] Rc[k] + zab[k][1] Rc[LP-k]   (xsym)
    const double* bcol;             // [o+1][JX_COL0_LD] step^2 beam[o+t][o+x], zero beyond x = o: what column 0 of a map row adds to output column x  (xsym)
    double* col0;                   // [walkers][o+1][NJ] what column 0 of the map adds to output columns 0..o of each job (xsym)
    int nblk3;                      // pass-3 blocks per walker (partials to sum in the tail)
    const cplx* tw_lp;              // [LP]   exp(-2 pi i n / LP)
    const cplx* tw_ls;              // [LS]   exp(-2 pi i n / LS)
    const cplx* tw_p;               // [LP+1] exp(-2 pi i k / P)
    const cplx* tw_s;               // [LS+1] exp(-2 pi i k / S)
    const double* taps;             // [ntap][Ph] real beam taps: tap[t][kx] multiplies rows r -+ t
    const cplx* hy;                 // [NJ][Sh] summed Hy weights of each job
    const int* urow;                // [NU]  map row transformed for distinct row u
    const int* umap;                // [S]   distinct-row index of map row m
    const int* jrow;                // [NJ]  conv row of job q (ascending)
    const int* seg;                 // [nseg][3] first conv row, number of rows, first job
};

#define JX_FIR_TILE 64              // output rows per chunk of pass 2
#define JX_FIR_NR 8                 // consecutive output rows per thread
#define JX_FIR_KX 32                // kx per slab
// beam half-widths o = (B-1)/2 with a register-window FIR instance (and an x-symmetric pass-3 pre-process)
#define JX_XSYM_MAXT 33              // taps (o + 1) of the widest beam the path takes ((B-1)/2 <= 32)
#define JX_COL0_XG 7                // output columns per block of the column-0 kernel
#define JX_COL0_LD 35               // row stride of its beam table (a multiple of JX_COL0_XG, >= JX_XSYM_MAXT)
#define JX_CT0_X 40                 // output columns per walker in the rho-minor column-0 array Ct0[w][JX_CT0_X][64]
#define JX_FIR_REG_O(X) X(4) X(5) X(13) X(27)          // all >= 4: the FIR's 8 accumulator chains need 2O+1 >= 8
#define JX_FIR_RING 128             // LDS ring of input rows (>= JX_FIR_TILE + 2 o)

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
__device__ __forceinline__ jx_cT<float> jx_ld(const float2* p) { const float2 v = *p; return jxcT<float>(v.x, v.y); }
__device__ __forceinline__ void jx_st(float2* p, jx_cT<float> v) { *p = make_float2(v.x, v.y); }
// LDS / memory pair type of a scalar type
template <typename T> struct jx_pair;
template <> struct jx_pair<double> { typedef double2 type; };
template <> struct jx_pair<float> { typedef float2 type; };

template <int L> struct jx_lay {
    static constexpr int L1 = jx_plan2<L>::L1, L2 = jx_plan2<L>::L2;
    static constexpr int L2P = L2 | 1;                     // padded inner stride (odd)
    static constexpr int SPAN = (L1 * L2P > L + 1 ? L1 * L2P : L + 1);
    static constexpr int RS = SPAN | 1;                    // row stride (odd), >= L + 1
    static constexpr int TMAX = L1 > L2 ? L1 : L2;
};

// step A on registers already loaded: x[n1] -> FFT over n1, twiddle, store to padded layout
template <int L, bool INV, typename T, typename T2>
__device__ __forceinline__ void jx_stepA_store(jx_cT<T>* x, int n2, T2* Mrow, const T2* tw) {
    constexpr int L1 = jx_lay<L>::L1, L2P = jx_lay<L>::L2P;
    jx_regfft<L1, INV>::run(x);
#pragma unroll
    for (int k1 = 0; k1 < L1; ++k1) {
        jx_cT<T> v = x[k1];
        if (k1 > 0) {
            const T2 w = tw[n2 * k1];                       // e^{-2 pi i n2 k1 / L}
            v = INV ? jxcT<T>(v.x * w.x + v.y * w.y, v.y * w.x - v.x * w.y) : jxcT<T>(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
        }
        jx_st(Mrow + k1 * L2P + n2, v);
    }
}

// step B: read the L2 samples of (row, k1) from the padded layout, transform, leave in y
template <int L, bool INV, typename T, typename T2>
__device__ __forceinline__ void jx_stepB_load(jx_cT<T>* y, int k1, const T2* Mrow) {
    constexpr int L2 = jx_lay<L>::L2, L2P = jx_lay<L>::L2P;
#pragma unroll
    for (int n2 = 0; n2 < L2; ++n2) y[n2] = jx_ld(Mrow + k1 * L2P + n2);
    jx_regfft<L2, INV>::run(y);
}

// ------------------------------------------------------------------------------------
// pass 1 (two-level): grid = (ceil(S / ROWS), walkers), 256 threads, ROWS = 256 / max(L1, L2)
// ------------------------------------------------------------------------------------
template <int LP, int ROWS>
// (LDS admits two blocks per CU = two waves per SIMD: registers are not the occupancy limit)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
jx_rowfft2_kernel(JxConv c, const double* __restrict__ img, size_t img_ld, size_t img_ws, cplx* __restrict__ Y) {
    typedef jx_lay<LP> Lay;
    constexpr int L1 = Lay::L1, L2 = Lay::L2, RS = Lay::RS;
    static_assert(ROWS * Lay::TMAX <= 256, "one task per thread");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    cplx* M = reinterpret_cast<cplx*>(sm);                  // [ROWS][RS]
    cplx* tw = M + ROWS * RS;                               // [LP]     e^{-2 pi i n / LP}
    cplx* twp = tw + LP;                                    // [LP + 1] e^{-2 pi i k / P}
    double* s_x0 = reinterpret_cast<double*>(twp + LP + 1);  // [ROWS] unpaired column 0 of each row (xsym)
    const int tid = threadIdx.x, nth = blockDim.x;
    // normal: block = (ROWS distinct rows from r0, walker w).  tmode: block = (distinct row r0, ROWS walkers from w).
    const int r0 = c.tmode ? blockIdx.x : blockIdx.x * ROWS, w = c.tmode ? blockIdx.y * ROWS : blockIdx.y;
    const int nrows = c.tmode ? min(ROWS, c.tn - w) : min(ROWS, c.NU - r0), half = c.S / 2;
    const size_t row_ws = c.tmode ? img_ws : 0, row_ld = c.tmode ? 0 : img_ld;       // what one step in `row` adds to the source address
    // twiddle tables: requested now, stored behind the row loads below (one round trip for everything)
    constexpr int NTW = (LP + 1 + 255) / 256;
    double tw_r[NTW][4];                                             // (scalars: an aggregate copy out of a register array ends up in scratch)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const cplx a = c.tw_lp[min(tid + i * 256, LP - 1)], b = c.tw_p[min(tid + i * 256, LP)];
        tw_r[i][0] = a.x; tw_r[i][1] = a.y; tw_r[i][2] = b.x; tw_r[i][3] = b.y;
    }

    const int rowA = tid / L2, n2 = tid - rowA * L2;
    const bool actA = tid < ROWS * L2 && rowA < nrows;
    jx_c x[L1];
    if (c.quad) {
        // quadrant row u = r0 + row holds q[a] = map[.][c +- a], a = 0..c (c = S/2, even).  Staged through the row's own
        // LDS slot with coalesced 16-byte loads; columns (2n, 2n+1) are (q[c-2n], q[c-2n-1]) left of the centre and
        // (q[2n-c], q[2n-c+1]) from it on.
        const int npair = (half >> 1) + 1;                            // q[0..c] in pairs (the spare column rides along)
        const double* qb = img + (size_t)w * img_ws + (size_t)r0 * img_ld;     // + row * (row_ws + row_ld)
        constexpr int NIT = (ROWS * (LP / 2 + 1) + 255) / 256;         // npair <= LP/2 + 1, blockDim = 256
        const int tot = nrows * npair;
        double2 stg[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {                               // every request before the first use
            const int e = min(tid + i * nth, tot - 1), row = e / npair, j = e - row * npair;
            stg[i] = *reinterpret_cast<const double2*>(qb + (size_t)row * (row_ws + row_ld) + 2 * j);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int e = tid + i * nth, row = e / npair, j = e - row * npair;
            // odd rows sit one double further: the 8-byte gathers of neighbouring rows then fall on different banks
            if (e < tot) {
                double* dst = reinterpret_cast<double*>(M + row * RS) + (row & 1) + 2 * j;
                dst[0] = stg[i].x; dst[1] = stg[i].y;
            }
        }
        __syncthreads();
        if (actA) {
            const double* q = reinterpret_cast<const double*>(M + rowA * RS) + (rowA & 1);
#pragma unroll
            for (int n1 = 0; n1 < L1; ++n1) {
                const int n = n1 * L2 + n2;
                x[n1] = jxc(0.0, 0.0);
                if (n < half) x[n1] = (2 * n < half) ? jxc(q[half - 2 * n], q[half - 2 * n - 1]) : jxc(q[2 * n - half], q[2 * n - half + 1]);
            }
        }
    } else if (actA) {
        const int u = c.tmode ? r0 : r0 + rowA;                     // == c.urow[u] without the dependent load
        const int mrow = c.mirror ? ((half + u < c.S) ? half + u : half - u) : u;
        const double* src = img + (size_t)(c.tmode ? w + rowA : w) * img_ws + (size_t)mrow * img_ld;
#pragma unroll
        for (int n1 = 0; n1 < L1; ++n1) {
            const int n = n1 * L2 + n2;
            x[n1] = jxc(0.0, 0.0);
            if (n < half) { const double2 v = *reinterpret_cast<const double2*>(src + 2 * n); x[n1] = jxc(v.x, v.y); }
        }
    }
    if (actA) {
        if (n2 == 0) s_x0[rowA] = (c.S & 1) ? 0.0 : x[0].x;           // column 0 has no mirror partner on an even side
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int n = tid + i * 256;
        if (n < LP) tw[n] = make_double2(tw_r[i][0], tw_r[i][1]);
        if (n <= LP) twp[n] = make_double2(tw_r[i][2], tw_r[i][3]);
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
    if (c.tmode) {
        // walker-minor output Rt[k][u][w]: consecutive lanes take consecutive walkers of the same k
        double* Rt = reinterpret_cast<double*>(Y);
        for (int e = tid; e < nrows * c.kact; e += nth) {             // (columns past the beam's band limit are never read)
            const int k = e / nrows, row = e - k * nrows;
            const cplx zk = M[row * RS + (k == LP ? 0 : k)];
            const cplx zc = c_conj(M[row * RS + (k == 0 ? 0 : LP - k)]);
            const cplx s = c_add(zk, zc), d = c_mul(c_sub(zk, zc), twp[k]);
            const cplx X = make_double2(0.5 * (s.x + d.y), 0.5 * (s.y - d.x));
            const int j = (k * (c.S >> 1)) % c.P;
            const cplx t = twp[j <= LP ? j : c.P - j];
            const double sn = j <= LP ? -t.y : t.y;
            Rt[((size_t)k * c.tKU + r0) * c.tW + w + row] = (X.x - s_x0[row]) * t.x - X.y * sn;
        }
        return;
    }
    for (int e = tid; e < nrows * Ph; e += nth) {
        const int row = e / Ph, k = e - row * Ph;
        const cplx zk = M[row * RS + (k == LP ? 0 : k)];
        const cplx zc = c_conj(M[row * RS + (k == 0 ? 0 : LP - k)]);
        const cplx s = c_add(zk, zc), d = c_mul(c_sub(zk, zc), twp[k]);
        const cplx X = make_double2(0.5 * (s.x + d.y), 0.5 * (s.y - d.x));
        if (c.xsym) {                                                 // R = Re[(X - x0) e^{+i phi}]
            const int j = (k * (c.S >> 1)) % c.P;                     // e^{+i phi} = conj(tw_P[j]) = tw_P[P - j]
            const cplx t = twp[j <= LP ? j : c.P - j];
            const double sn = j <= LP ? -t.y : t.y;
            reinterpret_cast<double*>(Y)[((size_t)w * c.NU + r0 + row) * c.fir_ld + k] = (X.x - s_x0[row]) * t.x - X.y * sn;
        } else {
            Y[((size_t)w * c.NU + r0 + row) * Ph + k] = X;
        }
    }
}

// ------------------------------------------------------------------------------------
// pass 3 (two-level): grid = (ceil(S / ROWS), walkers), 256 threads
// ------------------------------------------------------------------------------------
// x-symmetric mode: the unpaired column 0 of the map, convolved with the beam, reaches output columns 0..o only:
//     col0[w][x][q] = sum_t bcol[t][x] (x0[r_q - t] + x0[r_q + t]),   x0[m] = map[m][0],  r_q = conv row of job q.
// grid (walkers, ceil((o+1)/7)); pass 3 adds the result to the cropped rows in real space.
__global__ void __launch_bounds__(256)
jx_col0_kernel(JxConv c, const double* __restrict__ img, size_t img_ld, size_t img_ws, const double* __restrict__ xcol,
               double* __restrict__ col0) {
    constexpr int XG = JX_COL0_XG;                                   // output columns per block
    __shared__ double s_x0[1024 + 2 * JX_XSYM_MAXT];                 // zero margin of o rows on both sides
    const int tid = threadIdx.x, nth = blockDim.x, w = blockIdx.x, S = c.S, o = c.o, nt = o + 1, NJ = c.NJ;
    for (int m = tid; m < S + 2 * o; m += nth) {
        const int r = m - o;
        s_x0[m] = (r >= 0 && r < S) ? (c.quad ? xcol[(size_t)w * ((S >> 1) + 1) + abs(r - (S >> 1))]      // the map kernel's compact copy
                                               : img[(size_t)w * img_ws + (size_t)r * img_ld]) : 0.0;
    }
    __syncthreads();
    // consecutive lanes take consecutive jobs; the beam values are block-uniform (scalar loads, no LDS traffic)
    const double* bq = c.bcol + blockIdx.y * XG;                      // [o+1][JX_COL0_LD], zero beyond column o
    const int xb = blockIdx.y * XG;
    for (int q = tid; q < NJ; q += nth) {
        const int r = c.jrow[q] + o;
        double a[XG];
        const double s0 = s_x0[r];
#pragma unroll
        for (int j = 0; j < XG; ++j) a[j] = bq[j] * s0;
#pragma unroll 4
        for (int t = 1; t < nt; ++t) {
            const double sv = s_x0[r - t] + s_x0[r + t];
#pragma unroll
            for (int j = 0; j < XG; ++j) a[j] = fma(bq[t * JX_COL0_LD + j], sv, a[j]);
        }
#pragma unroll
        for (int j = 0; j < XG; ++j)
            if (xb + j < nt) col0[((size_t)w * nt + xb + j) * NJ + q] = a[j];             // [x][q]: coalesced over the jobs
    }
}

// TC: element type of the combined rows Ct / Ct0 in memory on the fused route (float in the fp32 variant; widened on load)
template <int LP, int LS, int ROWS, typename TC = double>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
jx_rowtf2_kernel(JxConv c, const cplx* __restrict__ C, cplx* __restrict__ part, double* __restrict__ tap_conv) {
    typedef jx_lay<LP> LayP;
    typedef jx_lay<LS> LayS;
    constexpr int P1 = LayP::L1, P2 = LayP::L2, P2P = LayP::L2P;
    constexpr int S1 = LayS::L1, S2 = LayS::L2, S2P = LayS::L2P;
    constexpr int RS = LayP::RS > LayS::RS ? LayP::RS : LayS::RS;
    static_assert(ROWS * LayP::TMAX <= 256 && ROWS * LayS::TMAX <= 256, "one task per thread");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    cplx* M = reinterpret_cast<cplx*>(sm);                  // [ROWS][RS]
    cplx* twp = M + ROWS * RS;                              // [LP]     e^{-2 pi i n / LP}
    cplx* tws = twp + LP;                                   // [LS]     e^{-2 pi i n / LS}
    const int tid = threadIdx.x, nth = blockDim.x;
    const int r0 = blockIdx.x * ROWS, w = blockIdx.y;
    const int S = c.S, Ph = c.Ph, Sh = c.Sh, NJ = c.NJ;             // r0: first job q of the block
    const int nrows = min(ROWS, NJ - r0);
    constexpr int NTW = ((LP > LS ? LP : LS) + 255) / 256;           // twiddle tables: requested now, stored behind the row loads
    double tw_r[NTW][4];                                             // (scalars: an aggregate copy out of a register array ends up in scratch)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const cplx a = c.tw_lp[min(tid + i * 256, LP - 1)], b = c.tw_ls[min(tid + i * 256, LS - 1)];
        tw_r[i][0] = a.x; tw_r[i][1] = a.y; tw_r[i][2] = b.x; tw_r[i][3] = b.y;
    }
    // Z[k] = (X[k] + conj X[LP-k]) + i e^{+2 pi i k/P} (X[k] - conj X[LP-k]) into the padded layout of n = k.
    // Four elements per thread and trip: their eight row reads are requested before anything is used.
    const cplx* Cblk = C + ((size_t)w * c.CROWS + r0) * Ph;
    const int npre = nrows * LP;
    double* s_s = reinterpret_cast<double*>(tws + LS);               // [ROWS][o + 1] column-0 terms of the block's jobs (xsym)
    constexpr int NSL = (ROWS * JX_XSYM_MAXT + 255) / 256;          // entries of s_s per thread (blockDim = 256)
    if (c.xsym && c.tmode) {
        // combined rows arrive rho-minor: Ct[w][k][64], the block's rows r0.. are 16-byte pairs of that last axis.
        // Work item = (pair (k, LP-k), two rows): NSUB consecutive lanes read the block's rows of one k (a 128-byte line for
        // 14 rows).  All trips' loads first.
        typedef typename jx_pair<TC>::type TC2;
        const TC* Cw = reinterpret_cast<const TC*>(C) + (size_t)w * Ph * 64 + r0;
        const TC* ct0 = reinterpret_cast<const TC*>(c.ct0);
        const int nt = c.o + 1, ns = nrows * nt;
        TC cz[NSL];                                                   // (storage type: widened where it is used, behind all the requests)
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const int e = min(tid + u * nth, ns - 1), row = e / nt, xo = e - row * nt;
            cz[u] = ct0[((size_t)w * JX_CT0_X + xo) * 64 + r0 + row];
        }
        constexpr int HR = (ROWS + 1) / 2;                            // row pairs
        constexpr int NSUB = HR <= 4 ? 4 : (HR <= 8 ? 8 : (HR <= 16 ? 16 : 32)), LSUB = NSUB == 4 ? 2 : (NSUB == 8 ? 3 : (NSUB == 16 ? 4 : 5));
        static_assert(HR <= 32, "row pairs of a block share one group of lanes");
        constexpr int NPAIR = LP / 2 + 1, NTRIP = (NPAIR * NSUB + 255) / 256;
        TC2 ra[NTRIP], rb[NTRIP];
#pragma unroll
        for (int i = 0; i < NTRIP; ++i) {
            const int e = min(tid + i * 256, NPAIR * NSUB - 1), k = e >> LSUB, sub = min(e & (NSUB - 1), HR - 1);
            TC2 va, vb;
            va.x = va.y = vb.x = vb.y = 0;
            if (k < c.kact) va = *reinterpret_cast<const TC2*>(Cw + (size_t)k * 64 + 2 * sub);
            if (LP - k < c.kact) vb = *reinterpret_cast<const TC2*>(Cw + (size_t)(LP - k) * 64 + 2 * sub);
            ra[i] = va;
            rb[i] = vb;
        }
#pragma unroll
        for (int i = 0; i < NTRIP; ++i) {
            const int e = tid + i * 256, k = e >> LSUB, sub = e & (NSUB - 1);
            if (e < NPAIR * NSUB && 2 * sub < nrows) {
                const int kp = LP - k, kq = kp == LP ? 0 : kp;
                const cplx za1 = c.zab[2 * k], zb1 = c.zab[2 * k + 1], za2 = c.zab[2 * kq], zb2 = c.zab[2 * kq + 1];
                const int n1a = k / P2, n2a = k - n1a * P2, n1b = kq / P2, n2b = kq - n1b * P2;
                const double r1[2] = {(double)ra[i].x, (double)ra[i].y}, r2[2] = {(double)rb[i].x, (double)rb[i].y};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 2 * sub + h;
                    if (row < nrows) {
                        M[row * RS + n1a * P2P + n2a] = make_double2(fma(za1.x, r1[h], zb1.x * r2[h]), fma(za1.y, r1[h], zb1.y * r2[h]));
                        if (kp != k && kp != LP)
                            M[row * RS + n1b * P2P + n2b] = make_double2(fma(za2.x, r2[h], zb2.x * r1[h]), fma(za2.y, r2[h], zb2.y * r1[h]));
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NSL; ++u) if (tid + u * nth < ns) s_s[tid + u * nth] = (double)cz[u];
    } else if (c.xsym) {
        // Z[k] = zab[k][0] Rc[k] + zab[k][1] Rc[LP-k], four elements per thread and trip as below
        const int ldr = c.fir_ld;                                    // row stride of the real arrays (Ph rounded up to whole cache lines)
        const double* Rblk = reinterpret_cast<const double*>(C) + ((size_t)w * c.CROWS + r0) * ldr;
        const int nt = c.o + 1, ns = nrows * nt;
        double cz[NSL];                                              // this block's slice of col0, requested first
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const int e = min(tid + u * nth, ns - 1), row = e / nt, xo = e - row * nt;
            cz[u] = c.col0[((size_t)w * nt + xo) * NJ + r0 + row];
        }
        // one thread per pair (k, LP-k): the same two reals of every row make Z[k] and Z[LP-k]; all rows requested at once
        for (int k = tid; 2 * k <= LP; k += nth) {
            const int kp = LP - k, kq = kp == LP ? 0 : kp;
            double r1[ROWS], r2[ROWS];
#pragma unroll
            for (int row = 0; row < ROWS; ++row) {
                const int rr = min(row, nrows - 1);
                r1[row] = Rblk[(size_t)rr * ldr + k];
                r2[row] = Rblk[(size_t)rr * ldr + kp];
            }
            const cplx za1 = c.zab[2 * k], zb1 = c.zab[2 * k + 1], za2 = c.zab[2 * kq], zb2 = c.zab[2 * kq + 1];
            const int n1a = k / P2, n2a = k - n1a * P2, n1b = kq / P2, n2b = kq - n1b * P2;
#pragma unroll
            for (int row = 0; row < ROWS; ++row) {
                if (row < nrows) {
                    M[row * RS + n1a * P2P + n2a] = make_double2(fma(za1.x, r1[row], zb1.x * r2[row]), fma(za1.y, r1[row], zb1.y * r2[row]));
                    if (kp != k && kp != LP)
                        M[row * RS + n1b * P2P + n2b] = make_double2(fma(za2.x, r2[row], zb2.x * r1[row]), fma(za2.y, r2[row], zb2.y * r1[row]));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NSL; ++u) if (tid + u * nth < ns) s_s[tid + u * nth] = cz[u];
    } else
    // (the e^{-2 pi i k/P} factors come from global memory with the same batch: an LDS copy would push
    // the block past 80 KB and halve the residency)
    for (int e0 = tid; e0 < npre; e0 += 4 * nth) {
        cplx xk[4], xc[4], tp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + u * nth, npre - 1), row = e / LP, k = e - row * LP;
            xk[u] = Cblk[(size_t)row * Ph + k];
            xc[u] = Cblk[(size_t)row * Ph + LP - k];
            tp[u] = c.tw_p[k];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * nth;
            if (e < npre) {
                const int row = e / LP, k = e - row * LP;
                const cplx xcc = c_conj(xc[u]);
                const cplx s = c_add(xk[u], xcc), d = c_mulc(c_sub(xk[u], xcc), tp[u]);
                const int n1 = k / P2, n2 = k - n1 * P2;
                M[row * RS + n1 * P2P + n2] = make_double2(s.x - d.y, s.y + d.x);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int n = tid + i * 256;
        if (n < LP) twp[n] = make_double2(tw_r[i][0], tw_r[i][1]);
        if (n < LS) tws[n] = make_double2(tw_r[i][2], tw_r[i][3]);
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
        // z[n] = (conv[2n], conv[2n+1]), n = k1 + P1 k2; the first LS of them are the cropped row
        if constexpr (P1 == S2 && P2 >= S1) {
            // The outputs this thread holds, n = k1 + S2 k2 (k2 < S1), are exactly the inputs of step A of the forward
            // transform for (row, n2 = k1): no LDS round trip and no barrier between the two transforms.
            __syncthreads();                                        // every thread has read its step-B inputs
            if (actB) {
                if (c.xsym) {                                        // column 0 reaches output columns 0..o only
                    const int o = c.o;
#pragma unroll
                    for (int k2 = 0; k2 < P2; ++k2) {
                        const int n = k1 + P1 * k2;
                        if (2 * n <= o) {
                            y[k2].x += s_s[rowB * (o + 1) + 2 * n];
                            if (2 * n + 1 <= o) y[k2].y += s_s[rowB * (o + 1) + 2 * n + 1];
                        }
                    }
                }
                if (tap_conv) {
#pragma unroll
                    for (int k2 = 0; k2 < P2; ++k2) {
                        const int n = k1 + P1 * k2;
                        if (n < LS)
                            *reinterpret_cast<double2*>(tap_conv + ((size_t)w * NJ + r0 + rowB) * S + 2 * n) = make_double2(y[k2].x, y[k2].y);
                    }
                }
            }
            if (actB) jx_stepA_store<LS, false>(y, k1, M + rowB * RS, tws);
            __syncthreads();
        } else {
            __syncthreads();
            if (actB) {
                if (c.xsym) {                                        // column 0 reaches output columns 0..o only
                    const int o = c.o;
#pragma unroll
                    for (int k2 = 0; k2 < P2; ++k2) {
                        const int n = k1 + P1 * k2;
                        if (2 * n <= o) {
                            y[k2].x += s_s[rowB * (o + 1) + 2 * n];
                            if (2 * n + 1 <= o) y[k2].y += s_s[rowB * (o + 1) + 2 * n + 1];
                        }
                    }
                }
                if (tap_conv) {
#pragma unroll
                    for (int k2 = 0; k2 < P2; ++k2) {
                        const int n = k1 + P1 * k2;
                        if (n < LS)
                            *reinterpret_cast<double2*>(tap_conv + ((size_t)w * NJ + r0 + rowB) * S + 2 * n) = make_double2(y[k2].x, y[k2].y);
                    }
                }
            }
            if (actB) {
#pragma unroll
                for (int k2 = 0; k2 < P2; ++k2) {
                    const int n = k1 + P1 * k2;
                    if (n < LS) { const int a2 = n / S2, b2 = n - a2 * S2; jx_st(M + rowB * RS + a2 * S2P + b2, y[k2]); }
                }
            }
            __syncthreads();
            const int rowA2 = tid / S2, n2b = tid - rowA2 * S2;
            const bool actA2 = tid < ROWS * S2 && rowA2 < nrows;
            jx_c x[S1];
            if (actA2) {
#pragma unroll
                for (int n1 = 0; n1 < S1; ++n1) x[n1] = jx_ld(M + rowA2 * RS + n1 * S2P + n2b);
                jx_stepA_store<LS, false>(x, n2b, M + rowA2 * RS, tws);
            }
            __syncthreads();
        }
    }
    {   // forward transform of length LS, step B
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
    {
        // Hy-weighted sum over the block's jobs, k = tid + 256 j.  The weights of every k of this thread are requested
        // before the first use (k = LS alone would otherwise cost a second round trip).
        constexpr int NKT = (LS + 1 + 255) / 256;
        cplx h[NKT][ROWS], tk[NKT];
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            const int k = min(tid + j * nth, Sh - 1);
            tk[j] = c.tw_s[k];
#pragma unroll
            for (int row = 0; row < ROWS; ++row) h[j][row] = c.hy[(size_t)min(r0 + row, NJ - 1) * Sh + k];
        }
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            const int k = tid + j * nth;
            if (k < Sh) {
                double zr = 0.0, zi = 0.0;
#pragma unroll
                for (int row = 0; row < ROWS; ++row) {
                    if (row < nrows) {
                        const cplx zk = M[row * RS + (k == LS ? 0 : k)];
                        const cplx zc = c_conj(M[row * RS + (k == 0 ? 0 : LS - k)]);
                        const cplx s = c_add(zk, zc), d = c_mul(c_sub(zk, zc), tk[j]);
                        const cplx x = make_double2(0.5 * (s.x + d.y), 0.5 * (s.y - d.x));
                        zr += x.x * h[j][row].x - x.y * h[j][row].y;
                        zi += x.x * h[j][row].y + x.y * h[j][row].x;
                    }
                }
                part[((size_t)w * c.nblk3 + blockIdx.x) * Sh + k] = make_double2(zr, zi);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// pass 2: FIR along rows, streaming.  grid = (ceil(Ph/32), walkers), 256 threads: lane & 31
// = kx in the slab, tid >> 5 = group of JX_FIR_NR consecutive output rows.  A block owns
// its kx slab for ALL conv jobs of the walker: input rows go through a 128-row LDS ring,
// each is fetched from global memory exactly once per segment (no halo re-reads), outputs
// are produced JX_FIR_TILE rows at a time.  Each thread slides over 2o + NR inputs with NR
// accumulators; the tap window lives in registers whose NAMES rotate (slot of logical tap j
// at step i is (j - i) mod NR), so the unrolled body has no moves, and the LDS reads of step
// i+1 are issued before the FMAs of step i.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
jx_beamfir_kernel(JxConv c, const cplx* __restrict__ Y, cplx* __restrict__ C) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int o = c.o, Ph = c.Ph, S = c.S;
    cplx* ring = reinterpret_cast<cplx*>(sm);                         // [128][32]
    double* taps = sm + (size_t)2 * JX_FIR_RING * JX_FIR_KX;          // [o+1][32]
    int* s_umap = reinterpret_cast<int*>(taps + (size_t)(o + 1) * JX_FIR_KX);   // [S]
    const int tid = threadIdx.x;
    const int lx = tid & 31, grp = tid >> 5;
    const int kx = blockIdx.x * JX_FIR_KX + lx, w = blockIdx.y;
    const bool kok = kx < Ph;
    const cplx* Yw = Y + (size_t)w * c.NU * Ph + kx;
    cplx* Cw = C + (size_t)w * c.CROWS * Ph + kx;
    for (int t = grp; t <= o; t += 8) taps[t * JX_FIR_KX + lx] = kok ? c.taps[(size_t)t * Ph + kx] : 0.0;
    for (int m = tid; m < S; m += 256) s_umap[m] = c.umap[m];
    const int nin = JX_FIR_TILE + 2 * o;
    constexpr int NPRE = JX_FIR_TILE / 8;                             // rows each group fetches per chunk
    const cplx zero = make_double2(0.0, 0.0);

    for (int sg = 0; sg < c.nseg; ++sg) {
        const int ra = c.seg[3 * sg], cnt = c.seg[3 * sg + 1], qa = c.seg[3 * sg + 2];
        const int base = ra - o;                                      // map row held by ring slot 0 (mod 128)
        __syncthreads();                                              // previous segment consumed; s_umap ready
        // prime the ring with the first nin rows: two batches of independent loads
        for (int rr0 = 0; rr0 < nin; rr0 += 8 * NPRE) {
            cplx pre[NPRE];
#pragma unroll
            for (int u = 0; u < NPRE; ++u) {
                const int rr = rr0 + grp + 8 * u, m = base + rr;
                pre[u] = (kok && rr < nin && m >= 0 && m < S) ? Yw[(size_t)s_umap[m] * Ph] : zero;
            }
#pragma unroll
            for (int u = 0; u < NPRE; ++u) {
                const int rr = rr0 + grp + 8 * u;
                if (rr < nin) ring[(rr & (JX_FIR_RING - 1)) * JX_FIR_KX + lx] = pre[u];
            }
        }
        __syncthreads();
        for (int t0 = 0; t0 < cnt; t0 += JX_FIR_TILE) {               // t0: first output of the chunk, segment-relative
            const bool more = t0 + JX_FIR_TILE < cnt;
            // the next chunk's 64 input rows are requested now and land in registers while this chunk computes
            cplx pre[NPRE];
            if (more) {
#pragma unroll
                for (int u = 0; u < NPRE; ++u) {
                    const int m = base + t0 + nin + grp + 8 * u;
                    pre[u] = (kok && m >= 0 && m < S) ? Yw[(size_t)s_umap[m] * Ph] : zero;
                }
            }
            const int rb = t0 + grp * JX_FIR_NR;                      // this thread's first output, segment-relative
            if (rb < cnt) {
                double ar[JX_FIR_NR], ai[JX_FIR_NR], tp[JX_FIR_NR];
#pragma unroll
                for (int j = 0; j < JX_FIR_NR; ++j) { ar[j] = 0.0; ai[j] = 0.0; tp[j] = 0.0; }
                const int nstep = 2 * o + JX_FIR_NR;
                // input i of this thread is map row ra + rb - o + i = base + rb + i -> ring slot (rb + i) & 127
                cplx vnext = ring[(rb & (JX_FIR_RING - 1)) * JX_FIR_KX + lx];
                double tnext = taps[o * JX_FIR_KX + lx];
                for (int i0 = 0; i0 < nstep; i0 += JX_FIR_NR) {
#pragma unroll
                    for (int s = 0; s < JX_FIR_NR; ++s) {
                        const int i = i0 + s;
                        if (i < nstep) {
                            const cplx v = vnext;
                            tp[(JX_FIR_NR - s) % JX_FIR_NR] = tnext;   // logical j = 0 at step i
                            if (i + 1 < nstep) {
                                const int d = abs(i + 1 - o);
                                vnext = ring[((rb + i + 1) & (JX_FIR_RING - 1)) * JX_FIR_KX + lx];
                                tnext = (d <= o) ? taps[d * JX_FIR_KX + lx] : 0.0;
                            }
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
                    for (int j = 0; j < JX_FIR_NR; ++j)
                        if (rb + j < cnt) Cw[(size_t)(qa + rb + j) * Ph] = make_double2(ar[j], ai[j]);
                }
            }
            if (more) {
                __syncthreads();                                      // everyone done reading the rows about to be replaced
#pragma unroll
                for (int u = 0; u < NPRE; ++u) {
                    const int off = t0 + nin + grp + 8 * u;
                    ring[(off & (JX_FIR_RING - 1)) * JX_FIR_KX + lx] = pre[u];
                }
                __syncthreads();
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// pass 2, register form (beam half-width O known at compile time).  The taps are real, so
// the real and imaginary parts of a spectrum column are two independent real series: the
// [row][kx] complex arrays are read as [row][2 Ph] doubles and one lane owns one such
// column.  A wave = 64 adjacent columns of one run of consecutive conv jobs.  The 2O+1 most
// recent inputs live in registers (a window whose slot NAMES rotate with the unrolled phase,
// so nothing is ever moved), the O+1 taps of the lane's kx too; each output costs one
// coalesced 512-byte row load, requested DEPTH rows ahead, and 2O+1 FMAs.  No LDS traffic in
// the loop, no barrier.  1-D grid of ceil(slabs * walkers / 8) * 8 * runs workgroups of 64 threads.
//   runs [nrun][3]: first conv row, number of rows, first job
// ------------------------------------------------------------------------------------
#ifndef JX_FIR_MIN_DEPTH
#define JX_FIR_MIN_DEPTH 8
#endif
template <int W> constexpr int jx_fir_depth() {
    for (int d = JX_FIR_MIN_DEPTH; d < W; ++d) if (W % d == 0) return d;      // rows requested ahead (must divide W)
    return W;
}

template <int O, int PH>
struct jx_fir_phase {
    static constexpr int W = 2 * O + 1;
    // one output: the newest input (map row r + O) enters slot PH; input i = 0..2O sits in slot (PH + 1 + i) % W
    template <int I>
    static __device__ __forceinline__ void acc(const double (&win)[W], const double (&tap)[O + 1], double (&a)[8]) {
        if constexpr (I < W) {
            constexpr int slot = (PH + 1 + I) % W;
            constexpr int t = I < O ? O - I : I - O;
            if constexpr (I < 8) a[I] = tap[t] * win[slot];           // the 8 chains start as products: no zero fill
            else a[I & 7] = fma(tap[t], win[slot], a[I & 7]);
            acc<I + 1>(win, tap, a);
        }
    }
};

// One batch of W = 2O+1 phases, STRAIGHT-LINE: no branch at all, so that hipcc can keep counted
// s_waitcnt vmcnt(N) between the row requests and their use (with a branch per phase it falls
// back to vmcnt(0) everywhere, which serialises every phase behind its own store and request).
// Out-of-range rows are read from a clamped row and multiplied away; phases past the end of the
// run store into the walker's spare row (index NJ) of C.
template <int O, int PH>
__device__ __forceinline__ void jx_fir_run_phases(double (&win)[2 * O + 1], double (&fifo)[jx_fir_depth<2 * O + 1>()],
                                                  const double (&tap)[O + 1], int mirror, const double* Yw,
                                                  double* Cw, int ld, int S, int NJ, int r /*conv row of phase 0*/,
                                                  int q /*job of phase 0*/, int nleft) {
    constexpr int W = 2 * O + 1, D = jx_fir_depth<W>();
    if constexpr (PH < W) {
        // map row r + PH + O was requested D phases ago (raw, from a clamped row); it is masked only now,
        // at its first use, so that nothing waits on a request right after issuing it
        const int mu = r + PH + O;
        win[PH] = fifo[PH % D] * ((mu >= 0 && mu < S) ? 1.0 : 0.0);
        const int mc = min(max(mu + D, 0), S - 1);                    // wave-uniform: scalar path
        const int uc = mirror ? abs(mc - (S >> 1)) : mc;              // distinct-row index (== umap[mc]), SALU only
        fifo[PH % D] = Yw[(size_t)uc * ld];                           // the row needed D phases from now
        double a[8];
        jx_fir_phase<O, PH>::template acc<0>(win, tap, a);
        const int row = (PH < nleft) ? q + PH : NJ;                   // spare row swallows the overshoot
        Cw[(size_t)row * ld] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        __builtin_amdgcn_sched_barrier(0);                            // keep each phase's request where it is: hoisting
                                                                      // them in groups costs 130 more registers
        jx_fir_run_phases<O, PH + 1>(win, fifo, tap, mirror, Yw, Cw, ld, S, NJ, r, q, nleft);
    }
}

template <int O>
__global__ void __launch_bounds__(64)
jx_beamfir_reg_kernel(JxConv c, const int* __restrict__ runs, int nrun, int nwalk, const cplx* __restrict__ Y,
                      cplx* __restrict__ C) {
    constexpr int W = 2 * O + 1, D = jx_fir_depth<W>();
    static_assert(W >= 8, "the accumulator chains assume at least 8 taps");
    const int lane = threadIdx.x, S = c.S, ld = c.fir_ld;             // Ph real columns (xsym) or 2 Ph (re, im interleaved)
    // XCD-aware decode of the 1-D grid: workgroups are dealt round-robin over the 8 XCDs, so ids that agree
    // mod 8 share an L2.  All runs of one (walker, column slab) unit get the same id mod 8 and consecutive
    // id / 8, so the 2O halo rows two neighbouring runs both read are served by that L2 (speed only).
    const int nslab = (ld + 63) / 64;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int unit = (seq / nrun) * 8 + xcd, run = seq - (seq / nrun) * nrun;
    if (unit >= nslab * nwalk) return;
    const int w = unit / nslab, slab = unit - w * nslab;
    // lanes past the last column redo the last column (same loads, same value, same address): no predication
    const int col = min(slab * 64 + lane, ld - 1);
    const int r0 = runs[3 * run], cnt = runs[3 * run + 1], q0 = runs[3 * run + 2];
    const int mirror = c.mirror, cen = S >> 1;
    const double* Yw = reinterpret_cast<const double*>(Y) + (size_t)w * c.NU * ld + col;
    double* Cw = reinterpret_cast<double*>(C) + (size_t)w * c.CROWS * ld + col;
    double tap[O + 1];
    const int tcol = c.xsym ? min(col, c.Ph - 1) : (col >> 1);        // (columns past Ph are line padding)
#pragma unroll
    for (int t = 0; t <= O; ++t) tap[t] = c.taps[(size_t)t * c.Ph + tcol];
    double win[W], fifo[D];
    // window before phase 0 of conv row r0: input i (map row r0 - O + i), i = 0..2O-1, sits in slot (1 + i) % W;
    // the row r0 + O (i = 2O) arrives through the fifo at phase 0, then r0+O+1.. for the following phases.
    // All 2O + D requests go out back to back (clamped rows, raw values); the masks follow behind a
    // scheduling barrier (otherwise hipcc pairs every request with its mask multiply and waits in between).
#pragma unroll
    for (int i = 0; i < 2 * O; ++i) {
        const int mc = min(max(r0 - O + i, 0), S - 1);
        win[(1 + i) % W] = Yw[(size_t)(mirror ? abs(mc - cen) : mc) * ld];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int mc = min(max(r0 + O + d, 0), S - 1);
        fifo[d] = Yw[(size_t)(mirror ? abs(mc - cen) : mc) * ld];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2 * O; ++i) {
        const int m = r0 - O + i;
        win[(1 + i) % W] *= (m >= 0 && m < S) ? 1.0 : 0.0;
    }
    win[0] = 0.0;
    __builtin_amdgcn_sched_barrier(0);
    for (int t0 = 0; t0 < cnt; t0 += W)
        jx_fir_run_phases<O, 0>(win, fifo, tap, mirror, Yw, Cw, ld, S, c.NJ, r0 + t0, q0 + t0, cnt - t0);
}

// ------------------------------------------------------------------------------------
// Low-rank form of the transfer-function weights.  The weights Hy[job][kx] of pass 3 factor as
// sum_rho U[rho][job] v_rho[kx] with a few dozen terms (jx_tables.hpp lowrank_factor, to rounding), and
// everything between the FIR and the weighting is linear and the same for every job, so
//     Z(kx) = sum_q Hy[q](kx) X_q(kx) = sum_rho v_rho(kx) T( sum_q U[rho][q] C_q )(kx):
// the jobs are combined FIRST and pass 3 transforms r rows per walker instead of NJ.
// This kernel is the combination  D[w][rho][j] = sum_q U[rho][q] B[w][q][j]  as a batched GEMM on the fp64
// matrix cores (v_mfma_f64_16x16x4: A = U tile, lane l holds U[rho0 + (l & 15)][4 s + (l >> 4)] for every
// k-step s of its share, kept in registers for the whole launch; B lane l = B[q = 4 s + (l >> 4)][j = j0 + (l & 15)];
// D lane l, register g = D[rho0 + (l >> 4) + 4 g][j0 + (l & 15)]).  Persistent blocks walk the (walker, column tile) tasks.
// Generic strides serve both the FIR rows (j = kx) and the column-0 terms (j = output column x).
// ------------------------------------------------------------------------------------
#define JX_LR_KS 72                 // k-steps per task, B values of all of them in flight at once: jobs <= 4 * JX_LR_KS
#define JX_LR_LDS_MAX (150 * 1024)  // the U fragments of every rho tile live in LDS
struct JxGemmSeg {                  // one run of batches of the GEMM kernel: matrices, operands, results and their strides
    const double* A; const double* B; double* D;
    long long a_batch, b_batch, d_batch, dj;                          // per-batch steps; step between result columns
    int nbatch;
    int acc;                                                          // 1: D += (the second half of a K range split over two launches)
};
struct JxLowrank {
    const double* U;                // [RP][KQ] sigma u, zero padded (RP multiple of 16, KQ = 4 ks) -- host-side handle, the kernel takes JxGemmSeg
    int r, ks, KQ, nq;              // rank, k-steps, padded and true job count
};

typedef double jx_v4d __attribute__((ext_vector_type(4)));

// One wave per column tile, all NTR rho tiles of it (NTR accumulators): every B value is requested once per CU, its NTR
// A fragments come from LDS (fragment order [tile][k-step][lane]).  KS = k-steps compiled in (>= lr.ks, the A fragments
// beyond lr.ks are zero): no branch inside the load and MFMA sequences, no barrier between the tasks of one A tile.
// TB: element type of the B operand and of the result in memory (double, or float for the fp32 variant: values are widened
// on load, the products and sums stay fp64, the result is rounded once on store).  The A matrices are always fp64.
template <int KS, int NTR, typename TB = double>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(1, 2)))
jx_lowrank_kernel(JxLowrank lr, JxGemmSeg sg0, JxGemmSeg sg1, long long bws, long long bq, long long bj,
                  long long dws, long long dr, int ncols, int nwalk) {
    extern __shared__ __attribute__((aligned(16))) double s_a[];      // [NTR][KS][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    // Work: batches (each with its own A matrix and B / D base: the fused FIR + combination has one per column kx) of
    // units (walker, 16-column tile); a group = nwave consecutive units of one batch, one per wave.  Persistent blocks
    // take CONTIGUOUS ranges of groups, so a block changes its A tile (refill of the LDS copy, two barriers) rarely.
    const int ntile = (ncols + 15) >> 4, nunit = ntile * nwalk, gpb = (nunit + nwave - 1) / nwave;
    const int nbatch = sg0.nbatch + sg1.nbatch;                       // batches of the second segment follow the first's
    const long long G = (long long)nbatch * gpb;
    const int g0 = (int)(G * blockIdx.x / gridDim.x), g1 = (int)(G * (blockIdx.x + 1) / gridDim.x);
    if (g0 >= g1) return;
    const long long step = 4 * bq;
    // B rows 4 s + lk, s < KS.  Rows past the last one are read too (A = 0 there): the caller keeps 4 KS rows behind every
    // B block allocated and finite.  Each B register is refilled with the row half a task ahead (this task's or the next
    // one's) right after its MFMAs: a value is requested H k-steps before it is used.
    auto bptr = [&](int g) {
        const int batch = g / gpb, unit = min((g - batch * gpb) * nwave + wave, nunit - 1);
        const int w = unit / ntile, jt = unit - w * ntile;
        const bool second = batch >= sg0.nbatch;                      // wave-uniform
        const TB* Bb = second ? reinterpret_cast<const TB*>(sg1.B) + (size_t)(batch - sg0.nbatch) * sg1.b_batch
                              : reinterpret_cast<const TB*>(sg0.B) + (size_t)batch * sg0.b_batch;
        return Bb + (size_t)w * bws + (size_t)min(jt * 16 + li, ncols - 1) * bj + (size_t)lk * bq;
    };
    // one LDS base per rho tile: the k-step offsets then fit the 16-bit immediate of ds_read (a single base would need an
    // address register for every (tile, step) past 64 KB)
    int sat[NTR];                                                    // element offsets into s_a
#pragma unroll
    for (int t = 0; t < NTR; ++t) {
        sat[t] = lane + t * KS * 64;
        asm volatile("" : "+v"(sat[t]));                              // opaque: keeps the bases apart
    }
    constexpr int H = KS / 2;                                         // B values in flight (must divide KS): half a task ahead
    TB b[H];                                                           // (kept in the storage type: a float is widened when it is USED, half a
                                                                       //  task after its request -- widening at the request would wait for it at once)
    const TB* pl = bptr(g0);                                           // running load pointer (no table of row addresses in registers)
#pragma unroll
    for (int u = 0; u < H; ++u) { b[u] = *pl; pl += step; }
    int cur = -1;
    for (int g = g0; g < g1; ++g) {
        const int batch = g / gpb;
        if (batch != cur) {                                           // block-uniform
            if (cur >= 0) __syncthreads();                            // every wave is done with the old tile
            const double* Ab = batch >= sg0.nbatch ? sg1.A + (size_t)(batch - sg0.nbatch) * sg1.a_batch : sg0.A + (size_t)batch * sg0.a_batch;
            for (int e = threadIdx.x; e < NTR * KS * 64; e += blockDim.x) {
                const int l = e & 63, s = (e >> 6) % KS, t = (e >> 6) / KS;
                s_a[e] = (s < lr.ks) ? Ab[(size_t)(t * 16 + (l & 15)) * lr.KQ + 4 * s + (l >> 4)] : 0.0;
            }
            __syncthreads();
            cur = batch;
        }
        const TB* pbn = bptr(min(g + 1, g1 - 1));                      // past the end: a valid address, never used
        jx_v4d acc[NTR];
#pragma unroll
        for (int t = 0; t < NTR; ++t) acc[t] = jx_v4d{0.0, 0.0, 0.0, 0.0};
        // A fragments: the next step's NTR values are read from LDS while this step's MFMAs run (explicit two-deep buffer;
        // left to itself the scheduler hoists whole groups of reads and spills)
        double ac[NTR], an[NTR];
#pragma unroll
        for (int t = 0; t < NTR; ++t) ac[t] = s_a[sat[t]];
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            if (u + 1 < KS) {
#pragma unroll
                for (int t = 0; t < NTR; ++t) an[t] = s_a[sat[t] + (u + 1) * 64];
            }
            const double bu = (double)b[u % H];
#pragma unroll
            for (int t = 0; t < NTR; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[t], bu, acc[t], 0, 0, 0);
            if (u + H == KS) pl = pbn;                               // the ring moves on to the next task's rows
            b[u % H] = *pl; pl += step;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NTR; ++t) ac[t] = an[t];
        }
        const int unit = (g - batch * gpb) * nwave + wave;
        const int w = unit / ntile, jt = unit - w * ntile;
        if (unit < nunit && jt * 16 + li < ncols) {
            const bool second = batch >= sg0.nbatch;
            TB* Dp = (second ? reinterpret_cast<TB*>(sg1.D) + (size_t)(batch - sg0.nbatch) * sg1.d_batch
                             : reinterpret_cast<TB*>(sg0.D) + (size_t)batch * sg0.d_batch)
                     + (size_t)w * dws + (size_t)(jt * 16 + li) * (second ? sg1.dj : sg0.dj);
#pragma unroll
            for (int t = 0; t < NTR; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int rho = t * 16 + lk + 4 * gq;
                    if (rho < lr.r) Dp[(size_t)rho * dr] = (TB)((second ? sg1.acc : sg0.acc) ? (double)Dp[(size_t)rho * dr] + acc[t][gq] : acc[t][gq]);
                }
        }
    }
}
#define JX_LR_BUCKETS(X) X(24) X(40) X(68) X(72)
#define JX_LR_KINDS(X) X(24, 1) X(24, 2) X(24, 3) X(24, 4) X(40, 1) X(40, 2) X(40, 3) X(40, 4) X(68, 1) X(68, 2) X(68, 3) X(68, 4) \
    X(72, 1) X(72, 2) X(72, 3) X(72, 4)
// the kinds instantiated for fp32 operands as well (the fused route of the even sides: rank 17..64)
#define JX_LR_KINDS_F32(X) X(24, 2) X(24, 3) X(24, 4) X(40, 2) X(40, 3) X(40, 4) X(68, 2) X(68, 3) X(68, 4) X(72, 2) X(72, 3) X(72, 4)

// ------------------------------------------------------------------------------------
// tail of the hand-written path.  One 256-thread block per walker:
//   Z[kc]  = sum over pass-3 blocks of the partial sums (fixed order)
//   row[k] = sum_kc Re(Z[kc] e^{2 pi i kc (S/2 + k)/S}),  k = 0..S/2-1   (joxsz_funcs.py:467,472)
//            as ONE inverse real transform of length S = 2 LS (half-length complex FFT in registers + LDS)
//   bright = row * cfac;  chi^2 against the flux points;  log-posterior  (joxsz_funcs.py:473-493, 544)
// The one-sided sum above is x[m] = V0 + 2 sum Re(V[k] e^{..}) + V[LS] (-1)^m with V0 = Re Z0, V[k] = Z[k]/2,
// V[LS] = Re Z[LS];  z[n] = x[2n] + i x[2n+1] is the inverse FFT of
//   Zc[k] = (V[k] + conj V[LS-k]) + i e^{+2 pi i k/S} (V[k] - conj V[LS-k]),  k = 0..LS-1.
// ------------------------------------------------------------------------------------
template <int LS>
__global__ void __launch_bounds__(256)
jx_tail_fft_kernel(JxDev c, JxConv cv, const cplx* __restrict__ zpart, int nblk, const double* __restrict__ cfac, const double* __restrict__ sz0,
                   const double* __restrict__ base, double* __restrict__ logp, int w0,
                   double* __restrict__ tap_row, double* __restrict__ tap_bright, double* __restrict__ tap_chisq,
                   double* __restrict__ tap_parts) {
    typedef jx_lay<LS> Lay;
    constexpr int L1 = Lay::L1, L2 = Lay::L2, RS = Lay::RS;
    constexpr int NK = (LS + 1 + 255) / 256;                         // spectrum entries per thread
    __shared__ cplx s_M[RS];                                          // (16-byte aligned: every LDS access below is b128 or b64)
    __shared__ cplx s_tw[LS];                                         // e^{-2 pi i n / LS}
    __shared__ cplx s_v[LS + 1];                                      // Z
    __shared__ double s_prof[LS];
    __shared__ double s_red[8];
    const int tid = threadIdx.x, w = blockIdx.x, Sh = LS + 1, nrow = LS;       // nblk: pass-3 blocks per walker of this launch

    // everything this thread will want from global memory before the first barrier is requested up front
    double twr[NK][2];
#pragma unroll
    for (int j = 0; j < NK; ++j) { const cplx t = cv.tw_ls[min(tid + j * 256, LS - 1)]; twr[j][0] = t.x; twr[j][1] = t.y; }
    double wk[L1][2];                                                 // e^{-2 pi i k/S}, k = n1 L2 + tid (step-A threads)
#pragma unroll
    for (int n1 = 0; n1 < L1; ++n1) { const cplx t = cv.tw_s[min(n1 * L2 + tid, LS)]; wk[n1][0] = t.x; wk[n1][1] = t.y; }
    const cplx* Zp = zpart + (size_t)w * nblk * Sh;
    double zr[NK], zi[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) { zr[j] = 0.0; zi[j] = 0.0; }
    for (int b0 = 0; b0 < nblk; b0 += 8) {
        cplx v[NK][8];
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int u = 0; u < 8; ++u) v[j][u] = Zp[(size_t)min(b0 + u, nblk - 1) * Sh + min(tid + j * 256, Sh - 1)];
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int u = 0; u < 8; ++u) if (b0 + u < nblk) { zr[j] += v[j][u].x; zi[j] += v[j][u].y; }
    }
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const int k = tid + j * 256;
        if (k < Sh) s_v[k] = make_double2(zr[j], zi[j]);
        if (k < LS) s_tw[k] = make_double2(twr[j][0], twr[j][1]);
    }
    __syncthreads();
    if (tid < L2) {
        jx_c x[L1];
#pragma unroll
        for (int n1 = 0; n1 < L1; ++n1) {
            const int k = n1 * L2 + tid;
            const cplx a = s_v[k], b = s_v[LS - k];
            // V[k], conj V[LS-k] with the end points real
            const double vx = (k == 0) ? a.x : 0.5 * a.x, vy = (k == 0) ? 0.0 : 0.5 * a.y;
            const double px = (k == 0) ? b.x : 0.5 * b.x, py = (k == 0) ? 0.0 : -0.5 * b.y;
            const double sx = vx + px, sy = vy + py, dx = vx - px, dy = vy - py;
            // i e^{+2 pi i k/S} (dx + i dy),  e^{+..} = conj(wk)
            const double cr = wk[n1][0], ci = -wk[n1][1];
            const double ex = dx * cr - dy * ci, ey = dx * ci + dy * cr;
            x[n1] = jxc(sx - ey, sy + ex);
        }
        jx_stepA_store<LS, true>(x, tid, s_M, s_tw);
    }
    __syncthreads();
    if (tid < L1) {
        jx_c y[L2];
        jx_stepB_load<LS, true>(y, tid, s_M);
#pragma unroll
        for (int k2 = 0; k2 < L2; ++k2) {
            const int n = tid + L1 * k2, k = 2 * n - LS;             // z[n] = (x[2n], x[2n+1]); row index k = column - S/2
            if (k >= 0) {
                const double2 cf = *reinterpret_cast<const double2*>(cfac + (size_t)w * nrow + k);
                if (tap_row) *reinterpret_cast<double2*>(tap_row + (size_t)w * nrow + k) = make_double2(y[k2].x, y[k2].y);
                const double b0v = y[k2].x * cf.x, b1v = y[k2].y * cf.y;
                s_prof[k] = b0v; s_prof[k + 1] = b1v;
                if (tap_bright) *reinterpret_cast<double2*>(tap_bright + (size_t)w * nrow + k) = make_double2(b0v, b1v);
            }
        }
    }
    __syncthreads();
    // chi^2: eight lanes per flux point, each over every eighth radius
    double part = 0.0;
    for (int d = tid >> 3; d < c.nflux; d += 32) {
        const double* e = c.emat + (size_t)d * nrow;
        double m = 0.0;
#pragma unroll 8
        for (int k = tid & 7; k < nrow; k += 8) m = fma(e[k], s_prof[k], m);
        m += __shfl_xor(m, 1, 64); m += __shfl_xor(m, 2, 64); m += __shfl_xor(m, 4, 64);
        const double z = (c.flux[c.nflux + d] - m) / c.flux[2 * c.nflux + d];
        const double z2 = z * z;
        if ((tid & 7) == 0 && z2 == z2) part += z2;                  // np.nansum drops NaN terms
    }
    const double chisq = jx_block_sum(part, s_red);
    if (tid == 0) {
        const double ll = -chisq / 2.0 + (sz0 ? sz0[w] : 0.0);
        const double b = base[w];
        double tot = (b == -INFINITY) ? -INFINITY : b + ll;
        if (tot != tot) tot = -INFINITY;                             // never hand NaN to the sampler
        logp[w0 + w] = tot;
        if (tap_chisq) tap_chisq[w] = chisq;
        if (tap_parts) tap_parts[(size_t)w * 4 + 1] = ll;
    }
}

// FIR along rows on the real row spectra for beam widths without a register-window instance.  Only the routes with
// separate kernels get here (the beam-convolved-map tap, JOXSZ_FUSED=0): one thread per output element, plain sums.
__global__ void __launch_bounds__(256)
jx_beamfir_real_kernel(JxConv c, const double* __restrict__ Y, double* __restrict__ C) {
    const int q = blockIdx.x, w = blockIdx.y, S = c.S, o = c.o, ld = c.fir_ld, r = c.jrow[q];
    const double* Yw = Y + (size_t)w * c.NU * ld;
    for (int k = threadIdx.x; k < c.Ph; k += blockDim.x) {
        double a = 0.0;
        for (int d = -o; d <= o; ++d) {
            const int m = r + d;
            if (m >= 0 && m < S) a = fma(c.taps[(size_t)abs(d) * c.Ph + k], Yw[(size_t)c.umap[m] * ld + k], a);
        }
        C[((size_t)w * c.CROWS + q) * ld + k] = a;
    }
}

// mirror the quadrant of distinct pixels into the full S x S Compton-y map (parity tap only)
__global__ void __launch_bounds__(256)
jx_expand_quad_kernel(const double* __restrict__ quad, size_t q_ld, size_t q_ws, int S, double* __restrict__ full /*[W][S][S]*/) {
    const int r = blockIdx.x, w = blockIdx.y, c = S >> 1;
    const double* src = quad + (size_t)w * q_ws + (size_t)abs(r - c) * q_ld;
    double* dst = full + ((size_t)w * S + r) * S;
    for (int x = threadIdx.x; x < S; x += blockDim.x) dst[x] = src[abs(x - c)];
}

// expand job rows to the full S x S beam-convolved map (parity tap only)
__global__ void __launch_bounds__(256)
jx_expand_rows_kernel(const double* __restrict__ jobs /*[W][NJ][S]*/, const int* __restrict__ rowjob /*[S]*/, int S, int NJ,
                      double* __restrict__ full /*[W][S][S]*/) {
    const int r = blockIdx.x, w = blockIdx.y;
    const double* src = jobs + ((size_t)w * NJ + rowjob[r]) * S;
    double* dst = full + ((size_t)w * S + r) * S;
    for (int x = threadIdx.x; x < S; x += blockDim.x) dst[x] = src[x];
}

// odd-side route, parity tap only: conv[w][r][x] = ccj[|x - c|][job of row r][w]
__global__ void __launch_bounds__(256)
jx_expand_odd_conv_kernel(const double* __restrict__ ccj /*[a][RPj][tW]*/, const int* __restrict__ rowjob /*[S]*/, int S, int RPj, long long tW,
                          double* __restrict__ full /*[W][S][S]*/) {
    const int r = blockIdx.x, w = blockIdx.y, c = S >> 1, q = rowjob[r];
    double* dst = full + ((size_t)w * S + r) * S;
    for (int x = threadIdx.x; x < S; x += blockDim.x) dst[x] = ccj[((size_t)abs(x - c) * RPj + q) * tW + w];
}
