// Exact form of the SZ side (round 5; default).  Between the Compton-y profile and the extracted map row every step of the
// reference is linear with constant coefficients (joxsz_funcs.py:460-467, row of :472):
//     f       = interp1d(+-r_pp, (y, y), 'cubic')            mirrored not-a-knot spline: moments M = G y, samples A y_k + B y_k+1 + C M_k + D M_k+1
//     y_2d    = f(d_mat)                                     every pixel of the S x S map
//     conv_2d = fftconvolve(y_2d, beam_2d, 'same') * step^2
//     map_out = real(ifft2(fft2(conv_2d) * filtering));      out[x] = map_out[S//2, S//2 + x]
// so out = Wy y with ONE constant nrow x Nk matrix, built once in jx_finalize on the host from the caller's d_mat, beam image and
// filter (jxt::exact_row_operator: every pixel, every radius, any beam image, any real transfer function, sums in long double;
// nothing is truncated or sub-sampled and there is one form for every input).  Per walker the path keeps what depends on its
// parameters: the pressure profile (jx_prep_kernel), its forward Abel transform and the Compton-y scale y = y_scale A pp
// (joxsz_funcs.py:457-459), the product with Wy, and the tail (joxsz_funcs.py:472-479, 538).
//
//                           (With an odd number of ordinate tiles the last one -- the cheapest -- has no partner: in the timed path its share of
//                           the row is folded into the row product as an operator on the profile, jxt::exact_fold_layout, and the others pair up.)
//   jx_ordrow_kernel        one block = 16 walkers (one matrix-core tile) x one PAIR of 16-ordinate column tiles (p, last - p: the
//                           Abel matrix is triangular, the two k-ranges add up to the same length for every pair); its four waves
//                           take every fourth 16-radius step of the k-range each, straight from memory (no LDS staging, no barrier
//                           in the loop), and meet in LDS.  The block then multiplies ITS 32 ordinates into the row operator --
//                           a partial row [16][outputs] per pair -- so the ordinates never have to be read back.
//   jx_rowsum_tail_kernel   adds a walker tile's partial rows in pair order, then conversion factors, the not-a-knot spline to the
//                           data radii (a constant nflux x nrow matrix), chi^2, total, and -- inside jx_sample -- the acceptance
//                           of the stretch move.  One block per 16 walkers.
//   jx_rowop_tail_kernel    the same two steps as ONE block per 16 walkers reading the ordinates back (K cut over eight waves):
//                           fewer, longer blocks; kept as the reference form of the pair-wise kernel (JOXSZ_X_PAIRWISE=0).
//
// Every sum is grouped by the problem alone (steps of a wave, waves of a block, pairs of a tile, in that order), never by the
// launch: a walker's result does not depend on its position in whatever batch.
//
// Operand layouts (v_mfma_f64_16x16x4: A lane l = A[l & 15][4 s + (l >> 4)], B lane l = B[4 s + (l >> 4)][l & 15], D register g of
// lane l = D[(l >> 4) + 4 g][l & 15]).  The order of the K index is free as long as A and B agree on it: a lane takes FOUR
// consecutive values of its walker in one 32-byte load, v[w][16 s + 4 lk + e], e = 0..3, and feeds element e to sub-step e; the
// operators are stored to match (jxt::exact_row_layout, jxt::abel_ordinate_layout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jx_kernels.hpp"

typedef double jx_ro_v4d __attribute__((ext_vector_type(4)));

struct JxRowOp {
    int n;                         // walkers of this launch
    int nS;                        // macro steps of 16 ordinates (Nkp / 16) = column tiles of the ordinate product
    int ldy;                       // doubles per walker of the ordinate array (= Nkp)
    int ng;                        // output groups of this launch: those the data-radii spline reads, or all of them (row / brightness taps)
    int nuse;                      // outputs the data-radii matrix reads (<= 16 NXT ng)
    int ldr;                       // doubles per walker of the row in LDS (odd, > nuse)
    int lde;                       // doubles per data radius of the data-radii matrix in LDS; 0: read from memory
    int dbg;                       // diagnostic build only (make ABLATIONS=1; JOXSZ_X_DBG): 1 no matrix instructions in the ordinate product, 2 no row product, 4 no partial-row stores, 8 no operator loads
    int ldpp, nSj, npair;          // ordinate product: doubles per profile (16 nSj, zeros behind the grid), macro steps of 16 radii, column-tile pairs
    int nfold, s0f;                // folded form (odd nS, timed path): macro steps of 16 radii of the last ordinate tile's share of the row, the first of them; 0: none
    const double* Wfk;             // [ng_all][nfold][4][64][NXT]  the last ordinate tile's share of the row as an operator on the profile (jxt::exact_fold_layout)
    const double* Opk;             // [ng_all][nS][4][64][NXT]     row operator (jxt::exact_row_layout)
    const double* Typ;             // [nSj][nS][64][4]             ordinate operator (jxt::abel_ordinate_layout)
    const double* pp;              // [n][ldpp] pressure profiles (jx_prep_kernel)
    double* y;                     // [tW][ldy] ordinates, walker-major; null: not stored (timed path: nothing reads them)
    double* P;                     // [walker tiles][npair][ng_all][16][16 NXT] partial rows
    long long* stamps;             // diagnostic build only: [blocks][8] wall-clock stamps (100 MHz) of jx_ordrow_kernel's phases
};
#ifdef JOXSZ_ABLATIONS
#define JX_STAMP(g, k) do { if ((g).stamps && threadIdx.x == 0) (g).stamps[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define JX_STAMP(g, k) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------------------------
// The tail on a walker tile's row in LDS (s_row[w][x], x < nuse, conversion not yet applied): joxsz_funcs.py:473, 476, 478, 538.
// Called by every thread of the block.  s_z: [waves][16][nflux + 1] LDS scratch; s_E: [nflux][lde] the data-radii matrix (filled by
// the caller; lde = 0: read from memory).  The nflux x nuse matrix-vector products of the 16 walkers: a wave takes a quarter (an
// eighth) of the outputs x, a lane one walker and every fourth data radius, so a brightness value read from LDS serves up to ND
// multiply-adds; the waves' partial sums are added in wave order.
// ------------------------------------------------------------------------------------------------------------------
template <int ND, bool ELDS>
__device__ __forceinline__ void jx_row_dots(const JxDev& c, const JxRowOp& g, const double* __restrict__ s_row, double* __restrict__ s_z,
                                            const double* __restrict__ s_E, int i0) {
    const int tid = threadIdx.x, nwv = blockDim.x >> 6, wv = tid >> 6, lane = tid & 63, w = lane & 15, dq = lane >> 4;
    const int nflux = c.nflux, nuse = g.nuse, nrow = c.nrow;
    const int xw = (nuse + nwv - 1) / nwv, x0 = min(wv * xw, nuse), x1 = min(nuse, x0 + xw);
    double acc[ND];
    int eo[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) { acc[i] = 0.0; eo[i] = min(dq + 4 * (i0 + i), nflux - 1) * (ELDS ? g.lde : nrow); }   // (a data radius beyond the last repeats it: never stored)
    const double* __restrict__ br = s_row + (size_t)w * g.ldr;
    const double* __restrict__ em = c.emat;
    for (int x = x0; x < x1; ++x) {
        const double b = br[x];
#pragma unroll
        for (int i = 0; i < ND; ++i) acc[i] = fma(ELDS ? s_E[eo[i] + x] : em[eo[i] + x], b, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int d = dq + 4 * (i0 + i);
        if (d < nflux) s_z[((size_t)wv * 16 + w) * (nflux + 1) + d] = acc[i];
    }
}

template <bool CONVERTED /* s_row already carries the conversion factors (and the brightness tap is written) */>
__device__ __forceinline__ void jx_row_epilogue(const JxDev& c, const JxRowOp& g, int wb, double* s_row, double* s_z, const double* s_E,
                                                const double* __restrict__ cfac, const double* __restrict__ sz0, const double* __restrict__ base,
                                                double* __restrict__ logp, int w0, double* __restrict__ tap_bright, double* __restrict__ tap_chisq,
                                                double* __restrict__ tap_parts, const JxSm& smv) {
    const int tid = threadIdx.x, nth = blockDim.x, nrow = c.nrow, nflux = c.nflux, nuse = g.nuse, ldr = g.ldr;
    // (what the last steps read from memory is requested here, ahead of the products)
    const bool fin = tid < 16 && wb + tid < g.n;
    const double pre_base = fin ? base[wb + tid] : 0.0, pre_sz0 = (fin && sz0) ? sz0[wb + tid] : 0.0;
    const double pre_xl = (fin && c.xr_split) ? c.xr_out[2 * (size_t)(wb + tid)] : 0.0, pre_xb = (fin && c.xr_split) ? c.xr_out[2 * (size_t)(wb + tid) + 1] : 0.0;
    double pre_f[2], pre_e[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int d = min((tid + u * nth) >> 4, nflux - 1); pre_f[u] = c.flux[nflux + d]; pre_e[u] = c.flux[2 * nflux + d]; }
    if (!CONVERTED) {
        for (int idx = tid; idx < 16 * nuse; idx += nth) {
            const int w = idx / nuse, x = idx - w * nuse;
            const size_t o = (size_t)min(wb + w, g.n - 1) * nrow + x;
            const double b = s_row[w * ldr + x] * cfac[o];
            s_row[w * ldr + x] = b;
            if (tap_bright && wb + w < g.n) tap_bright[o] = b;
        }
        __syncthreads();
    }
    const int ni = (nflux + 3) >> 2;                            // data radii per lane
    if (ni <= 5) { if (g.lde) jx_row_dots<5, true>(c, g, s_row, s_z, s_E, 0); else jx_row_dots<5, false>(c, g, s_row, s_z, s_E, 0); }
    else for (int i0 = 0; i0 < ni; i0 += 8) { if (g.lde) jx_row_dots<8, true>(c, g, s_row, s_z, s_E, i0); else jx_row_dots<8, false>(c, g, s_row, s_z, s_E, i0); }
    __syncthreads();
    const int nwv = nth >> 6;
    for (int p = tid, u = 0; p < 16 * nflux; p += nth, ++u) {
        const int w = p & 15, d = p >> 4;
        double m = 0.0;
        for (int v = 0; v < nwv; ++v) m += s_z[((size_t)v * 16 + w) * (nflux + 1) + d];
        const double z = ((u < 2 ? pre_f[u & 1] : c.flux[nflux + d]) - m) / (u < 2 ? pre_e[u & 1] : c.flux[2 * nflux + d]);
        const double z2 = z * z;
        s_z[(size_t)w * (nflux + 1) + d] = (z2 == z2) ? z2 : 0.0;       // np.nansum drops NaN terms
    }
    __syncthreads();
    if (tid < 16 && wb + tid < g.n) {
        const int w = wb + tid;
        double chisq = 0.0;
        for (int d = 0; d < nflux; ++d) chisq += s_z[(size_t)tid * (nflux + 1) + d];
        const double ll = -chisq / 2.0 + pre_sz0;
        double b = pre_base;
        // (two-block form of the per-walker kernel: the priors arrive here, the Cash log-likelihood and its verdict beside them)
        if (c.xr_split && b != -INFINITY) b = (pre_xb != 0.0) ? -INFINITY : b + pre_xl;
        double tot = (b == -INFINITY) ? -INFINITY : b + ll;
        if (tot != tot) tot = -INFINITY;                         // never hand NaN to the sampler
        logp[w0 + w] = tot;
        if (smv.on) {
            // accept or reject the proposal of walker w0 + w of this half (jx_sm_accept_kernel's arithmetic and random number)
            const int i = w0 + w;
            uint32_t r[4];
            jx_philox((uint32_t)i, (uint32_t)smv.iter2, 1u, 0u, (uint32_t)smv.seed, (uint32_t)(smv.seed >> 32), r);
            const double u3 = jx_u01(r[0], r[1]);
            const double lnpdiff = __dadd_rn(__dmul_rn((double)(smv.ndim - 1), log(smv.zz[i])), __dsub_rn(tot, smv.lp[smv.s1 + i]));
            if (isfinite(tot) && log(u3) < lnpdiff) {
                for (int dd = 0; dd < smv.ndim; ++dd) smv.x[(size_t)(smv.s1 + i) * smv.ndim + dd] = smv.q[(size_t)i * smv.ndim + dd];
                smv.lp[smv.s1 + i] = tot;
                smv.nacc[smv.s1 + i] += 1;
            }
        }
        if (tap_chisq) tap_chisq[w] = chisq;
        if (tap_parts) tap_parts[(size_t)w * 4 + 1] = ll;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Ordinate product + this pair's share of the row product.
//   grid: 8 x ceil(walker tiles / 8) x npair blocks of 256 threads; block id & 7 picks the walker tiles of one XCD (the profiles
//   of a tile and its partial rows then stay in one L2: speed only)
//   LDS: [4 waves][2 tiles][16][16] partial ordinates, then [16][36] the pair's ordinates
// ------------------------------------------------------------------------------------------------------------------
#define JX_ORD_LDS_DOUBLES (4 * 2 * 256 + 16 * 36)

template <int NXT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
jx_ordrow_kernel(JxRowOp g) {
    extern __shared__ __attribute__((aligned(16))) double sm_or[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntw = (g.n + 15) >> 4;
    const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
    const int p = jj % g.npair, wt = (jj / g.npair) * 8 + xcd;
    if (wt >= ntw) return;
    JX_STAMP(g, 0);
    const int wb = wt * 16;
    const int q = 2 * g.npair - 1 - p;                          // the pair's upper tile (may lie beyond the last: then p alone)
    const bool hasq = q < g.nS;
    double* s1 = sm_or;                                         // [4][2][16][16]
    double* s_yt = sm_or + 4 * 2 * 256;                         // [16][36]

    // ---- ordinate product: y[w][16 t + c] = sum_j Ty[j][16 t + c] pp[w][j] for t = p, q; steps s = p + wv, p + wv + 4, ... of 16 radii
    const int wq = min(wb + li, g.n - 1);                       // (a walker beyond the launch repeats the last: finite work, never stored)
    const double* __restrict__ ppl = g.pp + (size_t)wq * g.ldpp + 4 * lk;
    const double* __restrict__ tb = g.Typ + (size_t)lane * 4;
    jx_ro_v4d ap[2], aq[2];                                     // two chains per tile (sub-steps e and e + 2 share one)
#pragma unroll
    for (int e = 0; e < 2; ++e) { ap[e] = jx_ro_v4d{0.0, 0.0, 0.0, 0.0}; aq[e] = jx_ro_v4d{0.0, 0.0, 0.0, 0.0}; }
    constexpr int NTT = (NXT + 3) / 4;
    double bpv[NTT][4], bqv[NTT][4];
    auto load_op = [&](int gi) {
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const int t = min(wv + 4 * tt, NXT - 1);
            const double* __restrict__ ob = g.Opk + ((size_t)gi * g.nS * 256 + lane) * NXT + t;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bpv[tt][e] = ob[((size_t)p * 4 + e) * 64 * NXT];
                bqv[tt][e] = ob[((size_t)(hasq ? q : p) * 4 + e) * 64 * NXT];
            }
        }
    };

    auto load_a = [&](int s) -> jx_ro_v4d { return *reinterpret_cast<const jx_ro_v4d*>(ppl + 16 * s); };   // (rows padded with zeros to 16 nSj)
    auto load_b = [&](int s, int t) -> jx_ro_v4d { return *reinterpret_cast<const jx_ro_v4d*>(tb + ((size_t)s * g.nS + t) * 256); };
    {
        int s = p + wv;
        jx_ro_v4d a0, bp0, bq0, a1, bp1, bq1;
        const int tq = hasq ? q : p;                            // (no upper tile: its loads repeat the lower one's, its products are skipped)
        if (s < g.nSj) { a0 = load_a(s); bp0 = load_b(s, p); bq0 = load_b(s, tq); }
#ifdef JOXSZ_ABLATIONS
        if (g.stamps) { if (a0[0] + bp0[0] + bq0[0] == 1.234e300) ap[0][0] = 1.0; JX_STAMP(g, 1); }
#endif
        for (; s < g.nSj; s += 8) {
            const int s1n = s + 4, s2n = s + 8;
            if (s1n < g.nSj) { a1 = load_a(s1n); bp1 = load_b(s1n, p); bq1 = load_b(s1n, tq); }
            if (!JX_DBG(g, 1)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ap[e & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[e], bp0[e], ap[e & 1], 0, 0, 0);
            } else ap[0] += a0 + bp0 + bq0;
            if (hasq && s >= q && !JX_DBG(g, 1)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) aq[e & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[e], bq0[e], aq[e & 1], 0, 0, 0);
            }
            if (s1n < g.nSj) {
                if (s2n < g.nSj) { a0 = load_a(s2n); bp0 = load_b(s2n, p); bq0 = load_b(s2n, tq); }
                if (!JX_DBG(g, 1)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ap[e & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[e], bp1[e], ap[e & 1], 0, 0, 0);
                } else ap[0] += a1 + bp1 + bq1;
                if (hasq && s1n >= q && !JX_DBG(g, 1)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) aq[e & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[e], bq1[e], aq[e & 1], 0, 0, 0);
                }
            }
        }
    }
    JX_STAMP(g, 2);
    {
        const jx_ro_v4d yp = ap[0] + ap[1], yq = aq[0] + aq[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1[((wv * 2 + 0) * 16 + lk + 4 * r) * 16 + li] = yp[r];
            s1[((wv * 2 + 1) * 16 + lk + 4 * r) * 16 + li] = yq[r];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 512; idx += 256) {
        const int t = idx >> 8, w = (idx >> 4) & 15, cc = idx & 15;
        const double v = ((s1[((0 * 2 + t) * 16 + w) * 16 + cc] + s1[((1 * 2 + t) * 16 + w) * 16 + cc]) + s1[((2 * 2 + t) * 16 + w) * 16 + cc]) + s1[((3 * 2 + t) * 16 + w) * 16 + cc];
        s_yt[w * 36 + 16 * t + cc] = v;
        const int tile = t ? q : p;
        if (g.y && tile < g.nS && wb + w < g.n) g.y[(size_t)(wb + w) * g.ldy + 16 * tile + cc] = v;
    }
    __syncthreads();
    JX_STAMP(g, 3);

    // ---- this pair's share of the row product: P[x] = sum over the pair's 32 ordinates of Wy[x][k] y[k]; wave v owns the output tiles v, v + 4
    const jx_ro_v4d yp = *reinterpret_cast<const jx_ro_v4d*>(s_yt + li * 36 + 4 * lk);
    const jx_ro_v4d yq = *reinterpret_cast<const jx_ro_v4d*>(s_yt + li * 36 + 16 + 4 * lk);
    for (int gi = 0; gi < (JX_DBG(g, 2) ? 0 : g.ng); ++gi) {
        load_op(gi);
        // folded form: pair p also takes the macro steps p, p + npair, ... of the last ordinate tile's share of the row, an operator on the
        // profile (requested with the row operator above: one trip; requesting both ahead of the reduction was measured twice: slower)
        const bool fold = p < g.nfold;
        jx_ro_v4d af = jx_ro_v4d{0.0, 0.0, 0.0, 0.0};
        double bfv[NTT][4];
        if (fold) {
            af = load_a(g.s0f + p);
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                const int t = min(wv + 4 * tt, NXT - 1);
                const double* __restrict__ ob = g.Wfk + ((((size_t)gi * g.nfold + p) * 4) * 64 + lane) * NXT + t;
#pragma unroll
                for (int e = 0; e < 4; ++e) bfv[tt][e] = ob[(size_t)e * 64 * NXT];
            }
        }
        double* __restrict__ Pb = g.P + (((size_t)wt * g.npair + p) * g.ng + gi) * (size_t)(256 * NXT);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const int t = wv + 4 * tt;
            if (t < NXT) {                                       // (wave-uniform)
                jx_ro_v4d c0 = jx_ro_v4d{0.0, 0.0, 0.0, 0.0}, c1 = jx_ro_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int e = 0; e < 4; ++e) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(yp[e], bpv[tt][e], c0, 0, 0, 0);
                if (hasq) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(yq[e], bqv[tt][e], c1, 0, 0, 0);
                }
                if (fold) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[e], bfv[tt][e], c1, 0, 0, 0);
                    for (int sf = p + g.npair; sf < g.nfold; sf += g.npair) {        // (more macro steps than pairs: short grids only)
                        const jx_ro_v4d a2 = load_a(g.s0f + sf);
                        const double* __restrict__ ob = g.Wfk + ((((size_t)gi * g.nfold + sf) * 4) * 64 + lane) * NXT + t;
#pragma unroll
                        for (int e = 0; e < 4; ++e) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[e], ob[(size_t)e * 64 * NXT], c1, 0, 0, 0);
                    }
                }
                const jx_ro_v4d cs = c0 + c1;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (!JX_DBG(g, 4) || cs[r] == 1.234e300) Pb[(size_t)(lk + 4 * r) * (16 * NXT) + 16 * t + li] = cs[r];
            }
        }
    }
    JX_STAMP(g, 4);
}

// ------------------------------------------------------------------------------------------------------------------
// The pairs' partial rows added in pair order, then the tail.  One block of 256 threads per 16 walkers.
//   LDS: [16][ldr] row, [16][nflux + 1] chi^2 terms, [nflux][lde] data-radii matrix
// ------------------------------------------------------------------------------------------------------------------
#define JX_RST_LDS_DOUBLES(ldr, nflux, lde) ((size_t)16 * (ldr) + (size_t)8 * 16 * ((nflux) + 1) + (size_t)(nflux) * (lde))

#define JX_RST_THREADS 512
template <int NXT>
__global__ void __launch_bounds__(JX_RST_THREADS)
jx_rowsum_tail_kernel(JxDev c, JxRowOp g, const double* __restrict__ cfac, const double* __restrict__ sz0, const double* __restrict__ base,
                      double* __restrict__ logp, int w0, double* __restrict__ tap_row, double* __restrict__ tap_bright,
                      double* __restrict__ tap_chisq, double* __restrict__ tap_parts, JxSm smv) {
    extern __shared__ __attribute__((aligned(16))) double sm_rs[];
    constexpr int NX = 16 * NXT, MAXC = 3;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int wt = blockIdx.x, wb = wt * 16, nrow = c.nrow, nuse = g.nuse, ldr = g.ldr;
    double* s_row = sm_rs;
    double* s_z = s_row + (size_t)16 * ldr;
    double* s_E = s_z + (size_t)8 * 16 * (c.nflux + 1);
    // The block's reads -- the data-radii matrix, the pairs' partial rows, the conversion factors -- are independent of one another and
    // all of them cold (the previous kernel's lines left the L2 at its end): requested together, they cost one trip to memory, not three.
    if (g.lde)
        for (int i = tid; i < c.nflux * nuse; i += nth) { const int d = i / nuse, x = i - d * nuse; s_E[d * g.lde + x] = c.emat[(size_t)d * nrow + x]; }
    const double* __restrict__ Pt = g.P + (size_t)wt * g.npair * g.ng * (size_t)(16 * NX);
    const size_t ps = (size_t)g.ng * (16 * NX);
    const int total = g.ng * 16 * NX;
    for (int c0 = 0; c0 < total; c0 += MAXC * nth) {
        double a[MAXC], cf[MAXC];
        int rem[MAXC], gi[MAXC];
        bool ok[MAXC];
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            const int idx = c0 + u * nth + tid;
            ok[u] = idx < total;
            const int ic = ok[u] ? idx : 0;
            gi[u] = ic / (16 * NX); rem[u] = ic - gi[u] * (16 * NX);
            a[u] = 0.0;
            const int w = rem[u] / NX, X = gi[u] * NX + (rem[u] - w * NX);
            cf[u] = cfac[(size_t)min(wb + w, g.n - 1) * nrow + min(X, nrow - 1)];
        }
#pragma unroll 4
        for (int pr = 0; pr < g.npair; ++pr) {                    // (added in pair order)
#pragma unroll
            for (int u = 0; u < MAXC; ++u) a[u] += Pt[(size_t)pr * ps + (size_t)gi[u] * (16 * NX) + rem[u]];
        }
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            if (!ok[u]) continue;
            const int w = rem[u] / NX, X = gi[u] * NX + (rem[u] - w * NX);
            const double b = a[u] * cf[u];
            if (X < nuse) s_row[w * ldr + X] = b;
            if (X < nrow && wb + w < g.n) {
                const size_t o = (size_t)(wb + w) * nrow + X;
                if (tap_row) tap_row[o] = a[u];
                if (tap_bright) tap_bright[o] = b;
            }
        }
    }
    __syncthreads();
    jx_row_epilogue<true>(c, g, wb, s_row, s_z, s_E, cfac, sz0, base, logp, w0, tap_bright, tap_chisq, tap_parts, smv);
}

// ------------------------------------------------------------------------------------------------------------------
// Reference form of the two steps above: one block per 16 walkers reads the ordinates back and owns every output (K cut over its
// eight waves, partial tiles added in wave order in LDS), the tail as its epilogue.
//   LDS: [8][16][LDP] partial tiles, row, chi^2 terms, data-radii matrix
// ------------------------------------------------------------------------------------------------------------------
#define JX_ROP_NW 8
#define JX_ROP_LDP(NXT) (16 * (NXT) + 16)      // row stride of a wave's partial tile: the four 16-lane rows of a store land on different banks
#define JX_ROP_LDS_DOUBLES(NXT, ldr, nflux, lde) ((size_t)JX_ROP_NW * 16 * JX_ROP_LDP(NXT) + JX_RST_LDS_DOUBLES(ldr, nflux, lde))

template <int NXT>
__global__ void __launch_bounds__(64 * JX_ROP_NW)
jx_rowop_tail_kernel(JxDev c, JxRowOp g, const double* __restrict__ cfac, const double* __restrict__ sz0, const double* __restrict__ base,
                     double* __restrict__ logp, int w0, double* __restrict__ tap_row, double* __restrict__ tap_bright,
                     double* __restrict__ tap_chisq, double* __restrict__ tap_parts, JxSm smv) {
    extern __shared__ __attribute__((aligned(16))) double sm_ro[];
    constexpr int NW = JX_ROP_NW, LDP = JX_ROP_LDP(NXT), NX = 16 * NXT;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb = blockIdx.x * 16;
    const int nrow = c.nrow, nflux = c.nflux, nuse = g.nuse, ldr = g.ldr;
    double* s_part = sm_ro;                                   // [NW][16][LDP]
    double* s_row = s_part + (size_t)NW * 16 * LDP;           // [16][ldr]
    double* s_z = s_row + (size_t)16 * ldr;                   // [8][16][nflux + 1]
    double* s_E = s_z + (size_t)8 * 16 * (nflux + 1);         // [nflux][lde]

    // the data-radii matrix comes into LDS behind the operand loads of the product (joxsz_funcs.py:476 as a constant matrix)
    if (g.lde)
        for (int i = tid; i < nflux * nuse; i += nth) { const int d = i / nuse, x = i - d * nuse; s_E[d * g.lde + x] = c.emat[(size_t)d * nrow + x]; }

    const int wq = min(wb + li, g.n - 1);                     // (a walker beyond the launch repeats the last: finite work, never stored)
    const double* __restrict__ yl = g.y + (size_t)wq * g.ldy + 4 * lk;
    const int s_lo = (wv * g.nS) / NW, s_hi = ((wv + 1) * g.nS) / NW;
    for (int gi = 0; gi < g.ng; ++gi) {
        jx_ro_v4d acc[NXT];
#pragma unroll
        for (int t = 0; t < NXT; ++t) acc[t] = jx_ro_v4d{0.0, 0.0, 0.0, 0.0};
        const double* __restrict__ ob = g.Opk + ((size_t)gi * g.nS * 256 + lane) * NXT;
        // operands one macro step ahead of their use
        jx_ro_v4d a0, a1;
        double b0[4][NXT], b1[4][NXT];
        auto fetch = [&](int s, jx_ro_v4d& a, double (&b)[4][NXT]) {
            a = *reinterpret_cast<const jx_ro_v4d*>(yl + 16 * s);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NXT; ++t) b[e][t] = ob[((size_t)s * 4 + e) * 64 * NXT + t];
        };
        auto mul = [&](const jx_ro_v4d& a, const double (&b)[4][NXT]) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NXT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], b[e][t], acc[t], 0, 0, 0);
        };
        if (s_lo < s_hi) {
            fetch(s_lo, a0, b0);
            for (int s = s_lo; s < s_hi; s += 2) {
                fetch(min(s + 1, s_hi - 1), a1, b1);
                mul(a0, b0);
                if (s + 1 < s_hi) {
                    fetch(min(s + 2, s_hi - 1), a0, b0);
                    mul(a1, b1);
                }
            }
        }
        // the waves' partial tiles meet in LDS and are added in wave order
#pragma unroll
        for (int t = 0; t < NXT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) s_part[((size_t)wv * 16 + lk + 4 * q) * LDP + 16 * t + li] = acc[t][q];
        __syncthreads();
        for (int idx = tid; idx < 16 * NX; idx += nth) {
            const int w = idx / NX, x = idx - w * NX;
            double v = 0.0;
#pragma unroll
            for (int u = 0; u < NW; ++u) v += s_part[((size_t)u * 16 + w) * LDP + x];
            const int X = gi * NX + x;
            if (X < nuse) s_row[w * ldr + X] = v;
            if (X < nrow && wb + w < g.n) {
                const size_t o = (size_t)(wb + w) * nrow + X;
                if (tap_row) tap_row[o] = v;
                if (tap_bright && X >= nuse) tap_bright[o] = v * cfac[o];
            }
        }
        __syncthreads();
    }
    jx_row_epilogue<false>(c, g, wb, s_row, s_z, s_E, cfac, sz0, base, logp, w0, tap_bright, tap_chisq, tap_parts, smv);
}

// Profile taps off the ordinate product's own array (radial grids too long for the Abel kernel): y[w][k], ab = y / y_scale
__global__ void __launch_bounds__(256)
jx_unpack_ordinates_kernel(const double* __restrict__ y, int ldy, int N, double y_scale, double* __restrict__ tap_y, double* __restrict__ tap_ab) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (k >= N) return;
    const double v = (k < ldy) ? y[(size_t)w * ldy + k] : 0.0;
    if (tap_y) tap_y[(size_t)w * N + k] = v;
    if (tap_ab) tap_ab[(size_t)w * N + k] = v / y_scale;
}
