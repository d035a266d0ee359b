// Pass 1 of the default route, second form: the Compton-y map is never written.  A block takes the cubic-spline
// coefficients of NW walkers (left in HBM/L2 by the Abel kernel, 32 bytes per radial interval), evaluates ONE distinct
// map row u of each of them straight into LDS (joxsz_funcs.py:460-462), and turns it into the real cosine spectrum
// the beam convolution needs (joxsz_funcs.py:464) with a QUARTER-length complex FFT:
//
//   A distinct row is even about the map centre: x[n] = q[|n|], |n| <= amax, and what the FIR/GEMM stage wants is
//       R(k) = sum_n x[n] cos(2 pi k n / P),   k < kact                    (P = 2 LP: padded transform length).
//   With e[m] = x[2m] (even) and d[m] = x[2m+1] - x[2m-1] (odd), y = e + d is a REAL sequence of length LP whose
//   spectrum is Y[k] = A[k] + 2i sin(2 pi k/P) B[k], A/B the (real) transforms of the even-/odd-indexed samples, and
//       R(k) = A[k] + B[k],   R(LP - k) = A[k] - B[k]                      (Cooley, Lewis & Welch 1970).
//   The real sequence y goes through a complex FFT of length Q = LP/2 = P/4 (z[j] = y[2j] + i y[2j+1]) and the usual
//   split.  B[0] (the plain sum of the odd-indexed samples) is the one term the transform cannot give; it is summed
//   beside the FFT by the wave that has no FFT task.  scripts/proto/dct_quarter.py is the numpy statement of the steps.
//
// Work layout (Q = L1 L2, two-level FFT with both factors in registers, jx_regfft.hpp; TPR = max(L1, L2)):
//   E     every wave takes the block's walkers in turn; lane g owns the four samples a = 4g..4g+3 of the row (table
//         entries in registers, reused for all walkers; coefficient address = scalar walker base + per-lane offset),
//         gets its left neighbour's last three by a one-lane DPP shift, and writes z[g] and z[Q-g] (16-byte LDS stores).
//   A/B   thread (walker, n2) / (walker, k1): the two FFT levels, in place in the walker's LDS row.
//   post  lane = (4 consecutive k) x (16 walkers): Y[k], Y[Q-k] from Z[k], Z[Q-k]; up to four outputs per pair, stored
//         walker-minor Rt[k][u][w] (128-byte runs) for the GEMM.
// Persistent over rows: block (walker group, row class rc) takes rows u = rc, rc + nrc, ...; walker groups that share
// blockIdx % 8 share an XCD and so an L2 that holds their coefficients (speed only).
#pragma once
#include <hip/hip_runtime.h>
#include "jx_conv.hpp"

struct JxDct {
    int NU;                         // distinct rows
    int nrc;                        // row classes (gridDim = 8 * ceil(ngroups / 8) * nrc)
    int n;                          // walkers of this launch
    int kact;                       // outputs k < kact are stored
    int gl;                         // last group of four samples that holds data: floor(amax / 4)  (== the kernel's GL)
    int na4;                        // table entries per row: 256 * npass
    int has_x0;                     // even map side: the unpaired column 0 (a = S/2) goes to x0t, not into the transform
    int N;                          // radial grid points (slot N = zero outside the grid)
    long long cf_ws;                // doubles per walker in cf
    long long tW, tKU;              // walker stride / padded row count of Rt
    const int* dk;                  // [NU][na4] byte offset (slot * 32) of each sample's interval
    const double* dt;               // [NU][na4] local abscissa
    const int* x0k; const double* x0t_t;   // [NU] the same for column 0
    const cplx* tw_q;               // [Q] e^{-2 pi i n / Q}
    const double* pk;               // [Q/2 + 1][4]: cos/2, -sin/2 of 2 pi k / LP; 1/(2 sin(2 pi k/P)) (0 for k = 0); 1/(2 sin(2 pi (Q-k)/P))
};

template <int Q> struct jx_dct_lay {
    static constexpr int L1 = jx_plan2<Q>::L1, L2 = jx_plan2<Q>::L2, L2P = L2 | 1;
    static constexpr int TPR = L1 > L2 ? L1 : L2;
    static constexpr int BASE = L1 * L2P > Q ? L1 * L2P : Q;
    // the dump slot of lanes with nothing to store: the padding slot (n1 = 0, n2 = L2) when the inner stride is padded, else one more
    static constexpr bool PAD = L2P != L2;
    static constexpr int DUMP = PAD ? L2 : BASE, SPAN = PAD ? BASE : BASE + 1;
    // row stride (16-byte slots) congruent to TPR modulo 16: with thread = walker * TPR + i the slot index of every
    // access below is thread + const (mod 16), i.e. no two lanes of a 16-lane group share a bank
    static constexpr int RS = SPAN + ((TPR - SPAN) % 16 + 16) % 16;
    static constexpr int zslot(int j) { return (j / L2) * L2P + (j % L2); }
};

// geometry of the evaluation phase for AMAX + 1 samples per row (compile time: LS2 = AMAX + 1)
template <int Q, int NS> struct jx_dct_geo {
    static constexpr int AMAX = NS - 1, GL = AMAX / 4;
    static constexpr int NPASS = (GL + 1 + 63) / 64;
    static constexpr int GW = (64 * NPASS - 1 < Q / 2) ? 64 * NPASS - 1 : Q / 2;    // last group written by its own lane
    static constexpr bool TAIL = (GL == GW) && (GW < Q / 2);                          // group GW + 1 is written by lane GW
    static constexpr int ZLO = TAIL ? GW + 2 : GW + 1, NZFILL = Q - 2 * ZLO + 1;       // slots ZLO .. Q - ZLO stay zero
    static_assert(GL <= GW, "every group with samples has a lane");
    static_assert(GL + 2 <= Q / 2, "z[g] and z[Q-g] of g = 0..GL+1 stay in their own halves");
};

__device__ __forceinline__ double jx_cubic(const char* cfw, unsigned koff, double t) {
    const double2 c0 = *reinterpret_cast<const double2*>(cfw + koff);
    const double2 c1 = *reinterpret_cast<const double2*>(cfw + koff + 16);
    return fma(t, fma(t, fma(t, c1.y, c1.x), c0.y), c0.x);
}

// lane i gets `src` of lane i-1; lane 0 keeps `own0` (DPP wave_shr:1 without bound control leaves the destination alone)
__device__ __forceinline__ double jx_shr1(double own0, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(own0), __double2loint(src), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(own0), __double2hiint(src), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double jx_lane63(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

template <int LP, int NS, int NW, int NT>
__global__ void __launch_bounds__(NT)
jx_rowdct_kernel(JxDct d, const double* __restrict__ cf, double* __restrict__ Rt, double* __restrict__ x0t) {
    constexpr int Q = LP / 2;
    typedef jx_dct_lay<Q> Lay;
    typedef jx_dct_geo<Q, NS> Geo;
    constexpr int L1 = Lay::L1, L2 = Lay::L2, L2P = Lay::L2P, TPR = Lay::TPR, RS = Lay::RS;
    constexpr int NWAVE = NT / 64, NPASS = Geo::NPASS, WPW = (NW + NWAVE - 1) / NWAVE;
    constexpr int DUMP = Lay::DUMP;
    static_assert(NW * TPR <= NT, "one FFT task per thread");
    static_assert(NW == 16, "the post-processing lane map assumes 16 walkers per block");
    static_assert(NW * TPR <= 64 * (NWAVE - 1) || NWAVE == 1, "the last wave has no FFT task (it sums B[0])");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    cplx* M = reinterpret_cast<cplx*>(sm);                   // [NW][RS]
    cplx* tw = M + NW * RS;                                  // [Q]
    double* s_pk = reinterpret_cast<double*>(tw + Q);        // [Q/2 + 1][4]
    double* s_bs = s_pk + 4 * (Q / 2 + 1);                   // [NW][64] per-lane sums of the odd-indexed samples
    double* s_b0 = s_bs + NW * 64;                           // [NW] B[0]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (walker group, row class); groups with equal blockIdx % 8 share an XCD
    const int ngroups = (d.n + NW - 1) / NW, gp8 = (ngroups + 7) >> 3;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int grp = (seq % gp8) * 8 + xcd, rc = seq / gp8;
    if (grp >= ngroups) return;
    const int w0 = grp * NW;

    for (int i = tid; i < Q; i += NT) tw[i] = d.tw_q[i];
    for (int i = tid; i < 4 * (Q / 2 + 1); i += NT) s_pk[i] = d.pk[i];

    // ---- per-thread constants of the evaluation: LDS byte offsets (within a walker's row) of this lane's outputs
    unsigned e_j1[NPASS], e_j2[NPASS];
    bool e_on[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int g = lane + 64 * p;
        e_on[p] = g <= Geo::GW;
        const int gc = min(g, Geo::GW), gq = Q - gc;
        e_j1[p] = 16u * (unsigned)((2 * gc < Q) ? Lay::zslot(gc) : DUMP);           // first kind: j = g <= (Q-1)/2
        e_j2[p] = 16u * (unsigned)((gc > 0) ? Lay::zslot(gq) : DUMP);               // second kind: j = Q - g, g >= 1
    }
    constexpr unsigned e_j1t = 16u * Lay::zslot(Geo::GW + 1), e_j2t = 16u * Lay::zslot(Q - Geo::GW - 1);
    const bool e_tail = Geo::TAIL && lane == (Geo::GW & 63);
    // walker bases of this wave (byte pointers; a missing walker repeats the last one and is never stored)
    const char* cfb[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) cfb[i] = reinterpret_cast<const char*>(cf + (size_t)min(w0 + min(wq + NWAVE * i, NW - 1), d.n - 1) * d.cf_ws);

    // ---- FFT roles
    const int frow = tid / TPR, fidx = tid - frow * TPR;
    const bool actA = frow < NW && fidx < L2, actB = frow < NW && fidx < L1;
    cplx* Mrow = M + frow * RS;

    // ---- post-processing: 4 consecutive k x 16 walkers per wave
    constexpr int KPI = 4 * NWAVE, NIT = (Q / 2 + 1 + KPI - 1) / KPI;
    const int pkk = lane & 3, pw = lane >> 2;
    const bool wok = w0 + pw < d.n;
    const unsigned kstr8 = (unsigned)(d.tKU * d.tW * 8);      // bytes between consecutive k of Rt (k * kstr8 < 2^32: checked on the host)
    char* Rw = reinterpret_cast<char*>(Rt + w0 + pw);
    const cplx* Mw_post = M + pw * RS;

    for (int u = rc; u < d.NU; u += d.nrc) {
        // ---------------- E: evaluate the row of every walker, build z ----------------
        int4 kb[NPASS];
        double tq[NPASS][4];
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const size_t e = (size_t)u * d.na4 + 4 * (lane + 64 * p);
            kb[p] = *reinterpret_cast<const int4*>(d.dk + e);
            const double2 ta = *reinterpret_cast<const double2*>(d.dt + e), tb = *reinterpret_cast<const double2*>(d.dt + e + 2);
            tq[p][0] = ta.x; tq[p][1] = ta.y; tq[p][2] = tb.x; tq[p][3] = tb.y;
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int w = wq + NWAVE * i;                     // wave-uniform
            if (w < NW) {
                char* Mw = reinterpret_cast<char*>(M + w * RS);
                double bs = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;    // carry: samples of lane 63 of the previous pass
#pragma unroll
                for (int p = 0; p < NPASS; ++p) {
                    const double v0 = jx_cubic(cfb[i], (unsigned)kb[p].x, tq[p][0]), v1 = jx_cubic(cfb[i], (unsigned)kb[p].y, tq[p][1]);
                    const double v2 = jx_cubic(cfb[i], (unsigned)kb[p].z, tq[p][2]), v3 = jx_cubic(cfb[i], (unsigned)kb[p].w, tq[p][3]);
                    // left neighbour's last three samples q[4g-1], q[4g-2], q[4g-3]; q[-a] = q[a] at the centre
                    const double m1 = jx_shr1(p == 0 ? v1 : c3, v3), m2 = jx_shr1(p == 0 ? v2 : c2, v2), m3 = jx_shr1(p == 0 ? v3 : c1, v1);
                    if (p + 1 < NPASS) { c1 = jx_lane63(v1); c2 = jx_lane63(v2); c3 = jx_lane63(v3); }
                    bs += v1 + v3;
                    if (e_on[p]) {
                        *reinterpret_cast<double2*>(Mw + e_j1[p]) = make_double2(v0 + v1 - m1, v2 + v3 - v1);
                        *reinterpret_cast<double2*>(Mw + e_j2[p]) = make_double2(v0 - v1 + m1, m2 - m1 + m3);
                    }
                    if (Geo::TAIL && p == NPASS - 1 && e_tail) {   // the group behind the last one has no samples of its own
                        *reinterpret_cast<double2*>(Mw + e_j1t) = make_double2(-v3, 0.0);
                        *reinterpret_cast<double2*>(Mw + e_j2t) = make_double2(v3, v2 - v3 + v1);
                    }
                }
#pragma unroll
                for (int e0 = 0; e0 < Geo::NZFILL; e0 += 64) {
                    const int j = Geo::ZLO + e0 + lane;
                    if (j <= Q - Geo::ZLO) *reinterpret_cast<double2*>(Mw + 16 * ((j / L2) * L2P + (j % L2))) = make_double2(0.0, 0.0);
                }
                s_bs[w * 64 + lane] = bs;
            }
        }
        if (d.has_x0 && tid < NW && w0 + tid < d.n)
            x0t[(size_t)u * d.tW + w0 + tid] = jx_cubic(reinterpret_cast<const char*>(cf + (size_t)(w0 + tid) * d.cf_ws), (unsigned)d.x0k[u], d.x0t_t[u]);
        __syncthreads();

        // ---------------- FFT of length Q, two levels, in place; the last wave sums B[0] meanwhile ----------------
        if (actA) {
            jx_c x[L1];
#pragma unroll
            for (int n1 = 0; n1 < L1; ++n1) x[n1] = jx_ld(Mrow + n1 * L2P + fidx);
            jx_stepA_store<Q, false>(x, fidx, Mrow, tw);
        }
        if (wq == NWAVE - 1) {
            // lane = (walker, quarter): 16 partial sums each, then two exchanges inside the quad
            const double2* bp = reinterpret_cast<const double2*>(s_bs + (lane >> 2) * 64 + (lane & 3) * 16);
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const double2 v = bp[i]; a0 += v.x; a1 += v.y; }
            double a = a0 + a1;
            a += __shfl_xor(a, 1, 64);
            a += __shfl_xor(a, 2, 64);
            if ((lane & 3) == 0) s_b0[lane >> 2] = 2.0 * a;
        }
        __syncthreads();
        {
            jx_c y[L2];
            if (actB) jx_stepB_load<Q, false>(y, fidx, Mrow);
            __syncthreads();
            if (actB) {
#pragma unroll
                for (int k2 = 0; k2 < L2; ++k2) jx_st(Mrow + fidx + L1 * k2, y[k2]);
            }
        }
        __syncthreads();

        // ---------------- split, R(k) = A + B, walker-minor stores ----------------
        {
            const double b0 = s_b0[pw];
            char* Ru = Rw + (size_t)u * d.tW * 8;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int k = it * KPI + 4 * wq + pkk;
                if (k <= Q / 2) {
                    const cplx zk = Mw_post[k], zq = Mw_post[k == 0 ? 0 : Q - k];
                    const double2 pa = *reinterpret_cast<const double2*>(s_pk + 4 * k), pb = *reinterpret_cast<const double2*>(s_pk + 4 * k + 2);
                    const double sx = zk.x + zq.x, sy = zk.y - zq.y, dx = zk.x - zq.x, dy = zk.y + zq.y;   // s = zk + conj zq, dd = zk - conj zq
                    const double tx = fma(pa.x, dx, -pa.y * dy), ty = fma(pa.x, dy, pa.y * dx);           // t/2 = (w_k / 2) dd
                    const double Ak = fma(0.5, sx, ty), Aq = fma(0.5, sx, -ty), Ik = fma(0.5, sy, -tx), Iq = fma(-0.5, sy, -tx);
                    const double bk = (k == 0) ? b0 : pb.x * Ik, bq = pb.y * Iq;
                    if (wok) {
                        const unsigned ok = (unsigned)k * kstr8;
                        if (k < d.kact) *reinterpret_cast<double*>(Ru + ok) = Ak + bk;
                        if (LP - k < d.kact) *reinterpret_cast<double*>(Ru + (unsigned)(LP - k) * kstr8) = Ak - bk;
                        if (2 * k != Q) {
                            if (Q - k < d.kact) *reinterpret_cast<double*>(Ru + (unsigned)(Q - k) * kstr8) = Aq + bq;
                            if (k > 0 && Q + k < d.kact) *reinterpret_cast<double*>(Ru + (unsigned)(Q + k) * kstr8) = Aq - bq;
                        }
                    }
                }
            }
        }
        __syncthreads();                                        // the rows are free for the next u
    }
}
