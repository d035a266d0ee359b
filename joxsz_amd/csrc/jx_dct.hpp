// Pass 1 of the default route, second form: the Compton-y map is never written.  A block takes the spline ordinates
// and moments (y_k, M_k) of NW walkers (left in HBM/L2 by the Abel kernel, 16 bytes per knot), evaluates ONE distinct
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
//   E     every wave takes the block's walkers in turn.  Lane l evaluates the samples a = l, l + 64, l + 128, l + 192 (so
//         that one load instruction of the wave covers 64 neighbouring samples, i.e. a few cache lines of the walker's
//         (y, M) array: the vector-memory pipe, not arithmetic, bounds this phase); table entries live in registers
//         and serve all walkers; address = scalar walker base + per-lane offset.  The samples pass through the
//         walker's own (still unused) LDS row to the lane that owns the group a = 4g..4g+3, which writes z[g], z[Q-g].
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
    long long s_kstr; int n_in;     // MODE 1: doubles between consecutive samples of the source array, valid samples (the rest are 0)
    long long tW, tKU;              // walker stride / padded row count of Rt
    const int* dk;                  // [NU][na4] byte offset (16 k) of each sample's interval in a walker's (y, M) array
    const double* dw;               // [NU][na4][4] weights of y_k, y_{k+1}, M_k, M_{k+1}
    const int* x0k; const double* x0w;     // [NU], [NU][4] the same for column 0
    const cplx* tw_q;               // [Q] e^{-2 pi i n / Q}
    int dbg;                        // timing experiments (JOXSZ_DCT_DBG): 1 = perfectly coalesced sample addresses (wrong results)
    unsigned long long* stamps;     // diagnostic build only (JOXSZ_DCT_STAMPS=1): [blocks][8] cycle counts of the phases of wave 0
    const double* pk;               // [Q/2 + 1][4]: cos/2, -sin/2 of 2 pi k / LP; 1/(2 sin(2 pi k/P)) (0 for k = 0); 1/(2 sin(2 pi (Q-k)/P))
};

// geometry of the evaluation phase for NS = amax + 1 samples per row
#ifndef JX_DCT_EW
#define JX_DCT_EW 8                // walkers whose spline requests are in flight together in the evaluation phase
#endif
template <int Q, int NS> struct jx_dct_geo {
    static constexpr int AMAX = NS - 1, GL = AMAX / 4;
    static constexpr int NPASS = (GL + 1 + 63) / 64;
    static constexpr int GW = (64 * NPASS - 1 < Q / 2) ? 64 * NPASS - 1 : Q / 2;    // last group written by its own lane
    static constexpr bool TAIL = (GL == GW) && (GW < Q / 2);                          // group GW + 1 is written by lane GW
    static constexpr int ZLO = TAIL ? GW + 2 : GW + 1, NZFILL = Q - 2 * ZLO + 1;      // slots ZLO .. Q - ZLO stay zero
    static constexpr int NEV = 4 * GW + 4;                                            // samples a < NEV are evaluated (zero weights past amax)
    // FULL: samples up to the Nyquist index a = LP = 2Q (NS = 2Q + 1).  The last group g = Q/2 then also needs q[LP + j] =
    // q[LP - j], j = 1..3 (the sequence is even about LP as well as about 0), which the loader provides.
    static constexpr bool FULL = NS == 2 * Q + 1;
    static_assert(GL <= GW, "every group with samples has a lane");
    static_assert(FULL || GL + 2 <= Q / 2, "z[g] and z[Q-g] of g = 0..GL+1 stay in their own halves");
    static_assert(!FULL || Q % 2 == 0, "the full-range form assumes an even quarter length");
};

template <int Q, int NS> struct jx_dct_lay {
    static constexpr int L1 = jx_plan2<Q>::L1, L2 = jx_plan2<Q>::L2, L2P = L2 | 1;
    static constexpr int TPR = L1 > L2 ? L1 : L2;
    static constexpr int BASE = L1 * L2P > Q ? L1 * L2P : Q;
    // the dump slot of lanes with nothing to store: the padding slot (n1 = 0, n2 = L2) when the inner stride is padded, else one more
    static constexpr bool PAD = L2P != L2;
    static constexpr int DUMP = PAD ? L2 : BASE, SPAN0 = PAD ? BASE : BASE + 1;
    // the row also holds the walker's NEV samples (8 bytes each) during the evaluation
    static constexpr int NEVS = (jx_dct_geo<Q, NS>::NEV + 1) / 2, SPAN = SPAN0 > NEVS ? SPAN0 : NEVS;
    // row stride (16-byte slots) congruent to TPR modulo 16: with thread = walker * TPR + i the slot index of every
    // access below is thread + const (mod 16), i.e. no two lanes of a 16-lane group share a bank
    static constexpr int RS = SPAN + ((TPR - SPAN) % 16 + 16) % 16;
    static constexpr int zslot(int j) { return (j / L2) * L2P + (j % L2); }
};

// f = A y_k + B y_{k+1} + C M_k + D M_{k+1} from the walker's (y, M) pairs
// (TS: storage type of the pairs; koff is the byte offset of knot k in an fp64 array, 16 k)
template <typename TS>
__device__ __forceinline__ double jx_spline4(const char* ym, unsigned koff, double A, double B, double C, double D) {
    typedef typename jx_pair<TS>::type S2;
    const unsigned ko = koff / (16 / (unsigned)sizeof(S2));
    const S2 p0 = *reinterpret_cast<const S2*>(ym + ko);
    const S2 p1 = *reinterpret_cast<const S2*>(ym + ko + sizeof(S2));
    return fma(D, (double)p1.y, fma(C, (double)p0.y, fma(B, (double)p1.x, A * (double)p0.x)));
}

// MODE 0: the samples of row u are evaluated from the walkers' spline arrays cf (pass 1).
// MODE 1: the samples are read from a walker-minor array cf[k][u][w] (k < n_in; row stride tW, sample stride s_kstr): the
//         same transform taken of a band-limited real-even spectrum is its inverse (odd map sides: combined rows back to
//         real space, joxsz_funcs.py:464-467 without a transform of the odd length S).
// T: arithmetic and storage type of the spline arrays, the samples, the transform and the output (double: the reference's;
//    float: the fp32 variant of BASELINE configs[4] -- the tables stay fp64 and are rounded as they are loaded).  The evaluation
//    is bound by the bytes the vector L1 returns to the registers: 32 per (sample, walker) in fp64, 16 in fp32.
template <int LP, int NS, int NW, int NT, int MODE, typename T = double>
__global__ void __launch_bounds__(NT)
jx_rowdct_kernel(JxDct d, const void* __restrict__ src_v, void* __restrict__ out_v, void* __restrict__ x0t_v) {
    typedef typename jx_pair<T>::type T2;
    const T* cf = reinterpret_cast<const T*>(src_v);                // MODE 0: spline arrays (y_k, M_k), stored as T
    T* Rt = reinterpret_cast<T*>(out_v);
    T* x0t = reinterpret_cast<T*>(x0t_v);
    constexpr int Q = LP / 2;
    typedef jx_dct_lay<Q, NS> Lay;
    typedef jx_dct_geo<Q, NS> Geo;
    constexpr int L1 = Lay::L1, L2 = Lay::L2, L2P = Lay::L2P, TPR = Lay::TPR, RS = Lay::RS;
    constexpr int NWAVE = NT / 64, NPASS = Geo::NPASS, WPW = (NW + NWAVE - 1) / NWAVE;
    constexpr int DUMP = Lay::DUMP;
    constexpr bool LEAN = LP >= 576;
    static_assert(NW * TPR <= NT, "one FFT task per thread");
    static_assert(NW == 16 || NW == 8, "the post-processing lane map: 4 k x 16 walkers or 8 k x 8 walkers per wave");
    static_assert(NW * TPR <= 64 * (NWAVE - 1) || NWAVE == 1, "the last wave has no FFT task (it sums B[0])");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    T2* M = reinterpret_cast<T2*>(sm);                       // [NW][RS]
    T2* tw = M + NW * RS;                                    // [Q]
    T* s_pk = reinterpret_cast<T*>(tw + Q);                  // [Q/2 + 1][4]
    T* s_bs = s_pk + 4 * (Q / 2 + 1);                        // [NW][64] per-lane sums of the odd-indexed samples
    T* s_b0 = s_bs + NW * 64;                                // [NW] B[0]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (walker group, row class); groups with equal blockIdx % 8 share an XCD
    const int ngroups = (d.n + NW - 1) / NW, gp8 = (ngroups + 7) >> 3;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int grp = (seq % gp8) * 8 + xcd, rc = seq / gp8;
    if (grp >= ngroups) return;
    const int w0 = grp * NW;

    for (int i = tid; i < Q; i += NT) { const cplx t = d.tw_q[i]; T2 v; v.x = (T)t.x; v.y = (T)t.y; tw[i] = v; }
    for (int i = tid; i < 4 * (Q / 2 + 1); i += NT) s_pk[i] = (T)d.pk[i];

    // ---- per-thread constants of the evaluation: LDS byte offsets (within a walker's row) of this lane's outputs
    unsigned e_j1[NPASS], e_j2[NPASS];
    bool e_on[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int g = lane + 64 * p;
        e_on[p] = g <= Geo::GW;
        const int gc = min(g, Geo::GW), gq = Q - gc;
        e_j1[p] = (unsigned)sizeof(T2) * (unsigned)((2 * gc < Q) ? Lay::zslot(gc) : DUMP);   // first kind: j = g <= (Q-1)/2
        e_j2[p] = (unsigned)sizeof(T2) * (unsigned)((gc > 0) ? Lay::zslot(gq) : DUMP);       // second kind: j = Q - g, g >= 1
    }
    constexpr unsigned e_j1t = (unsigned)sizeof(T2) * Lay::zslot(Geo::GW + 1), e_j2t = (unsigned)sizeof(T2) * Lay::zslot(Q - Geo::GW - 1);
    const bool e_tail = Geo::TAIL && lane == (Geo::GW & 63);
    // ---- FFT roles
    const int frow = tid / TPR, fidx = tid - frow * TPR;
    const bool actA = frow < NW && fidx < L2, actB = frow < NW && fidx < L1;
    T2* Mrow = M + frow * RS;

    // ---- post-processing: 4 consecutive k x 16 walkers per wave
    constexpr int KPW = 64 / NW;                                 // consecutive k per wave and trip
    constexpr int KPI = KPW * NWAVE, NIT = (Q / 2 + 1 + KPI - 1) / KPI;
    const int pkk = lane & (KPW - 1), pw = lane / KPW;
    const bool wok = w0 + pw < d.n;
    const unsigned kstr8 = (unsigned)(d.tKU * d.tW * sizeof(T));   // bytes between consecutive k of Rt (k * kstr8 < 2^32: checked on the host)
    char* Rw = reinterpret_cast<char*>(Rt + w0 + pw);
    const T2* Mw_post = M + pw * RS;

    // evaluation (MODE 0): byte offset of every walker's spline array from the group's first (wave-uniform -> scalar registers)
    const char* cf0 = reinterpret_cast<const char*>(cf) + (size_t)w0 * d.cf_ws * sizeof(T);
    unsigned woff[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) woff[i] = (unsigned)min(i, d.n - 1 - w0) * (unsigned)(d.cf_ws * sizeof(T));   // a missing walker repeats the last one: never stored
    constexpr int NSLT = (MODE == 0) ? (Geo::NEV + NT - 1) / NT : 1;
    unsigned t_kb[NSLT];
    double2 t_wa[NSLT], t_wb[NSLT];
    if (MODE == 0 && rc < d.NU) {
#pragma unroll
        for (int sl = 0; sl < NSLT; ++sl) {
            const size_t e = (size_t)rc * d.na4 + min(tid + NT * sl, Geo::NEV - 1);
            t_kb[sl] = (unsigned)d.dk[e];
            t_wa[sl] = *reinterpret_cast<const double2*>(d.dw + 4 * e);
            t_wb[sl] = *reinterpret_cast<const double2*>(d.dw + 4 * e + 2);
        }
    }
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0;
    const bool stamping = d.stamps != nullptr;
#define JX_STAMP(i) if (stamping) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; }
    if (stamping) st_t = __builtin_amdgcn_s_memtime();
    for (int u = rc; u < d.NU; u += d.nrc) {
        // ---------------- E: the samples of row u of every walker, then z ----------------
        // z[g], z[Q-g] of walker w from its samples q[a] in the walker's LDS row (wave-private: LDS operations of one wave
        // complete in order, so the reads see the stores before them and the z stores come after every read)
        auto build_z = [&](int w) {
            char* Mw = reinterpret_cast<char*>(M + w * RS);
            const T* qa = reinterpret_cast<const T*>(Mw);
            T2 qv[NPASS][3];
            T q3[NPASS];
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const int g = min(lane + 64 * p, Geo::GW), gm = max(g, 1);
                qv[p][0] = *reinterpret_cast<const T2*>(qa + 4 * g);          // q[4g], q[4g+1]
                qv[p][1] = *reinterpret_cast<const T2*>(qa + 4 * g + 2);      // q[4g+2], q[4g+3]
                qv[p][2] = *reinterpret_cast<const T2*>(qa + 4 * gm - 2);     // q[4g-2], q[4g-1]
                q3[p] = qa[4 * gm - 3];                                        // q[4g-3]
            }
            T bs = 0;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const T v0 = qv[p][0].x, v1 = qv[p][0].y, v2 = qv[p][1].x, v3 = qv[p][1].y;
                const bool centre = (p == 0) && lane == 0;                          // q[-a] = q[a]
                const T m1 = centre ? v1 : qv[p][2].y, m2 = centre ? v2 : qv[p][2].x, m3 = centre ? v3 : q3[p];
                if (!Geo::FULL || lane + 64 * p < Q / 2) bs += v1 + v3;   // (the group at the Nyquist index holds mirror copies)
                if (e_on[p]) {
                    T2 za, zb;
                    za.x = v0 + v1 - m1; za.y = v2 + v3 - v1; zb.x = v0 - v1 + m1; zb.y = m2 - m1 + m3;
                    *reinterpret_cast<T2*>(Mw + e_j1[p]) = za;
                    *reinterpret_cast<T2*>(Mw + e_j2[p]) = zb;
                }
                if (Geo::TAIL && p == NPASS - 1 && e_tail) {   // the group behind the last one has no samples of its own
                    T2 za, zb;
                    za.x = -v3; za.y = 0; zb.x = v3; zb.y = v2 - v3 + v1;
                    *reinterpret_cast<T2*>(Mw + e_j1t) = za;
                    *reinterpret_cast<T2*>(Mw + e_j2t) = zb;
                }
            }
#pragma unroll
            for (int e0 = 0; e0 < Geo::NZFILL; e0 += 64) {
                const int j = Geo::ZLO + e0 + lane;
                T2 zz; zz.x = 0; zz.y = 0;
                if (j <= Q - Geo::ZLO) *reinterpret_cast<T2*>(Mw + sizeof(T2) * ((j / L2) * L2P + (j % L2))) = zz;
            }
            s_bs[w * 64 + lane] = bs;
        };
        // (two halves: all of a wave's walkers are read before any of them is written, so that the LDS round trips of the
        //  walkers overlap; the compiler cannot see that the walkers' rows do not alias)
        struct ZSrc { T2 qv[NPASS][3]; T q3[NPASS]; };
        auto read_q = [&](int w, ZSrc& zs) {
            const T* qa = reinterpret_cast<const T*>(M + w * RS);
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const int g = min(lane + 64 * p, Geo::GW), gm = max(g, 1);
                zs.qv[p][0] = *reinterpret_cast<const T2*>(qa + 4 * g);          // q[4g], q[4g+1]
                zs.qv[p][1] = *reinterpret_cast<const T2*>(qa + 4 * g + 2);      // q[4g+2], q[4g+3]
                zs.qv[p][2] = *reinterpret_cast<const T2*>(qa + 4 * gm - 2);     // q[4g-2], q[4g-1]
                zs.q3[p] = qa[4 * gm - 3];                                        // q[4g-3]
            }
        };
        auto write_z = [&](int w, const ZSrc& zs) {
            char* Mw = reinterpret_cast<char*>(M + w * RS);
            T bs = 0;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const T v0 = zs.qv[p][0].x, v1 = zs.qv[p][0].y, v2 = zs.qv[p][1].x, v3 = zs.qv[p][1].y;
                const bool centre = (p == 0) && lane == 0;                          // q[-a] = q[a]
                const T m1 = centre ? v1 : zs.qv[p][2].y, m2 = centre ? v2 : zs.qv[p][2].x, m3 = centre ? v3 : zs.q3[p];
                if (!Geo::FULL || lane + 64 * p < Q / 2) bs += v1 + v3;   // (the group at the Nyquist index holds mirror copies)
                if (e_on[p]) {
                    T2 za, zb;
                    za.x = v0 + v1 - m1; za.y = v2 + v3 - v1; zb.x = v0 - v1 + m1; zb.y = m2 - m1 + m3;
                    *reinterpret_cast<T2*>(Mw + e_j1[p]) = za;
                    *reinterpret_cast<T2*>(Mw + e_j2[p]) = zb;
                }
                if (Geo::TAIL && p == NPASS - 1 && e_tail) {   // the group behind the last one has no samples of its own
                    T2 za, zb;
                    za.x = -v3; za.y = 0; zb.x = v3; zb.y = v2 - v3 + v1;
                    *reinterpret_cast<T2*>(Mw + e_j1t) = za;
                    *reinterpret_cast<T2*>(Mw + e_j2t) = zb;
                }
            }
#pragma unroll
            for (int e0 = 0; e0 < Geo::NZFILL; e0 += 64) {
                const int j = Geo::ZLO + e0 + lane;
                T2 zz; zz.x = 0; zz.y = 0;
                if (j <= Q - Geo::ZLO) *reinterpret_cast<T2*>(Mw + sizeof(T2) * ((j / L2) * L2P + (j % L2))) = zz;
            }
            s_bs[w * 64 + lane] = bs;
        };
        auto build_z_all = [&]() {
            if constexpr (NPASS == 1) {
                ZSrc zs[WPW];
#pragma unroll
                for (int i = 0; i < WPW; ++i) {
                    const int w = wq + NWAVE * i;
                    if (NW % NWAVE == 0 || w < NW) read_q(w, zs[i]);
                }
#pragma unroll
                for (int i = 0; i < WPW; ++i) {
                    const int w = wq + NWAVE * i;
                    if (NW % NWAVE == 0 || w < NW) write_z(w, zs[i]);
                }
            } else {                                             // rows with more groups per lane: one walker at a time (registers)
#pragma unroll
                for (int i = 0; i < WPW; ++i) {
                    const int w = wq + NWAVE * i;
                    if (NW % NWAVE == 0 || w < NW) build_z(w);
                }
            }
        };
        if constexpr (MODE == 0) {
            // thread = sample slot: a = tid, tid + NT, ... (one load instruction of a wave covers 64 neighbouring samples:
            // a few cache lines of the walker's array); its table entry lives in registers and serves all NW walkers.
            // The walkers stream through in groups of EW: the (y, M) requests of a whole group are in flight together,
            // the next group's go out as soon as this one's registers are free.
            constexpr int NSL = (Geo::NEV + NT - 1) / NT, EW = JX_DCT_EW;
            static_assert(NW % EW == 0, "walker groups of the evaluation");
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) {
                const int a = tid + NT * sl;
                const bool live = a < Geo::NEV;
                const unsigned kb = t_kb[sl] / (16 / (unsigned)sizeof(T2));      // byte offset of the knot in an array of T pairs
                const T w0y = (T)t_wa[sl].x, w1y = (T)t_wa[sl].y, w0m = (T)t_wb[sl].x, w1m = (T)t_wb[sl].y;
                T* qdst = reinterpret_cast<T*>(M) + a;                        // + walker * (RS * sizeof(T2)): immediate offsets
                T2 ld[EW][2];
#pragma unroll
                for (int g0 = 0; g0 < NW; g0 += EW) {
#pragma unroll
                    for (int i = 0; i < EW; ++i) {
                        const unsigned off = kb + woff[g0 + i];               // scalar walker offset + per-lane slot offset (32 bits)
                        ld[i][0] = *reinterpret_cast<const T2*>(cf0 + off);
#ifdef JX_EXP_HALF_LOADS
                        ld[i][1] = ld[i][0];
#else
                        ld[i][1] = *reinterpret_cast<const T2*>(cf0 + off + sizeof(T2));
#endif
                    }
#pragma unroll
                    for (int i = 0; i < EW; ++i) {
                        const T v = fma(w1m, (T)ld[i][1].y, fma(w0m, (T)ld[i][0].y, fma(w1y, (T)ld[i][1].x, w0y * (T)ld[i][0].x)));
                        if (live) qdst[(g0 + i) * (RS * 2)] = v;               // (RS T2 slots = 2 RS samples per row)
                    }
                    __builtin_amdgcn_sched_barrier(0);            // (keeps the next group's requests behind this group's use: registers)
                }
            }
            // the next row's table entries are requested now and arrive while this row is transformed
            if (u + d.nrc < d.NU) {
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) {
                    const size_t e = (size_t)(u + d.nrc) * d.na4 + min(tid + NT * sl, Geo::NEV - 1);
                    t_kb[sl] = (unsigned)d.dk[e];
                    t_wa[sl] = *reinterpret_cast<const double2*>(d.dw + 4 * e);
                    t_wb[sl] = *reinterpret_cast<const double2*>(d.dw + 4 * e + 2);
                }
            }
            JX_STAMP(6)
            __syncthreads();
            JX_STAMP(7)
            build_z_all();
        } else {
            // lanes = NW walkers x NT/NW samples: 128-byte (64-byte) runs of the walker-minor source
            constexpr int KPT = NT / NW, NLD = (Geo::NEV + KPT - 1) / KPT;
            const int sw = tid & (NW - 1), sk = tid / NW;
            const T* sp = reinterpret_cast<const T*>(src_v) + (size_t)u * d.tW + min(w0 + sw, d.n - 1);
            T sv[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int k = sk + KPT * i, km = (Geo::FULL && k > LP) ? 2 * LP - k : k;     // even about the Nyquist index
                sv[i] = sp[(size_t)min(km, d.n_in - 1) * d.s_kstr];
            }
            T* qa = reinterpret_cast<T*>(M + sw * RS);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int k = sk + KPT * i, km = (Geo::FULL && k > LP) ? 2 * LP - k : k;
                if (k < Geo::NEV) qa[k] = (km < d.n_in) ? sv[i] : (T)0;
            }
            __syncthreads();
            build_z_all();
        }
        JX_STAMP(0)
        if (MODE == 0 && d.has_x0 && tid < NW && w0 + tid < d.n)
            x0t[(size_t)u * d.tW + w0 + tid] = (T)jx_spline4<T>(reinterpret_cast<const char*>(cf + (size_t)(w0 + tid) * d.cf_ws), (unsigned)d.x0k[u],
                                                             d.x0w[4 * u], d.x0w[4 * u + 1], d.x0w[4 * u + 2], d.x0w[4 * u + 3]);
        __syncthreads();
        JX_STAMP(1)

        // ---------------- FFT of length Q, two levels, in place; the last wave sums B[0] meanwhile ----------------
        T2* Mr = Mrow;
        int fi = fidx;
        if constexpr (LEAN) asm volatile("" : "+v"(Mr), "+v"(fi));
        if (actA) {
            jx_cT<T> x[L1];
#pragma unroll
            for (int n1 = 0; n1 < L1; ++n1) x[n1] = jx_ld(Mr + n1 * L2P + fi);
            jx_stepA_store<Q, false>(x, fi, Mr, (const T2*)tw);
        }
        if (wq == NWAVE - 1) {
            // lane = (walker, part): 64 / KPW partial sums each, then exchanges inside the group of KPW lanes
            const T2* bp = reinterpret_cast<const T2*>(s_bs + (lane / KPW) * 64 + (lane & (KPW - 1)) * (64 / KPW));
            T a0 = 0, a1 = 0;
#pragma unroll
            for (int i = 0; i < 32 / KPW; ++i) { const T2 v = bp[i]; a0 += v.x; a1 += v.y; }
            T a = a0 + a1;
#pragma unroll
            for (int m = 1; m < KPW; m <<= 1) a += __shfl_xor(a, m, 64);
            if ((lane & (KPW - 1)) == 0) s_b0[lane / KPW] = (T)2 * a;
        }
        JX_STAMP(2)
        __syncthreads();
        {
            jx_cT<T> y[L2];
            if (actB) jx_stepB_load<Q, false>(y, fi, (const T2*)Mr);
            __syncthreads();
            if (actB) {
#pragma unroll
                for (int k2 = 0; k2 < L2; ++k2) jx_st(Mr + fi + L1 * k2, y[k2]);
            }
        }
        JX_STAMP(3)
        __syncthreads();

        // ---------------- split, R(k) = A + B, walker-minor stores ----------------
        {
            const T b0 = s_b0[pw];
            // LEAN instances (long rows): the values below that do not depend on the row are recomputed per row instead of
            // living in registers across the whole loop -- the kernel otherwise spills (the short-row instances are faster
            // with them hoisted: 233 registers, no spill)
            int kb0 = KPW * wq + pkk;
            if constexpr (LEAN) asm volatile("" : "+v"(kb0));
            char* Ru = Rw + (size_t)u * d.tW * sizeof(T);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int k = it * KPI + kb0;
                if (k <= Q / 2) {
                    const T2 zk = Mw_post[k], zq = Mw_post[k == 0 ? 0 : Q - k];
                    const T2 pa = *reinterpret_cast<const T2*>(s_pk + 4 * k), pb = *reinterpret_cast<const T2*>(s_pk + 4 * k + 2);
                    const T sx = zk.x + zq.x, sy = zk.y - zq.y, dx = zk.x - zq.x, dy = zk.y + zq.y;   // s = zk + conj zq, dd = zk - conj zq
                    const T tx = fma(pa.x, dx, -pa.y * dy), ty = fma(pa.x, dy, pa.y * dx);            // t/2 = (w_k / 2) dd
                    const T Ak = fma((T)0.5, sx, ty), Aq = fma((T)0.5, sx, -ty), Ik = fma((T)0.5, sy, -tx), Iq = fma((T)-0.5, sy, -tx);
                    const T bk = (k == 0) ? b0 : pb.x * Ik, bq = pb.y * Iq;
                    if (wok && !(d.dbg & 2)) {
                        const unsigned ok = (unsigned)k * kstr8;
                        // (streamed out past the cache: these 0.35 GB per launch would otherwise push the walkers' spline arrays,
                        //  which every row re-reads, out of the XCD's L2)
                        if (k < d.kact) __builtin_nontemporal_store((T)(Ak + bk), reinterpret_cast<T*>(Ru + ok));
                        if (LP - k < d.kact) __builtin_nontemporal_store((T)(Ak - bk), reinterpret_cast<T*>(Ru + (unsigned)(LP - k) * kstr8));
                        if (2 * k != Q) {
                            if (Q - k < d.kact) __builtin_nontemporal_store((T)(Aq + bq), reinterpret_cast<T*>(Ru + (unsigned)(Q - k) * kstr8));
                            if (k > 0 && Q + k < d.kact) __builtin_nontemporal_store((T)(Aq - bq), reinterpret_cast<T*>(Ru + (unsigned)(Q + k) * kstr8));
                        }
                    }
                }
            }
        }
        JX_STAMP(4)
        __syncthreads();                                        // the rows are free for the next u
        JX_STAMP(5)
    }
    if (stamping && tid == 0) for (int i = 0; i < 8; ++i) d.stamps[(size_t)blockIdx.x * 8 + i] = st_acc[i];
#undef JX_STAMP
}
