// Device kernels of the JoXSZ log-posterior for gfx950 (CDNA4, wave64).  All arithmetic is IEEE fp64 like the reference
// (the fp32 variant of the contracted route narrows the spline arrays and the sample evaluation only).
//
//   jx_prep_kernel          theta -> parameter vector, priors, mass veto, pressure and T_SZ profiles,
//                           Compton->mJy/beam factors, X-ray counts + Cash likelihood (every call with taps; pow() form; calc_integ)
//   jx_walker2_kernel       the same work of the TIMED path as two lean roles per walker in one launch (same bits)
//   jx_abel_gemm_kernel     Abel integral, Compton-y scale and spline moments of a launch as one fp64 matrix-core product
//   jx_abel_map_sym_kernel  FUSED gNFW profile -> Abel integral -> Compton y -> cubic spline -> S x S map (symmetric
//   jx_abel_map_kernel      d_mat / any d_mat): the profile taps, the y_2d tap, the rocFFT sequence, the full-map measurement
//   jx_beam_mul_kernel      spectrum *= beam spectrum (rocFFT sequence on rocFFT's own 2-D plans)
//   jx_tail_kernel          rocFFT sequence: extracted row of the filtered map (from the window's spectrum, or from its column sums when the
//                           transforms are jx_fft.hpp's), conversion, chi^2, total
//   jx_operator_*_kernel    collapsed route (jx_set_route)
//   jx_sm_*_kernel          device-resident stretch move
// The exact form's kernels live in jx_exact.hpp, the literal sequence's transforms in jx_fft.hpp, the contracted forms' (rounds 3-4) in jx_mix.hpp.
//
// JOXSZ_DBG (JxDev::dbg; diagnostic build only, make ABLATIONS=1) holds timing-only ablation switches of the map kernel;
// results are wrong when any is set: 1 skip phases 1-4, 2 skip phase 5, 4 phase 5 = stores only,
// 8 skip the Abel sums, 16 skip the spline moments, 32 skip the pow() of the profile,
// 64 non-temporal stores (with 4), 128 Abel loop without the reciprocal square root.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "jx_fastmath.hpp"
#include <stdint.h>

// Timing-only ablations of the map kernel (JxDev::dbg, environment JOXSZ_DBG) exist in a diagnostic build only
// (make ABLATIONS=1): in the shipped library JX_DBG is a compile-time false and the kernels carry none of these branches.
#ifdef JOXSZ_ABLATIONS
#define JX_DBG(c, bits) ((c).dbg & (bits))
#else
#define JX_DBG(c, bits) false
#endif
#ifdef JOXSZ_ABLATIONS
#define JX_PSTAMP(c, k) do { if ((c).stamps && threadIdx.x == 0) (c).stamps[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define JX_PSTAMP(c, k) do { } while (0)
#endif
#define JX_MAX_PAR 19
// All LDS lives in the dynamic region, 16-byte aligned, with 16-byte-multiple carve offsets: a static
// __shared__ in front of it shifts the base and every ds_read/write_b128 is then replayed at ~64 cycles
// (cdna_hip_programming.md Guideline 17).  The first JX_LDS_HDR doubles hold the small per-block scalars.
#define JX_LDS_DECL extern __shared__ __attribute__((aligned(16))) double sm[]
#define JX_LDS_HDR 32          // [0..18] parameter vector, [20..27] reduction scratch, [28] int flag
#ifdef JX_PREP_THREADS_OVERRIDE
#define JX_PREP_THREADS JX_PREP_THREADS_OVERRIDE
#else
#define JX_PREP_THREADS 256
#endif
#define JX_TAIL_THREADS 256

// semantic slots of the parameter vector (joxsz_amd/problem.py PAR_SLOTS)
enum {
    P_LOGN0 = 0, P_BETA, P_LOGRC, P_LOGRS, P_ALPHA, P_EPS, P_GAMMA, P_LOGTR, P_Z,
    P_P0, P_A, P_B, P_C, P_RP, P_BACKSCALE, P_CALIB, P_LOGN02, P_BETA2, P_LOGRC2
};

enum { REJ_BOX = 1, REJ_MASS = 2, REJ_RCRS = 4, REJ_XRAY = 8 };

struct JxDev {
    // sizes
    int S, N, B, P, Ph, Sh, nrow, nt, nflux, nconv, nann, nband, ntab, npar, ndim;
    int ne_mode, exclude_unphy_mass, sz_only;
    int K;                       // half-bandwidth of the spline moment operator
    int map_split;               // row slabs per walker in the Abel+map kernel
    double y_scale;              // kpc_cm * sigma_T / m_e            (joxsz_funcs.py:459)
    double inv_h_mean;           // (N-1)/(r_N - r_1): first guess of the interval index in the generic map kernel
    double inv_dlnT;             // (ntab-1)/(lnT[ntab-1] - lnT[0]): first guess of the interval of the count-rate tables
    // constant tensors (device)
    const double* r_pp;          // [N]
    const double* d_mat;         // [S*S]
    long long img_ld, img_ws;    // row and walker strides (doubles) of the y-map image
    int dbg;                     // timing-only ablations (JOXSZ_DBG): 1 skip phases 1-4, 2 skip phase 5, 4 phase 5 stores only
    const double* fm_tab;        // [JX_FM_TABLE_DOUBLES] tables of jx_fastmath.hpp (jxt::fastmath_tables)
    int xr_split;                // 1: jx_prep_kernel runs as 2 n blocks -- n for everything but the X-ray side, n for the X-ray side alone -- and the tail adds the two
    double* xr_out;              // [chunk][2] Cash log-likelihood and its reject flag (xr_split)
    int prep_pow;                // 1 (JOXSZ_PREP_POW=1): the prep kernel evaluates the profiles with pow() as written in the reference
    const double* lr_pp;         // [N] log(r_pp)
    const double* sz_pack;       // jx_walker2_kernel: the tables of its two roles, each role's in one run (layouts at the kernel)
    const double* xr_pack;
    const double* inject_pp;     // operator build only: [nlaunch][N] pressure profiles that replace press_fun(theta) (else null)
    long long* stamps;           // diagnostic build only (make ABLATIONS=1, JOXSZ_X_STAMPS): [blocks][8] wall-clock stamps of jx_prep_kernel's phases
    int pp_ld;                   // doubles per walker of jx_prep_kernel's profile output (0: N; the exact form pads its rows to whole 16-radius steps)
    int fast_map, q_na, q_nb;    // symmetric-map form: table sizes (|ix-c|, |iy-c|)
    double* xcol;                // quad mode: copy of the quadrant's last column (map column 0): [chunk][q_nb], or walker-minor
    long long xcol_ld;           //   [q_nb padded][xcol_ld] when xcol_ld > 0 (fused FIR path)
    int pairw, nlaunch;          // quad mode: walkers per block (1 or 2: two coefficient sets share every table entry), walkers of this launch
    int quad;                    // 1: the image is the quadrant [q_nb][img_ld] (|iy-c|, |ix-c|) alone, nothing mirrored
    int calc_integ;              // joxsz_funcs.py:480-484: cint = integ_wp . press_fun(r_pp) (Simpson rule, spline value at 0, Compton
    const double* integ_wp;      //   scaling and Abel integral folded into one weight per radius); chi^2 term ((cint - mu)/sig)^2
    double integ_mu, integ_sig;
    double* cf_out;              // not null: the kernel stops after phase 3 and leaves the spline ordinates and moments (y_k, M_k),
    long long cf_ws;             //   k < N, here: [nlaunch][cf_ws] doubles (the contracted route evaluates the map samples from them)
    int cf_tr;                   //   1: walker-minor instead, cf_out[(k * cf_ws + w) * 2 + {0, 1}] (cf_ws = walker stride; contracted route)
    const int* q_k;              // [q_nb*q_na] coefficient slot of each quadrant radius
    const double* q_t;           // [q_nb*q_na] local abscissa within that slot
    const double* abel_tab;      // [N][4] (r_j, cj_j, dg_j, sp_j):  A[i][j] = cj_j / sqrt(r_j^2 - r_i^2) for j >= i+2,
                                 //        A[i][i] = dg_i (analytic end cell), A[i][i+1] = sp_i (half cell + end cell)
    const double* gband;         // [(2K+1)*N] gband[(k+K)*N+i] = G[i][i+k]
    const double* bhat;          // [P*Ph*2]
    const double* htab;          // [S*Sh*2]
    const double* twid;          // [S*2] cos, sin of 2 pi m / S
    const double* hw;            // [nt]   h(0) = sum hw[k] t[k]     (joxsz_funcs.py:470-473)
    const double* emat;          // [nflux*nrow]
    const double* flux;          // [3*nflux]
    const double* conv_T; const double* conv_v;   // [nconv]
    const double* par_vals; const double* par_min; const double* par_max;
    const double* par_mu; const double* par_sigma; const double* par_lnorm;   // par_lnorm = -log(sqrt(2 pi) sigma)
    const int* par_kind; const int* thawed_idx;
    const double* x_r_ne; const double* x_r_T; const double* projvols; const double* cts;
    const double* areascales; const double* exposures; const double* backrates;
    const double* geomarea; const double* lnT; const double* lnrate;
};

// ------------------------------------------------------------------------------------
// physics helpers
// ------------------------------------------------------------------------------------

// joxsz_funcs.py:275-287
__device__ __forceinline__ double jx_press(const double* p, double r) {
    const double x = r / p[P_RP];
    return p[P_P0] / (pow(x, p[P_C]) * pow(1.0 + pow(x, p[P_A]), (p[P_B] - p[P_C]) / p[P_A]));
}

// joxsz_funcs.py:375-395 (mydens_vikhFunction, single and double beta) with its radius-independent factors (five pow()
// of parameters only) taken out of the per-radius loops: pc = {n0^2, rc, rs, n02^2, rc2}
__device__ __forceinline__ void jx_ne_consts(const double* p, int mode, double* pc) {
    const double n0 = pow(10.0, p[P_LOGN0]);
    pc[0] = n0 * n0; pc[1] = pow(10.0, p[P_LOGRC]); pc[2] = pow(10.0, p[P_LOGRS]);
    const double n02 = (mode == 1) ? pow(10.0, p[P_LOGN02]) : 0.0;
    pc[3] = n02 * n02; pc[4] = (mode == 1) ? pow(10.0, p[P_LOGRC2]) : 1.0;
}
__device__ __forceinline__ double jx_ne_pc(const double* p, const double* pc, double r, int mode) {
    const double x = r / pc[1];
    double res = pc[0] * ((p[P_ALPHA] == 0.0) ? 1.0 : pow(x, -p[P_ALPHA])) /      // (alpha is frozen at 0 by default: x^-0 == 1 exactly)
                 (pow(1.0 + x * x, 3.0 * p[P_BETA] - p[P_ALPHA] / 2.0) *
                  pow(1.0 + pow(r / pc[2], p[P_GAMMA]), p[P_EPS] / p[P_GAMMA]));
    if (mode == 1) {
        const double x2 = r / pc[4];
        res += pc[3] / pow(1.0 + x2 * x2, 3.0 * p[P_BETA2]);
    }
    return sqrt(res);
}
// 1 / n_e: all the grid pass needs of the density (T_SZ = P / n_e, the mass profile ~ 1 / n_e).  Single-beta model: the square
// root and the division go into the exponent, n_e^-1 = n_0^-1 exp(+E / 2); double-beta: through jx_ne_log.
template <class M>
__device__ __forceinline__ double jx_inv_ne_log(const M& m, const double* p, const double* pl, double r, double lr, int mode) {
    if (mode == 1) return 1.0 / jx_ne_log(m, p, pl, r, lr, mode);
    const double x = r / pl[3];
    const double u = m.e(p[P_GAMMA] * (lr - pl[4]));
    return pl[10] * m.e(0.5 * (p[P_ALPHA] * (lr - pl[2]) + pl[5] * m.l(1.0 + x * x) + pl[6] * m.l(1.0 + u)));
}


// The two profiles in log form.  Every power in joxsz_funcs.py:275-287 and :375-395 has a positive base, so
//     press = P0 exp(-(c lx + ((b-c)/a) log(1 + x^a))),   x^a = exp(a lx),   lx = log r - log r_p
//     ne^2  = n0^2 exp(-(alpha lc + (3 beta - alpha/2) log(1 + (r/rc)^2) + (eps/gamma) log(1 + exp(gamma (log r - log r_s)))))
// (+ the second beta term): three log() and four exp() per radius instead of six pow(), which is what the prep kernel's
// time goes into.  The exponents are O(10) and carry an absolute error of a few 1e-16 each, so the values agree with
// the pow() forms to ~1e-14 relative (the pow() forms stay available: JOXSZ_PREP_POW=1).
// pl = {log r_p, (b-c)/a, log r_c, r_c, log r_s, 3 beta - alpha/2, eps/gamma, n0^2, n02^2, r_c2}
// exp / log of the profile chains: the device library's, or the table-driven pair of jx_fastmath.hpp (tables in LDS)
struct JxMathLib {
    __device__ __forceinline__ double e(double x) const { return exp(x); }
    __device__ __forceinline__ double l(double x) const { return log(x); }
};
struct JxMathTab {
    JxFm t;
    __device__ __forceinline__ double e(double x) const { return jx_fm_exp(t, x); }
    __device__ __forceinline__ double l(double x) const { return jx_fm_log(t, x); }
};
template <class M>
__device__ __forceinline__ void jx_prof_consts(const M& m, const double* p, int mode, double* pl) {
    const double ln10 = 2.30258509299404568402;
    pl[0] = m.l(p[P_RP]);
    pl[1] = (p[P_B] - p[P_C]) / p[P_A];
    pl[2] = p[P_LOGRC] * ln10;
    pl[3] = m.e(pl[2]);
    pl[4] = p[P_LOGRS] * ln10;
    pl[5] = 3.0 * p[P_BETA] - p[P_ALPHA] / 2.0;
    pl[6] = p[P_EPS] / p[P_GAMMA];
    pl[7] = m.e(2.0 * ln10 * p[P_LOGN0]);
    pl[8] = (mode == 1) ? m.e(2.0 * ln10 * p[P_LOGN02]) : 0.0;
    pl[9] = (mode == 1) ? m.e(ln10 * p[P_LOGRC2]) : 1.0;
    pl[10] = m.e(-ln10 * p[P_LOGN0]);              // 1 / sqrt(pl[7])
}
// pressure and x^a at radius r (lr = log r)
template <class M>
__device__ __forceinline__ double jx_press_log(const M& m, const double* p, const double* pl, double lr, double* xa_out) {
    const double lx = lr - pl[0];
    const double xa = m.e(p[P_A] * lx);
    *xa_out = xa;
    return p[P_P0] * m.e(-(p[P_C] * lx + pl[1] * m.l(1.0 + xa)));
}
template <class M>
__device__ __forceinline__ double jx_ne_log(const M& m, const double* p, const double* pl, double r, double lr, int mode) {
    const double x = r / pl[3];
    const double u = m.e(p[P_GAMMA] * (lr - pl[4]));
    double res = pl[7] * m.e(-(p[P_ALPHA] * (lr - pl[2]) + pl[5] * m.l(1.0 + x * x) + pl[6] * m.l(1.0 + u)));
    if (mode == 1) {
        const double x2 = r / pl[9];
        res += pl[8] * m.e(-3.0 * p[P_BETA2] * m.l(1.0 + x2 * x2));
    }
    return sqrt(res);
}

// linear interp1d with fill_value='extrapolate' (joxsz_main.py:109)
__device__ __forceinline__ double jx_convert(const JxDev& c, double T) {
    int hi = 1;
    while (hi < c.nconv - 1 && T > c.conv_T[hi]) ++hi;
    const int lo = hi - 1;
    const double slope = (c.conv_v[hi] - c.conv_v[lo]) / (c.conv_T[hi] - c.conv_T[lo]);
    return slope * (T - c.conv_T[lo]) + c.conv_v[lo];
}

// the same from copies of the two small tables (LDS): the search is a chain of dependent reads, a memory round trip each otherwise
__device__ __forceinline__ double jx_convert_tab(const double* ct, const double* cv, int nconv, double T) {
    int hi = 1;
    while (hi < nconv - 1 && T > ct[hi]) ++hi;
    const int lo = hi - 1;
    const double slope = (cv[hi] - cv[lo]) / (ct[hi] - ct[lo]);
    return slope * (T - ct[lo]) + cv[lo];
}

// np.interp(x, xp, fp) -- clamped linear interpolation -- for two tables on one grid: identical arithmetic per table, one
// interval search
__device__ __forceinline__ void jx_interp_clamped2(const double* xp, const double* f0, const double* f1, int n, double x, double inv_dx_mean,
                                                   double* o0, double* o1) {
    if (x != x) { *o0 = x; *o1 = x; return; }
    if (x <= xp[0]) { *o0 = f0[0]; *o1 = f1[0]; return; }
    if (x >= xp[n - 1]) { *o0 = f0[n - 1]; *o1 = f1[n - 1]; return; }
    // the interval [xp[lo], xp[lo + 1]) that holds x: a guess from the grid's mean spacing (exact for the uniform ln T grid of
    // mbproj2's count-rate tables, joxsz_funcs.py:669), corrected by comparisons -- the bisection's answer for any increasing
    // grid, in two dependent loads instead of seven
    int lo = (int)((x - xp[0]) * inv_dx_mean);
    lo = max(0, min(n - 2, lo));
    while (lo > 0 && xp[lo] > x) --lo;
    while (lo < n - 2 && xp[lo + 1] <= x) ++lo;
    const int hi = lo + 1;
    const double dx = xp[hi] - xp[lo], t = x - xp[lo];
    *o0 = (f0[hi] - f0[lo]) / dx * t + f0[lo];
    *o1 = (f1[hi] - f1[lo]) / dx * t + f1[lo];
}

// (nw: the waves that take part, the first nw of the block; every one of them must call -- two block barriers inside)
__device__ __forceinline__ double jx_block_sum(double v, double* red /*[>=nw]*/, int nw = 0) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (nw <= 0) nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < nw; ++k) t += red[k];
    return t;
}

__device__ __forceinline__ int jx_block_or(int v, int* redi) {
    __syncthreads();
    if (threadIdx.x == 0) *redi = 0;
    __syncthreads();
    if (v) atomicOr(redi, v);
    __syncthreads();
    return *redi;
}

// The stretch move folded into the likelihood's own kernels (jx_sample on the contracted route): the per-walker kernel draws the
// proposal of its walker itself (what jx_sm_propose_kernel does, same arithmetic, same random numbers) and the tail accepts or
// rejects it (jx_sm_accept_kernel) -- two launches fewer per half step.  on = 0: a plain evaluation.
struct JxSm {
    int on, ndim, half, s1, s2, iter2;
    double a;
    unsigned long long seed;
    double* x;                   // [W][ndim] positions of the whole ensemble
    double* q;                   // [half][ndim] proposals of this half step
    double* zz;                  // [half] stretch factors
    double* lp;                  // [W] log-posteriors of the ensemble
    long long* nacc;             // [W] acceptance counters
};
__device__ __forceinline__ void jx_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out);
__device__ __forceinline__ double jx_u01(uint32_t hi, uint32_t lo);

// full parameter vector of batch walker gw into p[0..JX_MAX_PAR) (LDS): the current values with the thawed ones
// replaced by theta (updateThawed, joxsz_funcs.py:516).  Ends with a barrier.
__device__ __forceinline__ void jx_load_params(const JxDev& c, const double* __restrict__ theta, int gw, double* p, const JxSm* sm = nullptr) {
    const int tid = threadIdx.x;
    // (all three requests first: one round trip to memory instead of two on the critical path of every per-walker kernel)
    const double pv = (tid < c.npar) ? c.par_vals[tid] : 0.0;
    const int ti = (tid < c.ndim) ? c.thawed_idx[tid] : -1;
    double tv = 0.0;
    if (sm && sm->on) {
        // the proposal of walker gw of this half (jx_sm_propose_kernel's arithmetic: separately rounded operations, replayable on the host)
        if (tid < c.ndim) {
            uint32_t r[4];
            jx_philox((uint32_t)gw, (uint32_t)sm->iter2, 0u, 0u, (uint32_t)sm->seed, (uint32_t)(sm->seed >> 32), r);
            const double u1 = jx_u01(r[0], r[1]), u2 = jx_u01(r[2], r[3]);
            const double t = __dadd_rn(__dmul_rn(sm->a - 1.0, u1), 1.0);
            const double z = __ddiv_rn(__dmul_rn(t, t), sm->a);
            int j = (int)__dmul_rn(u2, (double)sm->half);
            j = min(j, sm->half - 1);
            const double xp = sm->x[(size_t)(sm->s2 + j) * c.ndim + tid], xi = sm->x[(size_t)(sm->s1 + gw) * c.ndim + tid];
            tv = __dsub_rn(xp, __dmul_rn(__dsub_rn(xp, xi), z));
            sm->q[(size_t)gw * c.ndim + tid] = tv;
            if (tid == 0) sm->zz[gw] = z;
        }
    } else if (tid < c.ndim) tv = theta[(size_t)gw * c.ndim + tid];
    if (tid < JX_MAX_PAR) p[tid] = pv;
    __syncthreads();
    if (ti >= 0) p[ti] = tv;
    __syncthreads();
}

// ------------------------------------------------------------------------------------
// The X-ray side of a walker: calcProfiles + Cash (joxsz_funcs.py:527-532, 495-505; mbproj2 Fit.calcProfiles).  Called by every thread of a
// block (block barriers inside); its tables arrive through a struct of pointers -- global memory (jx_prep_kernel), or the copies an X-ray
// block of jx_walker2_kernel staged in LDS at its start (every table but the count-rate tables).  The terms of the Cash sum are added in
// ONE order whatever the block's size: 64 consecutive pairs per shuffle tree, the trees' sums in sequence.
// ------------------------------------------------------------------------------------
struct JxXrTab {
    const double *x_r_ne, *x_r_T, *lnT, *lnrate, *projvols, *areascales, *exposures, *backrates, *geomarea, *cts;
};
// dst[0, n) <- src[0, n) by the block's threads, U loads per thread requested before the first is stored (one trip to memory per U x threads)
template <int U>
__device__ __forceinline__ void jx_copy_batched(double* __restrict__ dst, const double* __restrict__ src, int n, int tid, int nth) {
    for (int i0 = 0; i0 < n; i0 += U * nth) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[min(i0 + u * nth + tid, n - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = i0 + u * nth + tid; if (i < n) dst[i] = v[u]; }
    }
}

template <bool POW, class M>
__device__ __forceinline__ void jx_xray_side(const JxDev& c, const JxXrTab& xt, const double* p, const double* pl, const double* pc, const M& mt, int tid, int nth,
                                             double* s_x, double* s_ne, double* s_T, double* s_rate, double* s_term, double* red, int* redi,
                                             size_t w, double* __restrict__ tap_xprofs, double* xlike_out, int* xbad_out) {
    constexpr bool logform = !POW;
    if (logform && 3 * c.nann <= nth) {
        // three jobs per shell side by side: density at the n_e radii, pressure at the T radii, density at the T radii (only
        // where the two radii differ), then T_X = P / n_e * 10^log(T_X/T_SZ)
        const int grp = tid / c.nann, k = tid - grp * c.nann;
        if (grp < 3) {
            const double rn = xt.x_r_ne[k], r = xt.x_r_T[k];
            if (grp == 0) s_ne[k] = jx_ne_log(mt, p, pl, rn, mt.l(rn), c.ne_mode);
            else if (grp == 1) { double xa; s_T[k] = jx_press_log(mt, p, pl, mt.l(r), &xa) * mt.e(2.30258509299404568402 * p[P_LOGTR]); }
            else s_x[k] = (rn == r) ? 0.0 : jx_ne_log(mt, p, pl, r, mt.l(r), c.ne_mode);
        }
        __syncthreads();
        if (tid < c.nann) s_T[tid] = s_T[tid] / ((xt.x_r_ne[tid] == xt.x_r_T[tid]) ? s_ne[tid] : s_x[tid]);   // T_X
    } else if (tid < c.nann) {
        const double rn = xt.x_r_ne[tid], r = xt.x_r_T[tid];
        if (logform) {
            double xa;
            const double lr = mt.l(r);
            s_ne[tid] = jx_ne_log(mt, p, pl, rn, (rn == r) ? lr : mt.l(rn), c.ne_mode);
            s_T[tid] = jx_press_log(mt, p, pl, lr, &xa) / jx_ne_log(mt, p, pl, r, lr, c.ne_mode) * mt.e(2.30258509299404568402 * p[P_LOGTR]);   // T_X
        } else {
            s_ne[tid] = jx_ne_pc(p, pc, rn, c.ne_mode);
            s_T[tid] = jx_press(p, r) / jx_ne_pc(p, pc, r, c.ne_mode) * pow(10.0, p[P_LOGTR]);   // T_X
        }
    }
    __syncthreads();
    JX_PSTAMP(c, 2);
    const int nba = c.nband * c.nann;
    for (int q = tid; q < nba; q += nth) {
        const int b = q / c.nann, j = q - b * c.nann;
        const double lt = mt.l(s_T[j]);
        const double* tab = xt.lnrate + (size_t)b * 2 * c.ntab;
        double i0, i1;                     // one search of the temperature grid serves both metallicity tables
        jx_interp_clamped2(xt.lnT, tab, tab + c.ntab, c.ntab, lt, c.inv_dlnT, &i0, &i1);
        const double z0 = mt.e(i0), z1 = mt.e(i1);
        s_rate[q] = (z0 + (z1 - z0) * p[P_Z]) * s_ne[j] * s_ne[j];
    }
    __syncthreads();
    JX_PSTAMP(c, 3);
    int bad = 0;
    for (int q = tid; q < nba; q += nth) {
        const int b = q / c.nann, i = q - b * c.nann;
        double proj = 0.0;
        for (int j = 0; j < c.nann; ++j) proj += xt.projvols[i * c.nann + j] * s_rate[b * c.nann + j];
        const double ae = xt.areascales[q] * xt.exposures[q];
        const double model = proj * ae + xt.backrates[q] * xt.geomarea[i] * ae * p[P_BACKSCALE];
        if (tap_xprofs) tap_xprofs[w * nba + q] = model;
        if (!(model > 0.0)) bad = 1;       // np.array(profs).min() > 0 fails (NaN included)
        const double ct = xt.cts[q];
        s_term[q] = (ct == ct) ? ct * mt.l(model) - model : 0.0;
    }
    JX_PSTAMP(c, 4);
    int xbad = jx_block_or(bad, redi);     // (its barriers order s_term as well)
    JX_PSTAMP(c, 5);
    if (tid < 64) {
        double tot = 0.0;
        for (int q0 = 0; q0 < nba; q0 += 64) {
            double v = (q0 + tid < nba) ? s_term[q0 + tid] : 0.0;
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            tot += v;
        }
        if (tid == 0) red[0] = tot;
    }
    __syncthreads();
    const double xlike = red[0];
    __syncthreads();
    // cashLogLikelihood returns -inf for a non-finite sum; per band in the reference, a non-finite band makes the total non-finite as well
    if (!(fabs(xlike) <= 1.79769313486231570e308)) xbad = 1;
    *xlike_out = xlike; *xbad_out = xbad;
}

// ------------------------------------------------------------------------------------
// K0: per-walker scalar work.  One 256-thread block per walker.
//   base  [W]  parprior + model prior + X-ray log-likelihood, or -inf when rejected
//   cfac  [W, nrow]  convert([h(0), t_prof]) * calibration   (joxsz_funcs.py:473)
//   optional taps: tprof [W,nrow], xprofs [W,nband,nann], parts [W,4]
// ------------------------------------------------------------------------------------
// POW: the profiles with pow() as written in the reference (JOXSZ_PREP_POW=1) instead of through their exponents -- a
// compile-time choice: the code of the form not taken would be a quarter of the kernel and its registers the kernel's.
// FM: exp and log of the log form through the tables of jx_fastmath.hpp (staged in LDS behind the conversion table): 16 and 22
// vector instructions instead of the device library's 38 and 95; held to 2 ulp (JOXSZ_PREP_FASTMATH=0: the library's).
template <bool POW, bool FM = false>
__global__ void __launch_bounds__(JX_PREP_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
jx_prep_kernel(JxDev c, const double* __restrict__ theta, int w0,
               double* __restrict__ base, double* __restrict__ cfac, double* __restrict__ pp_out /*[chunk][N] or null*/,
               double* __restrict__ sz0 /*[chunk] integrated-Compton term of the SZ log-likelihood, or null*/,
               double* __restrict__ tap_tprof, double* __restrict__ tap_xprofs, double* __restrict__ tap_parts, double* __restrict__ tap_integ,
               JxSm smv) {
    JX_LDS_DECL;
    double* p = sm;
    double* red = sm + 20;
    int& redi = *reinterpret_cast<int*>(sm + 28);
    // xr_split: the dependent phases of a walker are two chains that share nothing but its parameters -- priors, grid pass, mass veto and
    // conversion factors on one side, the X-ray model and its Cash sum on the other.  Run as two blocks (of half the threads) they take
    // max(the two) instead of the sum, and the arithmetic of the one hides the waits of the other.  Blocks [0, n): all but the X-ray
    // side; blocks [n, 2n): the X-ray side alone (-> xr_out); the tail adds them.
    const int nblk = c.xr_split ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const bool xonly = c.xr_split && (int)blockIdx.x >= nblk;      // this block: the X-ray side alone
    const bool noxr = c.xr_split && !xonly;                        // this block: everything else
    const int w = xonly ? (int)blockIdx.x - nblk : (int)blockIdx.x;   // walker within the chunk
    const int gw = w0 + w;                    // walker within the batch
    const int tid = threadIdx.x, nth = blockDim.x;
    JX_PSTAMP(c, 0);

    double* s_m = sm + JX_LDS_HDR;            // [N] mass profile
    double* s_t = s_m + c.N;                  // [N] T_SZ on r_pp[:nt]
    double* s_ne = s_t + c.N;                 // [nann]
    double* s_T = s_ne + c.nann;              // [nann]
    double* s_rate = s_T + c.nann;            // [nband*nann]
    double* s_conv = s_rate + c.nband * c.nann;   // [2 nconv] the Compton -> mJy/beam table (temperatures, factors)
    double* s_fm = s_conv + 2 * c.nconv;          // [JX_FM_TABLE_DOUBLES] exp / log tables (FM)
    double* s_term = s_fm + JX_FM_TABLE_DOUBLES;  // [nband*nann] terms of the Cash sum
    static_assert(!(POW && FM), "the tables serve the log form");
    typename std::conditional<FM, JxMathTab, JxMathLib>::type mt;
    if constexpr (FM) { mt.t.et = s_fm; mt.t.lt = s_fm + JX_FM_EXP_N; }

    // the constants of this thread's prior are requested before the parameters are assembled (they do not depend on them)
    const bool has_par = tid < c.npar;
    const int pk_kind = has_par ? c.par_kind[tid] : 0;
    const double pk_a = has_par ? (pk_kind == 1 ? c.par_mu[tid] : c.par_min[tid]) : 0.0;
    const double pk_b = has_par ? (pk_kind == 1 ? c.par_sigma[tid] : c.par_max[tid]) : 0.0;
    const double pk_ln = has_par ? c.par_lnorm[tid] : 0.0;
    for (int i = tid; i < 2 * c.nconv; i += nth) s_conv[i] = (i < c.nconv) ? c.conv_T[i] : c.conv_v[i - c.nconv];   // (visible behind the barriers of jx_load_params)
    if (FM) for (int i = tid; i < JX_FM_TABLE_DOUBLES; i += nth) s_fm[i] = c.fm_tab[i];
    jx_load_params(c, theta, gw, p, &smv);
    JX_PSTAMP(c, 1);
    double pc[5] = {0, 1, 1, 0, 1};           // radius-independent factors of the density (every thread its own copy)
    if (POW) jx_ne_consts(p, c.ne_mode, pc);

    constexpr bool logform = !POW;
    double pl[11];
    jx_prof_consts(mt, p, c.ne_mode, pl);
    double pr = 0.0, parprior = 0.0;
    int rej = 0;
    if (!xonly) {                              // (a block of the X-ray side alone goes straight to that side)
    // ---- priors on every parameter (joxsz_funcs.py:518) ----
    if (has_par) {
        const double v = p[tid];
        if (pk_kind == 1) {
            const double sg = pk_b;
            if (sg <= 0.0) { rej |= REJ_BOX; }
            else {
                const double z = (v - pk_a) / sg;
                pr = pk_ln - 0.5 * z * z;
            }
        } else if (v < pk_a || v > pk_b) rej |= REJ_BOX;
        if (v != v) rej |= REJ_BOX;           // NaN parameter: reject (emcee cannot use NaN)
    }
    parprior = jx_block_sum(pr, red);
    JX_PSTAMP(c, 2);
    if (!(fabs(parprior) <= 1.79769313486231570e308)) rej |= REJ_BOX;     // joxsz_funcs.py:519-520: a non-finite prior returns -inf at once

    // ---- model prior: r_c <= r_s (joxsz_funcs.py:397-407) ----
    if (tid == 0 && (POW ? (pow(10.0, p[P_LOGRC]) > pow(10.0, p[P_LOGRS])) : (p[P_LOGRC] > p[P_LOGRS]))) rej |= REJ_RCRS;   // 10^x is monotonic

    // ---- one pass over the radial grid: pressure (joxsz_funcs.py:275-287), the hydrostatic-mass profile of the
    //      monotonicity veto (joxsz_funcs.py:522-525, 428-437) and T_SZ on r_pp[:nt] (joxsz_funcs.py:469).  The pressure
    //      derivative (joxsz_funcs.py:289-301) is the pressure times -(c + b x^a) / (r (1 + x^a)): no powers of its own.
    const bool veto = c.exclude_unphy_mass != 0;
    const int nprof = (veto || pp_out || c.calc_integ) ? c.N : c.nt;
    double ci = 0.0;                          // this thread's share of integ_wp . pp
    const int nt_ = c.nt, mode_ = c.ne_mode;
    const size_t ppo = (size_t)w * (c.pp_ld ? c.pp_ld : c.N);
    if (logform) {
        // Two radii per trip, straight-line: the evaluations are chains of dependent fp64 operations, and two independent
        // chains in flight per lane are what keeps the SIMD issuing when few waves are resident.  (The second index is
        // clamped instead of branched on, the density is evaluated for every radius, and the stores come after both.)
        for (int i = tid; i < nprof; i += 2 * nth) {
            const int i2 = min(i + nth, nprof - 1);
            const bool two = i + nth < nprof;
            const double ra = c.r_pp[i], rb = c.r_pp[i2], lra = c.lr_pp[i], lrb = c.lr_pp[i2];
            double xaa, xab;
            const double pa = jx_press_log(mt, p, pl, lra, &xaa), pb = jx_press_log(mt, p, pl, lrb, &xab);
            const double ia = jx_inv_ne_log(mt, p, pl, ra, lra, mode_), ib = jx_inv_ne_log(mt, p, pl, rb, lrb, mode_);      // 1 / n_e
            // positive constant factors of mass_fun cannot change the sign test:  -r^2 / n_e dP/dr ~ P (c + b x^a) r / ((1 + x^a) n_e)
            const double ma = pa * (p[P_C] + p[P_B] * xaa) * ra * ia / (1.0 + xaa);
            const double mb = pb * (p[P_C] + p[P_B] * xab) * rb * ib / (1.0 + xab);
            if (pp_out) { pp_out[ppo + i] = pa; if (two) pp_out[ppo + i2] = pb; }
            if (c.calc_integ) { ci = fma(c.integ_wp[i], pa, ci); if (two) ci = fma(c.integ_wp[i2], pb, ci); }
            if (veto) { s_m[i] = ma; if (two) s_m[i2] = mb; }
            if (i < nt_) s_t[i] = pa * ia;
            if (two && i2 < nt_) s_t[i2] = pb * ib;
        }
    } else {
        for (int i = tid; i < nprof; i += nth) {
            const double r = c.r_pp[i];
            const double x = r / p[P_RP];
            const double xa = pow(x, p[P_A]);
            const double press = p[P_P0] / (pow(x, p[P_C]) * pow(1.0 + xa, (p[P_B] - p[P_C]) / p[P_A]));   // == jx_press(p, r)
            if (pp_out) pp_out[ppo + i] = press;
            if (c.calc_integ) ci = fma(c.integ_wp[i], press, ci);
            if (veto || i < nt_) {
                const double ne = jx_ne_pc(p, pc, r, mode_);
                if (veto) s_m[i] = press * (p[P_C] + p[P_B] * xa) / (r * (1.0 + xa)) * r * r / ne;
                if (i < nt_) s_t[i] = press / ne;
            }
        }
    }
    __syncthreads();
    JX_PSTAMP(c, 3);
    if (veto) {
        for (int i = tid; i < c.N; i += nth) {
            double g;                          // np.gradient(m, 1)
            if (i == 0) g = s_m[1] - s_m[0];
            else if (i == c.N - 1) g = s_m[c.N - 1] - s_m[c.N - 2];
            else g = (s_m[i + 1] - s_m[i - 1]) / 2.0;
            if (!(g > 0.0)) rej |= REJ_MASS;
        }
    }

    // ---- integrated Compton parameter and its chi^2 term (joxsz_funcs.py:480-484; np.nansum drops a NaN) ----
    if (c.calc_integ) {
        const double cint = jx_block_sum(ci, red);
        if (tid == 0) {
            const double z = (cint - c.integ_mu) / c.integ_sig;
            const double z2 = z * z;
            if (sz0) sz0[w] = (z2 == z2) ? -0.5 * z2 : 0.0;
            if (tap_integ) tap_integ[w] = cint;
        }
    }

    JX_PSTAMP(c, 4);
    // ---- h(0), conversion factors (joxsz_funcs.py:470-473) ----
    double part = 0.0;
    for (int k = tid; k < c.nt; k += nth) part += c.hw[k] * s_t[k];
    const double t0 = jx_block_sum(part, red);
    for (int k = tid; k < c.nrow; k += nth) {
        const double T = (k == 0) ? t0 : s_t[k - 1];
        cfac[(size_t)w * c.nrow + k] = jx_convert_tab(s_conv, s_conv + c.nconv, c.nconv, T) * p[P_CALIB];
        if (tap_tprof) tap_tprof[(size_t)w * c.nrow + k] = T;
    }
    __syncthreads();
    JX_PSTAMP(c, 5);
    }                                          // (!xonly)

    // ---- X-ray: calcProfiles + Cash (joxsz_funcs.py:527-532, 495-505) ----
    double xlike = 0.0;
    int xbad = 0;
    if (!c.sz_only && !noxr) {
        const JxXrTab xt{c.x_r_ne, c.x_r_T, c.lnT, c.lnrate, c.projvols, c.areascales, c.exposures, c.backrates, c.geomarea, c.cts};
        // (s_m, the mass profile, is dead by now: two block reductions since its last read)
        jx_xray_side<POW>(c, xt, p, pl, pc, mt, tid, nth, s_m, s_ne, s_T, s_rate, s_term, red, &redi, (size_t)w, tap_xprofs, &xlike, &xbad);
        if (xbad) rej |= REJ_XRAY;
    }
    if (xonly) {                               // the X-ray side alone: its sum and its verdict for the tail
        if (tid == 0) { c.xr_out[2 * (size_t)w] = xlike; c.xr_out[2 * (size_t)w + 1] = xbad ? 1.0 : 0.0; }
        JX_PSTAMP(c, 6);
        return;
    }
    const int rejall = jx_block_or(rej, &redi);
    if (tid == 0) {
        double prior = parprior;
        // the reference returns early on REJ_BOX / REJ_MASS; REJ_RCRS and REJ_XRAY add -inf
        double b = (rejall != 0) ? -INFINITY : (prior + xlike);
        base[w] = b;
        if (tap_parts) {
            tap_parts[(size_t)w * 4 + 0] = (rejall & REJ_XRAY) ? -INFINITY : xlike;
            tap_parts[(size_t)w * 4 + 2] = (rejall & (REJ_BOX | REJ_RCRS)) ? -INFINITY : prior;
            tap_parts[(size_t)w * 4 + 3] = (double)rejall;
        }
    }
    JX_PSTAMP(c, 6);
}

// ------------------------------------------------------------------------------------
// The per-walker work of the TIMED path as two lean roles in one launch (what jx_prep_kernel's two-block form does, same arithmetic, same
// bits): blocks [0, n) -- parameters, priors, grid pass (pressure, mass veto, T_SZ), h(0), conversion factors; blocks [n, 2n) -- the X-ray
// side alone.  jx_prep_kernel serves every call with taps, the pow() form, the integrated-Compton term, long grids; it is held to 128
// registers for its occupancy and spills when anything is added to it (profiles/r05_xray_in_tail.log).  Here every table a role reads
// -- exp / log tables, the radii and their logarithms, the h(0) weights, the conversion table; or exp / log tables and the small tables of the
// X-ray side -- comes from ONE packed array (jx_finalize) in ONE batched copy into LDS at the block's start, beside the parameter loads: a
// block pays one trip to memory in front of its first barrier instead of one per phase (the count-rate tables stay in memory: two gathers
// behind the temperature search, which runs on the LDS copy of the grid).
//   sz_pack  [JX_FM_TABLE_DOUBLES + 2 N + nt + 2 nconv]    fm | r_pp | log r_pp | hw | conv_T | conv_v
//   xr_pack  [JX_FM_TABLE_DOUBLES + 3 nann + ntab + nann^2 + 4 nband nann]    fm | x_r_ne | x_r_T | geomarea | lnT | projvols | areascales | exposures | backrates | cts
// ------------------------------------------------------------------------------------
#define JX_W2_THREADS 128
#define JX_SZ_PACK_DOUBLES(c) ((size_t)JX_FM_TABLE_DOUBLES + 2 * (size_t)(c).N + (c).nt + 2 * (size_t)(c).nconv)
#define JX_XR_PACK_DOUBLES(c) ((size_t)JX_FM_TABLE_DOUBLES + 3 * (size_t)(c).nann + (c).ntab + (size_t)(c).nann * (c).nann + 4 * (size_t)(c).nband * (c).nann)
#define JX_W2_LDS_DOUBLES(c) (JX_LDS_HDR + ((JX_SZ_PACK_DOUBLES(c) + (size_t)(c).N + (c).nt) > (JX_XR_PACK_DOUBLES(c) + 3 * (size_t)(c).nann + 2 * (size_t)(c).nband * (c).nann) \
                                             ? (JX_SZ_PACK_DOUBLES(c) + (size_t)(c).N + (c).nt) : (JX_XR_PACK_DOUBLES(c) + 3 * (size_t)(c).nann + 2 * (size_t)(c).nband * (c).nann)) + 8)

__global__ void __launch_bounds__(JX_W2_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
jx_walker2_kernel(JxDev c, const double* __restrict__ theta, int w0, double* __restrict__ base, double* __restrict__ cfac, double* __restrict__ pp_out, JxSm smv) {
    JX_LDS_DECL;
    double* p = sm;
    double* red = sm + 20;
    int& redi = *reinterpret_cast<int*>(sm + 28);
    const int nblk = (int)(gridDim.x >> 1);
    const bool xonly = (int)blockIdx.x >= nblk;
    const int w = xonly ? (int)blockIdx.x - nblk : (int)blockIdx.x, gw = w0 + w;
    const int tid = threadIdx.x, nth = JX_W2_THREADS;
    JX_PSTAMP(c, 0);
    double* s_pack = sm + JX_LDS_HDR;                      // the role's packed tables, the exp / log tables first
    JxMathTab mt;
    mt.t.et = s_pack; mt.t.lt = s_pack + JX_FM_EXP_N;
    if (xonly) {
        // ---- the X-ray side alone ----
        jx_copy_batched<12>(s_pack, c.xr_pack, (int)JX_XR_PACK_DOUBLES(c), tid, nth);
        jx_load_params(c, theta, gw, p, &smv);
        JX_PSTAMP(c, 1);
        const int nann = c.nann, nba = c.nband * c.nann;
        const double* t = s_pack + JX_FM_TABLE_DOUBLES;
        const JxXrTab xt{t, t + nann, t + 3 * nann, c.lnrate, t + 3 * nann + c.ntab, t + 3 * nann + c.ntab + nann * nann, t + 3 * nann + c.ntab + nann * nann + nba,
                         t + 3 * nann + c.ntab + nann * nann + 2 * nba, t + 2 * nann, t + 3 * nann + c.ntab + nann * nann + 3 * nba};
        double* s_ne = s_pack + JX_XR_PACK_DOUBLES(c);
        double* s_T = s_ne + nann;
        double* s_x = s_T + nann;
        double* s_rate = s_x + nann;
        double* s_term = s_rate + nba;
        double pl[11];
        jx_prof_consts(mt, p, c.ne_mode, pl);
        const double pc[5] = {0, 1, 1, 0, 1};
        double xlike = 0.0;
        int xbad = 0;
        jx_xray_side<false>(c, xt, p, pl, pc, mt, tid, nth, s_x, s_ne, s_T, s_rate, s_term, red, &redi, (size_t)w, nullptr, &xlike, &xbad);
        if (tid == 0) { c.xr_out[2 * (size_t)w] = xlike; c.xr_out[2 * (size_t)w + 1] = xbad ? 1.0 : 0.0; }
        JX_PSTAMP(c, 6);
        return;
    }
    // ---- everything but the X-ray side ----
    const int N = c.N, nt = c.nt;
    const double* s_r = s_pack + JX_FM_TABLE_DOUBLES;      // [N]
    const double* s_lr = s_r + N;                           // [N]
    const double* s_hw = s_lr + N;                          // [nt]
    const double* s_conv = s_hw + nt;                       // [2 nconv]
    double* s_m = s_pack + JX_SZ_PACK_DOUBLES(c);          // [N] mass profile
    double* s_t = s_m + N;                                  // [nt] T_SZ on r_pp[:nt]
    const bool has_par = tid < c.npar;
    const int pk_kind = has_par ? c.par_kind[tid] : 0;
    const double pk_a = has_par ? (pk_kind == 1 ? c.par_mu[tid] : c.par_min[tid]) : 0.0;
    const double pk_b = has_par ? (pk_kind == 1 ? c.par_sigma[tid] : c.par_max[tid]) : 0.0;
    const double pk_ln = has_par ? c.par_lnorm[tid] : 0.0;
    jx_copy_batched<20>(s_pack, c.sz_pack, (int)JX_SZ_PACK_DOUBLES(c), tid, nth);
    jx_load_params(c, theta, gw, p, &smv);
    JX_PSTAMP(c, 1);
    double pl[11];
    jx_prof_consts(mt, p, c.ne_mode, pl);
    double pr = 0.0;
    int rej = 0;
    // priors on every parameter (joxsz_funcs.py:518)
    if (has_par) {
        const double v = p[tid];
        if (pk_kind == 1) {
            const double sg = pk_b;
            if (sg <= 0.0) { rej |= REJ_BOX; }
            else {
                const double z = (v - pk_a) / sg;
                pr = pk_ln - 0.5 * z * z;
            }
        } else if (v < pk_a || v > pk_b) rej |= REJ_BOX;
        if (v != v) rej |= REJ_BOX;           // NaN parameter: reject (emcee cannot use NaN)
    }
    const double parprior = jx_block_sum(pr, red);
    JX_PSTAMP(c, 2);
    if (!(fabs(parprior) <= 1.79769313486231570e308)) rej |= REJ_BOX;     // joxsz_funcs.py:519-520
    if (tid == 0 && p[P_LOGRC] > p[P_LOGRS]) rej |= REJ_RCRS;             // model prior: r_c <= r_s (joxsz_funcs.py:397-407; 10^x is monotonic)
    // one pass over the radial grid, two radii per trip (jx_prep_kernel's arithmetic)
    const bool veto = c.exclude_unphy_mass != 0;
    const int mode_ = c.ne_mode;
    const size_t ppo = (size_t)w * (c.pp_ld ? c.pp_ld : N);
    for (int i = tid; i < N; i += 2 * nth) {
        const int i2 = min(i + nth, N - 1);
        const bool two = i + nth < N;
        const double ra = s_r[i], rb = s_r[i2], lra = s_lr[i], lrb = s_lr[i2];
        double xaa, xab;
        const double pa = jx_press_log(mt, p, pl, lra, &xaa), pb = jx_press_log(mt, p, pl, lrb, &xab);
        const double ia = jx_inv_ne_log(mt, p, pl, ra, lra, mode_), ib = jx_inv_ne_log(mt, p, pl, rb, lrb, mode_);      // 1 / n_e
        const double ma = pa * (p[P_C] + p[P_B] * xaa) * ra * ia / (1.0 + xaa);
        const double mb = pb * (p[P_C] + p[P_B] * xab) * rb * ib / (1.0 + xab);
        pp_out[ppo + i] = pa; if (two) pp_out[ppo + i2] = pb;
        if (veto) { s_m[i] = ma; if (two) s_m[i2] = mb; }
        if (i < nt) s_t[i] = pa * ia;
        if (two && i2 < nt) s_t[i2] = pb * ib;
    }
    __syncthreads();
    JX_PSTAMP(c, 3);
    if (veto) {
        for (int i = tid; i < N; i += nth) {
            double g;                          // np.gradient(m, 1)
            if (i == 0) g = s_m[1] - s_m[0];
            else if (i == N - 1) g = s_m[N - 1] - s_m[N - 2];
            else g = (s_m[i + 1] - s_m[i - 1]) / 2.0;
            if (!(g > 0.0)) rej |= REJ_MASS;
        }
    }
    JX_PSTAMP(c, 4);
    // h(0), conversion factors (joxsz_funcs.py:470-473)
    double part = 0.0;
    for (int k = tid; k < nt; k += nth) part += s_hw[k] * s_t[k];
    const double t0 = jx_block_sum(part, red);
    for (int k = tid; k < c.nrow; k += nth) {
        const double T = (k == 0) ? t0 : s_t[k - 1];
        cfac[(size_t)w * c.nrow + k] = jx_convert_tab(s_conv, s_conv + c.nconv, c.nconv, T) * p[P_CALIB];
    }
    JX_PSTAMP(c, 5);
    const int rejall = jx_block_or(rej, &redi);
    if (tid == 0) base[w] = (rejall != 0) ? -INFINITY : (parprior + 0.0);   // (the X-ray term arrives from the other block: the tail adds the two)
    JX_PSTAMP(c, 6);
}

// ------------------------------------------------------------------------------------
// The pressure profile on the radial grid alone (joxsz_funcs.py:275-287, 453): what the spline-array product waits for.
// One block per walker; the same arithmetic as the grid pass of jx_prep_kernel (which then runs beside the SZ chain on
// a second stream: priors, vetoes, X-ray side, conversion factors are consumed by the tail only).
// ------------------------------------------------------------------------------------
template <bool POW>
__global__ void __launch_bounds__(128)
jx_pp_kernel(JxDev c, const double* __restrict__ theta, int w0, double* __restrict__ pp_out /*[chunk][N]*/) {
    JX_LDS_DECL;
    double* p = sm;
    const int w = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    jx_load_params(c, theta, w0 + w, p);
    const int N_ = c.N;
    double* __restrict__ out = pp_out + (size_t)w * N_;
    if (POW) {
        for (int i = tid; i < N_; i += nth) out[i] = jx_press(p, c.r_pp[i]);
    } else {
        double pl[11];
        const JxMathLib mt;
        jx_prof_consts(mt, p, c.ne_mode, pl);
        // (two radii per trip, like the grid pass of jx_prep_kernel: two independent chains per lane)
        for (int i = tid; i < N_; i += 2 * nth) {
            const int i2 = min(i + nth, N_ - 1);
            double xa, xb;
            const double pa = jx_press_log(mt, p, pl, c.lr_pp[i], &xa), pb = jx_press_log(mt, p, pl, c.lr_pp[i2], &xb);
            out[i] = pa;
            if (i + nth < N_) out[i2] = pb;
        }
    }
}

// ------------------------------------------------------------------------------------
// K1: FUSED profile -> Abel -> y -> spline -> map.  grid = (chunk walkers * map_split).
//
// Phase 1  pp_j = gNFW(r_j)                       -> LDS            (joxsz_funcs.py:453)
// Phase 2  ab_i = sum_{j>=i} A[i][j] pp_j         thread-per-(half-)row, weights streamed
//          coalesced from the shared table (L2), pp broadcast from LDS (joxsz_funcs.py:457)
// Phase 3  y = y_scale * ab ; M = G_band y        (joxsz_funcs.py:459-460)
// Phase 4  per-interval cubic coefficients        -> LDS
// Phase 5  y_2d[p] = spline(d_mat[p]) for the block's rows, coalesced 16-byte stores into
//          the padded FFT image (joxsz_funcs.py:462).  This is the HBM-write-bound phase:
//          S*S*8 bytes per walker.  Two forms:
//          * symmetric (d_mat[iy][ix] depends on |iy-c|, |ix-c| only, which is what
//            centdistmat builds, joxsz_funcs.py:78-88): each wave evaluates the half row
//            |ix-c| = 0..na-1 once from the precomputed (interval, offset) table of the
//            shared radius grid, stages it in LDS and stores the two mirrored rows;
//          * generic: any d_mat, interval search per pixel.
//
// LDS coefficient slots: k = 0..N-2 intervals [r_k, r_{k+1}] in t = x - r_k;
// k = N-1 the centre interval [-r_0, r_0] in t = x (even polynomial); k = N zero (outside).
// ------------------------------------------------------------------------------------
// fast 1/sqrt(x) in fp64: hardware estimate y0 (relative error e ~ 2^-23) refined by one third-order
// (Halley) step  y = y0 (1 + e/2 + 3 e^2/8),  e = 1 - x y0^2  ->  error O(e^3), i.e. rounding only.
__device__ __forceinline__ double jx_rsqrt(double x) {
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y0, y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

// sum over j = j0, j0+4, ... < N of q_j / sqrt(r_j^2 - ri2): four independent chains in flight
__device__ __forceinline__ double jx_abel_row(const double2* s_rq, double ri2, int j0, int N) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int j = j0;
    for (; j + 12 < N; j += 16) {
        const double2 q0 = s_rq[j], q1 = s_rq[j + 4], q2 = s_rq[j + 8], q3 = s_rq[j + 12];
        const double t0 = jx_rsqrt(q0.x - ri2), t1 = jx_rsqrt(q1.x - ri2), t2 = jx_rsqrt(q2.x - ri2), t3 = jx_rsqrt(q3.x - ri2);
        a0 = fma(q0.y, t0, a0); a1 = fma(q1.y, t1, a1); a2 = fma(q2.y, t2, a2); a3 = fma(q3.y, t3, a3);
    }
    for (; j < N; j += 4) {
        const double2 q0 = s_rq[j];
        a0 = fma(q0.y, jx_rsqrt(q0.x - ri2), a0);
    }
    return (a0 + a1) + (a2 + a3);
}

__device__ __forceinline__ void jx_profile_to_coefs(const JxDev& c, const double* p, int w, bool taps, double* s_r,
                                                    double* s_pp, double* s_y, double* s_M, double* s_cf, double2* s_rq,
                                                    double2* s_ds, double* tap_pp, double* tap_ab, double* tap_y) {
    const int N = c.N, tid = threadIdx.x, nth = blockDim.x;
    // Phase 1: profile; s_rq[j] = (r_j^2, q_j) with q_j = cj_j * pp_j; the grid table (r, cj, dg, sp) comes
    // in one 32-byte load per knot so that the launch pays one cold-miss latency for it, not four.
    for (int j = tid; j < N; j += nth) {
        const double2 t0 = *reinterpret_cast<const double2*>(c.abel_tab + 4 * (size_t)j);
        const double2 t1 = *reinterpret_cast<const double2*>(c.abel_tab + 4 * (size_t)j + 2);
        const double r = t0.x;
        s_r[j] = r;
        s_ds[j] = t1;
        const double v = c.inject_pp ? c.inject_pp[(size_t)w * N + j] : (JX_DBG(c, 32) ? r : jx_press(p, r));
        s_pp[j] = v;
        s_rq[j] = make_double2(r * r, t0.y * v);
        if (taps && tap_pp) tap_pp[(size_t)w * N + j] = v;
    }
    __syncthreads();

    // Phase 2: Abel integral, weights generated on the fly from the radius grid in LDS:
    //   ab_i = dg_i pp_i + sp_i pp_{i+1} + sum_{j>=i+2} q_j / sqrt(r_j^2 - r_i^2)
    // Rows i and N-1-i are paired (their trapezoid sums have N-3 terms together) and each pair is
    // shared by 4 adjacent lanes (terms e = sub, sub+4, ...), reduced with two shuffles.
    const int npair = (N + 1) >> 1;
    for (int p0 = 0; p0 < npair; p0 += (nth >> 2)) {
        const int pr = p0 + (tid >> 2), sub = tid & 3;
        double acc1 = 0.0, acc2 = 0.0;
        const int i1 = pr, i2 = N - 1 - pr;
        if (pr < npair && !JX_DBG(c, 8)) {
            if (JX_DBG(c, 128)) {            // ablation: same loop without the reciprocal square root
                const double ri1 = s_rq[i1].x, ri2 = s_rq[i2].x;
                for (int j = i1 + 2 + sub; j < N; j += 4) { const double2 q = s_rq[j]; acc1 = fma(q.y, q.x - ri1, acc1); }
                for (int j = i2 + 2 + sub; j < N; j += 4) { const double2 q = s_rq[j]; acc2 = fma(q.y, q.x - ri2, acc2); }
            } else {
                acc1 = jx_abel_row(s_rq, s_rq[i1].x, i1 + 2 + sub, N);
                if (i2 != i1) acc2 = jx_abel_row(s_rq, s_rq[i2].x, i2 + 2 + sub, N);
            }
        }
        acc1 += __shfl_xor(acc1, 1, 64); acc1 += __shfl_xor(acc1, 2, 64);
        acc2 += __shfl_xor(acc2, 1, 64); acc2 += __shfl_xor(acc2, 2, 64);
        if (pr < npair && sub == 0) {
            const double2 ds1 = s_ds[i1];
            double ab = fma(ds1.x, s_pp[i1], acc1);
            if (i1 + 1 < N) ab = fma(ds1.y, s_pp[i1 + 1], ab);
            s_y[i1] = c.y_scale * ab;
            if (taps && tap_ab) tap_ab[(size_t)w * N + i1] = ab;
            if (i2 != i1) {
                const double2 ds2 = s_ds[i2];
                double ab2 = fma(ds2.x, s_pp[i2], acc2);
                if (i2 + 1 < N) ab2 = fma(ds2.y, s_pp[i2 + 1], ab2);
                s_y[i2] = c.y_scale * ab2;
                if (taps && tap_ab) tap_ab[(size_t)w * N + i2] = ab2;
            }
        }
    }
    __syncthreads();
    if (taps && tap_y) for (int i = tid; i < N; i += nth) tap_y[(size_t)w * N + i] = s_y[i];

    // Phase 3: spline moments through the banded operator, two lanes per row
    const int K = c.K;
    for (int i0 = 0; i0 < N; i0 += (nth >> 1)) {
        const int i = i0 + (tid >> 1), hsel = tid & 1;
        double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
        if (i < N && !JX_DBG(c, 16)) {
            const int k0 = max(-K, -i), k1 = min(K, N - 1 - i);
            const int km = (k0 + k1 + 1) >> 1;
            const int ka = hsel ? km : k0, kb = hsel ? k1 + 1 : km;      // [ka, kb)
            int k = ka;
            for (; k + 3 < kb; k += 4) {
                const double g0 = c.gband[(size_t)(k + K) * N + i], g1 = c.gband[(size_t)(k + 1 + K) * N + i],
                             g2 = c.gband[(size_t)(k + 2 + K) * N + i], g3 = c.gband[(size_t)(k + 3 + K) * N + i];
                m0 = fma(g0, s_y[i + k], m0);     m1 = fma(g1, s_y[i + k + 1], m1);
                m2 = fma(g2, s_y[i + k + 2], m2); m3 = fma(g3, s_y[i + k + 3], m3);
            }
            for (; k < kb; ++k) m0 = fma(c.gband[(size_t)(k + K) * N + i], s_y[i + k], m0);
        }
        double m = (m0 + m1) + (m2 + m3);
        m += __shfl_xor(m, 1, 64);
        if (i < N && hsel == 0) s_M[i] = m;
    }
    __syncthreads();
    if (c.cf_out) {                                         // (y_k, M_k) pairs out: the contracted route evaluates the map samples from them
        double2* o = reinterpret_cast<double2*>(c.cf_out) + (c.cf_tr ? (size_t)w : (size_t)w * (c.cf_ws >> 1));
        const size_t ks = c.cf_tr ? (size_t)c.cf_ws : 1;
        for (int k = tid; k < N; k += nth) o[k * ks] = make_double2(s_y[k], s_M[k]);
        return;
    }

    // Phase 4: cubic coefficients
    for (int k = tid; k < N + 1; k += nth) {
        double c0, c1, c2, c3;
        if (k < N - 1) {
            const double h = s_r[k + 1] - s_r[k];
            const double y0 = s_y[k], y1 = s_y[k + 1], m0 = s_M[k], m1 = s_M[k + 1];
            c0 = y0;
            c1 = (y1 - y0) / h - h * (2.0 * m0 + m1) / 6.0;
            c2 = 0.5 * m0;
            c3 = (m1 - m0) / (6.0 * h);
        } else if (k == N - 1) {                            // centre: y_0 + M_0 (x^2 - r_0^2)/2
            c0 = s_y[0] - 0.5 * s_M[0] * s_r[0] * s_r[0];
            c1 = 0.0; c2 = 0.5 * s_M[0]; c3 = 0.0;
        } else {
            c0 = c1 = c2 = c3 = 0.0;                        // outside [-r_N, r_N]: fill value 0
        }
        s_cf[4 * k + 0] = c0; s_cf[4 * k + 1] = c1; s_cf[4 * k + 2] = c2; s_cf[4 * k + 3] = c3;
    }
    __syncthreads();
}

// The same four phases for TWO walkers at once (quadrant mode, two walkers per block).  What the two share is computed
// once -- the Abel weights 1/sqrt(r_j^2 - r_i^2) of phase 2, the loads of the banded spline operator in phase 3 -- and
// the rest gives every lane two independent dependency chains instead of one.  Each walker's sums run over the same
// terms in the same order as in jx_profile_to_coefs: the results are the same bit patterns.
__device__ __forceinline__ void jx_abel_row2(const double2* s_rq, const double* s_qB, double ri2, int j0, int N, double* oA, double* oB) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    int j = j0;
    for (; j + 12 < N; j += 16) {
        const double2 q0 = s_rq[j], q1 = s_rq[j + 4], q2 = s_rq[j + 8], q3 = s_rq[j + 12];
        const double u0 = s_qB[j], u1 = s_qB[j + 4], u2 = s_qB[j + 8], u3 = s_qB[j + 12];
        const double t0 = jx_rsqrt(q0.x - ri2), t1 = jx_rsqrt(q1.x - ri2), t2 = jx_rsqrt(q2.x - ri2), t3 = jx_rsqrt(q3.x - ri2);
        a0 = fma(q0.y, t0, a0); a1 = fma(q1.y, t1, a1); a2 = fma(q2.y, t2, a2); a3 = fma(q3.y, t3, a3);
        b0 = fma(u0, t0, b0); b1 = fma(u1, t1, b1); b2 = fma(u2, t2, b2); b3 = fma(u3, t3, b3);
    }
    for (; j < N; j += 4) {
        const double2 q0 = s_rq[j];
        const double t0 = jx_rsqrt(q0.x - ri2);
        a0 = fma(q0.y, t0, a0);
        b0 = fma(s_qB[j], t0, b0);
    }
    *oA = (a0 + a1) + (a2 + a3);
    *oB = (b0 + b1) + (b2 + b3);
}

// scratch (doubles, Ne = N rounded up to even): s_r[Ne] s_ppA[Ne] s_ppB[Ne] s_yA[Ne] s_yB[Ne] s_qB[Ne] s_rq[2 Ne] s_ds[2 Ne];
// the spline moments of phase 3 go where s_rq was (dead after phase 2).
#define JX_MAP_SCRATCH2_DOUBLES(N) (10 * JX_MAP_NE(N) + JX_LDS_HDR)
__device__ __forceinline__ void jx_profile_to_coefs2(const JxDev& c, const double* pA, const double* pB, int w, bool taps,
                                                     double* scratch, double* s_cfA, double* s_cfB,
                                                     double* tap_pp, double* tap_ab, double* tap_y) {
    const int N = c.N, tid = threadIdx.x, nth = blockDim.x, Ne = (N + 1) & ~1;
    double* s_r = scratch;
    double* s_ppA = s_r + Ne;
    double* s_ppB = s_ppA + Ne;
    double* s_yA = s_ppB + Ne;
    double* s_yB = s_yA + Ne;
    double* s_qB = s_yB + Ne;
    double2* s_rq = reinterpret_cast<double2*>(s_qB + Ne);
    double2* s_ds = s_rq + Ne;
    double* s_MA = reinterpret_cast<double*>(s_rq);
    double* s_MB = s_MA + Ne;

    // Phase 1
    for (int j = tid; j < N; j += nth) {
        const double2 t0 = *reinterpret_cast<const double2*>(c.abel_tab + 4 * (size_t)j);
        const double2 t1 = *reinterpret_cast<const double2*>(c.abel_tab + 4 * (size_t)j + 2);
        const double r = t0.x;
        s_r[j] = r;
        s_ds[j] = t1;
        const double vA = c.inject_pp ? c.inject_pp[(size_t)w * N + j] : (JX_DBG(c, 32) ? r : jx_press(pA, r));
        const double vB = c.inject_pp ? c.inject_pp[(size_t)(w + 1) * N + j] : (JX_DBG(c, 32) ? r : jx_press(pB, r));
        s_ppA[j] = vA;
        s_ppB[j] = vB;
        s_rq[j] = make_double2(r * r, t0.y * vA);
        s_qB[j] = t0.y * vB;
        if (taps && tap_pp) { tap_pp[(size_t)w * N + j] = vA; tap_pp[(size_t)(w + 1) * N + j] = vB; }
    }
    __syncthreads();

    // Phase 2
    const int npair = (N + 1) >> 1;
    for (int p0 = 0; p0 < npair; p0 += (nth >> 2)) {
        const int pr = p0 + (tid >> 2), sub = tid & 3;
        double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
        const int i1 = pr, i2 = N - 1 - pr;
        if (pr < npair && !JX_DBG(c, 8)) {
            jx_abel_row2(s_rq, s_qB, s_rq[i1].x, i1 + 2 + sub, N, &a1, &b1);
            if (i2 != i1) jx_abel_row2(s_rq, s_qB, s_rq[i2].x, i2 + 2 + sub, N, &a2, &b2);
        }
        a1 += __shfl_xor(a1, 1, 64); a1 += __shfl_xor(a1, 2, 64);
        a2 += __shfl_xor(a2, 1, 64); a2 += __shfl_xor(a2, 2, 64);
        b1 += __shfl_xor(b1, 1, 64); b1 += __shfl_xor(b1, 2, 64);
        b2 += __shfl_xor(b2, 1, 64); b2 += __shfl_xor(b2, 2, 64);
        if (pr < npair && sub == 0) {
            const double2 ds1 = s_ds[i1];
            double abA = fma(ds1.x, s_ppA[i1], a1), abB = fma(ds1.x, s_ppB[i1], b1);
            if (i1 + 1 < N) { abA = fma(ds1.y, s_ppA[i1 + 1], abA); abB = fma(ds1.y, s_ppB[i1 + 1], abB); }
            s_yA[i1] = c.y_scale * abA;
            s_yB[i1] = c.y_scale * abB;
            if (taps && tap_ab) { tap_ab[(size_t)w * N + i1] = abA; tap_ab[(size_t)(w + 1) * N + i1] = abB; }
            if (i2 != i1) {
                const double2 ds2 = s_ds[i2];
                double ab2A = fma(ds2.x, s_ppA[i2], a2), ab2B = fma(ds2.x, s_ppB[i2], b2);
                if (i2 + 1 < N) { ab2A = fma(ds2.y, s_ppA[i2 + 1], ab2A); ab2B = fma(ds2.y, s_ppB[i2 + 1], ab2B); }
                s_yA[i2] = c.y_scale * ab2A;
                s_yB[i2] = c.y_scale * ab2B;
                if (taps && tap_ab) { tap_ab[(size_t)w * N + i2] = ab2A; tap_ab[(size_t)(w + 1) * N + i2] = ab2B; }
            }
        }
    }
    __syncthreads();
    if (taps && tap_y) for (int i = tid; i < N; i += nth) { tap_y[(size_t)w * N + i] = s_yA[i]; tap_y[(size_t)(w + 1) * N + i] = s_yB[i]; }

    // Phase 3
    const int K = c.K;
    for (int i0 = 0; i0 < N; i0 += (nth >> 1)) {
        const int i = i0 + (tid >> 1), hsel = tid & 1;
        double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0, n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
        if (i < N && !JX_DBG(c, 16)) {
            const int k0 = max(-K, -i), k1 = min(K, N - 1 - i);
            const int km = (k0 + k1 + 1) >> 1;
            const int ka = hsel ? km : k0, kb = hsel ? k1 + 1 : km;      // [ka, kb)
            int k = ka;
            for (; k + 3 < kb; k += 4) {
                const double g0 = c.gband[(size_t)(k + K) * N + i], g1 = c.gband[(size_t)(k + 1 + K) * N + i],
                             g2 = c.gband[(size_t)(k + 2 + K) * N + i], g3 = c.gband[(size_t)(k + 3 + K) * N + i];
                m0 = fma(g0, s_yA[i + k], m0);     m1 = fma(g1, s_yA[i + k + 1], m1);
                m2 = fma(g2, s_yA[i + k + 2], m2); m3 = fma(g3, s_yA[i + k + 3], m3);
                n0 = fma(g0, s_yB[i + k], n0);     n1 = fma(g1, s_yB[i + k + 1], n1);
                n2 = fma(g2, s_yB[i + k + 2], n2); n3 = fma(g3, s_yB[i + k + 3], n3);
            }
            for (; k < kb; ++k) {
                const double g = c.gband[(size_t)(k + K) * N + i];
                m0 = fma(g, s_yA[i + k], m0);
                n0 = fma(g, s_yB[i + k], n0);
            }
        }
        double m = (m0 + m1) + (m2 + m3), n = (n0 + n1) + (n2 + n3);
        m += __shfl_xor(m, 1, 64);
        n += __shfl_xor(n, 1, 64);
        if (i < N && hsel == 0) { s_MA[i] = m; s_MB[i] = n; }
    }
    __syncthreads();
    if (c.cf_out) {                                         // (y_k, M_k) pairs out: the contracted route evaluates the map samples from them
        double2* oA = reinterpret_cast<double2*>(c.cf_out) + (c.cf_tr ? (size_t)w : (size_t)w * (c.cf_ws >> 1));
        double2* oB = reinterpret_cast<double2*>(c.cf_out) + (c.cf_tr ? (size_t)(w + 1) : (size_t)(w + 1) * (c.cf_ws >> 1));
        const size_t ks = c.cf_tr ? (size_t)c.cf_ws : 1;
        for (int k = tid; k < N; k += nth) { oA[k * ks] = make_double2(s_yA[k], s_MA[k]); oB[k * ks] = make_double2(s_yB[k], s_MB[k]); }
        return;
    }

    // Phase 4
    for (int k = tid; k < 2 * (N + 1); k += nth) {
        const bool isB = k >= N + 1;
        const int kk = isB ? k - (N + 1) : k;
        const double* s_y = isB ? s_yB : s_yA;
        const double* s_M = isB ? s_MB : s_MA;
        double* s_cf = isB ? s_cfB : s_cfA;
        double c0, c1, c2, c3;
        if (kk < N - 1) {
            const double h = s_r[kk + 1] - s_r[kk];
            const double y0 = s_y[kk], y1 = s_y[kk + 1], m0 = s_M[kk], m1 = s_M[kk + 1];
            c0 = y0;
            c1 = (y1 - y0) / h - h * (2.0 * m0 + m1) / 6.0;
            c2 = 0.5 * m0;
            c3 = (m1 - m0) / (6.0 * h);
        } else if (kk == N - 1) {                           // centre: y_0 + M_0 (x^2 - r_0^2)/2
            c0 = s_y[0] - 0.5 * s_M[0] * s_r[0] * s_r[0];
            c1 = 0.0; c2 = 0.5 * s_M[0]; c3 = 0.0;
        } else {
            c0 = c1 = c2 = c3 = 0.0;                        // outside [-r_N, r_N]: fill value 0
        }
        s_cf[4 * kk + 0] = c0; s_cf[4 * kk + 1] = c1; s_cf[4 * kk + 2] = c2; s_cf[4 * kk + 3] = c3;
    }
    __syncthreads();
}

// LDS doubles needed by the profile-to-coefficients phases
// LDS layout of the map kernels (doubles): [header][cubic coefficients 4(N+1)] then the phase-1-4 scratch
// (knots, pp, y, M, (r^2,q), (dg,sp) = 8 N), which the symmetric kernel re-uses for its per-wave row buffers
// in phase 5: the block stays under 53 KB and three of them fit a CU.
#define JX_MAP_NE(N) ((size_t)(((N) + 1) & ~1))
#define JX_MAP_FIXED_DOUBLES(N) (JX_LDS_HDR + 4 * JX_MAP_NE(N) + 8)
#define JX_MAP_SCRATCH_DOUBLES(N) (8 * JX_MAP_NE(N))

template <bool VEC2>
__global__ void __launch_bounds__(1024)
jx_abel_map_kernel(JxDev c, const double* __restrict__ theta, int w0, double* __restrict__ img /*[chunk][P][P]*/,
                   double* __restrict__ tap_pp, double* __restrict__ tap_ab, double* __restrict__ tap_y) {
    JX_LDS_DECL;
    const int N = c.N;
    double* p = sm;
    const int Ne = (N + 1) & ~1;
    double* s_cf = sm + JX_LDS_HDR;                                   // [4(N+1)] cubic coefficients (live in phase 5)
    double* s_r = sm + JX_MAP_FIXED_DOUBLES(N);                       // [N] knots         -- scratch from here on
    double* s_pp = s_r + Ne;                                          // [N]
    double* s_y = s_pp + Ne;                                          // [N]
    double* s_M = s_y + Ne;                                           // [N]
    double2* s_rq = reinterpret_cast<double2*>(s_M + Ne);             // [N] (r^2, cj*pp)
    double2* s_ds = s_rq + Ne;                                        // [N] (dg, sp)

    const int tid = threadIdx.x, nth = blockDim.x;
    const int w = blockIdx.x / c.map_split;
    const int part = blockIdx.x - w * c.map_split;

    jx_load_params(c, theta, w0 + w, p);
    jx_profile_to_coefs(c, p, w, part == 0, s_r, s_pp, s_y, s_M, s_cf, s_rq, s_ds, tap_pp, tap_ab, tap_y);

    // Phase 5 (generic): the block's slab of rows
    const int S = c.S;
    const int rows_per = (S + c.map_split - 1) / c.map_split;
    const int row0 = part * rows_per, row1 = min(S, row0 + rows_per);
    const int lane = tid & 63, wv = tid >> 6, nwv = nth >> 6;
    double* out = img + (size_t)w * c.img_ws;
    const double r_first = s_r[0], r_last = s_r[N - 1];

    auto eval = [&](double d) -> double {
        // interp1d(..., bounds_error=False, fill_value=(0,0)): 0 outside [-r_N, r_N];
        // NaN radius propagates
        if (!(d <= r_last)) return (d != d) ? d : 0.0;
        int k;
        double t;
        if (d < r_first) { k = N - 1; t = d; }
        else {
            k = (int)((d - r_first) * c.inv_h_mean);
            k = min(max(k, 0), N - 2);
            while (k < N - 2 && d >= s_r[k + 1]) ++k;
            while (k > 0 && d < s_r[k]) --k;
            t = d - s_r[k];
        }
        const double* cf = s_cf + 4 * k;
        return fma(t, fma(t, fma(t, cf[3], cf[2]), cf[1]), cf[0]);
    };

    for (int iy = row0 + wv; iy < row1; iy += nwv) {
        const double* drow = c.d_mat + (size_t)iy * S;
        double* orow = out + (size_t)iy * c.img_ld;
        if (VEC2) {
            for (int ix = 2 * lane; ix < S; ix += 128) {
                const double2 d = *reinterpret_cast<const double2*>(drow + ix);
                double2 v;
                v.x = eval(d.x);
                v.y = eval(d.y);
                *reinterpret_cast<double2*>(orow + ix) = v;
            }
        } else {
            for (int ix = lane; ix < S; ix += 64) orow[ix] = eval(drow[ix]);
        }
    }
}

// Symmetric form of K1.  q_k / q_t: [q_nb][q_na] interval slot and local abscissa of the
// radius at (|iy-c|, |ix-c|), built once from d_mat and r_pp at jx_finalize.
//
// Phase 5 here is software-pipelined per wave: the (slot, abscissa) entries of the wave's
// NEXT half row are loaded into registers before the stores of the current two mirrored
// rows are issued, so the loads never queue behind those stores (vector memory operations
// return in order) and the stores stay in flight behind a counted vmcnt.  The half row is
// evaluated once (|ix-c| = 0..na-1), mirrored into a full row in the wave's private LDS
// buffer, and both rows iy = c+b and c-b are written from it with aligned 16-byte stores.
// NAIT = ceil(q_na / 64) register slots per lane.
template <bool VEC2, int NAIT>
__global__ void __launch_bounds__(1024)
jx_abel_map_sym_kernel(JxDev c, const double* __restrict__ theta, int w0, double* __restrict__ img,
                       double* __restrict__ tap_pp, double* __restrict__ tap_ab, double* __restrict__ tap_y) {
    JX_LDS_DECL;
    const int N = c.N;
    double* p = sm;
    const int Ne = (N + 1) & ~1;
    double* s_cf = sm + JX_LDS_HDR;                                   // [4(N+1)] cubic coefficients (live in phase 5)
    // quad mode with two walkers per block: the second walker's coefficients sit in front of the scratch
    const int npw = (c.pairw == 2) ? 2 : 1;
    double* s_cfB = sm + JX_MAP_FIXED_DOUBLES(N);
    double* s_r = sm + JX_MAP_FIXED_DOUBLES(N) + (npw == 2 ? 4 * JX_MAP_NE(N) + 8 : 0);   // [N] knots -- scratch from here on
    double* s_pp = s_r + Ne;                                          // [N]
    double* s_y = s_pp + Ne;                                          // [N]
    double* s_M = s_y + Ne;                                           // [N]
    double2* s_rq = reinterpret_cast<double2*>(s_M + Ne);             // [N] (r^2, cj*pp)
    double2* s_ds = s_rq + Ne;                                        // [N] (dg, sp)
    double* s_row = s_r;                                              // [nwaves][row_pad] over the dead scratch

    const int tid = threadIdx.x, nth = blockDim.x;
    const int blk = blockIdx.x / c.map_split, part = blockIdx.x - blk * c.map_split;
    const int w = blk * npw;
    const bool haveB = npw == 2 && w + 1 < c.nlaunch;                 // (an odd launch: the last block has one walker)

    const bool fused2 = haveB && !JX_DBG(c, 1 | 256);                // both walkers through phases 1-4 together (256: one after the other)
    if (fused2) {
        double* pB = s_r + 10 * JX_MAP_NE(N);
        jx_load_params(c, theta, w0 + w, p);
        jx_load_params(c, theta, w0 + w + 1, pB);
        jx_profile_to_coefs2(c, p, pB, w, part == 0, s_r, s_cf, s_cfB, tap_pp, tap_ab, tap_y);
    } else {
    jx_load_params(c, theta, w0 + w, p);
    if (!JX_DBG(c, 1)) jx_profile_to_coefs(c, p, w, part == 0, s_r, s_pp, s_y, s_M, s_cf, s_rq, s_ds, tap_pp, tap_ab, tap_y);
    else { for (int k = tid; k < 4 * (N + 1); k += nth) s_cf[k] = 0.0; __syncthreads(); }
    }
    if (haveB && !fused2) {
        __syncthreads();
        jx_load_params(c, theta, w0 + w + 1, p);
        if (!JX_DBG(c, 1)) jx_profile_to_coefs(c, p, w + 1, part == 0, s_r, s_pp, s_y, s_M, s_cfB, s_rq, s_ds, tap_pp, tap_ab, tap_y);
        else { for (int k = tid; k < 4 * (N + 1); k += nth) s_cfB[k] = 0.0; __syncthreads(); }
    }
    if (c.cf_out) return;                                   // phases 1-3 only: the spline ordinates and moments are in HBM
    if (JX_DBG(c, 2)) return;

    const int S = c.S, na = c.q_na, nb = c.q_nb, cc = c.S / 2;
    const int lane = tid & 63, wv = tid >> 6, nwv = nth >> 6;
    const int row_pad = (S + 3) & ~1;
    double* rowfull = s_row + (size_t)wv * row_pad;
    double* out = img + (size_t)w * c.img_ws;
    const int b_per = (nb + c.map_split - 1) / c.map_split;
    const int b0 = part * b_per, b1 = min(nb, b0 + b_per);

    if (c.quad && !JX_DBG(c, 4)) {
        // The distinct pixels only, walked as one flat array of (slot, abscissa) entries [b0*na, b1*na): every lane has
        // an entry in every trip (a row-wise walk leaves na mod 64 lanes idle), eight entries per lane in flight.
        const int e_end = b1 * na;
        constexpr int UN = 8;
        for (int e0 = b0 * na + tid; e0 < e_end; e0 += UN * nth) {
            int kk[UN];
            double tt[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int e = min(e0 + u * nth, e_end - 1);
                kk[u] = c.q_k[e];
                tt[u] = c.q_t[e];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int e = e0 + u * nth;
                if (e < e_end) {
                    const double t = tt[u];
                    const double* cf = s_cf + 4 * kk[u];
                    const double v = fma(t, fma(t, fma(t, cf[3], cf[2]), cf[1]), cf[0]);
                    const int bb = e / na, a = e - bb * na;
                    out[(size_t)bb * c.img_ld + a] = v;
                    if (a == na - 1) c.xcol[c.xcol_ld > 0 ? (size_t)bb * c.xcol_ld + w : (size_t)w * nb + bb] = v;
                    if (haveB) {                                      // the same table entry for the block's second walker
                        const double* cg = s_cfB + 4 * kk[u];
                        const double v2 = fma(t, fma(t, fma(t, cg[3], cg[2]), cg[1]), cg[0]);
                        out[c.img_ws + (size_t)bb * c.img_ld + a] = v2;
                        if (a == na - 1) c.xcol[c.xcol_ld > 0 ? (size_t)bb * c.xcol_ld + w + 1 : (size_t)(w + 1) * nb + bb] = v2;
                    }
                }
            }
        }
        return;
    }
    // (full rows: the block's walkers one after the other, each from its own coefficients)
    for (int ww = 0; ww < (haveB ? 2 : 1); ++ww) {
    const double* s_cfw = ww ? s_cfB : s_cf;
    out = img + (size_t)(w + ww) * c.img_ws;
    int kq[NAIT];
    double tq[NAIT];
    int b = b0 + wv;
    if (JX_DBG(c, 4)) {                                 // ablation: the store stream alone
        for (int ix = lane; ix < row_pad; ix += 64) rowfull[ix] = 1.0;
        __builtin_amdgcn_wave_barrier();
        for (; b < b1; b += nwv) {
            const int iy1 = cc + b, iy2 = cc - b;
            const bool do1 = iy1 < S, do2 = (b > 0) && (iy2 >= 0);
            double* orow1 = out + (size_t)iy1 * c.img_ld;
            double* orow2 = out + (size_t)iy2 * c.img_ld;
            if (JX_DBG(c, 64)) {
                typedef double jx_d2 __attribute__((ext_vector_type(2)));
                for (int ix = 2 * lane; ix + 1 < S; ix += 128) {
                    const jx_d2 v = *reinterpret_cast<const jx_d2*>(rowfull + ix);
                    if (do1) __builtin_nontemporal_store(v, reinterpret_cast<jx_d2*>(orow1 + ix));
                    if (do2) __builtin_nontemporal_store(v, reinterpret_cast<jx_d2*>(orow2 + ix));
                }
            } else {
                for (int ix = 2 * lane; ix + 1 < S; ix += 128) {
                    const double2 v = *reinterpret_cast<const double2*>(rowfull + ix);
                    if (do1) *reinterpret_cast<double2*>(orow1 + ix) = v;
                    if (do2) *reinterpret_cast<double2*>(orow2 + ix) = v;
                }
            }
        }
        continue;
    }
    if (b < b1) {
#pragma unroll
        for (int u = 0; u < NAIT; ++u) {
            const int a = lane + 64 * u;
            const bool ok = a < na;
            kq[u] = ok ? c.q_k[(size_t)b * na + a] : N;
            tq[u] = ok ? c.q_t[(size_t)b * na + a] : 0.0;
        }
    }
    for (; b < b1; b += nwv) {
        // evaluate the half row and mirror it into the full row
#pragma unroll
        for (int u = 0; u < NAIT; ++u) {
            const int a = lane + 64 * u;
            const double t = tq[u];
            const double* cf = s_cfw + 4 * kq[u];
            const double v = fma(t, fma(t, fma(t, cf[3], cf[2]), cf[1]), cf[0]);
            if (a < na) {
                if (cc + a < S) rowfull[cc + a] = v;
                if (cc - a >= 0) rowfull[cc - a] = v;
            }
        }
        // prefetch the next half row's table entries (older than the stores below)
        const int bn = b + nwv;
        if (bn < b1) {
#pragma unroll
            for (int u = 0; u < NAIT; ++u) {
                const int a = lane + 64 * u;
                const bool ok = a < na;
                kq[u] = ok ? c.q_k[(size_t)bn * na + a] : N;
                tq[u] = ok ? c.q_t[(size_t)bn * na + a] : 0.0;
            }
        }
        // rowfull is private to this wave; LDS operations of one wave complete in order
        __builtin_amdgcn_wave_barrier();
        const int iy1 = cc + b, iy2 = cc - b;
        const bool do1 = iy1 < S, do2 = (b > 0) && (iy2 >= 0);
        double* orow1 = out + (size_t)iy1 * c.img_ld;
        double* orow2 = out + (size_t)iy2 * c.img_ld;
        if (VEC2) {
            for (int ix = 2 * lane; ix < S; ix += 128) {
                const double2 v = *reinterpret_cast<const double2*>(rowfull + ix);
                if (do1) *reinterpret_cast<double2*>(orow1 + ix) = v;
                if (do2) *reinterpret_cast<double2*>(orow2 + ix) = v;
            }
        } else {
            for (int ix = lane; ix < S; ix += 64) {
                const double v = rowfull[ix];
                if (do1) orow1[ix] = v;
                if (do2) orow2[ix] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    }
}

// ------------------------------------------------------------------------------------
// K3: spectrum *= beam spectrum (complex, [chunk][P][Ph]).  Grid-stride, 16 B per lane.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
jx_beam_mul_kernel(double2* __restrict__ spec, const double2* __restrict__ bhat, size_t per_walker, size_t total) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const double2 b = bhat[i % per_walker];
        const double2 a = spec[i];
        double2 r;
        r.x = a.x * b.x - a.y * b.y;
        r.y = a.x * b.y + a.y * b.x;
        spec[i] = r;
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

// ------------------------------------------------------------------------------------
// K6+K7: transfer function, central row, conversion, chi^2, total.  One block per walker.
//   tfspec [chunk][S][Sh] = rfft2 of the beam-convolved S x S window (unnormalised)
//   Z[kc]  = sum_kr tfspec[kr][kc] * H[kr][kc]          (threads over kc: coalesced)
//   row[c] = sum_kc Re(Z[kc] e^{2 pi i kc c/S}),  c = S//2 .. S-1   (joxsz_funcs.py:467,472)
//   map_prof = row * cfac                                           (joxsz_funcs.py:473)
//   model_d = sum_k E[d][k] map_prof[k]                             (joxsz_funcs.py:476)
//   chisq = nansum(((flux - model)/err)^2); logp = base - chisq/2   (joxsz_funcs.py:478-479, 538)
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(JX_TAIL_THREADS)
jx_tail_kernel(JxDev c, const double2* __restrict__ tfspec, const double* __restrict__ zin /* [walker][2][Sh]: Z already summed (jx_fft.hpp), or null */,
               const double* __restrict__ cfac, const double* __restrict__ sz0,
               const double* __restrict__ base, double* __restrict__ logp, int w0,
               double* __restrict__ tap_row, double* __restrict__ tap_bright, double* __restrict__ tap_chisq,
               double* __restrict__ tap_parts) {
    JX_LDS_DECL;
    double* red = sm + 20;
    const int S = c.S, Sh = c.Sh, nrow = c.nrow;
    double* s_zr = sm + JX_LDS_HDR;    // [Sh]
    double* s_zi = s_zr + Sh;          // [Sh]
    double* s_prof = s_zi + Sh;        // [nrow]
    const int w = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const double2* X = tfspec + (size_t)w * S * Sh;
    const double2* H = reinterpret_cast<const double2*>(c.htab);

    if (zin) {
        for (int kc = tid; kc < Sh; kc += nth) {
            s_zr[kc] = zin[((size_t)w * 2) * Sh + kc];
            s_zi[kc] = zin[((size_t)w * 2 + 1) * Sh + kc];
        }
    } else {
        for (int kc = tid; kc < Sh; kc += nth) {
            double zr0 = 0.0, zi0 = 0.0, zr1 = 0.0, zi1 = 0.0;
            int kr = 0;
            for (; kr + 1 < S; kr += 2) {
                const double2 x0 = X[(size_t)kr * Sh + kc], h0 = H[(size_t)kr * Sh + kc];
                const double2 x1 = X[(size_t)(kr + 1) * Sh + kc], h1 = H[(size_t)(kr + 1) * Sh + kc];
                zr0 += x0.x * h0.x - x0.y * h0.y;  zi0 += x0.x * h0.y + x0.y * h0.x;
                zr1 += x1.x * h1.x - x1.y * h1.y;  zi1 += x1.x * h1.y + x1.y * h1.x;
            }
            for (; kr < S; ++kr) {
                const double2 x0 = X[(size_t)kr * Sh + kc], h0 = H[(size_t)kr * Sh + kc];
                zr0 += x0.x * h0.x - x0.y * h0.y;  zi0 += x0.x * h0.y + x0.y * h0.x;
            }
            s_zr[kc] = zr0 + zr1;
            s_zi[kc] = zi0 + zi1;
        }
    }
    __syncthreads();

    // the S roots of unity of the row's inverse transform: into LDS first (the loop below reads one per term at a data-dependent index)
    double2* tw = reinterpret_cast<double2*>(s_prof + ((nrow + 1) & ~1));
    {
        const double2* twg = reinterpret_cast<const double2*>(c.twid);
        for (int m = tid; m < S; m += nth) tw[m] = twg[m];
    }
    __syncthreads();
    const int c0 = S / 2;
    for (int k = tid; k < nrow; k += nth) {
        const int col = c0 + k;
        double acc = 0.0;
        int ph = 0;                                  // (kc * col) mod S
        for (int kc = 0; kc < Sh; ++kc) {
            const double2 t = tw[ph];
            acc += s_zr[kc] * t.x - s_zi[kc] * t.y;
            ph += col; if (ph >= S) ph -= S;
        }
        if (tap_row) tap_row[(size_t)w * nrow + k] = acc;
        const double b = acc * cfac[(size_t)w * nrow + k];
        s_prof[k] = b;
        if (tap_bright) tap_bright[(size_t)w * nrow + k] = b;
    }
    __syncthreads();

    double part = 0.0;
    for (int d = tid; d < c.nflux; d += nth) {
        const double* e = c.emat + (size_t)d * nrow;
        double m = 0.0;
        for (int k = 0; k < nrow; ++k) m = fma(e[k], s_prof[k], m);
        const double z = (c.flux[c.nflux + d] - m) / c.flux[2 * c.nflux + d];
        const double z2 = z * z;
        if (z2 == z2) part += z2;                    // np.nansum drops NaN terms
    }
    const double chisq = jx_block_sum(part, red);
    if (tid == 0) {
        const double ll = -chisq / 2.0 + (sz0 ? sz0[w] : 0.0);
        const double b = base[w];
        double tot = (b == -INFINITY) ? -INFINITY : b + ll;
        if (tot != tot) tot = -INFINITY;             // never hand NaN to the sampler
        logp[w0 + w] = tot;
        if (tap_chisq) tap_chisq[w] = chisq;
        if (tap_parts) tap_parts[(size_t)w * 4 + 1] = ll;
    }
}

// ------------------------------------------------------------------------------------
// Collapsed route (jx_set_route(ctx, JX_ROUTE_OPERATOR)).  Between the pressure profile and the extracted map row
// every step of joxsz_funcs.py:457-472 is linear with constant coefficients (Abel matrix, spline through fixed knots
// evaluated at fixed radii, beam convolution, transfer function, row extraction), so
//     map_row[x] = sum_j G[x][j] pp[j]
// with one constant nrow x N matrix.  G is not derived separately: jx_set_route pushes the N unit profiles through
// the kernels above (inject_pp) and stores their rows, Gt[j][x] (row stride ldg).  This kernel is then the whole SZ
// side of a walker: pp (left in HBM by the prep kernel, which needs it anyway) -> G pp -> conversion -> data radii ->
// chi^2 -> total (as jx_tail_kernel).
// One block of 256 threads for WPB walkers (4, 8 or 16 by launch size): a G entry fetched once serves all of them --
// every block streams the whole of G out of L2, which is what bounds this kernel, so large launches take more walkers
// per block.  The profiles pass through LDS in chunks of JC radii (512, 256, 128: the
// fewer walkers, the longer the chunk).  Each walker's sums run over j in the same
// order whatever WPB is: a result does not depend on the launch it was part of.
// LDS: [WPB][JC] profile chunk, [WPB][nrow_e] rows, 8 scratch.
// ------------------------------------------------------------------------------------
template <int WPB, int JC>
__global__ void __launch_bounds__(256)
jx_operator_kernel(JxDev c, const double* __restrict__ pp /*[launch][N], written by jx_prep_kernel*/, int w0, int n,
                   const double* __restrict__ Gt /*[N+4][ldg], the last four rows zero*/, int ldg,
                   const double* __restrict__ rows_t /*[launch / 32][nrow][32] G pp from jx_operator_mfma_kernel; or null*/,
                   const double* __restrict__ cfac, const double* __restrict__ sz0, const double* __restrict__ base, double* __restrict__ logp,
                   double* __restrict__ tap_row, double* __restrict__ tap_bright, double* __restrict__ tap_chisq,
                   double* __restrict__ tap_parts) {
    JX_LDS_DECL;
    const int N = c.N, nrow = c.nrow, Re = (nrow + 1) & ~1;
    double* s_pp = sm;                                  // [WPB][JC]
    double* s_prof = s_pp + WPB * JC;             // [WPB][Re]
    const int tid = threadIdx.x, nth = blockDim.x;
    const int wb = blockIdx.x * WPB;
    const int nw = min(WPB, n - wb);

    for (int xs = 0; xs < nrow; xs += nth) {            // slabs of 256 rows (one, up to S = 512)
        const int x = xs + tid;
        const bool live = x < nrow;
        double a[WPB][2];
#pragma unroll
        for (int k = 0; k < WPB; ++k) a[k][0] = a[k][1] = 0.0;
        if (rows_t && live) {
#pragma unroll
            for (int k = 0; k < WPB; ++k) {             // [walker / 32][x][walker % 32]: WPB divides 32, so one 32-block per thread block
                const int wq = wb + (k < nw ? k : 0);
                a[k][0] = rows_t[((size_t)(wq >> 5) * nrow + x) * 32 + (wq & 31)];
                a[k][1] = 0.0;
            }
        }
        for (int j0 = 0; j0 < (rows_t ? 0 : N); j0 += JC) {
            __syncthreads();                            // the previous chunk has been used up
            for (int q = tid; q < WPB * JC; q += nth) {
                const int k = q / JC, j = q - k * JC;
                // (a missing walker repeats the block's first: finite work, never stored)
                s_pp[q] = (j0 + j < N) ? pp[(size_t)(wb + (k < nw ? k : 0)) * N + j0 + j] : 0.0;
            }
            __syncthreads();
            if (live) {
                const int jn = min(JC, (N - j0 + 1) & ~1);          // even count: G has zero rows behind row N-1
                const double* g = Gt + (size_t)j0 * ldg + x;
#pragma unroll 4
                for (int j = 0; j < jn; j += 2) {           // (unrolled: eight G loads in flight per lane)
                    const double g0 = g[(size_t)j * ldg], g1 = g[(size_t)(j + 1) * ldg];
#pragma unroll
                    for (int k = 0; k < WPB; ++k) {
                        const double2 v = *reinterpret_cast<const double2*>(s_pp + k * JC + j);
                        a[k][0] = fma(g0, v.x, a[k][0]);
                        a[k][1] = fma(g1, v.y, a[k][1]);
                    }
                }
            }
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < WPB; ++k) {
                const double row = a[k][0] + a[k][1];
                double b = 0.0;
                if (k < nw) {
                    const size_t o = (size_t)(wb + k) * nrow + x;
                    b = row * cfac[o];
                    if (tap_row) tap_row[o] = row;
                    if (tap_bright) tap_bright[o] = b;
                }
                s_prof[k * Re + x] = b;
            }
        }
    }
    __syncthreads();

    // data radii and chi^2: one data point per wave at a time, lanes along the row (emat rows are contiguous).  Every
    // emat entry a wave will need is requested before the first is used (one L2 round trip instead of one per use).
    double part[WPB];
#pragma unroll
    for (int k = 0; k < WPB; ++k) part[k] = 0.0;
    const int lane = tid & 63, wv = tid >> 6, nwv = nth >> 6;
    constexpr int NXM = 4, NDM = 8;                     // up to 256 row entries per lane set, 8 data points per wave
    if (nrow <= 64 * NXM && c.nflux <= nwv * NDM) {
        double ev[NDM][NXM], fv[NDM], erv[NDM];
#pragma unroll
        for (int dd = 0; dd < NDM; ++dd) {
            const int d = wv + dd * nwv;
            fv[dd] = (d < c.nflux) ? c.flux[c.nflux + d] : 0.0;
            erv[dd] = (d < c.nflux) ? c.flux[2 * c.nflux + d] : 1.0;
#pragma unroll
            for (int i = 0; i < NXM; ++i) {
                const int x = lane + 64 * i;
                ev[dd][i] = (d < c.nflux && x < nrow) ? c.emat[(size_t)d * nrow + x] : 0.0;
            }
        }
        double pv[WPB][NXM];
#pragma unroll
        for (int k = 0; k < WPB; ++k)
#pragma unroll
            for (int i = 0; i < NXM; ++i) { const int x = lane + 64 * i; pv[k][i] = (x < nrow) ? s_prof[k * Re + x] : 0.0; }
#pragma unroll
        for (int dd = 0; dd < NDM; ++dd) {
            const int d = wv + dd * nwv;
            if (d < c.nflux) {                          // (wave-uniform)
                double m[WPB];
#pragma unroll
                for (int k = 0; k < WPB; ++k) {
                    m[k] = 0.0;
#pragma unroll
                    for (int i = 0; i < NXM; ++i) if (lane + 64 * i < nrow) m[k] = fma(ev[dd][i], pv[k][i], m[k]);
                }
#pragma unroll
                for (int k = 0; k < WPB; ++k)
                    for (int off = 32; off > 0; off >>= 1) m[k] += __shfl_xor(m[k], off, 64);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < WPB; ++k) {
                        const double z = (fv[dd] - m[k]) / erv[dd];
                        const double z2 = z * z;
                        if (z2 == z2) part[k] += z2;     // np.nansum drops NaN terms
                    }
                }
            }
        }
    } else {
        for (int d = wv; d < c.nflux; d += nwv) {
            const double* e = c.emat + (size_t)d * nrow;
            double m[WPB];
#pragma unroll
            for (int k = 0; k < WPB; ++k) m[k] = 0.0;
            for (int x = lane; x < nrow; x += 64) {
                const double ev = e[x];
#pragma unroll
                for (int k = 0; k < WPB; ++k) m[k] = fma(ev, s_prof[k * Re + x], m[k]);
            }
#pragma unroll
            for (int k = 0; k < WPB; ++k)
                for (int off = 32; off > 0; off >>= 1) m[k] += __shfl_xor(m[k], off, 64);
            if (lane == 0) {
                const double f = c.flux[c.nflux + d], er = c.flux[2 * c.nflux + d];
#pragma unroll
                for (int k = 0; k < WPB; ++k) {
                    const double z = (f - m[k]) / er;
                    const double z2 = z * z;
                    if (z2 == z2) part[k] += z2;         // np.nansum drops NaN terms
                }
            }
        }
    }
    // lane 0 of every wave holds its share of each walker's chi^2: one pass through LDS adds the (at most four) shares
    const double base_pre = (tid < nw) ? base[wb + tid] : 0.0;      // (requested ahead of the two barriers)
    __syncthreads();                                    // s_pp is free now
    double* s_part = s_pp;                              // [nwv][WPB]
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < WPB; ++k) s_part[wv * WPB + k] = part[k];
    }
    __syncthreads();
    if (tid < nw) {
        double chisq = 0.0;
        for (int v = 0; v < nwv; ++v) chisq += s_part[v * WPB + tid];
        const int w = wb + tid;
        const double ll = -chisq / 2.0 + (sz0 ? sz0[w] : 0.0);
        const double b = base_pre;
        double tot = (b == -INFINITY) ? -INFINITY : b + ll;
        if (tot != tot) tot = -INFINITY;                 // never hand NaN to the sampler
        logp[w0 + w] = tot;
        if (tap_chisq) tap_chisq[w] = chisq;
        if (tap_parts) tap_parts[(size_t)w * 4 + 1] = ll;
    }
}

// G pp of large launches on the fp64 matrix cores (v_mfma_f64_16x16x4; operand layouts as in jx_lowrank_kernel: A lane l
// = A[l & 15][4 s + (l >> 4)], B lane l = B[4 s + (l >> 4)][l & 15], D lane l register g = D[(l >> 4) + 4 g][l & 15]).
// A = G (rows x, K = radii j; read straight from L2, 128 contiguous bytes per k-row of a tile), B = the pressure
// profiles of 32 walkers (two 16-walker tiles sharing every A fragment), staged through LDS radius-major.  A block of
// four waves covers all rows: wave v owns the row tiles [v MT, (v+1) MT).  G is fetched once per 32 walkers.
// Output per block, walker-minor: rows_t[block][x][32 walkers] (64 KB contiguous per block).  (The sums run in the matrix core's order, not in jx_operator_kernel's: the
// two agree to rounding, not bit for bit.)
// LDS: [JX_OPM_JC][33] profile chunk.
#define JX_OPM_JC 128
#define JX_OPM_GPAD 32
typedef double jx_op_v4d __attribute__((ext_vector_type(4)));
template <int MT>
__global__ void __launch_bounds__(256)
jx_operator_mfma_kernel(const double* __restrict__ pp /*[launch][N]*/, int n, int N, int nrow,
                        const double* __restrict__ Gt /*[N + JX_OPM_GPAD][ldg]*/, int ldg, double* __restrict__ rows_t /*[launch / 32][nrow][32]*/) {
    JX_LDS_DECL;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb = blockIdx.x * 32;
    const int ntile = (nrow + 15) >> 4;
    jx_op_v4d acc[MT][2];
#pragma unroll
    for (int t = 0; t < MT; ++t) { acc[t][0] = jx_op_v4d{0.0, 0.0, 0.0, 0.0}; acc[t][1] = jx_op_v4d{0.0, 0.0, 0.0, 0.0}; }
    // A fragments three k-steps ahead of their use in a ring of four named slots (the k loop is unrolled by four, so no
    // value moves between registers and the waits count the requests in flight); G has JX_OPM_GPAD zero rows behind row N-1
    int toff[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) toff[t] = min(wv * MT + t, ntile - 1) * 16;   // (a tile past the last repeats it: never stored)
    const double* gb = Gt + (size_t)lk * ldg + li;
    double a[4][MT];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int t = 0; t < MT; ++t) a[u][t] = gb[(size_t)(4 * u) * ldg + toff[t]];
    // the profiles of the next chunk wait in registers while this chunk is multiplied
    constexpr int NST = 32 * JX_OPM_JC / 256;
    const int sl = tid / JX_OPM_JC, sj = tid % JX_OPM_JC;                       // consecutive threads: consecutive radii of one walker
    double stg[NST];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int l = sl + i * (256 / JX_OPM_JC);
            stg[i] = (wb + l < n && j0 + sj < N) ? pp[(size_t)(wb + l) * N + j0 + sj] : 0.0;
        }
    };
    fetch(0);
    int kg = 0;                                                                // global k-step
    for (int j0 = 0; j0 < N; j0 += JX_OPM_JC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NST; ++i) sm[sj * 33 + sl + i * (256 / JX_OPM_JC)] = stg[i];
        __syncthreads();
        if (j0 + JX_OPM_JC < N) fetch(j0 + JX_OPM_JC);
        const int kgroups = (min(JX_OPM_JC, N - j0) + 15) >> 4;
        for (int sg = 0; sg < kgroups; ++sg, kg += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int t = 0; t < MT; ++t) a[(u + 3) & 3][t] = gb[(size_t)(4 * (kg + u + 3)) * ldg + toff[t]];
                const double b0 = sm[(4 * (4 * sg + u) + lk) * 33 + li], b1 = sm[(4 * (4 * sg + u) + lk) * 33 + 16 + li];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    acc[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][t], b0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][t], b1, acc[t][1], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int tile = wv * MT + t;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int w = wb + nt * 16 + li;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int x = tile * 16 + lk + 4 * g;
                if (tile < ntile && x < nrow && w < n) rows_t[((size_t)blockIdx.x * nrow + x) * 32 + nt * 16 + li] = acc[t][nt][g];
            }
        }
    }
}

// Phases 2-3 of the Abel kernel for a whole launch as ONE matrix product on the fp64 matrix cores (default route).
// The forward Abel transform (PyAbel direct_transform, joxsz_funcs.py:457), the Compton-y scale and the moments of the
// mirrored cubic spline (interp1d 'cubic', joxsz_funcs.py:460) are linear in the pressure profile with constant
// coefficients, so
//     cf[w][2k] = y_k = sum_j Tm[j][2k] pp[w][j],     cf[w][2k+1] = M_k = sum_j Tm[j][2k+1] pp[w][j]
// with Tm built once at jx_finalize (y_scale A and G_band y_scale A, the second accumulated in long double).  pp comes
// from jx_prep_kernel, which evaluates the pressure on the whole grid anyway.  The kernels behind this one read cf as
// before.  Operand layouts as in jx_lowrank_kernel; here A = the profiles of 32 walkers (two 16-walker tiles, staged
// through LDS radius-major) and B = Tm tiles read straight from L2 two k-steps ahead of their use, so that D comes out
// walker-major: register g of lane l = D[walker (l >> 4) + 4 g][column l & 15], i.e. 128-byte runs of a walker's array.
// Tm is upper triangular up to the spline's band: column tile t (knots 8t .. 8t+7) has no entry above row 8t - K.  A
// wave therefore owns NPW PAIRS of tiles (p, last - p), whose k-ranges add up to the same length for every pair, and
// skips the matrix instructions of a tile above its first row (the fragment loads are clamped to that row instead:
// they stay in the L1).
#define JX_AG_R 8
#define JX_AG_ROWS(N) ((((N) + 4 * JX_AG_R - 1) / (4 * JX_AG_R) + 1) * (4 * JX_AG_R))
// TR = 1: the result goes out walker-minor for the contracted route (jx_mix.hpp): cf[(k * cf_ws + w) * 2 + {0: y_k, 1: M_k}],
// cf_ws = walker stride; columns >= ncol are not stored.
// NWT = 16-walker tiles per block (2, or 1 for launches too small to give every SIMD a wave otherwise: a walker's sums are
// the same either way).
// SINGLE = 1: a wave owns ONE column tile (tile 4 * blockIdx.y + wave; the four of a block adjacent, so their k-ranges differ
// by six steps at most) -- twice the blocks of the paired form, two per CU: one block's barriers and first-touch waits then
// overlap the other's matrix instructions, and the fp64 matrix instruction issues every 114 cycles from two waves per SIMD
// instead of every 156 from one.  Blocks with the longest k-ranges have the lowest blockIdx.y: dispatched first.
// RS: the radial sub-grid (a compile-time choice: the plain form keeps its straight-line staging loads)
template <int NPW, typename TO = double, int TR = 0, int NWT = 2, int SINGLE = 0, bool RS = false>
__global__ void __launch_bounds__(256)
jx_abel_gemm_kernel(const double* __restrict__ pp /*[launch][N]*/, int n, int N, const double* __restrict__ Tm /*[JX_AG_ROWS(N)][ldt], zero rows behind N-1*/,
                    int ldt, int K, int ntile, int npair, TO* __restrict__ cf /*[launch][cf_ws]; float for the fp32 variant (rounded once, on store)*/, long long cf_ws,
                    long long ncol = 0,
                    // radial sub-grid (DESIGN of round 4, 6.3): the rows of Tm are then N of the Npp radii of a profile (rsub[k] = the radius of row k), the
                    // interpolation to the others folded into Tm, and tks[t] = the first k-step of column tile t with entries (null: all radii)
                    const int* __restrict__ rsub = nullptr, int Npp = 0, const int* __restrict__ tks = nullptr) {
    JX_LDS_DECL;
    static_assert(!SINGLE || NPW == 1, "one tile per wave");
    constexpr int NTL = SINGLE ? 1 : 2 * NPW;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb = blockIdx.x * 16 * NWT, grp = blockIdx.y;
    const int ktot4 = (N + 4 * JX_AG_R - 1) / (4 * JX_AG_R) * JX_AG_R;          // k-steps, in whole groups of JX_AG_R
    int tile[NTL], ks[NTL], toff[NTL];
    const bool tk = RS || tks != nullptr;                      // the first k-step of a column tile from a table (radial sub-grid; ordinates-only operator of the exact form)
    if (SINGLE) {
        const int t = grp * 4 + wv;
        tile[0] = t;
        ks[0] = (t < ntile) ? (tk ? tks[t] : (max(0, 8 * t - K) >> 2)) : ktot4;
        toff[0] = min(t, ntile - 1) * 16;
    } else {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int p = (grp * 4 + wv) * NPW + i;
            tile[2 * i] = p; tile[(2 * i + 1) % NTL] = 2 * npair - 1 - p;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = tile[(2 * i + h) % NTL];
                ks[(2 * i + h) % NTL] = (p < npair && t < ntile) ? (tk ? tks[t] : (max(0, 8 * t - K) >> 2)) : ktot4;   // first k-step with entries (ktot4: none)
                toff[(2 * i + h) % NTL] = min(t, ntile - 1) * 16;
            }
        }
    }
    jx_op_v4d acc[NTL][NWT];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int nt = 0; nt < NWT; ++nt) acc[t][nt] = jx_op_v4d{0.0, 0.0, 0.0, 0.0};
    // the block's first chunk of radii: the first row of its lowest tile
    const int kminb = tk ? tks[min(grp * 4 * NPW, ntile - 1)] : (max(0, 8 * (grp * 4 * NPW) - K) >> 2);      // (tks does not decrease with the tile)
    const int j00 = (4 * kminb / JX_OPM_JC) * JX_OPM_JC;
    int kg = j00 >> 2;
    const double* gb = Tm + (size_t)lk * ldt + li;
    // Tm fragments JX_AG_R - 1 k-steps ahead of their use in a ring of JX_AG_R named slots (the k loop is unrolled by as
    // many, so no value is ever moved between registers and the waits count the requests still in flight): with one
    // wave per SIMD nothing else hides the L2 round trip
    constexpr int R = JX_AG_R;
    double a[R][NTL];
#pragma unroll
    for (int u = 0; u < R - 1; ++u)
#pragma unroll
        for (int t = 0; t < NTL; ++t) a[u][t] = gb[(size_t)(4 * max(kg + u, ks[t])) * ldt + toff[t]];
    // the profiles of the next chunk wait in registers while this chunk is multiplied
    constexpr int NST = 16 * NWT * JX_OPM_JC / 256;
    const int sl = tid / JX_OPM_JC, sj = tid % JX_OPM_JC;                       // consecutive threads: consecutive radii of one walker
    double stg[NST];
    auto fetch = [&](int j0) {
        const int rcol = RS ? rsub[min(j0 + sj, N - 1)] : 0;          // (this thread's radius of the chunk)
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int l = sl + i * (256 / JX_OPM_JC);
            if (RS) stg[i] = (wb + l < n && j0 + sj < N) ? pp[(size_t)(wb + l) * Npp + rcol] : 0.0;
            else stg[i] = (wb + l < n && j0 + sj < N) ? pp[(size_t)(wb + l) * N + j0 + sj] : 0.0;
        }
    };
    fetch(j00);
    for (int j0 = j00; j0 < N; j0 += JX_OPM_JC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NST; ++i) sm[sj * 33 + sl + i * (256 / JX_OPM_JC)] = stg[i];
        __syncthreads();
        if (j0 + JX_OPM_JC < N) fetch(j0 + JX_OPM_JC);
        const int kgroups = (min(JX_OPM_JC, N - j0) + 4 * R - 1) / (4 * R);
        for (int sg = 0; sg < kgroups; ++sg, kg += R) {
            // the group's profile fragments leave the LDS together (one wait per R k-steps, not one in front of every product)
            double pv[R][NWT];
#pragma unroll
            for (int u = 0; u < R; ++u)
#pragma unroll
                for (int nt = 0; nt < NWT; ++nt) pv[u][nt] = sm[(4 * (R * sg + u) + lk) * 33 + nt * 16 + li];
#pragma unroll
            for (int u = 0; u < R; ++u) {
#pragma unroll
                for (int t = 0; t < NTL; ++t) a[(u + R - 1) % R][t] = gb[(size_t)(4 * max(kg + u + R - 1, ks[t])) * ldt + toff[t]];
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    if (kg + u >= ks[t]) {
#pragma unroll
                        for (int nt = 0; nt < NWT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(pv[u][nt], a[u][t], acc[t][nt], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const long long col = (long long)tile[t] * 16 + li;
        if (ks[t] >= ktot4 || col >= (TR ? ncol : cf_ws)) continue;
#pragma unroll
        for (int nt = 0; nt < NWT; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int w = wb + nt * 16 + lk + 4 * g;
                if (w < n) {
                    if (TR) cf[((size_t)(col >> 1) * cf_ws + w) * 2 + (col & 1)] = (TO)acc[t][nt][g];
                    else cf[(size_t)w * cf_ws + col] = (TO)acc[t][nt][g];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Device-resident stretch move (the caller of the hot path, joxsz_funcs.py:593-622 through emcee's
// red/blue StretchMove).  Random numbers: Philox4x32-10 (Salmon et al. 2011), key = seed, counter =
// (walker slot, 2 * iteration + half, draw, 0); a uniform double is ((hi << 32 | lo) >> 11) * 2^-53.
// The arithmetic of the proposal is written without fused multiply-adds so that a host implementation with
// separately rounded operations (joxsz_amd/sampler.py::DeviceStretchMove.replay) reproduces it bit for bit.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void jx_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double jx_u01(uint32_t hi, uint32_t lo) {
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

// proposals for the `half` walkers [s1, s1 + half) against the complementary ensemble [s2, s2 + half)
__global__ void __launch_bounds__(256)
jx_sm_propose_kernel(const double* __restrict__ x, double* __restrict__ q, double* __restrict__ zz, int ndim, int half, int s1, int s2,
                     int iter2, double a, uint64_t seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    uint32_t r[4];
    jx_philox((uint32_t)i, (uint32_t)iter2, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const double u1 = jx_u01(r[0], r[1]), u2 = jx_u01(r[2], r[3]);
    const double t = __dadd_rn(__dmul_rn(a - 1.0, u1), 1.0);
    const double z = __ddiv_rn(__dmul_rn(t, t), a);                   // ((a-1) u + 1)^2 / a
    int j = (int)__dmul_rn(u2, (double)half);
    j = min(j, half - 1);
    const double* xp = x + (size_t)(s2 + j) * ndim;
    const double* xi = x + (size_t)(s1 + i) * ndim;
    for (int d = 0; d < ndim; ++d) q[(size_t)i * ndim + d] = __dsub_rn(xp[d], __dmul_rn(__dsub_rn(xp[d], xi[d]), z));
    zz[i] = z;
}

__global__ void __launch_bounds__(256)
jx_sm_accept_kernel(double* __restrict__ x, double* __restrict__ lp, const double* __restrict__ q, const double* __restrict__ lq,
                    const double* __restrict__ zz, long long* __restrict__ nacc, int ndim, int half, int s1, int iter2, uint64_t seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    uint32_t r[4];
    jx_philox((uint32_t)i, (uint32_t)iter2, 1u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const double u3 = jx_u01(r[0], r[1]);
    const double lnew = lq[i];
    const double lnpdiff = __dadd_rn(__dmul_rn((double)(ndim - 1), log(zz[i])), __dsub_rn(lnew, lp[s1 + i]));
    if (isfinite(lnew) && log(u3) < lnpdiff) {
        for (int d = 0; d < ndim; ++d) x[(size_t)(s1 + i) * ndim + d] = q[(size_t)i * ndim + d];
        lp[s1 + i] = lnew;
        nacc[s1 + i] += 1;
    }
}
