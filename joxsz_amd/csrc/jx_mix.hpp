// Contracted forms of the SZ side (rounds 3-4; the default until round 4, since round 5 behind the option JOXSZ_MIX_FORM=legacy|lowrank|full
// and kept for one round beside the exact form of jx_exact.hpp, which replaced them: DESIGN 10): the sum over map rows is taken BEFORE any
// transform, so no row is ever transformed and no row spectrum ever reaches HBM.
//
// Reference lines computed (joxsz_funcs.py:462-467, row of :472):
//     y_2d    = f(d_mat)                                     S x S samples of the mirrored cubic spline
//     conv_2d = fftconvolve(y_2d, beam_2d, 'same') * step^2
//     map_out = real(ifft2(fft2(conv_2d) * filtering));      out[x] = map_out[S//2, S//2 + x]
// y_2d[m][n] = Q[|m-c|][|n-c|] (mirror structure of centdistmat), Q the NU x NU quadrant of distinct samples, and the map
// from Q to out is linear with constant coefficients.  Two forms of it, chosen at jx_finalize by the cost of each:
//
//   low-rank form (separable beam image, smooth transfer function -- every BASELINE config):
//     step^2 beam[a][b] = sum_s by_s[a] bx_s[b]   (one term for the Gaussian branch of mybeam, joxsz_funcs.py:69-71)
//     Hy[q][kx] = sum_rho U[rho][q] v_rho[kx]     (transfer-function weights of the extracted row, truncated SVD)
//     stage 1  jx_rowmix_kernel   D[x'][j] = sum_u C[j][u] Q[u][x'],  j = (rho, s), R = r * ns values per column
//              -- the ONLY pass over the samples: each is evaluated from the walker's (y_k, M_k) in registers
//              (f = A y_k + B y_k+1 + C M_k + D M_k+1, joxsz_funcs.py:460-462) and goes straight into R fused
//              multiply-adds; the map itself is never stored
//     stage 2  jx_opgemm_kernel<LOAD>   out[x] = sum_{x',j} G[x][(x',j)] D[x'][j]   (beam along x, circular kernel of
//              term rho along the row, row extraction: one constant nrow x (NU R) matrix; fp64 matrix cores)
//   full form (any beam image, any real transfer function -- the reference's measured inputs, joxsz_main.py:59-60):
//     jx_opgemm_kernel<EVAL>   out[x] = sum_{(u,x')} Omega[x][(u,x')] Q[u][x'],  samples evaluated by the lanes that feed
//              them to the matrix cores; exact (no truncation), cost independent of smoothness
//   jx_tail_row_kernel  sums the K-slice partials in fixed order, conversion, data radii, chi^2, total (funcs:472-479, 538)
//
// Spline arrays arrive walker-minor: cft[k][w] = (y_k, M_k) of walker w as one double2 (jx_abel_gemm_kernel, TR = 1), so a
// wave whose lanes are 64 consecutive walkers reads one knot of all of them as 1 KB contiguous.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jx_kernels.hpp"

typedef double jx_mx_v4d __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------------
// Stage 1.  One wave = one column x' of the quadrant x 64 walkers (lane = walker).  The wave walks down the column,
// u = 0 .. NU-1; the radius grows with u, so the spline interval index never decreases and a knot fetched once serves
// every sample of its interval: the column is a list of SEGMENTS (one per knot interval, seg[j] = number of samples in
// it) and the knots sit in a ring of NS named register slots, fetched NS-2 segments before their first use (the segment
// loop is unrolled NS times: no value ever moves between registers and the waits count the requests in flight).
// Everything that does not depend on the walker is wave-uniform and arrives through the scalar unit: the four spline
// weights of the sample and the R coefficients C[u][0..R) of the row, which enter the multiply-adds as SGPR operands.
// Per sample and walker: 4 + R fp64 FMAs and on average 0.65 x 16 B through the vector L1.
//   cft  [N + pad][tW] double2      Dt [NU * R][tW]   (row x' * R + j, walker-minor)
// ------------------------------------------------------------------------------------------------------------------
struct JxMix {
    int NU, R, n;                  // quadrant side, combined rows per column, walkers of this launch
    long long tW;                  // walker stride of cft and Dt (multiple of 128)
    unsigned cft_bytes;            // size of cft (buffer descriptor range)
    int segld, wld, cld;           // strides: segment counts per column, samples per column (>= NU), C row (>= RT)
    int dbg;                       // diagnostic build only (make ABLATIONS=1): timing experiments, results are wrong
    int cper;                      // block -> (column, walker-group set): XCDs per set (8 / sets) when that divides, else 0
    int usplit;                    // pieces a column is walked in, one wave each (piece v = x' * usplit + h); their sums meet in LDS
    const int* urange;             // [NU * usplit] first row u of the piece | number of rows << 16
    const int* seg0;               // [NU * usplit]  first knot interval of the piece
    const int* nseg;               // [NU * usplit]  number of segments of the piece
    const int* seg;                // [NU * usplit][segld] samples per segment
    const double* w4;              // [NU][wld][4] weights (A, B, C, D) of sample u of column x'
    const double* Cm;              // [wld][cld]   C[u][j], zero padded
    const float* w4f;              // fp32 arithmetic (dtype 2): the same two tables rounded to fp32
    const float* Cmf;
};

// Profile taps off the matrix product's arrays (radial grids too long for the Abel kernel): y[w][k] = cft[k][w].x, ab = y / y_scale
__global__ void __launch_bounds__(256)
jx_unpack_splines_kernel(const double2* __restrict__ cft, long long tW, int N, double y_scale, double* __restrict__ tap_y, double* __restrict__ tap_ab) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (k >= N) return;
    const double y = cft[(size_t)k * tW + w].x;
    if (tap_y) tap_y[(size_t)w * N + k] = y;
    if (tap_ab) tap_ab[(size_t)w * N + k] = y / y_scale;
}

typedef unsigned jx_mx_u4 __attribute__((ext_vector_type(4)));
typedef unsigned jx_mx_u2 __attribute__((ext_vector_type(2)));
template <typename TC> __device__ __forceinline__ TC jx_mx_ldknot(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ double2 jx_mx_ldknot<double2>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const jx_mx_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    double2 r;
    r.x = __hiloint2double((int)v.y, (int)v.x);
    r.y = __hiloint2double((int)v.w, (int)v.z);
    return r;
}
template <> __device__ __forceinline__ float2 jx_mx_ldknot<float2>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const jx_mx_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}

template <int RT, int NS, typename TC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT <= 18 ? 6 : 1)))
jx_rowmix_kernel(JxMix m, const TC* __restrict__ cft, double* __restrict__ Dt) {
    static_assert(NS == 8, "one aligned 8-dword scalar load carries the sample counts of a group of NS segments");
    extern __shared__ __attribute__((aligned(16))) double sm_mix[];            // [gpb][usplit - 1][RT][64] sums of the later pieces
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
    // A block = one column x' and gpb walker groups, each group's column walked in `usplit` pieces by as many waves: what
    // bounds a small launch is the length of a wave's walk, not the arithmetic, so the walk is cut (the pieces' sums are
    // added in a fixed order through LDS at the end: a result does not depend on the launch).  The waves of a piece stream
    // the same weights and coefficients through the scalar cache.  XCD-aware: block id % 8 is the XCD; the blocks of a set
    // of walker groups share XCDs, so the set's spline arrays stay in those L2s.
    const int usp = m.usplit, gpb = wpb / usp;
    const int xh = wv % usp, gi = wv / usp;
    const int ngrp = (m.n + 63) >> 6, nq = (ngrp + gpb - 1) / gpb;
    const int id = blockIdx.x;
    int gq, xq;
    if (m.cper > 0) { const int xcd = id & 7, jj = id >> 3; gq = xcd % nq; xq = xcd / nq + m.cper * jj; }
    else { gq = id % nq; xq = id / nq; }
    const int grp = gq * gpb + gi;
    const bool active = grp < ngrp && xq < m.NU;               // (no early exit: every wave of the block meets the barrier below)
    const int xv = min(xq, m.NU - 1) * usp + xh;               // piece
    const int ur = __builtin_amdgcn_readfirstlane(m.urange[xv]), ubeg = ur & 0xffff, ucnt = ur >> 16;
    const size_t w = (size_t)grp * 64 + lane;
    const size_t tW = (size_t)m.tW;
    // knots through a buffer descriptor: per-lane byte offset in a VGPR that never changes, the knot's offset in an SGPR
    // -- no address arithmetic on the vector unit and no address registers for the compiler to recycle out of the ring
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<TC*>(cft), 0, m.cft_bytes, 0x00020000);
    const unsigned loff = (unsigned)((unsigned)grp * 64u + (unsigned)lane) * (unsigned)sizeof(TC);
    const unsigned kstride = (unsigned)(tW * sizeof(TC));
    const int k0 = __builtin_amdgcn_readfirstlane(m.seg0[xv]), nseg = active ? __builtin_amdgcn_readfirstlane(m.nseg[xv]) : 0;
    const int* __restrict__ sc = m.seg + (size_t)xv * m.segld;                 // [segld], zero padded to a multiple of NS
    const double* __restrict__ wp = m.w4 + ((size_t)(xv / usp) * m.wld + ubeg) * 4;
    const double* __restrict__ cp = m.Cm + (size_t)ubeg * RT;                  // rows of RT doubles
    // The column's weights and sample counts reach the wave through the scalar cache, which has no prefetch: the first touch of
    // a 64-byte line there is a wave-blocking wait all the way to memory (a column's streams are read by this block and the
    // other walker quads' blocks, on other XCDs, and by nobody before them).  One vector load with a lane per line pulls 4 KiB
    // of a stream into this XCD's L2 long before the scalar loads come for it; the loaded words are not used.  (3 % of the
    // kernel's time at 512^2; the rest of its scalar-side stall is the scalar cache's own latency, see DESIGN.md.)
    int pf[3] = {0, 0, 0};
    if (active) {
        const char* wb = reinterpret_cast<const char*>(wp);
        const int wbytes = ucnt * 32, sbytes = m.segld * 4;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int o = lane * 64 + k * 4096;
            if (o < wbytes) pf[k] = *reinterpret_cast<const int*>(wb + o);
        }
        if (lane * 64 < sbytes) pf[2] = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(sc) + lane * 64);
        for (int o = lane * 64 + 2 * 4096; o < wbytes; o += 4096) pf[2] |= *reinterpret_cast<const int*>(wb + o);   // (walks beyond 256 rows)
    }
    double acc[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) acc[j] = 0.0;
    unsigned wo = 0u, co = 0u;                                // byte offsets into the piece's weight and operator streams
    TC q[NS];
    unsigned kb = (unsigned)k0 * kstride;                    // (uniform) byte offset of the next knot to request
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) { q[i] = jx_mx_ldknot<TC>(rs, loff, kb); kb += kstride; }
    for (int s0 = 0; s0 < nseg; s0 += NS) {
        int cnt[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) cnt[j] = sc[s0 + j];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
#ifdef JOXSZ_ABLATIONS
            if (!(m.dbg & 1))                                 // timing experiment (wrong results): no knot requests in the loop
#endif
            q[(j + NS - 1) % NS] = jx_mx_ldknot<TC>(rs, loff, kb);
            kb += kstride;
            for (int i = 0; i < cnt[j]; ++i) {
                // (both scalar streams through one 32-bit byte offset each: s_load base, offset -- no 64-bit pointer arithmetic per sample)
                const double* __restrict__ wq = reinterpret_cast<const double*>(reinterpret_cast<const char*>(wp) + wo);
                const double* __restrict__ cq = reinterpret_cast<const double*>(reinterpret_cast<const char*>(cp) + co);
                const double wa = wq[0], wb = wq[1], wc = wq[2], wd = wq[3];
                double f = wa * (double)q[j].x;
                f = fma(wb, (double)q[(j + 1) % NS].x, f);
                f = fma(wc, (double)q[j].y, f);
                f = fma(wd, (double)q[(j + 1) % NS].y, f);
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[r] = fma(cq[r], f, acc[r]);
#ifdef JOXSZ_ABLATIONS
                if (!(m.dbg & 2))                             // timing experiment (wrong results): the scalar streams stand still
#endif
                { wo += 32u; co += 8u * RT; }
            }
        }
    }
    if (m.n < 0 && (pf[0] | pf[1] | pf[2]) == 0x5a5a1234) Dt[0] = 0.0;   // (never: keeps the prefetch loads alive)
    if (usp > 1) {
        if (xh > 0) {
            double* __restrict__ sp = sm_mix + ((size_t)(gi * (usp - 1) + xh - 1) * RT) * 64 + lane;
#pragma unroll
            for (int r = 0; r < RT; ++r) sp[r * 64] = acc[r];
        }
        __syncthreads();
        if (xh == 0) {
            for (int hh = 1; hh < usp; ++hh) {                  // fixed order: piece 0 + piece 1 (+ piece 2 ...)
                const double* __restrict__ sp = sm_mix + ((size_t)(gi * (usp - 1) + hh - 1) * RT) * 64 + lane;
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[r] += sp[r * 64];
            }
        }
    }
    if (xh == 0 && active) {
        double* __restrict__ dp = Dt + (size_t)xq * m.R * tW + w;
#pragma unroll
        for (int r = 0; r < RT; ++r)
            if (r < m.R) dp[(size_t)r * tW] = acc[r];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Stage 1 in fp32 arithmetic (jx_config.dtype = 2; BASELINE configs[4]'s fp32 variant): the walk of jx_rowmix_kernel with fp32
// knots, weights, operator and sums.  The R multiply-adds of a sample are R / 2 packed ones (v_pk_fma_f32: two rows of the
// operator as one 64-bit scalar operand, the sample in both halves), the hand-over of the column pieces and the rows written
// to Dt are fp32.  What this costs in accuracy is measured, not assumed: test_fp32_variant_tolerance_sweep, bench.py.
// ------------------------------------------------------------------------------------------------------------------
typedef float jx_mx_v2f __attribute__((ext_vector_type(2)));

template <int RT, int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8)))
jx_rowmix_f32_kernel(JxMix m, const float2* __restrict__ cft, float* __restrict__ Dt) {
    static_assert(NS == 8 && RT % 2 == 0, "segment counts come in groups of 8; operator rows in pairs");
    extern __shared__ __attribute__((aligned(16))) float sm_mixf[];            // [gpb][usplit - 1][RT][64] sums of the later pieces
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
    const int usp = m.usplit, gpb = wpb / usp;
    const int xh = wv % usp, gi = wv / usp;
    const int ngrp = (m.n + 63) >> 6, nq = (ngrp + gpb - 1) / gpb;
    const int id = blockIdx.x;
    int gq, xq;
    if (m.cper > 0) { const int xcd = id & 7, jj = id >> 3; gq = xcd % nq; xq = xcd / nq + m.cper * jj; }
    else { gq = id % nq; xq = id / nq; }
    const int grp = gq * gpb + gi;
    const bool active = grp < ngrp && xq < m.NU;
    const int xv = min(xq, m.NU - 1) * usp + xh;
    const int ur = __builtin_amdgcn_readfirstlane(m.urange[xv]), ubeg = ur & 0xffff, ucnt = ur >> 16;
    const size_t w = (size_t)grp * 64 + lane;
    const size_t tW = (size_t)m.tW;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(cft), 0, m.cft_bytes, 0x00020000);
    const unsigned loff = (unsigned)((unsigned)grp * 64u + (unsigned)lane) * (unsigned)sizeof(float2);
    const unsigned kstride = (unsigned)(tW * sizeof(float2));
    const int k0 = __builtin_amdgcn_readfirstlane(m.seg0[xv]), nseg = active ? __builtin_amdgcn_readfirstlane(m.nseg[xv]) : 0;
    const int* __restrict__ sc = m.seg + (size_t)xv * m.segld;
    const float* __restrict__ wp = m.w4f + ((size_t)(xv / usp) * m.wld + ubeg) * 4;
    const float* __restrict__ cp = m.Cmf + (size_t)ubeg * RT;
    int pf[3] = {0, 0, 0};                                                     // (L2 pre-touch of the column's scalar streams, see jx_rowmix_kernel)
    if (active) {
        const char* wb = reinterpret_cast<const char*>(wp);
        const int wbytes = ucnt * 16, sbytes = m.segld * 4;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int o = lane * 64 + k * 4096;
            if (o < wbytes) pf[k] = *reinterpret_cast<const int*>(wb + o);
        }
        if (lane * 64 < sbytes) pf[2] = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(sc) + lane * 64);
    }
    jx_mx_v2f acc[RT / 2];
#pragma unroll
    for (int j = 0; j < RT / 2; ++j) acc[j] = jx_mx_v2f{0.f, 0.f};
    float2 q[NS];
    unsigned kb = (unsigned)k0 * kstride;
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) { q[i] = jx_mx_ldknot<float2>(rs, loff, kb); kb += kstride; }
    for (int s0 = 0; s0 < nseg; s0 += NS) {
        int cnt[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) cnt[j] = sc[s0 + j];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            q[(j + NS - 1) % NS] = jx_mx_ldknot<float2>(rs, loff, kb);
            kb += kstride;
            for (int i = 0; i < cnt[j]; ++i) {
                const float wa = wp[0], wb = wp[1], wc = wp[2], wd = wp[3];
                float f = wa * q[j].x;
                f = fmaf(wb, q[(j + 1) % NS].x, f);
                f = fmaf(wc, q[j].y, f);
                f = fmaf(wd, q[(j + 1) % NS].y, f);
                const jx_mx_v2f ff = jx_mx_v2f{f, f};
#pragma unroll
                for (int r = 0; r < RT / 2; ++r) acc[r] = __builtin_elementwise_fma(jx_mx_v2f{cp[2 * r], cp[2 * r + 1]}, ff, acc[r]);
                wp += 4; cp += RT;
            }
        }
    }
    if (m.n < 0 && (pf[0] | pf[1] | pf[2]) == 0x5a5a1234) Dt[0] = 0.f;   // (never: keeps the prefetch loads alive)
    if (usp > 1) {
        if (xh > 0) {
            float* __restrict__ sp = sm_mixf + ((size_t)(gi * (usp - 1) + xh - 1) * RT) * 64 + lane;
#pragma unroll
            for (int r = 0; r < RT / 2; ++r) { sp[(2 * r) * 64] = acc[r].x; sp[(2 * r + 1) * 64] = acc[r].y; }
        }
        __syncthreads();
        if (xh == 0) {
            for (int hh = 1; hh < usp; ++hh) {                  // fixed order: piece 0 + piece 1 (+ piece 2 ...)
                const float* __restrict__ sp = sm_mixf + ((size_t)(gi * (usp - 1) + hh - 1) * RT) * 64 + lane;
#pragma unroll
                for (int r = 0; r < RT / 2; ++r) { acc[r].x += sp[(2 * r) * 64]; acc[r].y += sp[(2 * r + 1) * 64]; }
            }
        }
    }
    if (xh == 0 && active) {
        float* __restrict__ dp = Dt + (size_t)xq * m.R * tW + w;
#pragma unroll
        for (int r = 0; r < RT / 2; ++r) {
            if (2 * r < m.R) dp[(size_t)(2 * r) * tW] = acc[r].x;
            if (2 * r + 1 < m.R) dp[(size_t)(2 * r + 1) * tW] = acc[r].y;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Stage 1 on the fp64 matrix cores, for R <= 16 rows per column (one 16-row tile).  OPT-IN (JOXSZ_MIX_MFMA=1): measured at 100 us
// per 1024 walkers at 512^2 against the 81 us of jx_rowmix_kernel -- a wave's per-sample loop costs about as many cycles per
// instruction whatever the instruction's kind, and the staging, the group counter and the dispatch weigh as much as the 16
// multiply-adds they replace (DESIGN of round 4, 6.1; profiles/r04_stage1_mfma_*.log).  Same walk as jx_rowmix_kernel: one
// wave = one piece of column x' x 64 walkers, lane = walker, knots in the ring of NS named slots, the four spline weights
// of a sample through the scalar unit, the sample evaluated by its walker's lane (4 FMAs; joxsz_funcs.py:460-462).  What
// changes is the mixing: instead of R scalar-operand FMAs per sample, the samples of four consecutive rows u0 .. u0+3 go
// through four wave-private LDS rows into the B operand of v_mfma_f64_16x16x4 (lane l: row u0 + (l >> 4), walker
// 16 t + (l & 15) of tile t), the A operand is C[u0 + (l >> 4)][l & 15] -- 512 contiguous bytes of the row-major operator,
// which the block holds in LDS (the only vector-memory requests of the walk are then the knots, and their counted waits stay
// exact) -- and D[j][walker] accumulates in the matrix cores' layout (register g of lane l: j = 4 g + (l >> 4), walker
// 16 t + (l & 15)): four matrix instructions per four rows and 64 walkers.  The operands of a group are requested when its
// fourth sample has been written and used one group later (nothing waits on the LDS round trip; LDS serves a wave's requests
// in order, so the next group's samples may overwrite the rows at once).  The groups of four are a function of the piece
// alone, so a walker's sums do not depend on the launch.
//   sm: [crows][16] operator C | [gpb][JX_MXM_REGION(usplit)] per walker group: the sample rows of its waves (wave xh at
//       xh * JX_MXM_RING), later the sums of the later pieces [usplit - 1][16][64]
// ------------------------------------------------------------------------------------------------------------------
#define JX_MXM_LD 80
#define JX_MXM_RING (4 * JX_MXM_LD)
#define JX_MXM_REGION(usp) ((usp) > 1 ? (((usp) - 1) * 1024 > (usp) * JX_MXM_RING ? ((usp) - 1) * 1024 : (usp) * JX_MXM_RING) : JX_MXM_RING)

template <int NS, typename TC>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
jx_rowmix_mfma_kernel(JxMix m, int crows, const TC* __restrict__ cft, double* __restrict__ Dt) {
    static_assert(NS == 8, "one aligned 8-dword scalar load carries the sample counts of a group of NS segments");
    extern __shared__ __attribute__((aligned(16))) double sm_mix[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
    const int usp = m.usplit, gpb = wpb / usp;
    const int xh = wv % usp, gi = wv / usp;
    const int ngrp = (m.n + 63) >> 6, nq = (ngrp + gpb - 1) / gpb;
    const int id = blockIdx.x;
    int gq, xq;
    if (m.cper > 0) { const int xcd = id & 7, jj = id >> 3; gq = xcd % nq; xq = xcd / nq + m.cper * jj; }
    else { gq = id % nq; xq = id / nq; }
    const int grp = gq * gpb + gi;
    const bool active = grp < ngrp && xq < m.NU;               // (no early exit: every wave of the block meets the barriers below)
    const int xc = min(xq, m.NU - 1), xv = xc * usp + xh;      // column, piece
    const int ur = __builtin_amdgcn_readfirstlane(m.urange[xv]), ubeg = ur & 0xffff;
    const double* __restrict__ wp = m.w4 + ((size_t)xc * m.wld + ubeg) * 4;
    const size_t tW = (size_t)m.tW;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<TC*>(cft), 0, m.cft_bytes, 0x00020000);
    const unsigned loff = (unsigned)((unsigned)grp * 64u + (unsigned)lane) * (unsigned)sizeof(TC);
    const unsigned kstride = (unsigned)(tW * sizeof(TC));
    const int k0 = __builtin_amdgcn_readfirstlane(m.seg0[xv]), nseg = active ? __builtin_amdgcn_readfirstlane(m.nseg[xv]) : 0;
    const int* __restrict__ sc = m.seg + (size_t)xv * m.segld;                 // [segld], zero padded: whole groups of NS, one group ahead
    TC q[NS];
    unsigned kb = (unsigned)k0 * kstride;
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) { q[i] = jx_mx_ldknot<TC>(rs, loff, kb); kb += kstride; }
    int pf[3] = {0, 0, 0};                                                     // (L2 pre-touch of the column's scalar streams, see jx_rowmix_kernel)
    if (active) {
        const char* wb = reinterpret_cast<const char*>(wp);
        const int wbytes = (ur >> 16) * 32, sbytes = m.segld * 4;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int o = lane * 64 + k * 4096;
            if (o < wbytes) pf[k] = *reinterpret_cast<const int*>(wb + o);
        }
        if (lane * 64 < sbytes) pf[2] = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(sc) + lane * 64);
        for (int o = lane * 64 + 2 * 4096; o < wbytes; o += 4096) pf[2] |= *reinterpret_cast<const int*>(wb + o);
    }
    {   // the operator into LDS, once per block
        const double2* __restrict__ src = reinterpret_cast<const double2*>(m.Cm);
        double2* dst = reinterpret_cast<double2*>(sm_mix);
        for (int i = threadIdx.x; i < crows * 8; i += blockDim.x) dst[i] = src[i];
    }
    double* __restrict__ region = sm_mix + (size_t)crows * 16 + (size_t)gi * JX_MXM_REGION(usp);
    double* __restrict__ ring = region + xh * JX_MXM_RING;
    const double* __restrict__ bp = ring + (lane >> 4) * JX_MXM_LD + (lane & 15);   // B operand of tile t: bp[16 t]
    const double* __restrict__ ap = sm_mix + (size_t)ubeg * 16 + lane;             // A operand of the group that starts at row ubeg + c: ap[16 c]
    __syncthreads();
    jx_mx_v4d acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = jx_mx_v4d{0.0, 0.0, 0.0, 0.0};
    double bq[4] = {0.0, 0.0, 0.0, 0.0}, av = 0.0;                             // operands of the group that waits for its matrix instructions
    int c = 0;                                                                 // samples written so far
    auto boundary = [&]() {                                                    // c is a multiple of 4: rows c-4 .. c-1 are complete in the ring
#ifdef JOXSZ_ABLATIONS
        if (!(m.dbg & 4))
#endif
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bq[t], acc[t], 0, 0, 0);
#ifdef JOXSZ_ABLATIONS
        if (!(m.dbg & 16))
#endif
        {
#pragma unroll
        for (int t = 0; t < 4; ++t) bq[t] = bp[16 * t];
        av = ap[(c - 4) * 16];
        }
    };
    int cntn[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) cntn[j] = sc[j];
    for (int s0 = 0; s0 < nseg; s0 += NS) {
        int cnt[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) { cnt[j] = cntn[j]; cntn[j] = sc[s0 + NS + j]; }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
#ifdef JOXSZ_ABLATIONS
            if (!(m.dbg & 1))
#endif
            q[(j + NS - 1) % NS] = jx_mx_ldknot<TC>(rs, loff, kb);
            kb += kstride;
            for (int i = 0; i < cnt[j]; ++i) {
                const double wa = wp[0], wb = wp[1], wc = wp[2], wd = wp[3];
                double f = wa * (double)q[j].x;
                f = fma(wb, (double)q[(j + 1) % NS].x, f);
                f = fma(wc, (double)q[j].y, f);
                f = fma(wd, (double)q[(j + 1) % NS].y, f);
#ifdef JOXSZ_ABLATIONS
                if (!(m.dbg & 8))
#endif
                ring[(c & 3) * JX_MXM_LD + lane] = f;
#ifdef JOXSZ_ABLATIONS
                if (!(m.dbg & 2))
#endif
                wp += 4;
                ++c;
                if ((c & 3) == 0) boundary();
            }
        }
    }
    if (m.n < 0 && (pf[0] | pf[1] | pf[2]) == 0x5a5a1234) Dt[0] = 0.0;   // (never: keeps the prefetch loads alive)
    // the last, partial group: rows beyond the piece enter as zero samples (their operator rows are finite)
    if (c & 3) {
        while (c & 3) { ring[(c & 3) * JX_MXM_LD + lane] = 0.0; ++c; }
        boundary();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bq[t], acc[t], 0, 0, 0);
    if (usp > 1) {
        __syncthreads();                                        // (the sums go where the sample rows were)
        if (xh > 0) {
            double* __restrict__ sp = region + (size_t)(xh - 1) * 1024 + lane;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) sp[(t * 4 + g) * 64] = acc[t][g];
        }
        __syncthreads();
        if (xh == 0) {
            for (int hh = 1; hh < usp; ++hh) {                  // fixed order: piece 0 + piece 1 (+ piece 2 ...)
                const double* __restrict__ sp = region + (size_t)(hh - 1) * 1024 + lane;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[t][g] += sp[(t * 4 + g) * 64];
            }
        }
    }
    if (xh == 0 && active) {
        const int lk = lane >> 4, li = lane & 15;
        double* __restrict__ dp = Dt + (size_t)xq * m.R * tW + (size_t)grp * 64 + li;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j = 4 * g + lk;
            if (j < m.R) {
#pragma unroll
                for (int t = 0; t < 4; ++t) dp[(size_t)j * tW + 16 * t] = acc[t][g];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Stage 2 and the full form: P[ks][w][x] = sum_{kappa in slice ks} a_kappa(w) Op[x][kappa] on v_mfma_f64_16x16x4.
//   A operand (M = walkers): lane l holds a_kappa(w) for w = tile + (l & 15), kappa = 4 step + (l >> 4)
//       LOAD: a = Dt[kappa][w]                         (stage-1 output, 128-byte runs)
//       EVAL: a = the map sample kappa of walker w, evaluated here from cft (entry table in LDS)
//   B operand (N = outputs): Op[x][kappa], stored so that a lane's NXT tiles are contiguous: Opk[step][l >> 4][l & 15][tile]
//   D: register g of lane l = out[walker (l >> 4) + 4 g][x = l & 15]: a walker's row comes out in 128-byte runs
// Block = 4 waves = 128 walkers (2 tiles per wave) x NXT output tiles x one K slice; operands are fetched RD k-steps ahead
// into named register slots.  Blocks that share a K slice share an XCD (the slice of Dt and of Op reach one L2 only).
// ------------------------------------------------------------------------------------------------------------------
struct JxSamp { long long off; double a, b, c, d; long long pad; };      // off = knot * tW (elements of cft)

struct JxOpg {
    int n; long long tW;
    int ksplit, kper;              // K slices, k-steps per slice (multiple of the prefetch depth)
    int kmajor;                    // block -> unit mapping: 1 = XCD x works on the K slices x, x + 8, ... (8 or more slices), 0 = units dealt round
    int ntile, nog;                // output tiles in all (multiple of NXT), output groups = ntile / NXT
    int ldx;                       // doubles per partial row (>= 16 ntile)
    long long pstride;             // doubles between the partial rows of two K slices (>= tW ldx)
    const double* Op;              // [ksplit * kper + slack][4][16][ntile]
    const double* Dt;              // LOAD: [4 (ksplit kper + slack)][tW]
    const JxSamp* ent;             // EVAL: [4 (ksplit kper + slack)]
    const float* Opf;              // fp32 arithmetic (dtype 2): the operator and the stage-1 rows in fp32
    const float* Dtf;
};

#ifndef JX_OPG_RD
#define JX_OPG_RD 2
#endif
#define JX_OPG_ECH 256             // EVAL: k-steps of entries staged in LDS at a time (256 x 4 x 48 B = 48 KB)

template <int MODE /*0 LOAD, 1 EVAL*/, int NXT, typename TC>
__global__ void __launch_bounds__(256)
jx_opgemm_kernel(JxOpg g, const TC* __restrict__ cft, double* __restrict__ Pt) {
    extern __shared__ __attribute__((aligned(16))) double sm_opg[];
    constexpr int RD = JX_OPG_RD;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwb = (g.n + 127) >> 7, nunit = g.ksplit * g.nog;
    const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
    const int wb = jj % nwb;
    int ks, og;
    if (g.kmajor) {                                  // an XCD owns whole K slices: the slice of Dt is fetched into one L2 only
        const int lu = jj / nwb;
        og = lu % g.nog; ks = (lu / g.nog) * 8 + xcd;
        if (ks >= g.ksplit) return;
    } else {
        const int unit = (jj / nwb) * 8 + xcd;
        if (unit >= nunit) return;
        ks = unit / g.nog; og = unit - ks * g.nog;
    }
    const size_t tW = (size_t)g.tW;
    const size_t wbase = (size_t)wb * 128 + wv * 32 + li;
    const int s0 = ks * g.kper, s1 = s0 + g.kper;
    jx_mx_v4d acc[2][NXT];
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int t = 0; t < NXT; ++t) acc[mm][t] = jx_mx_v4d{0.0, 0.0, 0.0, 0.0};
    const double* __restrict__ opb = g.Op + ((size_t)lk * 16 + li) * g.ntile + (size_t)og * NXT;
    const size_t opstep = (size_t)64 * g.ntile;
    double b[RD][NXT], a[RD][2];
    TC c0[RD][2], c1[RD][2];
    double ew[RD][4];
    (void)c0; (void)c1; (void)ew;
    const JxSamp* esm = reinterpret_cast<const JxSamp*>(sm_opg);
    auto fetch = [&](int slot, int s) {
#pragma unroll
        for (int t = 0; t < NXT; ++t) b[slot][t] = opb[(size_t)s * opstep + t];
        if (MODE == 0) {
            const double* dp = g.Dt + (size_t)(4 * s + lk) * tW + wbase;
            a[slot][0] = dp[0]; a[slot][1] = dp[16];
        } else {
            const JxSamp& e = esm[4 * ((s - s0) % JX_OPG_ECH) + lk];
            const TC* cp = cft + (size_t)e.off + wbase;
            c0[slot][0] = cp[0]; c0[slot][1] = cp[16];
            c1[slot][0] = cp[tW]; c1[slot][1] = cp[tW + 16];
            ew[slot][0] = e.a; ew[slot][1] = e.b; ew[slot][2] = e.c; ew[slot][3] = e.d;
        }
    };
    auto stage_entries = [&](int sa) {                           // k-steps [sa, sa + ECH) of this slice into LDS
        __syncthreads();
        const double* src = reinterpret_cast<const double*>(g.ent + (size_t)4 * sa);
        for (int i = tid; i < JX_OPG_ECH * 4 * 6; i += 256) sm_opg[i] = src[i];
        __syncthreads();
    };
    // (EVAL: the prefetch never crosses a staging boundary: the slots are refilled after each re-staging)
    for (int sa = s0; sa < s1; sa += JX_OPG_ECH) {
        const int sb = (MODE == 1) ? min(s1, sa + JX_OPG_ECH) : s1;
        if (MODE == 1) stage_entries(sa);
#pragma unroll
        for (int u = 0; u < RD - 1; ++u) fetch(u, min(sa + u, sb - 1));
        for (int s = sa; s < sb; s += RD) {
#pragma unroll
            for (int u = 0; u < RD; ++u) {
                fetch((u + RD - 1) % RD, min(s + u + RD - 1, sb - 1));
                double av0, av1;
                if (MODE == 0) { av0 = a[u][0]; av1 = a[u][1]; }
                else {
                    av0 = ew[u][0] * (double)c0[u][0].x; av1 = ew[u][0] * (double)c0[u][1].x;
                    av0 = fma(ew[u][1], (double)c1[u][0].x, av0); av1 = fma(ew[u][1], (double)c1[u][1].x, av1);
                    av0 = fma(ew[u][2], (double)c0[u][0].y, av0); av1 = fma(ew[u][2], (double)c0[u][1].y, av1);
                    av0 = fma(ew[u][3], (double)c1[u][0].y, av0); av1 = fma(ew[u][3], (double)c1[u][1].y, av1);
                }
#pragma unroll
                for (int t = 0; t < NXT; ++t) {
                    acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av0, b[u][t], acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av1, b[u][t], acc[1][t], 0, 0, 0);
                }
            }
        }
        if (MODE == 0) break;
    }
    const size_t wrow = (size_t)wb * 128 + wv * 32 + lk;
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const size_t w = wrow + mm * 16 + 4 * gq;
            if (w < (size_t)g.n) {
                double* row = Pt + (size_t)ks * g.pstride + w * g.ldx + (size_t)(og * NXT) * 16 + li;
#pragma unroll
                for (int t = 0; t < NXT; ++t) row[t * 16] = acc[mm][t][gq];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// Stage 2 in fp32 arithmetic (dtype 2, low-rank form): the product of jx_opgemm_kernel<LOAD> on v_mfma_f32_16x16x4_f32, stage-1
// rows, operator and partial rows in fp32 (the tail adds the K slices in fp64).  Same blocks, K slices and XCD ownership.
//   D of the fp32 instruction: register g of lane l = out[walker 4 (l >> 4) + g][x = l & 15]
// ------------------------------------------------------------------------------------------------------------------
typedef float jx_mx_v4f __attribute__((ext_vector_type(4)));

template <int NXT>
__global__ void __launch_bounds__(256)
jx_opgemm_f32_kernel(JxOpg g, float* __restrict__ Pt) {
    constexpr int RD = JX_OPG_RD;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwb = (g.n + 127) >> 7, nunit = g.ksplit * g.nog;
    const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
    const int wb = jj % nwb;
    int ks, og;
    if (g.kmajor) {
        const int lu = jj / nwb;
        og = lu % g.nog; ks = (lu / g.nog) * 8 + xcd;
        if (ks >= g.ksplit) return;
    } else {
        const int unit = (jj / nwb) * 8 + xcd;
        if (unit >= nunit) return;
        ks = unit / g.nog; og = unit - ks * g.nog;
    }
    const size_t tW = (size_t)g.tW;
    const size_t wbase = (size_t)wb * 128 + wv * 32 + li;
    const int s0 = ks * g.kper, s1 = s0 + g.kper;
    jx_mx_v4f acc[2][NXT];
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int t = 0; t < NXT; ++t) acc[mm][t] = jx_mx_v4f{0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ opb = g.Opf + ((size_t)lk * 16 + li) * g.ntile + (size_t)og * NXT;
    const size_t opstep = (size_t)64 * g.ntile;
    float b[RD][NXT], a[RD][2];
    auto fetch = [&](int slot, int s) {
#pragma unroll
        for (int t = 0; t < NXT; ++t) b[slot][t] = opb[(size_t)s * opstep + t];
        const float* dp = g.Dtf + (size_t)(4 * s + lk) * tW + wbase;
        a[slot][0] = dp[0]; a[slot][1] = dp[16];
    };
#pragma unroll
    for (int u = 0; u < RD - 1; ++u) fetch(u, min(s0 + u, s1 - 1));
    for (int s = s0; s < s1; s += RD) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
            fetch((u + RD - 1) % RD, min(s + u + RD - 1, s1 - 1));
#pragma unroll
            for (int t = 0; t < NXT; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0], b[u][t], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1], b[u][t], acc[1][t], 0, 0, 0);
            }
        }
    }
    const size_t wrow = (size_t)wb * 128 + wv * 32 + 4 * lk;
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const size_t w = wrow + mm * 16 + gq;
            if (w < (size_t)g.n) {
                float* row = Pt + (size_t)ks * g.pstride + w * g.ldx + (size_t)(og * NXT) * 16 + li;
#pragma unroll
                for (int t = 0; t < NXT; ++t) row[t * 16] = acc[mm][t][gq];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// Tail: the extracted row arrives as nks partial rows P[ks][w][ldx] (one per K slice); they are added in a fixed order
// (a result does not depend on the launch it was part of), then conversion, data radii, chi^2 and total as in
// jx_tail_kernel (joxsz_funcs.py:472-479, 538).  One block per walker.
// ------------------------------------------------------------------------------------------------------------------
template <typename TP /* partial rows: double, or float behind the fp32 product */>
__global__ void __launch_bounds__(JX_TAIL_THREADS)
jx_tail_row_kernel(JxDev c, const TP* __restrict__ Pt, int nks, long long pstride /*elements between partials*/, int ldx,
                   int nuse /* outputs present in the partial rows: the ones the data-radii matrix reads (<= nrow; the taps get nrow) */,
                   const double* __restrict__ cfac, const double* __restrict__ sz0,
                   const double* __restrict__ base, double* __restrict__ logp, int w0,
                   double* __restrict__ tap_row, double* __restrict__ tap_bright, double* __restrict__ tap_chisq,
                   double* __restrict__ tap_parts, JxSm smv) {
    JX_LDS_DECL;
    double* red = sm + 20;
    const int nrow = c.nrow;
    double* s_prof = sm + JX_LDS_HDR;  // [nrow]
    const int w = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const TP* Pw = Pt + (size_t)w * ldx;
    for (int k = tid; k < nuse; k += nth) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int ks = 0;
        // (sixteen slices requested together -- one trip to memory instead of four -- and added in the order of the loop below)
        for (; ks + 15 < nks; ks += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (double)Pw[(size_t)(ks + u) * pstride + k];
#pragma unroll
            for (int u = 0; u < 16; u += 4) { a0 += v[u]; a1 += v[u + 1]; a2 += v[u + 2]; a3 += v[u + 3]; }
        }
        for (; ks + 3 < nks; ks += 4) {
            a0 += (double)Pw[(size_t)ks * pstride + k]; a1 += (double)Pw[(size_t)(ks + 1) * pstride + k];
            a2 += (double)Pw[(size_t)(ks + 2) * pstride + k]; a3 += (double)Pw[(size_t)(ks + 3) * pstride + k];
        }
        for (; ks < nks; ++ks) a0 += (double)Pw[(size_t)ks * pstride + k];
        const double acc = (a0 + a1) + (a2 + a3);
        if (tap_row) tap_row[(size_t)w * nrow + k] = acc;
        const double b = acc * cfac[(size_t)w * nrow + k];
        s_prof[k] = b;
        if (tap_bright) tap_bright[(size_t)w * nrow + k] = b;
    }
    __syncthreads();
    double part = 0.0;
    for (int dd = tid >> 3; dd < c.nflux; dd += nth >> 3) {         // eight lanes per flux point
        const double* e = c.emat + (size_t)dd * nrow;
        double m = 0.0;
        // (eight coefficients requested together, then their eight multiply-adds in the order of a plain loop: the loop is a
        //  chain of dependent memory round trips otherwise -- 7 of the kernel's 13 us)
        for (int k0 = tid & 7; k0 < nuse; k0 += 64) {
            double ev[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ev[u] = e[min(k0 + 8 * u, nuse - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (k0 + 8 * u < nuse) m = fma(ev[u], s_prof[k0 + 8 * u], m);
        }
        m += __shfl_xor(m, 1, 64); m += __shfl_xor(m, 2, 64); m += __shfl_xor(m, 4, 64);
        const double z = (c.flux[c.nflux + dd] - m) / c.flux[2 * c.nflux + dd];
        const double z2 = z * z;
        if ((tid & 7) == 0 && z2 == z2) part += z2;                  // np.nansum drops NaN terms
    }
    const double chisq = jx_block_sum(part, red);
    if (tid == 0) {
        const double ll = -chisq / 2.0 + (sz0 ? sz0[w] : 0.0);
        double b = base[w];
        // (two-block form of the per-walker kernel: the priors arrive here, the Cash log-likelihood and its verdict beside them)
        if (c.xr_split && b != -INFINITY) b = (c.xr_out[2 * (size_t)w + 1] != 0.0) ? -INFINITY : b + c.xr_out[2 * (size_t)w];
        double tot = (b == -INFINITY) ? -INFINITY : b + ll;
        if (tot != tot) tot = -INFINITY;             // never hand NaN to the sampler
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
