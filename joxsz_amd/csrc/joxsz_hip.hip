// C-ABI implementation (include/joxsz_hip.h) for gfx950: context, uploads, table
// building, rocFFT plans and the per-chunk launch sequence.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/joxsz_hip.h"
#include "jx_kernels.hpp"
#include "jx_conv.hpp"
#include "jx_dct.hpp"
#include "jx_mix.hpp"
#include "jx_tables.hpp"

namespace {

struct Plan3 {
    rocfft_plan beam_fwd = nullptr, beam_inv = nullptr, tf_fwd = nullptr;
    size_t work_bytes = 0;
};

struct EvSet {
    hipEvent_t e[7];
    int walkers;
    bool op;                           // operator route: only e[0], e[1], e[5] were recorded
    bool gemm;                         // e[6] (behind the FIR + combination GEMM of the fused route) was recorded
    bool p1only = false;               // timing mode 2: only e[2], e[3] (around pass 1) were recorded
};

}  // namespace

struct jx_ctx {
    jx_config cfg;
    bool finalized = false;
    std::string err;
    std::string devname;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;  // created by jx_create; `stream` may point at a caller's stream instead (jx_set_stream)

    // host copies of the uploaded tensors
    std::vector<std::vector<unsigned char>> host;
    std::vector<bool> have;

    // derived sizes
    int P = 0, Ph = 0, Sh = 0, nrow = 0, nt = 0, K = 0, chunk = 0, map_split = 1, map_threads = 256;
    size_t map_lds_bytes = 0;
    int64_t device_bytes = 0;

    // device constants
    std::vector<void*> dev_allocs;
    JxDev d;
    double* d_par_vals = nullptr;

    // work buffers (chunk capacity)
    double *d_base = nullptr, *d_cfac = nullptr;
    double *d_sz0 = nullptr, *d_sz0_op = nullptr, *t_integ = nullptr;   // integrated-Compton term per walker (calc_integ), its tap
    double *d_img = nullptr, *d_conv = nullptr;
    double2 *d_spec = nullptr, *d_tfspec = nullptr;
    // hand-written convolution (conv_mode 2)
    int conv_mode = 1;
    JxConv cv;
    JxConv cv_lr;                      // pass 3 over the r combined rows of the low-rank form (NJ = CROWS = r)
    JxLowrank lr;                      // lr.r == 0: every job goes through pass 3
    double *d_Clr = nullptr, *d_col0lr = nullptr;
    const cplx* lr_vt = nullptr;       // [r][Sh] right singular vectors as (v, 0)
    // fused FIR + job combination (one GEMM per column, walker-minor row spectra): tables, buffers, launch copy of JxConv
    JxLowrank lrf, lrf0;               // lrf.r == 0: not available.  A = Wk [Ph][RP][KU] / V0 [o+1][RP][KU]
    JxConv cv_f;
    int fused_bucket = 0, tW = 0, tKU = 0, kact = 0;
    double lr_tol = 1e-10;             // singular-value cut in use
    double lr_tol_override = 0.0;      // > 0: the cut to use (second finalize pass after the truncation probe asked for a tighter one)
    double trunc_est = -1.0;           // probe: largest difference between the extracted row on the truncated default route and on the
                                       //   route with every job and every column, relative to the row's largest entry (-1: not measured)
    int trunc_retried = 0;
    double *d_Rt = nullptr, *d_Ct = nullptr, *d_Ct0 = nullptr, *d_x0t = nullptr;
    std::vector<double> h_L, h_taps;   // finalize scratch: U [r][NJ], FIR taps [o+1][Ph]
    jxt::ConvRows h_rows;
    cplx *d_Y = nullptr, *d_C = nullptr, *d_part = nullptr;
    size_t p1_lds = 0, p2_lds = 0, p3_lds = 0;
    int p13_rows = 8, p1_rows = 8;
    int lr_bucket = 0;                // k-steps compiled into the low-rank kernel in use
    bool lr_sep = false;              // the separate combination kernel is available (else: fused route or one row per job)
    int fused_nh = 1;                 // K halves of the fused GEMM (two launches, the second accumulating, when NU/4 exceeds the buckets)
    int last_nblk3 = 0;               // pass-3 blocks per walker of the launch sequence just queued
    // pass 1 from the spline coefficients (jx_rowdct_kernel): no Compton-y map in HBM on the default route
    bool dct_ok = false;
    JxDct dct{};
    double* d_cf = nullptr;           // [chunk][cf_ws] spline ordinates and moments, Abel kernel -> jx_rowdct_kernel
    // the Abel kernel's phases 2-3 as one matrix product (jx_abel_gemm_kernel): operator, its geometry, the launch's pressure profiles
    std::vector<double> h_Tm;         // host copy until the route that uses it is known
    double* d_Tm = nullptr; int tm_ld = 0, tm_ntile = 0, tm_npair = 0;
    double* d_ppc = nullptr;          // [chunk][N] prep kernel -> jx_abel_gemm_kernel
    double* d_cf_tap = nullptr;       // fp32 contexts: fp64 spline arrays of the Abel kernel when it runs for the profile taps
    bool abel_gemm = false;
    size_t dct_lds = 0;
    void* samp_buf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // jx_sample work buffers (grow-only)
    size_t samp_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int dct_nw = 16;                  // walkers per block of the coefficient-fed pass 1 (8 for the longest rows: two blocks then share a CU's LDS)
    // odd map sides (the reference's own shapes): the transfer-function step in real space, no transform of length S
    bool f32 = false;                  // jx_config.dtype == 1: fp32 storage between the kernels, fp32 evaluation and pass-1 transform
    bool odd = false;
    JxDct dct3{};                      // combined rows back to real space (jx_rowdct_kernel, MODE 1)
    size_t dct3_lds = 0;
    JxLowrank lr2{};                   // second matrix product: real-space circular kernels x combined rows
    int o_nmg = 0, o_bucket2 = 0, o_RPc = 0, o_ldb = 0, o_nout = 0, o_nh2 = 1;
    double *d_Ctp = nullptr, *d_cc = nullptr, *d_D2 = nullptr;
    const double* d_Kp = nullptr;
    // beam-convolved-map tap on the odd-side route (built on first use): FIR-only operator, one row per job
    const double* d_Wfir = nullptr;
    double *d_Ctj = nullptr, *d_ccj = nullptr;
    int o_RPj = 0;
    int num_cu = 256;
    int* d_rowjob = nullptr;
    int* d_runs = nullptr;
    int nrun = 0, fir_reg = 0;
    double* t_convjobs = nullptr;
    double* t_conv = nullptr;
    double* t_y2d = nullptr;          // full Compton-y maps mirrored out of the quadrant (Y2D stage tap only)
    void* d_work = nullptr;
    size_t work_cap = 0;
    // batch staging for the host-pointer API
    double *d_theta = nullptr, *d_logp = nullptr;
    int batch_cap = 0;
    // taps
    double *t_pp = nullptr, *t_ab = nullptr, *t_y = nullptr, *t_row = nullptr, *t_bright = nullptr,
           *t_chisq = nullptr, *t_tprof = nullptr, *t_xprofs = nullptr, *t_parts = nullptr;

    // collapsed route (jx_set_route): Gt [N][g_ld], row j = map row of the unit pressure profile e_j
    int route = JX_ROUTE_MAP;
    double* d_G = nullptr;
    double* d_pp = nullptr;            // [op_cap][N] pressure profiles, prep kernel -> operator kernel
    double *d_base_op = nullptr, *d_cfac_op = nullptr;   // [op_cap], [op_cap][nrow]
    double* d_rows = nullptr;          // [op_cap / 32][nrow][32] G pp of large launches (jx_operator_mfma_kernel)
    int op_cap = 0;                    // walkers per launch on the operator route
    int g_ld = 0;

    // contracted route (conv_mode 3, jx_mix.hpp): sum over map rows before any transform
    JxMix mx{};                        // stage 1 tables
    JxOpg og{};                        // stage 2 (low-rank form) / full form
    int mix_form = 0;                  // 0 low-rank (stage 1 + stage 2), 1 full operator on the samples
    int mix_RT = 0, mix_nxt = 0;       // template instances in use
    int mix_r = 0, mix_ns = 0, mix_rank_full = 0, mix_ksteps = 0;
    long long mix_tW = 0;
    int mix_ncol = 0;                  // columns of the walker-minor spline array stored by the matrix product: 2 (N + pad)
    double *d_cft = nullptr, *d_Dt = nullptr, *d_Pt = nullptr;
    double mix_beam_tol = 0.0;

    ncclComm_t comm = nullptr;        // RCCL communicator of this rank (jx_comm_init_rank)
    int comm_rank = 0, comm_size = 1;

    std::map<int, Plan3> plans;
    rocfft_execution_info info = nullptr;

    // timing
    bool timing_on = false;
    int timing_mode = 0;              // jx_timing_enable: 1 = every stage, 2 = only the events around pass 1 of the hand-written route
    std::vector<EvSet> ev_inflight, ev_free;
    jx_timing acc{};
};

static int g_rocfft_refs = 0;

// beam half-widths (B-1)/2 for which the register-window FIR is instantiated

// two-level (register-blocked) forms: (LP, LS, rows per block in pass 1, rows per block in pass 3)
#define JX_CONV2_PAIRS(X) X(18, 16, 42, 42) X(48, 24, 32, 32) X(48, 32, 32, 32) X(96, 64, 21, 21) X(144, 128, 21, 16) X(288, 256, 14, 14) X(576, 512, 10, 8)

// coefficient-fed pass 1 (jx_rowdct_kernel): (LP = padded length / 2, NS =
// the generic instances for odd map sides: any sample count up to LP - 4 (LP, threads per block)
#define JX_DCT_ODD_SIZES(X) X(48, 192) X(96, 192) X(144, 256) X(288, 256) X(576, 384)
// samples per distinct row, threads per block: 16 walkers x max(L1, L2) FFT tasks + one wave without one)
#define JX_DCT_SIZES(X) X(48, 24, 192, 16) X(48, 32, 192, 16) X(96, 64, 192, 16) X(144, 128, 256, 16) X(288, 256, 256, 16) X(576, 512, 256, 8)
// LDS bytes of an instance: [NW][RS] complex rows + twiddles [LP/2] + split constants + per-lane odd sums + B[0]  (esz = sizeof(T))
template <int LP, int NS> static size_t dct_lds_bytes(int nw, size_t esz) {
    return 2 * esz * ((size_t)nw * jx_dct_lay<LP / 2, NS>::RS + LP / 2) + esz * (4 * (LP / 4 + 1) + (size_t)nw * 64 + nw);
}

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return (e_ == hipErrorOutOfMemory) ? JX_ERR_NOMEM : JX_ERR_HIP;                        \
        }                                                                                          \
    } while (0)

#define FFTCHK(ctx, call)                                                                          \
    do {                                                                                           \
        rocfft_status s_ = (call);                                                                 \
        if (s_ != rocfft_status_success) {                                                         \
            (ctx)->err = std::string(#call) + ": rocfft status " + std::to_string((int)s_);        \
            return JX_ERR_ROCFFT;                                                                  \
        }                                                                                          \
    } while (0)

static size_t tensor_bytes(const jx_config& c, int id) {
    const size_t f = sizeof(double);
    switch (id) {
        case JX_T_R_PP: return f * c.N;
        case JX_T_D_MAT: return f * (size_t)c.S * c.S;
        case JX_T_BEAM_2D: return f * (size_t)c.B * c.B;
        case JX_T_FILTERING: return f * (size_t)c.S * c.S;
        case JX_T_RADIUS: return f * c.S;
        case JX_T_FLUX_DATA: return f * 3 * c.nflux;
        case JX_T_CONV_T: case JX_T_CONV_V: return f * c.nconv;
        case JX_T_PAR_VALS: case JX_T_PAR_MIN: case JX_T_PAR_MAX: case JX_T_PAR_MU: case JX_T_PAR_SIGMA:
            return f * c.npar;
        case JX_T_PAR_KIND: return sizeof(int32_t) * c.npar;
        case JX_T_THAWED_IDX: return sizeof(int32_t) * c.ndim;
        case JX_T_X_R_NE: case JX_T_X_R_T: case JX_T_GEOMAREA: return f * c.nann;
        case JX_T_PROJVOLS: return f * (size_t)c.nann * c.nann;
        case JX_T_CTS: case JX_T_AREASCALES: case JX_T_EXPOSURES: case JX_T_BACKRATES:
            return f * (size_t)c.nband * c.nann;
        case JX_T_LNT: return f * c.ntab;
        case JX_T_LNRATE: return f * (size_t)c.nband * 2 * c.ntab;
        case JX_T_INTEG_W: return f * ((size_t)c.N + 1);
    }
    return 0;
}

static bool tensor_is_xray(int id) { return id >= JX_T_X_R_NE && id <= JX_T_LNRATE; }

template <typename T>
static int dev_put(jx_ctx* ctx, const T* src, size_t count, T** out) {
    void* p = nullptr;
    HIPCHK(ctx, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    ctx->dev_allocs.push_back(p);
    ctx->device_bytes += (int64_t)(count * sizeof(T));
    if (count) HIPCHK(ctx, hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *out = (T*)p;
    return JX_OK;
}

template <typename T>
static int dev_new(jx_ctx* ctx, size_t count, T** out, bool zero = false) {
    void* p = nullptr;
    HIPCHK(ctx, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    ctx->dev_allocs.push_back(p);
    ctx->device_bytes += (int64_t)(count * sizeof(T));
    if (zero) HIPCHK(ctx, hipMemset(p, 0, count * sizeof(T)));
    *out = (T*)p;
    return JX_OK;
}

template <typename T>
static std::vector<T> host_vec(jx_ctx* ctx, int id) {
    const auto& b = ctx->host[id];
    std::vector<T> v(b.size() / sizeof(T));
    if (!v.empty()) memcpy(v.data(), b.data(), b.size());
    return v;
}

// What the hand-written route needs for an odd map side, worked out before the route is chosen (so that `auto` can fall
// back to the rocFFT sequence): row bookkeeping, beam taps, low-rank form of the transfer-function weights, band limit.
struct OddPlan {
    bool ok = false;
    const char* why = "";
    int LP = 0, r = 0, kact = 0;
    jxt::ConvRows rows;
    std::vector<double> taps, L, V;
};

static bool dmat_is_mirror(const std::vector<double>& dm, int S) {
    const int c = S / 2;
    for (int iy = 0; iy < S; ++iy)
        for (int ix = 0; ix < S; ++ix) {
            const int b = std::abs(iy - c), a = std::abs(ix - c);
            const int jy = (c + b < S) ? c + b : c - b, jx = (c + a < S) ? c + a : c - a;
            if (memcmp(&dm[(size_t)iy * S + ix], &dm[(size_t)jy * S + jx], sizeof(double)) != 0) return false;
        }
    return true;
}

static void plan_odd(const jx_config& c, const std::vector<double>& beam, const std::vector<double>& filt,
                     const std::vector<double>& dm, double tol, OddPlan& pl) {
    const int S = c.S, B = c.B, o = (B - 1) / 2, Sh = S / 2 + 1;
    pl.LP = jxt::custom_conv_lp_odd(S, o);
    if (!pl.LP) { pl.why = "no padded length for this odd side"; return; }
    if (!jxt::beam_is_symmetric(beam, B)) { pl.why = "beam image not flip-symmetric"; return; }
    if (!dmat_is_mirror(dm, S)) { pl.why = "d_mat lacks the mirror structure of centdistmat"; return; }
    const int P = 2 * pl.LP, Ph = pl.LP + 1;
    jxt::conv_row_tables(S, o, true, pl.rows);
    jxt::beam_fir_taps(beam, B, P, c.step * c.step / (double)P, pl.taps);
    std::vector<double> hy;
    jxt::tf_hy_table(filt, S, hy);
    const int NJ = pl.rows.NJ;
    std::vector<double> A((size_t)NJ * Sh, 0.0);
    double maxre = 0.0, maxim = 0.0;
    for (int rr = 0; rr < S; ++rr) {
        const size_t q = pl.rows.rowjob[rr];
        for (int k = 0; k < Sh; ++k) {
            A[q * Sh + k] += hy[((size_t)rr * Sh + k) * 2];
            maxim = std::max(maxim, std::fabs(hy[((size_t)rr * Sh + k) * 2 + 1]));
        }
    }
    for (double v : A) maxre = std::max(maxre, std::fabs(v));
    if (!(maxim <= 1e-15 * maxre)) { pl.why = "transfer-function weights are not real"; return; }
    if (NJ < 16 || NJ > 600) { pl.why = "job count outside the matrix-product kernel's range"; return; }
    pl.r = jxt::lowrank_factor(A.data(), NJ, Sh, tol, pl.L, pl.V);
    // (a measured transfer function can have full rank, r = NJ: the products below then run over more rows, nothing else changes)
    if (!(pl.r > 0)) { pl.why = "transfer-function weights vanish"; return; }
    int kact = Ph;
    {
        const double band_tol = 0.03 * tol;
        double tmax = 0.0;
        for (double v : pl.taps) tmax = std::max(tmax, std::fabs(v));
        while (kact > 1) {
            double m = 0.0;
            for (int t = 0; t <= o; ++t) m = std::max(m, std::fabs(pl.taps[(size_t)t * Ph + kact - 1]));
            if (m > band_tol * tmax) break;
            --kact;
        }
    }
    pl.kact = kact;
    if ((S / 2 + 1 + 7) / 8 > JX_LR_KS) { pl.why = "map side beyond the second matrix product's k range (two halves)"; return; }
    pl.ok = true;
}

// librccl, loaded on first use (dlopen): the functions of the all-gather path only
namespace {
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
RcclApi g_rccl;

bool rccl_load() {
    if (g_rccl.h) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        g_rccl.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.h) break;
    }
    if (!g_rccl.h) { g_rccl.err = std::string("dlopen(librccl): ") + (dlerror() ? dlerror() : "?"); return false; }
#define JX_SYM(field, sym) *(void**)(&g_rccl.field) = dlsym(g_rccl.h, sym); if (!g_rccl.field) { g_rccl.err = std::string("librccl lacks ") + sym; g_rccl.h = nullptr; return false; }
    JX_SYM(GetUniqueId, "ncclGetUniqueId") JX_SYM(CommInitRank, "ncclCommInitRank") JX_SYM(AllGather, "ncclAllGather")
    JX_SYM(AllReduce, "ncclAllReduce") JX_SYM(CommDestroy, "ncclCommDestroy") JX_SYM(GetErrorString, "ncclGetErrorString")
#undef JX_SYM
    return true;
}
}  // namespace

#define NCCLCHK(ctx, call)                                                                         \
    do {                                                                                           \
        ncclResult_t r_ = (call);                                                                  \
        if (r_ != ncclSuccess) {                                                                   \
            (ctx)->err = std::string(#call) + ": " + g_rccl.GetErrorString(r_);                    \
            return JX_ERR_COMM;                                                                    \
        }                                                                                          \
    } while (0)

extern "C" {

int jx_comm_unique_id(void* id_out) {
    if (!id_out) return JX_ERR_INVALID;
    if (!rccl_load()) return JX_ERR_COMM;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return JX_ERR_COMM;
    static_assert(sizeof(id) == JX_COMM_ID_BYTES, "id size");
    memcpy(id_out, &id, sizeof(id));
    return JX_OK;
}

int jx_comm_init_rank(jx_ctx* ctx, const void* idp, int nranks, int rank) {
    if (!ctx || !idp || nranks < 1 || rank < 0 || rank >= nranks) return JX_ERR_INVALID;
    if (ctx->comm) { ctx->err = "jx_comm_init_rank: this context already has a communicator"; return JX_ERR_STATE; }
    if (!rccl_load()) { ctx->err = g_rccl.err; return JX_ERR_COMM; }
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    ncclUniqueId id;
    memcpy(&id, idp, sizeof(id));
    // RCCL prints a version banner on stdout when its first communicator comes up; a caller whose stdout is a protocol
    // (bench.py: one JSON line) must not see it, so file descriptor 1 points at stderr for the duration of the call
    fflush(stdout);
    const int saved = dup(1);
    if (saved >= 0) dup2(2, 1);
    const ncclResult_t r0 = g_rccl.CommInitRank(&ctx->comm, nranks, id, rank);
    fflush(stdout);
    if (saved >= 0) { dup2(saved, 1); close(saved); }
    NCCLCHK(ctx, r0);
    ctx->comm_rank = rank; ctx->comm_size = nranks;
    return JX_OK;
}

int jx_allgather_logp(jx_ctx* ctx, const double* send_dev, double* recv_dev, int count) {
    if (!ctx || !send_dev || !recv_dev || count < 0) return JX_ERR_INVALID;
    if (!ctx->comm) { ctx->err = "jx_allgather_logp before jx_comm_init_rank"; return JX_ERR_STATE; }
    if (count == 0) return JX_OK;
    NCCLCHK(ctx, g_rccl.AllGather(send_dev, recv_dev, (size_t)count, ncclDouble, ctx->comm, ctx->stream));
    return JX_OK;
}

int jx_comm_allreduce_max(jx_ctx* ctx, double* inout_dev, int count) {
    if (!ctx || !inout_dev || count < 1) return JX_ERR_INVALID;
    if (!ctx->comm) { ctx->err = "jx_comm_allreduce_max before jx_comm_init_rank"; return JX_ERR_STATE; }
    NCCLCHK(ctx, g_rccl.AllReduce(inout_dev, inout_dev, (size_t)count, ncclDouble, ncclMax, ctx->comm, ctx->stream));
    return JX_OK;
}

int jx_comm_destroy(jx_ctx* ctx) {
    if (!ctx) return JX_ERR_INVALID;
    if (!ctx->comm) return JX_OK;
    (void)hipStreamSynchronize(ctx->stream);
    NCCLCHK(ctx, g_rccl.CommDestroy(ctx->comm));
    ctx->comm = nullptr; ctx->comm_rank = 0; ctx->comm_size = 1;
    return JX_OK;
}

const char* jx_strerror(int s) {
    switch (s) {
        case JX_OK: return "ok";
        case JX_ERR_INVALID: return "invalid argument";
        case JX_ERR_STATE: return "call out of order";
        case JX_ERR_MISSING: return "required tensor not uploaded";
        case JX_ERR_HIP: return "HIP runtime error";
        case JX_ERR_ROCFFT: return "rocFFT error";
        case JX_ERR_NOMEM: return "out of memory";
        case JX_ERR_NODEVICE: return "no usable HIP device";
        case JX_ERR_UNSUPPORTED: return "unsupported size";
        case JX_ERR_COMM: return "RCCL error";
    }
    return "unknown status";
}

const char* jx_last_error(jx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int jx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* jx_device_name(jx_ctx* ctx) { return ctx ? ctx->devname.c_str() : ""; }

int jx_create(const jx_config* cfg, jx_ctx** out) {
    if (!cfg || !out) return JX_ERR_INVALID;
    *out = nullptr;
    if (cfg->abi_version != JX_ABI_VERSION) return JX_ERR_INVALID;
    const jx_config& c = *cfg;
    if (c.S < 7 || c.N < 4 || c.B < 1 || (c.B % 2) == 0 || c.nflux < 1 || c.nconv < 2) return JX_ERR_INVALID;
    if (c.npar != 16 && c.npar != 19) return JX_ERR_INVALID;
    if ((c.ne_mode == 1) != (c.npar == 19)) return JX_ERR_INVALID;
    if (c.ndim < 1 || c.ndim > c.npar) return JX_ERR_INVALID;
    if (!c.sz_only && (c.nann < 1 || c.nband < 1 || c.ntab < 2)) return JX_ERR_INVALID;
    if (c.N < c.S - c.S / 2) return JX_ERR_INVALID;       // r_pp[:nrow-1] must exist (joxsz_funcs.py:469)
    if (!(c.step > 0) || !(c.kpc_as > 0) || !(c.m_e > 0) || !(c.sigma_T > 0) || !(c.kpc_cm > 0)) return JX_ERR_INVALID;
    if (c.nann > 64 || c.nband > 64 || c.N > 4096 || c.S > 4096) return JX_ERR_UNSUPPORTED;
    if (c.dtype != 0 && c.dtype != 1) return JX_ERR_INVALID;
    if (c.calc_integ && !(c.integ_sig > 0)) return JX_ERR_INVALID;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return JX_ERR_NODEVICE;
    if (c.device < 0 || c.device >= ndev) return JX_ERR_NODEVICE;

    jx_ctx* ctx = new (std::nothrow) jx_ctx();
    if (!ctx) return JX_ERR_NOMEM;
    ctx->cfg = c;
    ctx->host.resize(JX_T_COUNT);
    ctx->have.assign(JX_T_COUNT, false);
    if (hipSetDevice(c.device) != hipSuccess) { delete ctx; return JX_ERR_NODEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c.device) != hipSuccess) { delete ctx; return JX_ERR_NODEVICE; }
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->devname = std::string(prop.name[0] ? prop.name : "AMD GPU") + " (" + prop.gcnArchName + ", " + std::to_string(prop.multiProcessorCount) + " CUs)";
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return JX_ERR_HIP; }
    ctx->own_stream = ctx->stream;
    if (g_rocfft_refs++ == 0) rocfft_setup();
    *out = ctx;
    return JX_OK;
}

int jx_upload(jx_ctx* ctx, int id, const void* host, size_t nbytes) {
    if (!ctx || !host) return JX_ERR_INVALID;
    if (ctx->finalized) { ctx->err = "jx_upload after jx_finalize"; return JX_ERR_STATE; }
    if (id < 0 || id >= JX_T_COUNT) return JX_ERR_INVALID;
    const size_t want = tensor_bytes(ctx->cfg, id);
    if (nbytes != want) {
        ctx->err = "tensor " + std::to_string(id) + ": got " + std::to_string(nbytes) + " bytes, expected " + std::to_string(want);
        return JX_ERR_INVALID;
    }
    ctx->host[id].assign((const unsigned char*)host, (const unsigned char*)host + nbytes);
    ctx->have[id] = true;
    return JX_OK;
}

static int make_plans(jx_ctx* ctx, int batch, Plan3** out) {
    auto it = ctx->plans.find(batch);
    if (it != ctx->plans.end()) { *out = &it->second; return JX_OK; }
    Plan3 pl;
    const size_t P = ctx->P, S = ctx->cfg.S, Sh = ctx->Sh;
    {   // beam convolution forward: real [P][P] -> hermitian [P][Ph]
        size_t len[2] = {P, P};
        FFTCHK(ctx, rocfft_plan_create(&pl.beam_fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                       rocfft_precision_double, 2, len, (size_t)batch, nullptr));
        FFTCHK(ctx, rocfft_plan_create(&pl.beam_inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                                       rocfft_precision_double, 2, len, (size_t)batch, nullptr));
    }
    {   // transfer function forward: real S x S window of the padded image (row stride P)
        rocfft_plan_description desc = nullptr;
        FFTCHK(ctx, rocfft_plan_description_create(&desc));
        size_t istr[2] = {1, P}, ostr[2] = {1, Sh};
        FFTCHK(ctx, rocfft_plan_description_set_data_layout(desc, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                            nullptr, nullptr, 2, istr, P * P, 2, ostr, S * Sh));
        size_t len[2] = {S, S};
        FFTCHK(ctx, rocfft_plan_create(&pl.tf_fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                       rocfft_precision_double, 2, len, (size_t)batch, desc));
        rocfft_plan_description_destroy(desc);
    }
    size_t w1 = 0, w2 = 0, w3 = 0;
    FFTCHK(ctx, rocfft_plan_get_work_buffer_size(pl.beam_fwd, &w1));
    FFTCHK(ctx, rocfft_plan_get_work_buffer_size(pl.beam_inv, &w2));
    FFTCHK(ctx, rocfft_plan_get_work_buffer_size(pl.tf_fwd, &w3));
    pl.work_bytes = std::max(w1, std::max(w2, w3));
    if (pl.work_bytes > ctx->work_cap) {
        if (ctx->d_work) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->d_work)); ctx->d_work = nullptr; }
        HIPCHK(ctx, hipMalloc(&ctx->d_work, pl.work_bytes));
        ctx->device_bytes += (int64_t)pl.work_bytes - (int64_t)ctx->work_cap;
        ctx->work_cap = pl.work_bytes;
    }
    if (ctx->work_cap) FFTCHK(ctx, rocfft_execution_info_set_work_buffer(ctx->info, ctx->d_work, ctx->work_cap));
    ctx->plans[batch] = pl;
    *out = &ctx->plans[batch];
    return JX_OK;
}

// the operator of jx_abel_gemm_kernel goes to the device once a route that evaluates the map rows from (y, M) exists
static int setup_abel_gemm(jx_ctx* ctx, int chunk) {
    if (ctx->h_Tm.empty() || ctx->d_Tm) return JX_OK;
    int rc;
    if ((rc = dev_put(ctx, ctx->h_Tm.data(), ctx->h_Tm.size(), &ctx->d_Tm))) return rc;
    if ((rc = dev_new(ctx, (size_t)chunk * ctx->cfg.N, &ctx->d_ppc, true))) return rc;
    std::vector<double>().swap(ctx->h_Tm);
    const char* e = getenv("JOXSZ_ABEL_GEMM");
    ctx->abel_gemm = !(e && atoi(e) == 0);
    return JX_OK;
}

// ---- contracted route (jx_mix.hpp): host-side plan.  Built before the route is chosen, so that `auto` can fall back.
#define JX_MIX_NS 8
#define JX_MIX_RTS(X) X(4) X(8) X(12) X(16) X(20) X(24) X(28) X(32) X(40) X(48) X(56) X(64)
#define JX_MIX_NXTS(X) X(1) X(2) X(3) X(4) X(5) X(6)
#define JX_MIX_KSPLIT_MAX 64
struct MixBuild {
    bool ok = false;
    std::string why;
    int NU = 0, r = 0, ns = 0, R = 0, RT = 0, nxt = 0, ntile = 0, nog = 0, ksteps = 0;
    size_t krows = 0;
    jxt::MixColumns cols;
    std::vector<double> Cm, Op;
    int cld = 0;
    double tol = 0.0, beam_tol = 0.0;
};

static void plan_mix(const jx_config& c, const std::vector<double>& beam, const std::vector<double>& filt, const std::vector<double>& Qtab,
                     int qn, bool mirror, const std::vector<double>& r, double tol, MixBuild& mb) {
    const int S = c.S, B = c.B, Sh = S / 2 + 1, nrow = S - S / 2;
    if (!mirror) { mb.why = "d_mat lacks the mirror structure of centdistmat"; return; }
    if (c.fft_pad != 0) { mb.why = "fft_pad is a parameter of the rocFFT sequence"; return; }
    const int NU = std::max(S / 2, S - 1 - S / 2) + 1;
    if (qn != NU) { mb.why = "quadrant table size"; return; }
    if (NU > 9 * 64) { mb.why = "map side beyond the symmetric map kernel's range"; return; }
    mb.NU = NU; mb.tol = tol; mb.beam_tol = 1e-14;
    if (!jxt::mix_column_tables(Qtab, qn, NU, r, mb.cols)) { mb.why = "pixel radii do not grow along the columns of d_mat"; return; }
    // transfer-function weights of the extracted row, real for a real point-symmetric filter
    std::vector<double> hy;
    jxt::tf_hy_table(filt, S, hy);
    std::vector<double> A((size_t)S * Sh);
    double maxre = 0.0, maxim = 0.0;
    for (size_t e = 0; e < A.size(); ++e) { A[e] = hy[2 * e]; maxre = std::max(maxre, std::fabs(hy[2 * e])); maxim = std::max(maxim, std::fabs(hy[2 * e + 1])); }
    if (!(maxim <= 1e-15 * maxre)) { mb.why = "transfer-function weights are not real"; return; }
    std::vector<double> U, V, by, bx;
    mb.r = jxt::lowrank_factor_qr(A.data(), S, Sh, tol, U, V);
    if (mb.r <= 0) { mb.why = "transfer-function weights vanish"; return; }
    mb.ns = jxt::beam_separable_terms(beam, B, c.step * c.step, mb.beam_tol, by, bx);
    if (mb.ns <= 0) { mb.why = "beam image vanishes"; return; }
    mb.R = mb.r * mb.ns;
#define JX_PICK(Rv) if (!mb.RT && mb.R <= Rv) mb.RT = Rv;
    JX_MIX_RTS(JX_PICK)
#undef JX_PICK
    if (!mb.RT) { mb.why = "rank of the separable form beyond the stage-1 kernel (" + std::to_string(mb.r) + " x " + std::to_string(mb.ns) + " terms)"; return; }
    mb.cld = mb.RT;
    jxt::mix_stage1_operator(U, mb.r, by, mb.ns, S, B, NU, mb.cols.wld, mb.cld, mb.Cm);
    // output tiling of stage 2: the instance with the least padded work
    const int tiles = (nrow + 15) / 16;
    double best = 1e300;
#define JX_PICK(Xv) { const int og = (tiles + Xv - 1) / Xv; const double cost = (double)og * Xv * (1.0 + 0.5 / Xv); if (cost < best) { best = cost; mb.nxt = Xv; mb.nog = og; } }
    JX_MIX_NXTS(JX_PICK)
#undef JX_PICK
    mb.ntile = mb.nog * mb.nxt;
    const size_t K = (size_t)NU * mb.R;
    mb.ksteps = (int)((K + 3) / 4);
    mb.krows = 4 * ((size_t)mb.ksteps + (size_t)JX_MIX_KSPLIT_MAX * JX_OPG_RD + JX_OPG_RD);
    jxt::mix_stage2_operator(V, mb.r, bx, mb.ns, S, B, NU, mb.krows, mb.ntile, mb.Op);
    mb.ok = true;
}

#define JX_LR_TOL_DEFAULT 1e-8
#define JX_TRUNC_BOUND 2e-10
static int finalize_impl(jx_ctx* ctx) {
    if (!ctx) return JX_ERR_INVALID;
    if (ctx->finalized) { ctx->err = "jx_finalize called twice"; return JX_ERR_STATE; }
    const jx_config& c = ctx->cfg;
    for (int id = 0; id < JX_T_COUNT; ++id) {
        if (c.sz_only && tensor_is_xray(id)) continue;
        if (id == JX_T_INTEG_W && !c.calc_integ) continue;
        if (!ctx->have[id]) { ctx->err = "tensor " + std::to_string(id) + " missing"; return JX_ERR_MISSING; }
    }
    HIPCHK(ctx, hipSetDevice(c.device));

    const int S = c.S, N = c.N, B = c.B;
    ctx->nrow = S - S / 2;
    ctx->nt = ctx->nrow - 1;
    ctx->Sh = S / 2 + 1;
    const int o = (B - 1) / 2;
    std::vector<double> r = host_vec<double>(ctx, JX_T_R_PP);
    const std::vector<double>& r_grid = r;
    for (int i = 0; i < N; ++i) {
        if (!(r[i] > 0) || (i && !(r[i] > r[i - 1]))) { ctx->err = "r_pp must be positive and increasing"; return JX_ERR_INVALID; }
    }
    // ---- does d_mat have the mirror structure d_mat[iy][ix] = Q[|iy-c|][|ix-c|] (what centdistmat builds)?
    std::vector<double> dm_h = host_vec<double>(ctx, JX_T_D_MAT);
    const int cc0 = S / 2, qn = std::max(cc0, S - 1 - cc0) + 1;
    std::vector<double> Qtab((size_t)qn * qn);
    bool dmat_mirror = true;
    {
        for (int b = 0; b < qn; ++b)
            for (int a = 0; a < qn; ++a) {
                const int iy = (cc0 + b < S) ? cc0 + b : cc0 - b, ix = (cc0 + a < S) ? cc0 + a : cc0 - a;
                Qtab[(size_t)b * qn + a] = dm_h[(size_t)iy * S + ix];
            }
        for (int iy = 0; iy < S && dmat_mirror; ++iy)
            for (int ix = 0; ix < S; ++ix) {
                const double q = Qtab[(size_t)std::abs(iy - cc0) * qn + std::abs(ix - cc0)];
                if (memcmp(&q, &dm_h[(size_t)iy * S + ix], sizeof(double)) != 0) { dmat_mirror = false; break; }
            }
    }
    // ---- which convolution: rocFFT sequence or the hand-written mixed-domain passes
    int want = c.conv_mode;
    if (const char* e = getenv("JOXSZ_CONV")) {
        if (!strcmp(e, "rocfft")) want = 1; else if (!strcmp(e, "custom")) want = 2; else if (!strcmp(e, "auto")) want = 0; else if (!strcmp(e, "mix")) want = 3;
    }
    if (want < 0 || want > 3) { ctx->err = "conv_mode must be 0, 1, 2 or 3"; return JX_ERR_INVALID; }
    std::vector<double> beam_h = host_vec<double>(ctx, JX_T_BEAM_2D);
    const bool oddS = (S & 1) != 0;
    OddPlan oplan;
    // singular-value cut of the transfer-function weights.  Small maps (a beam image comparable with the map: the extracted
    // row is then a small difference of large terms) keep every term above rounding, where it costs next to nothing; large
    // maps cut at 1e-10 (rank 46 instead of 61 at 512^2), which the truncation test of tests/test_gpu_parity.py bounds
    // singular-value cut of the transfer-function weights: 1e-8 of the largest leaves rank ~20 of ~290 at 512^2 and changes the
    // extracted row by ~1e-11, chi^2/2 by < 5e-9 (scripts/tol_scan.py); small maps keep every term above rounding (there the
    // 55-pixel beam is almost the map and the log-posterior a small difference of large terms); jx_finalize measures the
    // effect on the caller's data and tightens the cut when it is not small enough
    double lr_tol0 = (S < 400) ? 1e-13 : JX_LR_TOL_DEFAULT;
    if (const char* e = getenv("JOXSZ_LOWRANK_TOL")) { const double v2 = atof(e); if (v2 > 0.0 && v2 < 1e-6) lr_tol0 = v2; }
    if (ctx->lr_tol_override > 0.0) lr_tol0 = ctx->lr_tol_override;
    // contracted route first (jx_mix.hpp): every map side, odd ones included
    MixBuild mixb;
    if (want == 3 || want == 0) {
        plan_mix(c, beam_h, host_vec<double>(ctx, JX_T_FILTERING), Qtab, qn, dmat_mirror, r, lr_tol0, mixb);
        if (want == 3 && !mixb.ok) { ctx->err = "contracted route: " + mixb.why; return JX_ERR_UNSUPPORTED; }
        if (mixb.ok) want = 3;
    }
    if (oddS && want != 1 && want != 3 && c.fft_pad == 0 && JX_FIR_TILE + 2 * o <= JX_FIR_RING && !getenv("JOXSZ_ODD_ROCFFT"))
        plan_odd(c, beam_h, host_vec<double>(ctx, JX_T_FILTERING), host_vec<double>(ctx, JX_T_D_MAT), lr_tol0, oplan);
    const int lp_custom = oddS ? (oplan.ok ? oplan.LP : 0) : jxt::custom_conv_lp(S, o);
    const size_t fir_lds = sizeof(double) * ((size_t)2 * JX_FIR_RING * JX_FIR_KX + (size_t)(o + 1) * JX_FIR_KX) + sizeof(int) * (size_t)(S + 4);
    const bool eligible = lp_custom > 0 && jxt::beam_is_symmetric(beam_h, B) && JX_FIR_TILE + 2 * o <= JX_FIR_RING && c.fft_pad == 0;
    if (want == 2 && !eligible) {
        ctx->err = oddS ? std::string("hand-written convolution, odd map side: ") + (oplan.why[0] ? oplan.why : "fft_pad must be 0 and (B-1)/2 <= 32")
                        : std::string("hand-written convolution needs S/2 in {16,24,32,64,128,256,512}, a flip-symmetric beam with (B-1)/2 <= 32 and fft_pad = 0");
        return JX_ERR_UNSUPPORTED;
    }
    ctx->odd = oddS && eligible && want != 1 && want != 3;
    ctx->conv_mode = (want == 3) ? 3 : (want == 2 || (want == 0 && eligible)) ? 2 : 1;
    int P = c.fft_pad > 0 ? c.fft_pad : jxt::next_smooth_even(S + o);
    if (const char* e = getenv("JOXSZ_FFT_PAD")) { int v = atoi(e); if (v > 0 && ctx->conv_mode == 1) P = v; }
    if (ctx->conv_mode == 2) P = 2 * lp_custom;
    if (P < S + o) { ctx->err = "fft_pad smaller than S + (B-1)/2"; return JX_ERR_INVALID; }
    ctx->P = P;
    ctx->Ph = P / 2 + 1;

    std::vector<int32_t> thawed = host_vec<int32_t>(ctx, JX_T_THAWED_IDX);
    for (int k = 0; k < c.ndim; ++k)
        if (thawed[k] < 0 || thawed[k] >= c.npar) { ctx->err = "thawed_idx out of range"; return JX_ERR_INVALID; }

    JxDev& d = ctx->d;
    memset(&d, 0, sizeof(d));
    d.S = S; d.N = N; d.B = B; d.P = P; d.Ph = ctx->Ph; d.Sh = ctx->Sh; d.nrow = ctx->nrow; d.nt = ctx->nt;
    d.nflux = c.nflux; d.nconv = c.nconv; d.nann = c.nann; d.nband = c.nband; d.ntab = c.ntab;
    d.npar = c.npar; d.ndim = c.ndim; d.ne_mode = c.ne_mode; d.exclude_unphy_mass = c.exclude_unphy_mass;
    d.sz_only = c.sz_only;
    if (const char* e = getenv("JOXSZ_DBG")) d.dbg = atoi(e);      // timing-only ablations, results are wrong
    d.y_scale = c.kpc_cm * c.sigma_T / c.m_e;
    d.inv_h_mean = (double)(N - 1) / (r[N - 1] - r[0]);

    int rc;
    // ---- Abel weights in on-the-fly form (per-source factor, diagonal, first off-diagonal)
    {
        std::vector<double> cj, dg, sp;
        jxt::abel_onfly_tables(r, cj, dg, sp);
        std::vector<double> tab((size_t)N * 4);
        for (int j = 0; j < N; ++j) { tab[4 * j] = r[j]; tab[4 * j + 1] = cj[j]; tab[4 * j + 2] = dg[j]; tab[4 * j + 3] = sp[j]; }
        double* p;
        if ((rc = dev_put(ctx, tab.data(), tab.size(), &p))) return rc; d.abel_tab = p;
    }
    // ---- spline moment operator of the mirrored grid, stored as a band
    {
        std::vector<double> G;
        if (!jxt::mirrored_spline_op(r, G)) { ctx->err = "spline operator: singular system"; return JX_ERR_INVALID; }
        int K = jxt::band_halfwidth(G, N, 1e-20);
        ctx->K = d.K = K;
        std::vector<double> band((size_t)(2 * K + 1) * N, 0.0);
        for (int i = 0; i < N; ++i)
            for (int k = -K; k <= K; ++k) {
                const int j = i + k;
                if (j >= 0 && j < N) band[(size_t)(k + K) * N + i] = G[(size_t)i * N + j];
            }
        double* p; if ((rc = dev_put(ctx, band.data(), band.size(), &p))) return rc; d.gband = p;
        {
            // operator of jx_abel_gemm_kernel (jx_tables.hpp::abel_spline_operator)
            ctx->tm_ntile = (2 * N + 15) / 16;
            ctx->tm_npair = (ctx->tm_ntile + 1) / 2;
            ctx->tm_ld = 32 * ctx->tm_npair;
            jxt::abel_spline_operator(r, G, K, d.y_scale, JX_AG_ROWS(N), ctx->tm_ld, ctx->h_Tm);
        }
        if (c.calc_integ) {
            // cint = w . [f(0), y],  f(0) = y_0 - r_0^2/2 * (G y)_0 (value at 0 of the mirrored spline),  y = y_scale * A pp:
            // one weight per radius of the pressure profile
            std::vector<double> w = host_vec<double>(ctx, JX_T_INTEG_W), wy(N), A, wp(N, 0.0);
            for (int j = 0; j < N; ++j) wy[j] = w[j + 1] - w[0] * 0.5 * r[0] * r[0] * G[j];
            wy[0] += w[0];
            jxt::abel_matrix(r, A);
            for (int i = 0; i < N; ++i) {
                const double f = d.y_scale * wy[i];
                for (int j = i; j < N; ++j) wp[j] += f * A[(size_t)i * N + j];
            }
            if ((rc = dev_put(ctx, wp.data(), wp.size(), &p))) return rc;
            d.integ_wp = p; d.calc_integ = 1; d.integ_mu = c.integ_mu; d.integ_sig = c.integ_sig;
        }
    }
    // ---- h(0) weights: spline through (+-r_pp[:nt], t) evaluated at 0 (joxsz_funcs.py:470-473)
    {
        std::vector<double> rt(r.begin(), r.begin() + ctx->nt), G;
        if (!jxt::mirrored_spline_op(rt, G)) { ctx->err = "h(0) operator: singular system"; return JX_ERR_INVALID; }
        std::vector<double> hw(ctx->nt);
        for (int j = 0; j < ctx->nt; ++j) hw[j] = -0.5 * rt[0] * rt[0] * G[j];
        hw[0] += 1.0;
        double* p; if ((rc = dev_put(ctx, hw.data(), hw.size(), &p))) return rc; d.hw = p;
    }
    // ---- evaluation matrix of g at the data radii (joxsz_funcs.py:476)
    {
        std::vector<double> radius = host_vec<double>(ctx, JX_T_RADIUS);
        std::vector<double> xk(radius.begin() + S / 2, radius.end());
        for (size_t i = 1; i < xk.size(); ++i)
            if (!(xk[i] > xk[i - 1])) { ctx->err = "radius[S//2:] must be increasing"; return JX_ERR_INVALID; }
        std::vector<double> flux = host_vec<double>(ctx, JX_T_FLUX_DATA);
        std::vector<double> q(flux.begin(), flux.begin() + c.nflux), E;
        if (!jxt::nak_eval_matrix(xk, q, E)) { ctx->err = "profile spline: singular system"; return JX_ERR_INVALID; }
        double* p; if ((rc = dev_put(ctx, E.data(), E.size(), &p))) return rc; d.emat = p;
        if ((rc = dev_put(ctx, flux.data(), flux.size(), &p))) return rc; d.flux = p;
    }

    // ---- twiddles of the final inverse transform of the extracted row (both modes)
    {
        double* p;
        std::vector<double> tw((size_t)S * 2);
        for (int m = 0; m < S; ++m) { tw[2 * m] = std::cos(2.0 * jxt::kPi * m / S); tw[2 * m + 1] = std::sin(2.0 * jxt::kPi * m / S); }
        if ((rc = dev_put(ctx, tw.data(), tw.size(), &p))) return rc; d.twid = p;
    }
    if (ctx->conv_mode == 3) {
        ctx->lr_tol = lr_tol0;
    } else if (ctx->conv_mode == 1) {
        // ---- rocFFT sequence: beam spectrum and transfer-function row table
        std::vector<double> bh;
        jxt::beam_spectrum(beam_h, B, P, c.step * c.step / ((double)P * (double)P), bh);
        double* p; if ((rc = dev_put(ctx, bh.data(), bh.size(), &p))) return rc; d.bhat = p;
        std::vector<double> filt = host_vec<double>(ctx, JX_T_FILTERING), H;
        jxt::tf_row_table(filt, S, H);
        if ((rc = dev_put(ctx, H.data(), H.size(), &p))) return rc; d.htab = p;
    } else {
        // ---- hand-written passes: twiddles, real FIR taps of the beam, Hy table
        JxConv& cv = ctx->cv;
        memset(&cv, 0, sizeof(cv));
        cv.S = S; cv.Sh = ctx->Sh; cv.B = B; cv.o = o; cv.P = P; cv.Ph = ctx->Ph; cv.LP = P / 2; cv.LS = S / 2; cv.ntap = o + 1;
        memset(&ctx->lr, 0, sizeof(ctx->lr));
        memset(&ctx->lrf, 0, sizeof(ctx->lrf));
        memset(&ctx->lrf0, 0, sizeof(ctx->lrf0));
        if (ctx->odd) {
            // odd side: row bookkeeping and low-rank weights come from the plan; the matrices follow with the work buffers
            cv.NU = oplan.rows.NU; cv.NJ = oplan.rows.NJ; cv.nseg = oplan.rows.nseg; cv.CROWS = oplan.rows.NJ + 1; cv.mirror = 1;
            cv.xsym = 1; cv.fir_ld = (cv.Ph + 15) & ~15; cv.nblk3 = 0;
            ctx->lr_tol = lr_tol0;
            ctx->kact = oplan.kact;
            ctx->lr.r = oplan.r; ctx->lr.nq = cv.NJ;
            ctx->h_rows = oplan.rows; ctx->h_taps = oplan.taps;
            int* q;
            if ((rc = dev_put(ctx, oplan.rows.rowjob.data(), oplan.rows.rowjob.size(), &q))) return rc; ctx->d_rowjob = q;
        } else {
        ctx->p13_rows = ctx->p1_rows = 0;
#define JX_SEL2(LPv, LSv, R1v, R3v) if (cv.LP == LPv && cv.LS == LSv) { ctx->p1_rows = R1v; ctx->p13_rows = R3v; }
        JX_CONV2_PAIRS(JX_SEL2)
#undef JX_SEL2
        if (!ctx->p1_rows) { ctx->err = "no convolution kernels for this size"; return JX_ERR_UNSUPPORTED; }
        // row bookkeeping: distinct map rows, conv jobs, segments (identity tables without the mirror structure)
        bool use_mirror = dmat_mirror;
        if (const char* e = getenv("JOXSZ_CONV_NOSYM")) { if (atoi(e) > 0) use_mirror = false; }
        jxt::ConvRows rows;
        jxt::conv_row_tables(S, o, use_mirror, rows);
        cv.NU = rows.NU; cv.NJ = rows.NJ; cv.nseg = rows.nseg; cv.CROWS = rows.NJ + 1; cv.mirror = use_mirror ? 1 : 0;
        cv.nblk3 = (cv.NJ + ctx->p13_rows - 1) / ctx->p13_rows;
        {
            int* q;
            if ((rc = dev_put(ctx, rows.urow.data(), rows.urow.size(), &q))) return rc; cv.urow = q;
            if ((rc = dev_put(ctx, rows.umap.data(), rows.umap.size(), &q))) return rc; cv.umap = q;
            if ((rc = dev_put(ctx, rows.jrow.data(), rows.jrow.size(), &q))) return rc; cv.jrow = q;
            if ((rc = dev_put(ctx, rows.seg.data(), rows.seg.size(), &q))) return rc; cv.seg = q;
            if ((rc = dev_put(ctx, rows.rowjob.data(), rows.rowjob.size(), &q))) return rc; ctx->d_rowjob = q;
        }
        std::vector<double> v;
        double* p;
        jxt::twiddles(cv.LP, cv.LP, v); if ((rc = dev_put(ctx, v.data(), v.size(), &p))) return rc; cv.tw_lp = (const cplx*)p;
        jxt::twiddles(cv.LS, cv.LS, v); if ((rc = dev_put(ctx, v.data(), v.size(), &p))) return rc; cv.tw_ls = (const cplx*)p;
        jxt::twiddles(P, cv.LP + 1, v); if ((rc = dev_put(ctx, v.data(), v.size(), &p))) return rc; cv.tw_p = (const cplx*)p;
        jxt::twiddles(S, cv.LS + 1, v); if ((rc = dev_put(ctx, v.data(), v.size(), &p))) return rc; cv.tw_s = (const cplx*)p;
        jxt::beam_fir_taps(beam_h, B, P, c.step * c.step / (double)P, v);
        if ((rc = dev_put(ctx, v.data(), v.size(), &p))) return rc; cv.taps = p;
        ctx->h_taps = v; ctx->h_rows = rows;
        std::vector<double> filt = host_vec<double>(ctx, JX_T_FILTERING), hy;
        jxt::tf_hy_table(filt, S, hy);
        // Hy weights summed over the conv rows of each job
        std::vector<double> hyc((size_t)cv.NJ * cv.Sh * 2, 0.0);
        for (int r = 0; r < S; ++r) {
            const size_t q = rows.rowjob[r];
            for (int k = 0; k < cv.Sh * 2; ++k) hyc[q * cv.Sh * 2 + k] += hy[(size_t)r * cv.Sh * 2 + k];
        }
        if ((rc = dev_put(ctx, hyc.data(), hyc.size(), &p))) return rc; cv.hy = (const cplx*)p;
        // low-rank form of the weights (see jx_lowrank_kernel): real weights, few enough jobs for the register-held U tile
        {
            double tol = lr_tol0;           // (1e-10: the extracted row then agrees with the untruncated weights to ~1e-10, the log-posterior to ~1e-13)
            bool want = true;
            if (const char* e = getenv("JOXSZ_LOWRANK")) { if (atoi(e) == 0) want = false; }
            ctx->lr_tol = tol;
            double maxre = 0.0, maxim = 0.0;
            for (size_t e = 0; e < hyc.size(); e += 2) { maxre = std::max(maxre, std::fabs(hyc[e])); maxim = std::max(maxim, std::fabs(hyc[e + 1])); }
            if (want && cv.NJ >= 32 && cv.NJ <= 600 && maxim <= 1e-15 * maxre) {
                std::vector<double> A((size_t)cv.NJ * cv.Sh), L, Rt;
                for (size_t e = 0; e < A.size(); ++e) A[e] = hyc[2 * e];
                const int r = jxt::lowrank_factor(A.data(), cv.NJ, cv.Sh, tol, L, Rt);
                int bucket = 0;
#define JX_LR_PICK(K) if (!bucket && (cv.NJ + 3) / 4 <= K) bucket = K;
                JX_LR_BUCKETS(JX_LR_PICK)
#undef JX_LR_PICK
                ctx->lr_bucket = bucket;
                const size_t lds_need = (size_t)((r + 15) / 16) * bucket * 64 * sizeof(double);
                // the combination as its own kernel (behind the FIR kernel) holds the U tile for all jobs in LDS
                ctx->lr_sep = bucket && lds_need <= JX_LR_LDS_MAX;
                if (r > 0 && 2 * r <= cv.NJ && r <= 64) {
                    JxLowrank& lr = ctx->lr;
                    lr.r = r; lr.nq = cv.NJ; lr.ks = (cv.NJ + 3) / 4; lr.KQ = 4 * lr.ks;
                    const int RP = ((r + 15) / 16) * 16;
                    std::vector<double> U((size_t)RP * lr.KQ, 0.0), vt((size_t)r * cv.Sh * 2, 0.0);
                    for (int rho = 0; rho < r; ++rho) {
                        for (int q = 0; q < cv.NJ; ++q) U[(size_t)rho * lr.KQ + q] = L[(size_t)rho * cv.NJ + q];
                        for (int k = 0; k < cv.Sh; ++k) vt[((size_t)rho * cv.Sh + k) * 2] = Rt[(size_t)rho * cv.Sh + k];
                    }
                    if ((rc = dev_put(ctx, U.data(), U.size(), &p))) return rc; lr.U = p;
                    if ((rc = dev_put(ctx, vt.data(), vt.size(), &p))) return rc;
                    ctx->lr_vt = (const cplx*)p;
                    ctx->h_L = L;
                }
            }
        }
        ctx->p2_lds = fir_lds;
        // runs of the register-window FIR: every segment cut into pieces of at most `runlen` conv rows
        {
            int runlen = 128;
            if (const char* e = getenv("JOXSZ_FIR_RUN")) { int v2 = atoi(e); if (v2 >= 8) runlen = v2; }
            std::vector<int> runs;
            for (int sg = 0; sg < rows.nseg; ++sg) {
                const int ra = rows.seg[3 * sg], cnt = rows.seg[3 * sg + 1], qa = rows.seg[3 * sg + 2];
                const int pieces = (cnt + runlen - 1) / runlen, len = (cnt + pieces - 1) / pieces;
                for (int t0 = 0; t0 < cnt; t0 += len) { runs.push_back(ra + t0); runs.push_back(std::min(len, cnt - t0)); runs.push_back(qa + t0); }
            }
            ctx->nrun = (int)runs.size() / 3;
            int* q;
            if ((rc = dev_put(ctx, runs.data(), runs.size(), &q))) return rc; ctx->d_runs = q;
            ctx->fir_reg = 0;
#define JX_HAS_O(Ov) if (o == Ov) ctx->fir_reg = 1;
            JX_FIR_REG_O(JX_HAS_O)
#undef JX_HAS_O
            if (const char* e = getenv("JOXSZ_FIR_LDS")) { if (atoi(e) > 0) ctx->fir_reg = 0; }
        }
        // x-symmetric rows: one real array per row between the passes (needs the mirror structure of d_mat, which makes
        // every map row symmetric about column S/2, and the register-window FIR)
        cv.xsym = (use_mirror && S <= 1024 && o < JX_XSYM_MAXT) ? 1 : 0;       // (beam widths without a register FIR: jx_beamfir_real_kernel)
        if (const char* e = getenv("JOXSZ_CONV_XSYM")) { if (atoi(e) == 0) cv.xsym = 0; }
        cv.fir_ld = cv.xsym ? ((cv.Ph + 15) & ~15) : 2 * cv.Ph;
        if (cv.xsym) {
            // Z[k] = (X[k] + conj X[LP-k]) + i w (X[k] - conj X[LP-k]),  w = e^{+2 pi i k/P},  X[k] = e^{-i phi_k} Rc[k]
            const int LPn = cv.LP, cS = S / 2;
            auto cis = [&](long long num) {                           // e^{2 pi i num / P}, argument reduced on the integers
                const long long j = ((num % P) + P) % P;
                return std::complex<double>(std::cos(2.0 * jxt::kPi * (double)j / P), std::sin(2.0 * jxt::kPi * (double)j / P));
            };
            std::vector<double> zab((size_t)LPn * 4);
            const std::complex<double> I(0.0, 1.0);
            for (int k = 0; k < LPn; ++k) {
                const std::complex<double> wk = cis(k);
                const std::complex<double> za = cis(-(long long)k * cS) * (1.0 + I * wk);
                const std::complex<double> zb = cis((long long)(LPn - k) * cS) * (1.0 - I * wk);
                zab[4 * k] = za.real(); zab[4 * k + 1] = za.imag(); zab[4 * k + 2] = zb.real(); zab[4 * k + 3] = zb.imag();
            }
            if ((rc = dev_put(ctx, zab.data(), zab.size(), &p))) return rc; cv.zab = (const cplx*)p;
            const int nt = o + 1;
            std::vector<double> bc((size_t)nt * JX_COL0_LD, 0.0);
            for (int t = 0; t < nt; ++t)
                for (int x = 0; x < nt; ++x) bc[(size_t)t * JX_COL0_LD + x] = c.step * c.step * beam_h[(size_t)(o + t) * B + o + x];
            if ((rc = dev_put(ctx, bc.data(), bc.size(), &p))) return rc; cv.bcol = p;
        }
        }   // even side
    }
    // ---- plain copies
    {
        double* p; int* q;
#define PUTD(field, id) { std::vector<double> v = host_vec<double>(ctx, id); if ((rc = dev_put(ctx, v.data(), v.size(), &p))) return rc; d.field = p; }
#define PUTI(field, id) { std::vector<int32_t> v = host_vec<int32_t>(ctx, id); if ((rc = dev_put(ctx, v.data(), v.size(), &q))) return rc; d.field = q; }
        {
            std::vector<double> lr = host_vec<double>(ctx, JX_T_R_PP);
            for (double& v : lr) v = std::log(v);
            if ((rc = dev_put(ctx, lr.data(), lr.size(), &p))) return rc;
            d.lr_pp = p;
            d.prep_pow = getenv("JOXSZ_PREP_POW") && atoi(getenv("JOXSZ_PREP_POW")) ? 1 : 0;
        }
        PUTD(r_pp, JX_T_R_PP) PUTD(d_mat, JX_T_D_MAT) PUTD(conv_T, JX_T_CONV_T) PUTD(conv_v, JX_T_CONV_V)
        PUTD(par_vals, JX_T_PAR_VALS) PUTD(par_min, JX_T_PAR_MIN) PUTD(par_max, JX_T_PAR_MAX)
        PUTD(par_mu, JX_T_PAR_MU) PUTD(par_sigma, JX_T_PAR_SIGMA)
        {   // -log(sqrt(2 pi)) - log(sigma) of the Gaussian priors, once
            std::vector<double> sg = host_vec<double>(ctx, JX_T_PAR_SIGMA);
            for (double& v : sg) v = (v > 0.0) ? -0.5 * std::log(2.0 * 3.14159265358979323846) - std::log(v) : 0.0;
            if ((rc = dev_put(ctx, sg.data(), sg.size(), &p))) return rc;
            d.par_lnorm = p;
        }
        PUTI(par_kind, JX_T_PAR_KIND) PUTI(thawed_idx, JX_T_THAWED_IDX)
        ctx->d_par_vals = const_cast<double*>(d.par_vals);
        if (!c.sz_only) {
            PUTD(x_r_ne, JX_T_X_R_NE) PUTD(x_r_T, JX_T_X_R_T) PUTD(projvols, JX_T_PROJVOLS) PUTD(cts, JX_T_CTS)
            PUTD(areascales, JX_T_AREASCALES) PUTD(exposures, JX_T_EXPOSURES) PUTD(backrates, JX_T_BACKRATES)
            PUTD(geomarea, JX_T_GEOMAREA) PUTD(lnT, JX_T_LNT) PUTD(lnrate, JX_T_LNRATE)
        }
#undef PUTD
#undef PUTI
    }

    // ---- symmetric-map tables: d_mat[iy][ix] = Q[|iy-c|][|ix-c|] (what centdistmat builds)
    {
        const int na = qn;
        const std::vector<double>& Q = Qtab;
        bool sym = dmat_mirror;
        if (const char* e = getenv("JOXSZ_GENERIC_MAP")) { if (atoi(e) > 0) sym = false; }
        if (na > 9 * 64) sym = false;                 // register-resident half row: |ix-c| < 576
        d.fast_map = sym ? 1 : 0;
        d.q_na = d.q_nb = na;
        // x-symmetric convolution + symmetric map kernel: only the quadrant of distinct pixels is ever stored
        d.quad = (((ctx->conv_mode == 2 && ctx->cv.xsym) || ctx->conv_mode == 3) && sym && na == S / 2 + 1) ? 1 : 0;
        if (const char* e = getenv("JOXSZ_FULL_MAP")) { if (atoi(e) > 0) d.quad = 0; }
        ctx->cv.quad = d.quad;
        if (sym) {
            std::vector<int32_t> qk((size_t)na * na);
            std::vector<double> qt((size_t)na * na);
            for (size_t e = 0; e < Q.size(); ++e) {
                const double dd = Q[e];
                if (!(dd <= r[N - 1])) { qk[e] = N; qt[e] = (dd != dd) ? dd : 0.0; }       // fill value 0 / NaN
                else if (dd < r[0]) { qk[e] = N - 1; qt[e] = dd; }                           // centre interval
                else {
                    int k = (int)(std::upper_bound(r.begin(), r.end(), dd) - r.begin()) - 1;
                    k = std::max(0, std::min(N - 2, k));
                    qk[e] = k; qt[e] = dd - r[k];
                }
            }
            int* qi; double* qd;
            if ((rc = dev_put(ctx, qk.data(), qk.size(), &qi))) return rc; d.q_k = qi;
            if ((rc = dev_put(ctx, qt.data(), qt.size(), &qd))) return rc; d.q_t = qd;
        }
    }

    // ---- chunk capacity and work buffers
    const size_t per_walker = (ctx->conv_mode == 3)
        ? sizeof(double) * ((size_t)d.q_nb * (d.q_na + 16) + (size_t)mixb.krows + (size_t)(JX_MIX_KSPLIT_MAX + 1) * 16 * mixb.ntile + 4 * (size_t)N + 64)
        : (ctx->conv_mode == 1)
        ? sizeof(double) * ((size_t)P * P * 2 + (size_t)P * ctx->Ph * 2 + (size_t)S * ctx->Sh * 2)
        : ctx->odd ? sizeof(double) * ((size_t)d.q_nb * (d.q_na + 16) + (size_t)ctx->Ph * (ctx->cv.NU + 4) + (size_t)(ctx->Ph + 2 * ctx->nrow + 300) * 64)
        : sizeof(double) * ((d.quad ? (size_t)d.q_nb * (d.q_na + 16) : (size_t)S * S) + (size_t)(ctx->cv.NU + ctx->cv.CROWS) * ctx->cv.fir_ld + (size_t)ctx->cv.NJ * 28 + (size_t)ctx->cv.nblk3 * ctx->Sh * 2);
    d.img_ld = (ctx->conv_mode == 1) ? P : S;
    d.img_ws = (ctx->conv_mode == 1) ? (long long)P * P : (long long)S * S;
    if (d.quad) { d.img_ld = (d.q_na + 16) & ~15; d.img_ws = (long long)d.q_nb * d.img_ld; }   // rows start on cache lines, >= 1 spare column
    int chunk = c.max_batch > 0 ? c.max_batch : 1024;   // >= 4 map blocks per CU: launches desynchronise, stores overlap compute
    if (const char* e = getenv("JOXSZ_CHUNK")) { int v = atoi(e); if (v > 0) chunk = v; }
    const size_t budget = (size_t)24 << 30;
    while (chunk > 1 && per_walker * chunk > budget) chunk /= 2;
    ctx->chunk = chunk;
    int split = c.map_split > 0 ? c.map_split : 1;
    if (const char* e = getenv("JOXSZ_MAP_SPLIT")) { int v = atoi(e); if (v > 0) split = v; }
    ctx->map_split = d.map_split = std::min(split, S);
    ctx->map_threads = 512;                     // two 8-wave blocks per CU measured best (profiles/r01_sweeps.md)
    if (const char* e = getenv("JOXSZ_MAP_THREADS")) { int v = atoi(e); if (v >= 64 && v <= 1024 && v % 64 == 0) ctx->map_threads = v; }

    {
        const size_t LDS_MAX = 160 * 1024;
        auto need = [&](int threads) {
            size_t scratch = (d.pairw == 2) ? JX_MAP_SCRATCH2_DOUBLES(N) : JX_MAP_SCRATCH_DOUBLES(N);
            if (d.fast_map) scratch = std::max(scratch, (size_t)(threads / 64) * ((S + 3) & ~1));
            const size_t dbl = JX_MAP_FIXED_DOUBLES(N) + scratch + (d.pairw == 2 ? 4 * JX_MAP_NE(N) + 8 : 0);
            return dbl * sizeof(double);
        };
        // quadrant map: two walkers per block share every (slot, abscissa) table entry (half the table traffic per walker),
        // as long as two such blocks still fit a CU
        d.pairw = 1;
        if (d.quad) {
            d.pairw = 2;
            if (const char* e = getenv("JOXSZ_MAP_PAIR")) { if (atoi(e) == 0) d.pairw = 1; }
            if (d.pairw == 2 && need(ctx->map_threads) > (LDS_MAX - 2048) / 2) d.pairw = 1;
        }
        while (ctx->map_threads > 64 && need(ctx->map_threads) > LDS_MAX - 1024) ctx->map_threads /= 2;
        ctx->map_lds_bytes = need(ctx->map_threads);
        if (ctx->map_lds_bytes > LDS_MAX - 1024) { ctx->err = "radial grid too long for the LDS-resident spline"; return JX_ERR_UNSUPPORTED; }
        const int lds = (int)ctx->map_lds_bytes;
#define JX_ATTR(V, NA) HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_abel_map_sym_kernel<V, NA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds))
        JX_ATTR(true, 3); JX_ATTR(true, 5); JX_ATTR(true, 9); JX_ATTR(false, 3); JX_ATTR(false, 5); JX_ATTR(false, 9);
#undef JX_ATTR
    }

    if ((rc = dev_new(ctx, (size_t)chunk, &ctx->d_base))) return rc;
    if ((rc = dev_new(ctx, (size_t)chunk * ctx->nrow, &ctx->d_cfac))) return rc;
    if (c.calc_integ && (rc = dev_new(ctx, (size_t)chunk, &ctx->d_sz0, true))) return rc;
    if (ctx->conv_mode == 3) {
        // ---- contracted route: spline arrays walker-minor -> stage 1 (rows mixed per column) -> stage 2 (matrix cores) -> tail
        if (!d.quad) { ctx->err = "contracted route needs the quadrant map tables"; return JX_ERR_UNSUPPORTED; }
        if ((rc = dev_new(ctx, (size_t)chunk * d.img_ws, &ctx->d_img, true))) return rc;         // y_2d tap only
        if ((rc = dev_new(ctx, (size_t)chunk * d.q_nb, &d.xcol))) return rc;
        const long long tW = ((long long)chunk + 127) & ~127LL;
        ctx->mix_tW = tW;
        ctx->mix_form = 0; ctx->mix_RT = mixb.RT; ctx->mix_nxt = mixb.nxt; ctx->mix_r = mixb.r; ctx->mix_ns = mixb.ns;
        ctx->lr.r = mixb.r;                                     // (jx_get_truncation reports it)
        JxMix& mx = ctx->mx;
        memset(&mx, 0, sizeof(mx));
        mx.NU = mixb.NU; mx.R = mixb.R; mx.tW = tW; mx.segld = mixb.cols.segld; mx.wld = mixb.cols.wld; mx.cld = mixb.cld;
        int* qi; double* qd;
        if ((rc = dev_put(ctx, mixb.cols.seg0.data(), mixb.cols.seg0.size(), &qi))) return rc; mx.seg0 = qi;
        if ((rc = dev_put(ctx, mixb.cols.nseg.data(), mixb.cols.nseg.size(), &qi))) return rc; mx.nseg = qi;
        if ((rc = dev_put(ctx, mixb.cols.seg.data(), mixb.cols.seg.size(), &qi))) return rc; mx.seg = qi;
        if ((rc = dev_put(ctx, mixb.cols.w4.data(), mixb.cols.w4.size(), &qd))) return rc; mx.w4 = qd;
        if ((rc = dev_put(ctx, mixb.Cm.data(), mixb.Cm.size(), &qd))) return rc; mx.Cm = qd;
        JxOpg& og = ctx->og;
        memset(&og, 0, sizeof(og));
        og.tW = tW; og.ntile = mixb.ntile; og.nog = mixb.nog; og.ldx = 16 * mixb.ntile;
        if ((rc = dev_put(ctx, mixb.Op.data(), mixb.Op.size(), &qd))) return rc; og.Op = qd;
        ctx->mix_ncol = 2 * N;
        if ((size_t)16 * (N + 2 * JX_MIX_NS + 2) * tW >= ((size_t)1 << 32)) { ctx->err = "contracted route: launch too large for 32-bit knot offsets (lower max_batch)"; return JX_ERR_UNSUPPORTED; }
        mx.cft_bytes = (unsigned)((size_t)16 * (N + 2 * JX_MIX_NS + 2) * tW);
        if ((rc = dev_new(ctx, (size_t)2 * (N + 2 * JX_MIX_NS + 2) * tW, &ctx->d_cft, true))) return rc;
        if ((rc = dev_new(ctx, mixb.krows * (size_t)tW, &ctx->d_Dt, true))) return rc;
        og.Dt = ctx->d_Dt;
        if ((rc = dev_new(ctx, (size_t)JX_MIX_KSPLIT_MAX * tW * og.ldx, &ctx->d_Pt))) return rc;
        ctx->mix_ksteps = mixb.ksteps;
        ctx->dct.cf_ws = (2 * ((long long)N + 2) + 15) & ~15LL;
        if ((rc = dev_new(ctx, (size_t)chunk * ctx->dct.cf_ws, &ctx->d_cf, true))) return rc;    // Abel kernel's walker-major copy (profile taps)
        if ((rc = setup_abel_gemm(ctx, chunk))) return rc;
    } else if (ctx->conv_mode == 1) {
        if ((rc = dev_new(ctx, (size_t)chunk * P * P, &ctx->d_img, true))) return rc;     // padding stays zero for ever
        if ((rc = dev_new(ctx, (size_t)chunk * P * P, &ctx->d_conv))) return rc;
        if ((rc = dev_new(ctx, (size_t)chunk * P * ctx->Ph, &ctx->d_spec))) return rc;
        if ((rc = dev_new(ctx, (size_t)chunk * S * ctx->Sh, &ctx->d_tfspec))) return rc;
    } else if (ctx->odd) {
        // ---- odd map side: pass 1 (rows from the spline, real-even transform) -> matrix products per column (FIR +
        //      job combination, walker-minor result) -> inverse real-even transform of the combined rows -> matrix products
        //      with the real-space circular kernels of the transfer function -> sum over the combined rows in the tail
        const JxConv& cv = ctx->cv;
        if (!d.quad) { ctx->err = "odd-side route needs the quadrant map tables"; return JX_ERR_UNSUPPORTED; }
        if ((rc = dev_new(ctx, (size_t)chunk * d.img_ws, &ctx->d_img, true))) return rc;         // y_2d tap only
        if ((rc = dev_new(ctx, (size_t)chunk * d.q_nb, &d.xcol))) return rc;
        const int r = oplan.r, RP = ((r + 15) / 16) * 16, LPo = cv.LP, nout = S / 2 + 1;
        // K ranges beyond the largest compiled k-step bucket are split in two halves (the second launch accumulates)
        const int nh1 = ((cv.NU + 3) / 4 > JX_LR_KS) ? 2 : 1, KU = (cv.NU + 4 * nh1 - 1) / (4 * nh1) * (4 * nh1), KUh = KU / nh1;
        int fb = 0;
#define JX_LR_PICK(K) if (!fb && KUh / 4 <= K) fb = K;
        JX_LR_BUCKETS(JX_LR_PICK)
#undef JX_LR_PICK
        const int nh2 = ((nout + 3) / 4 > JX_LR_KS) ? 2 : 1, KQ2 = (nout + 4 * nh2 - 1) / (4 * nh2) * (4 * nh2), ks2 = KQ2 / nh2 / 4;
        int fb2 = 0;
#define JX_LR_PICK(K) if (!fb2 && ks2 <= K) fb2 = K;
        JX_LR_BUCKETS(JX_LR_PICK)
#undef JX_LR_PICK
        if (!fb || !fb2) { ctx->err = "odd-side route: matrix sizes beyond the compiled k-step buckets"; return JX_ERR_UNSUPPORTED; }
        const size_t tW = (chunk + 15) & ~15;
        ctx->tW = (int)tW; ctx->tKU = KU; ctx->fused_bucket = fb; ctx->fused_nh = nh1; ctx->o_nh2 = nh2;
        ctx->o_RPc = RP; ctx->o_nout = nout; ctx->o_bucket2 = fb2; ctx->o_ldb = (nout + 15) & ~15;
        // first product: Wk [kact][RP][KU]
        {
            std::vector<double> Wk;
            jxt::fused_row_operator(oplan.L, r, oplan.rows, S, cv.o, oplan.taps.data(), oplan.kact, cv.Ph, RP, KU, Wk);
            double* p2;
            if ((rc = dev_put(ctx, Wk.data(), Wk.size(), &p2))) return rc;
            ctx->lrf.U = p2; ctx->lrf.r = r; ctx->lrf.ks = KUh / 4; ctx->lrf.KQ = KU; ctx->lrf.nq = cv.NU;
        }
        // second product: K [nmg][r][64][KQ2]
        {
            std::vector<double> Kp;
            jxt::odd_rowspace_operator(oplan.V, r, S, KQ2, Kp, &ctx->o_nmg);
            double* p2;
            if ((rc = dev_put(ctx, Kp.data(), Kp.size(), &p2))) return rc;
            ctx->d_Kp = p2;
            ctx->lr2.U = p2; ctx->lr2.r = 64; ctx->lr2.ks = ks2; ctx->lr2.KQ = KQ2; ctx->lr2.nq = nout;
        }
        const size_t slack1 = (size_t)4 * fb - KUh + 4;
        if ((rc = dev_new(ctx, ((size_t)cv.Ph * KU + slack1) * tW, &ctx->d_Rt, true))) return rc;
        if ((rc = dev_new(ctx, (size_t)cv.Ph * RP * tW, &ctx->d_Ctp, true))) return rc;
        if ((rc = dev_new(ctx, ((size_t)KQ2 - 4 * ks2 + 4 * fb2 + 8) * RP * tW, &ctx->d_cc, true))) return rc;
        if ((rc = dev_new(ctx, (size_t)tW * r * ctx->o_ldb, &ctx->d_D2, true))) return rc;
        // transforms: forward from the spline (row = distinct map row), inverse from the combined rows (row = rho)
        jxt::DctTables dt;
        if (!jxt::dct_tables(Qtab, qn, qn, r_grid, S, LPo, dt) || dt.amax + 1 > LPo - 4 ||
            (unsigned long long)LPo * std::max(KU, RP) * tW * 8ull >= (1ull << 32)) {
            ctx->err = "odd-side route: sizes outside the transform kernel's ranges"; return JX_ERR_UNSUPPORTED;
        }
        {
            JxDct& dc = ctx->dct;
            memset(&dc, 0, sizeof(dc));
            dc.NU = cv.NU; dc.kact = oplan.kact; dc.gl = dt.gl; dc.na4 = dt.na4; dc.has_x0 = 0; dc.N = N;
            dc.cf_ws = (2 * ((long long)N + 2) + 15) & ~15LL;
            dc.tW = (long long)tW; dc.tKU = KU;
            int* qi; double* qd;
            // (the generic instance of the kernel walks 256 entries per pass of its own pass count: pad the table to it)
            const int npass_k = ((LPo - 5) / 4 + 1 + 63) / 64, na4k = 256 * npass_k;
            std::vector<int> dk2((size_t)cv.NU * na4k, 0);
            std::vector<double> dw2((size_t)cv.NU * na4k * 4, 0.0);
            for (int u = 0; u < cv.NU; ++u)
                for (int a = 0; a < std::min(dt.na4, na4k); ++a) {
                    dk2[(size_t)u * na4k + a] = dt.dk[(size_t)u * dt.na4 + a];
                    for (int j = 0; j < 4; ++j) dw2[((size_t)u * na4k + a) * 4 + j] = dt.dw[((size_t)u * dt.na4 + a) * 4 + j];
                }
            dc.na4 = na4k;
            if ((rc = dev_put(ctx, dk2.data(), dk2.size(), &qi))) return rc; dc.dk = qi;
            if ((rc = dev_put(ctx, dw2.data(), dw2.size(), &qd))) return rc; dc.dw = qd;
            if ((rc = dev_put(ctx, dt.pk.data(), dt.pk.size(), &qd))) return rc; dc.pk = qd;
            std::vector<double> tq;
            jxt::twiddles(LPo / 2, LPo / 2, tq);
            if ((rc = dev_put(ctx, tq.data(), tq.size(), &qd))) return rc; dc.tw_q = (const cplx*)qd;
            if ((rc = dev_new(ctx, (size_t)chunk * dc.cf_ws, &ctx->d_cf, true))) return rc;
            if ((rc = setup_abel_gemm(ctx, chunk))) return rc;
            JxDct& d3 = ctx->dct3;
            d3 = dc;
            d3.NU = r; d3.kact = nout; d3.tKU = RP; d3.n_in = oplan.kact; d3.s_kstr = (long long)RP * (long long)tW;
            d3.dk = nullptr; d3.dw = nullptr;
        }
        bool have = false;
#define JX_DCTO_ATTR(LPv, NTv) if (LPo == LPv) { have = true; \
            ctx->dct_lds = sizeof(cplx) * ((size_t)16 * jx_dct_lay<LPv / 2, LPv - 4>::RS + LPv / 2) + sizeof(double) * (4 * (LPv / 4 + 1) + 16 * 64 + 16); \
            HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_rowdct_kernel<LPv, LPv - 4, 16, NTv, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->dct_lds)); \
            ctx->dct3_lds = sizeof(cplx) * ((size_t)16 * jx_dct_lay<LPv / 2, LPv + 1>::RS + LPv / 2) + sizeof(double) * (4 * (LPv / 4 + 1) + 16 * 64 + 16); \
            HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_rowdct_kernel<LPv, LPv + 1, 16, NTv, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->dct3_lds)); }
        JX_DCT_ODD_SIZES(JX_DCTO_ATTR)
#undef JX_DCTO_ATTR
        if (!have) { ctx->err = "odd-side route: no transform kernel for this padded length"; return JX_ERR_UNSUPPORTED; }
#define JX_LR_ATTR(K, T) HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_lowrank_kernel<K, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)JX_LR_LDS_MAX));
        JX_LR_KINDS(JX_LR_ATTR)
#undef JX_LR_ATTR
        ctx->dct_ok = true;
    } else {
        const JxConv& cv = ctx->cv;
        if ((rc = dev_new(ctx, (size_t)chunk * d.img_ws, &ctx->d_img, true))) return rc;
        if (d.quad && (rc = dev_new(ctx, (size_t)chunk * d.q_nb, &d.xcol))) return rc;
        // (counts in complex elements: fir_ld doubles per row in either mode)
        if ((rc = dev_new(ctx, ((size_t)chunk * cv.NU * cv.fir_ld + 1) / 2, &ctx->d_Y, true))) return rc;
        // (zeroed, with slack rows: the low-rank combination reads up to 4 JX_LR_KS rows from a walker's first row, against U = 0)
        if ((rc = dev_new(ctx, ((size_t)chunk * cv.CROWS * cv.fir_ld + 1) / 2 + (size_t)2 * JX_LR_KS * cv.fir_ld, &ctx->d_C, true))) return rc;
        if (cv.xsym && (rc = dev_new(ctx, (size_t)chunk * cv.NJ * (cv.o + 1) + 4 * JX_LR_KS, &ctx->cv.col0, true))) return rc;
        if ((rc = dev_new(ctx, (size_t)chunk * cv.nblk3 * cv.Sh, &ctx->d_part))) return rc;
        if (ctx->lr.r > 0) {
            if (ctx->lr_sep && (rc = dev_new(ctx, (size_t)chunk * ctx->lr.r * cv.fir_ld, &ctx->d_Clr))) return rc;
            if (ctx->lr_sep && cv.xsym && (rc = dev_new(ctx, (size_t)chunk * (cv.o + 1) * ctx->lr.r, &ctx->d_col0lr))) return rc;
            ctx->cv_lr = ctx->cv;                                  // cv is final here
            ctx->cv_lr.hy = ctx->lr_vt;
            ctx->cv_lr.col0 = ctx->d_col0lr;
            ctx->cv_lr.NJ = ctx->lr.r; ctx->cv_lr.CROWS = ctx->lr.r;
            ctx->cv_lr.nblk3 = (ctx->lr.r + ctx->p13_rows - 1) / ctx->p13_rows;
            // fused FIR + combination: needs the real row spectra and the quadrant map (walker-minor column-0 copy)
            bool fuse = cv.xsym && d.quad;
            if (const char* e = getenv("JOXSZ_FUSED")) { if (atoi(e) == 0) fuse = false; }
            const int r = ctx->lr.r, RP = ((r + 15) / 16) * 16, nt = cv.o + 1;
            // K = distinct rows: one GEMM pass while NU/4 fits the compiled k-step buckets, else two halves (second accumulates)
            int nh = ((cv.NU + 3) / 4 <= JX_LR_KS) ? 1 : 2;
            const int KUh = (((cv.NU + nh - 1) / nh) + 3) & ~3, KU = nh * KUh;
            int fb = 0;
#define JX_LR_PICK(K) if (!fb && KUh / 4 <= K) fb = K;
            JX_LR_BUCKETS(JX_LR_PICK)
#undef JX_LR_PICK
            if (fuse && fb && (size_t)(RP / 16) * fb * 64 * sizeof(double) <= JX_LR_LDS_MAX) {
                std::vector<double> Wk, V0, bc((size_t)nt * JX_COL0_LD, 0.0);
                // band limit of the beam: past the last column with a tap above band_tol (0.03 of the singular-value cut:
                // 3e-12 by default) of the largest one the combined rows are dropped like the small singular values are;
                // those columns are neither stored by pass 1 nor multiplied (Ct stays at its zero fill)
                int kact = cv.Ph;
                {
                    const double band_tol = 0.03 * ctx->lr_tol;
                    double tmax = 0.0;
                    for (double v2 : ctx->h_taps) tmax = std::max(tmax, std::fabs(v2));
                    while (kact > 1) {
                        double m = 0.0;
                        for (int t = 0; t < nt; ++t) m = std::max(m, std::fabs(ctx->h_taps[(size_t)t * cv.Ph + kact - 1]));
                        if (m > band_tol * tmax) break;
                        --kact;
                    }
                    if (const char* e = getenv("JOXSZ_BANDLIMIT")) { if (atoi(e) == 0) kact = cv.Ph; }
                }
                ctx->kact = kact;
                jxt::fused_row_operator(ctx->h_L, r, ctx->h_rows, S, cv.o, ctx->h_taps.data(), kact, cv.Ph, RP, KU, Wk);
                for (int t = 0; t < nt; ++t)
                    for (int x = 0; x < nt; ++x) bc[(size_t)t * JX_COL0_LD + x] = c.step * c.step * beam_h[(size_t)(cv.o + t) * B + cv.o + x];
                jxt::fused_row_operator(ctx->h_L, r, ctx->h_rows, S, cv.o, bc.data(), nt, JX_COL0_LD, RP, KU, V0);
                double* p2;
                if (nh == 2) {                                     // [b][RP][KU] -> [half][b][RP][KUh]
                    for (std::vector<double>* T : {&Wk, &V0}) {
                        const size_t nb = T->size() / ((size_t)RP * KU);
                        std::vector<double> H2(T->size());
                        for (int h = 0; h < 2; ++h)
                            for (size_t b = 0; b < nb; ++b)
                                for (int rho = 0; rho < RP; ++rho)
                                    memcpy(&H2[(((size_t)h * nb + b) * RP + rho) * KUh], &(*T)[((size_t)b * RP + rho) * KU + (size_t)h * KUh], sizeof(double) * KUh);
                        T->swap(H2);
                    }
                }
                if ((rc = dev_put(ctx, Wk.data(), Wk.size(), &p2))) return rc;
                ctx->lrf.U = p2; ctx->lrf.r = r; ctx->lrf.ks = KUh / 4; ctx->lrf.KQ = KUh; ctx->lrf.nq = cv.NU;
                if ((rc = dev_put(ctx, V0.data(), V0.size(), &p2))) return rc;
                ctx->lrf0 = ctx->lrf; ctx->lrf0.U = p2;
                ctx->fused_bucket = fb; ctx->tKU = KU; ctx->tW = (chunk + 15) & ~15; ctx->fused_nh = nh;
                const size_t tW = ctx->tW, slack_rows = (size_t)4 * fb - KUh + 4;
                if ((rc = dev_new(ctx, ((size_t)cv.Ph * KU + slack_rows) * tW, &ctx->d_Rt, true))) return rc;
                if ((rc = dev_new(ctx, tW * cv.Ph * 64 + 64, &ctx->d_Ct, true))) return rc;
                if ((rc = dev_new(ctx, tW * JX_CT0_X * 64 + 64, &ctx->d_Ct0, true))) return rc;
                if ((rc = dev_new(ctx, ((size_t)KU + slack_rows) * tW, &ctx->d_x0t, true))) return rc;
                ctx->cv_f = ctx->cv_lr;
                ctx->cv_f.tmode = 1; ctx->cv_f.tW = ctx->tW; ctx->cv_f.tKU = KU; ctx->cv_f.ct0 = ctx->d_Ct0; ctx->cv_f.kact = kact;
            }
            ctx->h_L.clear(); ctx->h_taps.clear();
            // pass 1 straight from the spline coefficients: needs the fused route (walker-minor rows) and the quadrant table
            bool want_dct = ctx->lrf.r > 0 && d.quad;
            if (const char* e = getenv("JOXSZ_DCT")) { if (atoi(e) == 0) want_dct = false; }
            if (want_dct) {
                jxt::DctTables dt;
                bool have_kernel = false;
#define JX_DCT_HAS(LPv, NSv, NTv, NWv) if (cv.LP == LPv && cv.LS == NSv && S % 2 == 0) { have_kernel = true; ctx->dct_nw = NWv; }
                JX_DCT_SIZES(JX_DCT_HAS)
#undef JX_DCT_HAS
                if (have_kernel && jxt::dct_tables(Qtab, qn, qn, r_grid, S, cv.LP, dt) && dt.amax + 1 == cv.LS &&
                    (unsigned long long)cv.LP * ctx->tKU * ctx->tW * 8ull < (1ull << 32)) {
                    JxDct& dc = ctx->dct;
                    memset(&dc, 0, sizeof(dc));
                    dc.NU = cv.NU; dc.kact = ctx->kact; dc.gl = dt.gl; dc.na4 = dt.na4; dc.has_x0 = dt.has_x0; dc.N = N;
                    dc.cf_ws = (2 * ((long long)N + 2) + 15) & ~15LL;          // (y, M) pairs + two zero entries behind the last knot
                    dc.tW = ctx->tW; dc.tKU = ctx->tKU;
                    int* qi; double* qd;
                    if ((rc = dev_put(ctx, dt.dk.data(), dt.dk.size(), &qi))) return rc; dc.dk = qi;
                    if ((rc = dev_put(ctx, dt.dw.data(), dt.dw.size(), &qd))) return rc; dc.dw = qd;
                    if ((rc = dev_put(ctx, dt.x0k.data(), dt.x0k.size(), &qi))) return rc; dc.x0k = qi;
                    if ((rc = dev_put(ctx, dt.x0w.data(), dt.x0w.size(), &qd))) return rc; dc.x0w = qd;
                    if ((rc = dev_put(ctx, dt.pk.data(), dt.pk.size(), &qd))) return rc; dc.pk = qd;
                    std::vector<double> tq;
                    jxt::twiddles(cv.LP / 2, cv.LP / 2, tq);
                    if ((rc = dev_put(ctx, tq.data(), tq.size(), &qd))) return rc; dc.tw_q = (const cplx*)qd;
                    if ((rc = dev_new(ctx, (size_t)chunk * dc.cf_ws, &ctx->d_cf, true))) return rc;
            if ((rc = setup_abel_gemm(ctx, chunk))) return rc;
#define JX_DCT_ATTR(LPv, NSv, NTv, NWv) if (cv.LP == LPv && cv.LS == NSv) { \
                        ctx->dct_lds = dct_lds_bytes<LPv, NSv>(NWv, sizeof(double)); \
                        HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_rowdct_kernel<LPv, NSv, NWv, NTv, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
                    JX_DCT_SIZES(JX_DCT_ATTR)
#undef JX_DCT_ATTR
                    ctx->dct_ok = true;
                }
            }
        }
        if (c.dtype == 1) {
            const int ntr = (ctx->lrf.r + 15) / 16;
            if (!ctx->dct_ok || ntr < 2) {
                ctx->err = "dtype f32 is available on the default route of even map sides only (fused matrix products, rank >= 17)";
                return JX_ERR_UNSUPPORTED;
            }
            if (!ctx->abel_gemm) { ctx->err = "dtype f32 takes its spline arrays from the matrix product (JOXSZ_ABEL_GEMM=0 is the f64 build's switch)"; return JX_ERR_UNSUPPORTED; }
            if ((rc = dev_new(ctx, (size_t)chunk * ctx->dct.cf_ws, &ctx->d_cf_tap, true))) return rc;
            ctx->f32 = true;
#define JX_DCT_ATTR(LPv, NSv, NTv, NWv) if (cv.LP == LPv && cv.LS == NSv) { \
            ctx->dct_lds = dct_lds_bytes<LPv, NSv>(NWv, sizeof(float)); \
            HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_rowdct_kernel<LPv, NSv, NWv, NTv, 0, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->dct_lds)); }
            JX_DCT_SIZES(JX_DCT_ATTR)
#undef JX_DCT_ATTR
#define JX_LR_ATTR(K, T) HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_lowrank_kernel<K, T, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)JX_LR_LDS_MAX));
            JX_LR_KINDS_F32(JX_LR_ATTR)
#undef JX_LR_ATTR
        }
#define JX_ATTR2(LPv, LSv, R1v, R3v) if (cv.LP == LPv && cv.LS == LSv) { \
            constexpr int rs1 = jx_lay<LPv>::RS, rs3 = jx_lay<LPv>::RS > jx_lay<LSv>::RS ? jx_lay<LPv>::RS : jx_lay<LSv>::RS; \
            ctx->p1_lds = sizeof(cplx) * ((size_t)R1v * rs1 + 2 * LPv + 2) + sizeof(double) * R1v; \
            ctx->p3_lds = sizeof(cplx) * ((size_t)R3v * rs3 + LPv + LSv) + (cv.xsym ? sizeof(double) * R3v * (cv.o + 1) : 0); \
            HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_rowfft2_kernel<LPv, R1v>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->p1_lds)); \
            HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_rowtf2_kernel<LPv, LSv, R3v>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->p3_lds)); \
            HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_rowtf2_kernel<LPv, LSv, R3v, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->p3_lds)); }
        JX_CONV2_PAIRS(JX_ATTR2)
#undef JX_ATTR2
        HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_beamfir_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->p2_lds));
#define JX_LR_ATTR(K, T) HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_lowrank_kernel<K, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)JX_LR_LDS_MAX));
        JX_LR_KINDS(JX_LR_ATTR)
#undef JX_LR_ATTR
    }

    FFTCHK(ctx, rocfft_execution_info_create(&ctx->info));
    FFTCHK(ctx, rocfft_execution_info_set_stream(ctx->info, ctx->stream));
    // the zero fills above ran on the null stream, which the context's non-blocking stream does not wait for
    HIPCHK(ctx, hipDeviceSynchronize());
    if (c.dtype == 1 && !ctx->f32) {                             // never silently fall back to the fp64 arithmetic
        ctx->err = "dtype f32 is available on the default route of even map sides only (hand-written convolution, fused matrix products)";
        return JX_ERR_UNSUPPORTED;
    }
    ctx->finalized = true;
    return JX_OK;
}

static int ensure_batch(jx_ctx* ctx, int n) {
    if (n <= ctx->batch_cap) return JX_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_theta) { (void)hipFree(ctx->d_theta); (void)hipFree(ctx->d_logp); }
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_theta, sizeof(double) * (size_t)n * ctx->cfg.ndim));
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_logp, sizeof(double) * (size_t)n));
    ctx->batch_cap = n;
    return JX_OK;
}

static int get_evset(jx_ctx* ctx, EvSet* out) {
    if (!ctx->ev_free.empty()) { *out = ctx->ev_free.back(); ctx->ev_free.pop_back(); return JX_OK; }
    for (int k = 0; k < 7; ++k) HIPCHK(ctx, hipEventCreate(&out->e[k]));
    return JX_OK;
}

static int drain_events(jx_ctx* ctx) {
    if (ctx->ev_inflight.empty()) return JX_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& es : ctx->ev_inflight) {
        float ms[5] = {0, 0, 0, 0, 0}, tot;
        if (es.p1only) {                               // two events per launch sequence: the time-dominant kernel alone
            HIPCHK(ctx, hipEventElapsedTime(&ms[2], es.e[2], es.e[3]));
            ctx->acc.beam_fft_ms += ms[2];
            ctx->acc.launches += 1; ctx->acc.walkers += es.walkers;
            ctx->ev_free.push_back(es);
            continue;
        }
        if (es.op) {                                   // collapsed route: prep, then one kernel
            HIPCHK(ctx, hipEventElapsedTime(&ms[0], es.e[0], es.e[1]));
            HIPCHK(ctx, hipEventElapsedTime(&ms[4], es.e[1], es.e[5]));
        } else {
            for (int k = 0; k < 5; ++k) HIPCHK(ctx, hipEventElapsedTime(&ms[k], es.e[k], es.e[k + 1]));
        }
        HIPCHK(ctx, hipEventElapsedTime(&tot, es.e[0], es.e[5]));
        if (es.gemm && !es.op) { float g; HIPCHK(ctx, hipEventElapsedTime(&g, es.e[3], es.e[6])); ctx->acc.gemm_ms += g; }
        ctx->acc.prep_ms += ms[0]; ctx->acc.abel_map_ms += ms[1]; ctx->acc.beam_fft_ms += ms[2];
        ctx->acc.tf_fft_ms += ms[3]; ctx->acc.tail_ms += ms[4]; ctx->acc.total_ms += tot;
        ctx->acc.launches += 1; ctx->acc.walkers += es.walkers;
        ctx->ev_free.push_back(es);
    }
    ctx->ev_inflight.clear();
    return JX_OK;
}

struct Taps {
    double *pp = nullptr, *ab = nullptr, *y = nullptr, *row = nullptr, *bright = nullptr, *chisq = nullptr,
           *tprof = nullptr, *xprofs = nullptr, *parts = nullptr, *conv = nullptr, *integ = nullptr;
    bool need_img = false;            // the Compton-y map itself is wanted (y_2d tap, work-buffer hook): map kernel + pass 1 from the image
};


// true when this launch takes the fused route (decided before the map kernel: it changes where column 0 is copied to)
static bool use_fused(const jx_ctx* ctx, const double* tap_convjobs) {
    return ctx->conv_mode == 2 && ctx->lrf.r > 0 && !tap_convjobs;
}

// map -> pass 1 (walker-minor rows) -> one GEMM per column (FIR + job combination) -> pass 3 over the combined rows
static int launch_fused_conv(jx_ctx* ctx, int n, EvSet* es, bool dct) {
    const JxConv& cv = ctx->cv;
    const JxDev& d = ctx->d;
    hipStream_t st = ctx->stream;
    JxConv cf = ctx->cv_f;
    cf.tn = n;
    bool done = false;
    if (dct) {
        JxDct dc = ctx->dct;
        dc.n = n;
        const int ngroups = (n + ctx->dct_nw - 1) / ctx->dct_nw, gp8 = (ngroups + 7) / 8;
        // row classes: enough blocks to fill the device a few times over, at least ~8 rows per block when there are many
        int nrc = std::max(1, std::min(dc.NU, (8 * ctx->num_cu + 8 * gp8 - 1) / (8 * gp8)));
        if (const char* e = getenv("JOXSZ_DCT_NRC")) { int v = atoi(e); if (v > 0) nrc = std::min(v, dc.NU); }
        dc.nrc = nrc;
        if (const char* e = getenv("JOXSZ_DCT_DBG")) dc.dbg = atoi(e);
        const dim3 gd((unsigned)(8 * gp8 * nrc));
        static unsigned long long* stamp_buf = nullptr;
        if (getenv("JOXSZ_DCT_STAMPS")) {
            if (!stamp_buf) HIPCHK(ctx, hipMalloc((void**)&stamp_buf, sizeof(unsigned long long) * 8 * 65536));
            HIPCHK(ctx, hipMemsetAsync(stamp_buf, 0, sizeof(unsigned long long) * 8 * gd.x, st));
            if (gd.x <= 65536) dc.stamps = stamp_buf;
        }
        size_t dlds = ctx->dct_lds;
        if (const char* e = getenv("JOXSZ_DCT_LDS_KB")) dlds = std::max(dlds, (size_t)atoi(e) * 1024);     // occupancy experiments
#define JX_DCT_GO(LPv, NSv, NTv, NWv) if (!done && cv.LP == LPv && cv.LS == NSv) { \
            if (ctx->f32) hipLaunchKernelGGL((jx_rowdct_kernel<LPv, NSv, NWv, NTv, 0, float>), gd, dim3(NTv), dlds, st, dc, ctx->d_cf, ctx->d_Rt, ctx->d_x0t); \
            else hipLaunchKernelGGL((jx_rowdct_kernel<LPv, NSv, NWv, NTv, 0>), gd, dim3(NTv), dlds, st, dc, ctx->d_cf, ctx->d_Rt, ctx->d_x0t); \
            done = true; }
        JX_DCT_SIZES(JX_DCT_GO)
#undef JX_DCT_GO
        if (!done) { ctx->err = "no coefficient-fed pass-1 kernel for this size"; return JX_ERR_UNSUPPORTED; }
        if (dc.stamps) {                                        // diagnostic: mean cycles per phase of wave 0, over the blocks
            std::vector<unsigned long long> h((size_t)gd.x * 8);
            HIPCHK(ctx, hipStreamSynchronize(st));
            HIPCHK(ctx, hipMemcpy(h.data(), dc.stamps, h.size() * 8, hipMemcpyDeviceToHost));
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (unsigned b = 0; b < gd.x; ++b) for (int i = 0; i < 8; ++i) acc[i] += (double)h[(size_t)b * 8 + i];
            const double rows = (double)dc.NU * ngroups;
            fprintf(stderr, "[dct stamps] cycles per (row, 16 walkers), wave 0: E %.0f | x0+barrier %.0f | stepA %.0f | barrier+stepB %.0f | barrier+post %.0f | barrier %.0f\n",
                    acc[0] / rows, acc[1] / rows, acc[2] / rows, acc[3] / rows, acc[4] / rows, acc[5] / rows);
            fprintf(stderr, "[dct stamps]   inside E: evaluation %.0f | barrier %.0f | (z build = E - these)\n", acc[6] / rows, acc[7] / rows);
        }
    }
    const dim3 g1(cv.NU, (n + ctx->p1_rows - 1) / ctx->p1_rows);
#define JX_P1(LPv, LSv, R1v, R3v) if (!done && cv.LP == LPv && cv.LS == LSv) { \
        hipLaunchKernelGGL((jx_rowfft2_kernel<LPv, R1v>), g1, dim3(256), ctx->p1_lds, st, cf, ctx->d_img, (size_t)d.img_ld, (size_t)d.img_ws, \
                           reinterpret_cast<cplx*>(ctx->d_Rt)); done = true; }
    JX_CONV2_PAIRS(JX_P1)
#undef JX_P1
    if (!done) { ctx->err = "no pass-1 kernel for this size"; return JX_ERR_UNSUPPORTED; }
    if (es) HIPCHK(ctx, hipEventRecord(es->e[3], st));
    {
        const JxLowrank& lr = ctx->lrf;
        const int ntr = (lr.r + 15) / 16, ncols = (n + 15) & ~15, ntile = ncols / 16;
        const size_t lds = (size_t)ntr * ctx->fused_bucket * 64 * sizeof(double);
        const long long tW = ctx->tW, KU = ctx->tKU, RP = 16LL * ntr, nt = cv.o + 1;
        (void)ntile;
        // one launch: the kx batches of the row spectra, then the output-column batches of the column-0 terms
        // (K split in halves for large maps: the second launch adds its half of the distinct rows to the first's results)
        const long long KUh = KU / ctx->fused_nh;
        for (int h = 0; h < ctx->fused_nh; ++h) {
            // (operand offsets in bytes: the elements are floats in the fp32 variant)
            const size_t esz = ctx->f32 ? sizeof(float) : sizeof(double);
            const double* Bh = reinterpret_cast<const double*>(reinterpret_cast<const char*>(ctx->d_Rt) + (size_t)h * KUh * tW * esz);
            const double* B0h = reinterpret_cast<const double*>(reinterpret_cast<const char*>(ctx->d_x0t) + (size_t)h * KUh * tW * esz);
            const JxGemmSeg s0{lr.U + (size_t)h * ctx->kact * RP * KUh, Bh, ctx->d_Ct, RP * KUh, KU * tW, 64LL,
                               (long long)cv.Ph * 64, ctx->kact, h};
            const JxGemmSeg s1{ctx->lrf0.U + (size_t)h * nt * RP * KUh, B0h, ctx->d_Ct0, RP * KUh, 0LL, 64LL,
                               (long long)JX_CT0_X * 64, (int)nt, h};
#define JX_LR_GO(K, T) if (ctx->fused_bucket == K && ntr == T && !ctx->f32) \
            hipLaunchKernelGGL((jx_lowrank_kernel<K, T>), dim3(ctx->num_cu), dim3(512), lds, st, lr, s0, s1, 0LL, tW, 1LL, 0LL, 1LL, ncols, 1);
            JX_LR_KINDS(JX_LR_GO)
#undef JX_LR_GO
#define JX_LR_GO(K, T) if (ctx->fused_bucket == K && ntr == T && ctx->f32) \
            hipLaunchKernelGGL((jx_lowrank_kernel<K, T, float>), dim3(ctx->num_cu), dim3(512), lds, st, lr, s0, s1, 0LL, tW, 1LL, 0LL, 1LL, ncols, 1);
            JX_LR_KINDS_F32(JX_LR_GO)
#undef JX_LR_GO
        }
    }
    if (es && !es->p1only) { HIPCHK(ctx, hipEventRecord(es->e[6], st)); es->gemm = true; }
    ctx->last_nblk3 = cf.nblk3;
    done = false;
    const dim3 g3(cf.nblk3, n);
#define JX_P3(LPv, LSv, R1v, R3v) if (!done && cv.LP == LPv && cv.LS == LSv) { \
        if (ctx->f32) hipLaunchKernelGGL((jx_rowtf2_kernel<LPv, LSv, R3v, float>), g3, dim3(256), ctx->p3_lds, st, cf, reinterpret_cast<const cplx*>(ctx->d_Ct), \
                           ctx->d_part, (double*)nullptr); \
        else hipLaunchKernelGGL((jx_rowtf2_kernel<LPv, LSv, R3v>), g3, dim3(256), ctx->p3_lds, st, cf, reinterpret_cast<const cplx*>(ctx->d_Ct), \
                           ctx->d_part, (double*)nullptr); \
        done = true; }
    JX_CONV2_PAIRS(JX_P3)
#undef JX_P3
    if (es && !es->p1only) HIPCHK(ctx, hipEventRecord(es->e[4], st));
    return JX_OK;
}

// Odd map side: pass 1 -> matrix products per column -> inverse transform of the combined rows -> matrix products with
// the real-space kernels of the transfer function.  The tail (jx_tail_odd_kernel) sums the partial rows over rho.
static int launch_odd_conv(jx_ctx* ctx, int n, EvSet* es) {
    const JxConv& cv = ctx->cv;
    hipStream_t st = ctx->stream;
    const int ngroups = (n + 15) / 16, gp8 = (ngroups + 7) / 8, ncols = (n + 15) & ~15;
    const long long tW = ctx->tW, KU = ctx->tKU, RP = ctx->o_RPc;
    const JxGemmSeg none{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0};
    bool done = false;
    {   // pass 1: rows from the spline, real-even transform
        JxDct dc = ctx->dct;
        dc.n = n;
        dc.nrc = std::max(1, std::min(dc.NU, (8 * ctx->num_cu + 8 * gp8 - 1) / (8 * gp8)));
        const dim3 gd((unsigned)(8 * gp8 * dc.nrc));
#define JX_DCT_GO(LPv, NTv) if (!done && cv.LP == LPv) { \
            hipLaunchKernelGGL((jx_rowdct_kernel<LPv, LPv - 4, 16, NTv, 0>), gd, dim3(NTv), ctx->dct_lds, st, dc, ctx->d_cf, ctx->d_Rt, (double*)nullptr); done = true; }
        JX_DCT_ODD_SIZES(JX_DCT_GO)
#undef JX_DCT_GO
        if (!done) { ctx->err = "no pass-1 kernel for this odd size"; return JX_ERR_UNSUPPORTED; }
    }
    if (es) HIPCHK(ctx, hipEventRecord(es->e[3], st));
    // FIR + job combination: Ctp[kx][rho][w] = sum_u W_kx[rho][u] Rt[kx][u][w], 64 rows rho per launch
    for (int g0 = 0; g0 < ctx->lrf.r; g0 += 64)
        for (int h = 0; h < ctx->fused_nh; ++h) {                 // (K halves of large maps: the second launch adds to the first's results)
            JxLowrank lr = ctx->lrf;
            lr.r = std::min(64, ctx->lrf.r - g0);
            const int ntr = (lr.r + 15) / 16;
            const size_t lds = (size_t)ntr * ctx->fused_bucket * 64 * sizeof(double);
            const long long Kh = 4LL * lr.ks;
            const JxGemmSeg s0{ctx->lrf.U + (size_t)g0 * KU + (size_t)h * Kh, ctx->d_Rt + (size_t)h * Kh * tW, ctx->d_Ctp + (size_t)g0 * tW,
                               RP * KU, KU * tW, RP * tW, 1LL, ctx->kact, h};
#define JX_LR_GO(K, T) if (ctx->fused_bucket == K && ntr == T) \
            hipLaunchKernelGGL((jx_lowrank_kernel<K, T>), dim3(ctx->num_cu), dim3(512), lds, st, lr, s0, none, 0LL, tW, 1LL, 0LL, tW, ncols, 1);
            JX_LR_KINDS(JX_LR_GO)
#undef JX_LR_GO
        }
    if (es && !es->p1only) { HIPCHK(ctx, hipEventRecord(es->e[6], st)); es->gemm = true; }
    {   // combined rows back to real space: cc[a][rho][w], a = 0..S/2 (offset from the centre column)
        JxDct d3 = ctx->dct3;
        d3.n = n;
        d3.nrc = std::max(1, std::min(d3.NU, (4 * ctx->num_cu + 8 * gp8 - 1) / (8 * gp8)));
        const dim3 gd((unsigned)(8 * gp8 * d3.nrc));
        done = false;
#define JX_DCT_GO(LPv, NTv) if (!done && cv.LP == LPv) { \
            hipLaunchKernelGGL((jx_rowdct_kernel<LPv, LPv + 1, 16, NTv, 1>), gd, dim3(NTv), ctx->dct3_lds, st, d3, ctx->d_Ctp, ctx->d_cc, (double*)nullptr); done = true; }
        JX_DCT_ODD_SIZES(JX_DCT_GO)
#undef JX_DCT_GO
    }
    {   // D2[w][rho][b] = sum_a K[rho][b][a] cc[a][rho][w], 64 rows b per launch
        const int r = ctx->lrf.r, nout = ctx->o_nout;
        const long long ldb = ctx->o_ldb, KQ2 = ctx->lr2.KQ;
        for (int mg = 0; mg < ctx->o_nmg; ++mg)
            for (int h = 0; h < ctx->o_nh2; ++h) {
                JxLowrank lr = ctx->lr2;
                lr.r = std::min(64, nout - 64 * mg);
                const int ntr = (lr.r + 15) / 16;
                const size_t lds = (size_t)ntr * ctx->o_bucket2 * 64 * sizeof(double);
                const long long Kh = 4LL * lr.ks;
                const JxGemmSeg s0{ctx->d_Kp + (size_t)mg * r * 64 * KQ2 + (size_t)h * Kh, ctx->d_cc + (size_t)h * Kh * RP * tW, ctx->d_D2 + 64 * mg,
                                   64 * KQ2, tW, ldb, (long long)r * ldb, r, h};
#define JX_LR_GO(K, T) if (ctx->o_bucket2 == K && ntr == T) \
                hipLaunchKernelGGL((jx_lowrank_kernel<K, T>), dim3(ctx->num_cu), dim3(512), lds, st, lr, s0, none, 0LL, RP * tW, 1LL, 0LL, 1LL, ncols, 1);
                JX_LR_KINDS(JX_LR_GO)
#undef JX_LR_GO
            }
    }
    if (es && !es->p1only) HIPCHK(ctx, hipEventRecord(es->e[4], st));
    return JX_OK;
}

// Beam-convolved map (joxsz_funcs.py:464) on the odd-side route, for the parity tap only: the FIR along rows without the
// job combination (identity in place of U), every job's row back to real space, then mirrored out to S x S.
static int launch_odd_conv_tap(jx_ctx* ctx, int n) {
    const JxConv& cv = ctx->cv;
    if (ctx->fused_nh > 1) { ctx->err = "conv_2d tap: not built for map sides whose matrix products run in two K halves (use conv = rocfft for this tap)"; return JX_ERR_UNSUPPORTED; }
    hipStream_t st = ctx->stream;
    const int NJ = cv.NJ, RPj = ((NJ + 15) / 16) * 16, S = cv.S, nout = S / 2 + 1;
    const long long tW = ctx->tW, KU = ctx->tKU;
    int rc;
    if (!ctx->d_Wfir) {
        std::vector<double> I((size_t)NJ * NJ, 0.0), Wf;
        for (int q = 0; q < NJ; ++q) I[(size_t)q * NJ + q] = 1.0;
        jxt::fused_row_operator(I, NJ, ctx->h_rows, S, cv.o, ctx->h_taps.data(), ctx->kact, cv.Ph, RPj, (int)KU, Wf);
        double* p2;
        if ((rc = dev_put(ctx, Wf.data(), Wf.size(), &p2))) return rc;
        ctx->d_Wfir = p2;
        ctx->o_RPj = RPj;
        if ((rc = dev_new(ctx, (size_t)cv.Ph * RPj * tW, &ctx->d_Ctj, true))) return rc;
        if ((rc = dev_new(ctx, (size_t)(nout + 8) * RPj * tW, &ctx->d_ccj, true))) return rc;
        if ((unsigned long long)cv.LP * RPj * tW * 8ull >= (1ull << 32)) { ctx->err = "conv_2d tap: launch too large (lower max_batch)"; return JX_ERR_UNSUPPORTED; }
    }
    const int ncols = (n + 15) & ~15, ngroups = (n + 15) / 16, gp8 = (ngroups + 7) / 8;
    const JxGemmSeg none{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0};
    for (int g0 = 0; g0 < NJ; g0 += 64) {
        JxLowrank lr = ctx->lrf;
        lr.r = std::min(64, NJ - g0);
        const int ntr = (lr.r + 15) / 16;
        const size_t lds = (size_t)ntr * ctx->fused_bucket * 64 * sizeof(double);
        const JxGemmSeg s0{ctx->d_Wfir + (size_t)g0 * KU, ctx->d_Rt, ctx->d_Ctj + (size_t)g0 * tW, (long long)RPj * KU, KU * tW, (long long)RPj * tW, 1LL, ctx->kact, 0};
#define JX_LR_GO(K, T) if (ctx->fused_bucket == K && ntr == T) \
        hipLaunchKernelGGL((jx_lowrank_kernel<K, T>), dim3(ctx->num_cu), dim3(512), lds, st, lr, s0, none, 0LL, tW, 1LL, 0LL, tW, ncols, 1);
        JX_LR_KINDS(JX_LR_GO)
#undef JX_LR_GO
    }
    JxDct d3 = ctx->dct3;
    d3.n = n; d3.NU = NJ; d3.tKU = RPj; d3.s_kstr = (long long)RPj * tW;
    d3.nrc = std::max(1, std::min(d3.NU, (4 * ctx->num_cu + 8 * gp8 - 1) / (8 * gp8)));
    const dim3 gd((unsigned)(8 * gp8 * d3.nrc));
    bool done = false;
#define JX_DCT_GO(LPv, NTv) if (!done && cv.LP == LPv) { \
        hipLaunchKernelGGL((jx_rowdct_kernel<LPv, LPv + 1, 16, NTv, 1>), gd, dim3(NTv), ctx->dct3_lds, st, d3, ctx->d_Ctj, ctx->d_ccj, (double*)nullptr); done = true; }
    JX_DCT_ODD_SIZES(JX_DCT_GO)
#undef JX_DCT_GO
    hipLaunchKernelGGL(jx_expand_odd_conv_kernel, dim3(S, n), dim3(256), 0, st, ctx->d_ccj, ctx->d_rowjob, S, RPj, (long long)tW, ctx->t_conv);
    return JX_OK;
}

static int launch_custom_conv(jx_ctx* ctx, int n, double* tap_convjobs, EvSet* es) {
    const JxConv& cv = ctx->cv;
    const JxDev& d = ctx->d;
    hipStream_t st = ctx->stream;
    bool done = false;
    const dim3 g1((cv.NU + ctx->p1_rows - 1) / ctx->p1_rows, n);
#define JX_P1(LPv, LSv, R1v, R3v) if (!done && cv.LP == LPv && cv.LS == LSv) { \
        hipLaunchKernelGGL((jx_rowfft2_kernel<LPv, R1v>), g1, dim3(256), ctx->p1_lds, st, cv, ctx->d_img, (size_t)d.img_ld, (size_t)d.img_ws, ctx->d_Y); done = true; }
    JX_CONV2_PAIRS(JX_P1)
#undef JX_P1
    if (!done) { ctx->err = "no pass-1 kernel for this size"; return JX_ERR_UNSUPPORTED; }
    if (cv.xsym)
        hipLaunchKernelGGL(jx_col0_kernel, dim3(n, (cv.o + JX_COL0_XG) / JX_COL0_XG), dim3(256), 0, st, cv, ctx->d_img, (size_t)d.img_ld, (size_t)d.img_ws, d.xcol, cv.col0);
    if (ctx->fir_reg) {
        const int nslab = (cv.fir_ld + 63) / 64, units = nslab * n;
        const dim3 g2((unsigned)(((units + 7) / 8) * 8 * ctx->nrun));
#define JX_FIRREG(Ov) if (cv.o == Ov) hipLaunchKernelGGL((jx_beamfir_reg_kernel<Ov>), g2, dim3(64), 0, st, cv, ctx->d_runs, ctx->nrun, n, ctx->d_Y, ctx->d_C);
        JX_FIR_REG_O(JX_FIRREG)
#undef JX_FIRREG
    } else if (cv.xsym) {
        hipLaunchKernelGGL(jx_beamfir_real_kernel, dim3(cv.NJ, n), dim3(256), 0, st, cv, reinterpret_cast<const double*>(ctx->d_Y),
                           reinterpret_cast<double*>(ctx->d_C));
    } else {
        const dim3 g2((cv.Ph + JX_FIR_KX - 1) / JX_FIR_KX, n);
        hipLaunchKernelGGL(jx_beamfir_kernel, g2, dim3(256), ctx->p2_lds, st, cv, ctx->d_Y, ctx->d_C);
    }
    if (es) HIPCHK(ctx, hipEventRecord(es->e[3], st));
    done = false;
    // the beam-convolved map is only materialised job by job (parity tap) without the low-rank combination
    const bool lowrank = ctx->lr.r > 0 && ctx->lr_sep && !tap_convjobs;
    ctx->last_nblk3 = lowrank ? ctx->cv_lr.nblk3 : cv.nblk3;
    if (lowrank) {
        const JxLowrank& lr = ctx->lr;
        const int ntr = (lr.r + 15) / 16, threads = 512;
        const size_t lds = (size_t)ntr * ctx->lr_bucket * 64 * sizeof(double);
        const dim3 blocks(ctx->num_cu);
        const long long ld = cv.fir_ld, nt = cv.o + 1;
        const JxGemmSeg none{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0};
        const JxGemmSeg sr{lr.U, reinterpret_cast<const double*>(ctx->d_C), ctx->d_Clr, 0, 0, 0, 1LL, 1, 0};
        const JxGemmSeg sc{lr.U, cv.col0, ctx->d_col0lr, 0, 0, 0, (long long)lr.r, 1, 0};
#define JX_LR_GO(K, T) if (ctx->lr_bucket == K && ntr == T) { \
            hipLaunchKernelGGL((jx_lowrank_kernel<K, T>), blocks, dim3(threads), lds, st, lr, sr, none, (long long)cv.CROWS * ld, ld, 1LL, \
                               (long long)lr.r * ld, ld, cv.xsym ? cv.Ph : 2 * cv.Ph, n); \
            if (cv.xsym) \
                hipLaunchKernelGGL((jx_lowrank_kernel<K, T>), blocks, dim3(threads), lds, st, lr, sc, none, nt * cv.NJ, 1LL, (long long)cv.NJ, \
                                   nt * lr.r, 1LL, (int)nt, n); }
        JX_LR_KINDS(JX_LR_GO)
#undef JX_LR_GO
    }
    const JxConv& c3 = lowrank ? ctx->cv_lr : cv;
    const cplx* in3 = lowrank ? reinterpret_cast<const cplx*>(ctx->d_Clr) : ctx->d_C;
    const dim3 g3b(c3.nblk3, n);
#define JX_P3(LPv, LSv, R1v, R3v) if (!done && cv.LP == LPv && cv.LS == LSv) { \
        hipLaunchKernelGGL((jx_rowtf2_kernel<LPv, LSv, R3v>), g3b, dim3(256), ctx->p3_lds, st, c3, in3, ctx->d_part, tap_convjobs); done = true; }
    JX_CONV2_PAIRS(JX_P3)
#undef JX_P3
    if (tap_convjobs)
        hipLaunchKernelGGL(jx_expand_rows_kernel, dim3(cv.S, n), dim3(256), 0, st, tap_convjobs, ctx->d_rowjob, cv.S, cv.NJ, ctx->t_conv);
    if (es && !es->p1only) HIPCHK(ctx, hipEventRecord(es->e[4], st));
    return JX_OK;
}

// Contracted route (jx_mix.hpp): spline arrays (walker-minor) -> stage 1 -> stage 2 -> tail.  es: the launch's event set or null.
static int launch_mix(jx_ctx* ctx, int n, EvSet* es) {
    hipStream_t st = ctx->stream;
    const bool f32 = false;
    (void)f32;
    if (ctx->mix_form == 0) {
        JxMix mx = ctx->mx;
        mx.n = n;
#ifdef JOXSZ_ABLATIONS
        if (const char* e = getenv("JOXSZ_MIX_DBG")) mx.dbg = atoi(e);
#endif
        int wpb = 4;                                              // waves per block: they share a column's scalar stream
        if (const char* e = getenv("JOXSZ_MIX_WPB")) { const int v = atoi(e); if (v >= 1 && v <= 16) wpb = v; }
        const int ngrp = (n + 63) / 64;
        wpb = std::min(wpb, ngrp);
        const int nq = (ngrp + wpb - 1) / wpb;
        mx.cper = (nq <= 8 && 8 % nq == 0) ? 8 / nq : 0;
        const dim3 g1((unsigned)(mx.cper ? 8 * ((mx.NU + mx.cper - 1) / mx.cper) : nq * mx.NU));
        bool done = false;
#define JX_MIX_GO(Rv) if (!done && ctx->mix_RT == Rv) { \
            hipLaunchKernelGGL((jx_rowmix_kernel<Rv, JX_MIX_NS, double2>), g1, dim3(64 * wpb), 0, st, mx, reinterpret_cast<const double2*>(ctx->d_cft), ctx->d_Dt); done = true; }
        JX_MIX_RTS(JX_MIX_GO)
#undef JX_MIX_GO
        if (!done) { ctx->err = "no stage-1 kernel for this rank"; return JX_ERR_UNSUPPORTED; }
    }
    if (es) HIPCHK(ctx, hipEventRecord(es->e[3], st));
    {
        JxOpg og = ctx->og;
        og.n = n;
        const int nwb = (n + 127) / 128;
        int ksplit = (2 * ctx->num_cu + nwb * og.nog - 1) / (nwb * og.nog);
        if (const char* e = getenv("JOXSZ_MIX_KSPLIT")) { const int v = atoi(e); if (v > 0) ksplit = v; }
        ksplit = std::max(1, std::min(std::min(ksplit, JX_MIX_KSPLIT_MAX), (ctx->mix_ksteps + 7) / 8));
        int kper = (ctx->mix_ksteps + ksplit - 1) / ksplit;
        kper = (kper + JX_OPG_RD - 1) / JX_OPG_RD * JX_OPG_RD;
        ksplit = (ctx->mix_ksteps + kper - 1) / kper;
        og.ksplit = ksplit; og.kper = kper;
        ctx->og.ksplit = ksplit;                                  // (the tail sums this many partials)
        const int nunit = ksplit * og.nog;
        const dim3 g2((unsigned)(8 * nwb * ((nunit + 7) / 8)));
        bool done = false;
#define JX_OPG_GO(Xv) if (!done && ctx->mix_nxt == Xv) { \
            hipLaunchKernelGGL((jx_opgemm_kernel<0, Xv, double2>), g2, dim3(256), 0, st, og, reinterpret_cast<const double2*>(ctx->d_cft), ctx->d_Pt); done = true; }
        JX_MIX_NXTS(JX_OPG_GO)
#undef JX_OPG_GO
        if (!done) { ctx->err = "no stage-2 kernel for this output tiling"; return JX_ERR_UNSUPPORTED; }
    }
    if (es && !es->p1only) HIPCHK(ctx, hipEventRecord(es->e[4], st));
    return JX_OK;
}

// One chunk: walkers [w0, w0+n) of the batch whose thetas live at theta_dev.
static int run_chunk(jx_ctx* ctx, const double* theta_dev, double* logp_dev, int w0, int n, const Taps& t) {
    const JxDev& d = ctx->d;
    const bool op_route = ctx->route == JX_ROUTE_OPERATOR && !t.pp && !d.inject_pp;      // stage taps and the operator build: map route
    Plan3* pl = nullptr;
    int rc = (ctx->conv_mode == 1 && !op_route) ? make_plans(ctx, n, &pl) : JX_OK;
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    EvSet es;
    // timing mode 2 records the two events around pass 1 only (default route of the hand-written convolution); elsewhere it records nothing
    const bool tm = ctx->timing_on && ctx->timing_mode != 2;
    const bool tm2 = ctx->timing_on && ctx->timing_mode == 2 && ctx->conv_mode >= 2 && !op_route;
    // the operator route has no per-walker work buffers beyond these three, so its launches can be much larger than a chunk
    double* base_buf = op_route ? ctx->d_base_op : ctx->d_base;
    double* cfac_buf = op_route ? ctx->d_cfac_op : ctx->d_cfac;
    double* sz0_buf = op_route ? ctx->d_sz0_op : ctx->d_sz0;           // null unless calc_integ
    // default route: the map rows are evaluated inside pass 1 from the coefficients (no image), unless the image is asked for
    const bool mix = !op_route && ctx->conv_mode == 3;
    if (mix && t.conv) { ctx->err = "conv_2d tap: not available on the contracted route (use conv = rocfft for this tap)"; return JX_ERR_UNSUPPORTED; }
    const bool dct = mix || (!op_route && ctx->dct_ok && (ctx->odd || (use_fused(ctx, t.conv) && !t.need_img)));
    // ... and the coefficients come from one matrix product, unless the profile taps are asked for (they live in the Abel kernel)
    // (fp32 contexts keep the spline arrays in float and always take them from the matrix product -- from the injected profiles
    //  when the operator is being built; the Abel kernel then runs beside it only to serve the profile taps)
    const bool f32cf = ctx->f32 && dct;
    const bool want_abel_taps = t.pp || t.ab || t.y || t.need_img;
    const bool ag = dct && ctx->abel_gemm && (f32cf || (!want_abel_taps && !d.inject_pp));
    if (tm || tm2) {
        if (ctx->ev_inflight.size() > 2048 && (rc = drain_events(ctx))) return rc;
        if ((rc = get_evset(ctx, &es))) return rc;
        es.walkers = n;
        es.op = false;
        es.gemm = false;
        es.p1only = tm2;
        if (tm) HIPCHK(ctx, hipEventRecord(es.e[0], st));
    }
    {
        const size_t sh = sizeof(double) * (JX_LDS_HDR + (size_t)2 * d.N + 2 * d.nann + (size_t)d.nband * d.nann + 8);
        hipLaunchKernelGGL(jx_prep_kernel, dim3(n), dim3(JX_PREP_THREADS), sh, st, d, theta_dev, w0,
                           base_buf, cfac_buf, op_route ? ctx->d_pp : ((ag && !d.inject_pp) ? ctx->d_ppc : (double*)nullptr), sz0_buf, t.tprof, t.xprofs, t.parts, t.integ);
    }
    if (tm) HIPCHK(ctx, hipEventRecord(es.e[1], st));
    if (op_route) {
        // collapsed route: the SZ side is one kernel (no map, no transforms)
        es.op = true;
        const int nrow = d.nrow, Re = (nrow + 1) & ~1;
        // large launches: G pp on the fp64 matrix cores (G fetched once per 32 walkers), the rest in the kernel behind it
        const int mt = (nrow + 63) / 64;                                     // row tiles per wave
        const bool wide = n >= 4096 && (mt == 1 || mt == 2 || mt == 4 || mt == 8) && !getenv("JOXSZ_OP_NARROW");
        if (wide) {
            const size_t shw = sizeof(double) * JX_OPM_JC * 33;
#define JX_OPM_GO(MTv) hipLaunchKernelGGL((jx_operator_mfma_kernel<MTv>), dim3((n + 31) / 32), dim3(256), shw, st, ctx->d_pp, n, d.N, nrow, \
                           ctx->d_G, ctx->g_ld, ctx->d_rows)
            if (mt == 1) JX_OPM_GO(1); else if (mt == 2) JX_OPM_GO(2); else if (mt == 4) JX_OPM_GO(4); else JX_OPM_GO(8);
#undef JX_OPM_GO
        }
        const double* rows_t = wide ? ctx->d_rows : nullptr;
        // walkers per block (x radii per LDS chunk = 2048): more of them spare G traffic; behind the matrix-core product there is none to spare
        int wpb = wide ? 4 : ((n >= 4096) ? 16 : (n >= 2048 ? 8 : 4));
        while (wpb > 4 && sizeof(double) * (2048 + (size_t)wpb * Re + 8) > 64 * 1024) wpb >>= 1;
        const size_t sh = sizeof(double) * (2048 + (size_t)wpb * Re + 8);
#define JX_OP_GO(WPBv) hipLaunchKernelGGL((jx_operator_kernel<WPBv, 2048 / WPBv>), dim3((n + WPBv - 1) / WPBv), dim3(256), sh, st, d, ctx->d_pp, w0, n, \
                           ctx->d_G, ctx->g_ld, rows_t, cfac_buf, sz0_buf, base_buf, logp_dev, t.row, t.bright, t.chisq, t.parts)
        if (wpb == 16) JX_OP_GO(16); else if (wpb == 8) JX_OP_GO(8); else JX_OP_GO(4);
#undef JX_OP_GO
        if (tm) {
            HIPCHK(ctx, hipEventRecord(es.e[5], st));
            ctx->ev_inflight.push_back(es);
        }
        HIPCHK(ctx, hipGetLastError());
        return JX_OK;
    }
    if (ctx->f32 && !op_route && !dct) { ctx->err = "dtype f32: the map and beam-convolved-map taps exist in the f64 build of the context only"; return JX_ERR_UNSUPPORTED; }
    if (ag) {
        const dim3 grid((n + 31) / 32, (ctx->tm_npair + 3) / 4);
        const double* pp_src = d.inject_pp ? d.inject_pp : ctx->d_ppc;
        if (mix) hipLaunchKernelGGL((jx_abel_gemm_kernel<1, double, 1>), grid, dim3(256), sizeof(double) * JX_OPM_JC * 33, st, pp_src, n, d.N, ctx->d_Tm, ctx->tm_ld,
                                    d.K, ctx->tm_ntile, ctx->tm_npair, ctx->d_cft, ctx->mix_tW, (long long)ctx->mix_ncol);
        else if (f32cf) hipLaunchKernelGGL((jx_abel_gemm_kernel<1, float>), grid, dim3(256), sizeof(double) * JX_OPM_JC * 33, st, pp_src, n, d.N, ctx->d_Tm, ctx->tm_ld,
                                      d.K, ctx->tm_ntile, ctx->tm_npair, reinterpret_cast<float*>(ctx->d_cf), ctx->dct.cf_ws);
        else hipLaunchKernelGGL((jx_abel_gemm_kernel<1>), grid, dim3(256), sizeof(double) * JX_OPM_JC * 33, st, pp_src, n, d.N, ctx->d_Tm, ctx->tm_ld,
                                d.K, ctx->tm_ntile, ctx->tm_npair, ctx->d_cf, ctx->dct.cf_ws);
    }
    if (!ag || (f32cf && want_abel_taps)) {
        const bool vec2 = (d.S % 2 == 0) && (d.P % 2 == 0);
        const int npw = (d.quad && d.pairw == 2) ? 2 : 1;
        const dim3 grid0(((n + npw - 1) / npw) * d.map_split), block(ctx->map_threads);
        if (d.fast_map) {
            const size_t sh = ctx->map_lds_bytes;
            JxDev dm = d;                                          // fused route: column 0 is copied walker-minor
            dm.nlaunch = n;
            if (use_fused(ctx, t.conv) && ctx->d_x0t) { dm.xcol = ctx->d_x0t; dm.xcol_ld = ctx->tW; }   // (odd sides have no unpaired column)
            const int nait = (d.q_na + 63) / 64;
#define JX_SYM_LAUNCH(V, NA) hipLaunchKernelGGL((jx_abel_map_sym_kernel<V, NA>), grid, block, sh, st, dm, theta_dev, w0, ctx->d_img, t.pp, t.ab, t.y)
#define JX_SYM_PICK() { if (vec2) { if (nait <= 3) JX_SYM_LAUNCH(true, 3); else if (nait <= 5) JX_SYM_LAUNCH(true, 5); else JX_SYM_LAUNCH(true, 9); } \
                        else      { if (nait <= 3) JX_SYM_LAUNCH(false, 3); else if (nait <= 5) JX_SYM_LAUNCH(false, 5); else JX_SYM_LAUNCH(false, 9); } }
            if (dct && t.need_img) {                               // the Compton-y map tap on a route that does not store the map
                const dim3 grid = grid0;
                JX_SYM_PICK()
            }
            if (dct) { dm.cf_out = f32cf ? ctx->d_cf_tap : ctx->d_cf; dm.cf_ws = ctx->dct.cf_ws; dm.map_split = 1; }   // phases 1-3 only: spline out (fp32 contexts: to a scratch array, for the taps' sake)
            // contracted route: the Abel kernel's arrays go straight to the walker-minor array when the matrix product is off
            // (JOXSZ_ABEL_GEMM=0, operator build), else to the walker-major scratch (only the profile taps are wanted from it)
            if (mix && !ag) { dm.cf_out = ctx->d_cft; dm.cf_ws = ctx->mix_tW; dm.cf_tr = 1; }
            const dim3 grid(dct ? (unsigned)((n + npw - 1) / npw) : grid0.x);
            JX_SYM_PICK()
#undef JX_SYM_PICK
#undef JX_SYM_LAUNCH
        } else {
            const size_t sh = ctx->map_lds_bytes;
            if (vec2) hipLaunchKernelGGL(jx_abel_map_kernel<true>, grid0, block, sh, st, d, theta_dev, w0, ctx->d_img, t.pp, t.ab, t.y);
            else hipLaunchKernelGGL(jx_abel_map_kernel<false>, grid0, block, sh, st, d, theta_dev, w0, ctx->d_img, t.pp, t.ab, t.y);
        }
    }
    if (tm || tm2) HIPCHK(ctx, hipEventRecord(es.e[2], st));
    const cplx* zpart = nullptr;
    int nblk = 0;
    if (ctx->conv_mode == 1) {
        {
            void* in[1] = {ctx->d_img};
            void* out[1] = {ctx->d_spec};
            FFTCHK(ctx, rocfft_execute(pl->beam_fwd, in, out, ctx->info));
            const size_t per = (size_t)d.P * d.Ph, total = per * n;
            const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
            hipLaunchKernelGGL(jx_beam_mul_kernel, dim3(blocks), dim3(256), 0, st, ctx->d_spec, (const double2*)d.bhat, per, total);
            void* in2[1] = {ctx->d_spec};
            void* out2[1] = {ctx->d_conv};
            FFTCHK(ctx, rocfft_execute(pl->beam_inv, in2, out2, ctx->info));
        }
        if (tm) HIPCHK(ctx, hipEventRecord(es.e[3], st));
        {
            void* in[1] = {ctx->d_conv};
            void* out[1] = {ctx->d_tfspec};
            FFTCHK(ctx, rocfft_execute(pl->tf_fwd, in, out, ctx->info));
        }
        if (tm) HIPCHK(ctx, hipEventRecord(es.e[4], st));
    } else {
        if (ctx->odd && !dct) { ctx->err = "odd map side: the hand-written route needs the coefficient-fed pass 1"; return JX_ERR_UNSUPPORTED; }
        EvSet* esp = (tm || tm2) ? &es : nullptr;
        int rc2 = mix ? launch_mix(ctx, n, esp) : ctx->odd ? launch_odd_conv(ctx, n, esp)
                : use_fused(ctx, t.conv) ? launch_fused_conv(ctx, n, esp, dct) : launch_custom_conv(ctx, n, t.conv, esp);
        if (rc2) return rc2;
        zpart = ctx->d_part;
        nblk = ctx->last_nblk3;
    }
    if (mix) {
        const size_t sh = sizeof(double) * (JX_LDS_HDR + (size_t)d.nrow + 8);
        hipLaunchKernelGGL(jx_tail_row_kernel, dim3(n), dim3(JX_TAIL_THREADS), sh, st, d, ctx->d_Pt, ctx->og.ksplit, (long long)ctx->mix_tW * ctx->og.ldx, ctx->og.ldx,
                           ctx->d_cfac, ctx->d_sz0, ctx->d_base, logp_dev, w0, t.row, t.bright, t.chisq, t.parts);
        if (tm) HIPCHK(ctx, hipEventRecord(es.e[5], st));
        if (tm || tm2) ctx->ev_inflight.push_back(es);
        HIPCHK(ctx, hipGetLastError());
        return JX_OK;
    }
    if (ctx->odd) {
        if (t.conv && (rc = launch_odd_conv_tap(ctx, n))) return rc;
        const size_t sh = sizeof(double) * (JX_LDS_HDR + (size_t)d.nrow + 8);
        hipLaunchKernelGGL(jx_tail_odd_kernel, dim3(n), dim3(JX_TAIL_THREADS), sh, st, d, ctx->d_D2, ctx->lrf.r, ctx->o_ldb, ctx->d_cfac, ctx->d_sz0,
                           ctx->d_base, logp_dev, w0, t.row, t.bright, t.chisq, t.parts);
        if (tm) HIPCHK(ctx, hipEventRecord(es.e[5], st));
        if (tm || tm2) ctx->ev_inflight.push_back(es);
        HIPCHK(ctx, hipGetLastError());
        return JX_OK;
    }
    bool tail_done = false;
    if (ctx->conv_mode == 2 && d.nrow == ctx->cv.LS && !getenv("JOXSZ_TAIL_DFT")) {
        const JxConv& cv = ctx->cv;
#define JX_TAILF(LPv, LSv, R1v, R3v) if (!tail_done && cv.LP == LPv && cv.LS == LSv) { \
            hipLaunchKernelGGL((jx_tail_fft_kernel<LSv>), dim3(n), dim3(256), 0, st, d, cv, zpart, nblk, ctx->d_cfac, ctx->d_sz0, ctx->d_base, logp_dev, w0, \
                               t.row, t.bright, t.chisq, t.parts); tail_done = true; }
        JX_CONV2_PAIRS(JX_TAILF)
#undef JX_TAILF
    }
    if (!tail_done) {
        const size_t sh = sizeof(double) * (JX_LDS_HDR + (size_t)2 * d.Sh + d.nrow + 8);
        hipLaunchKernelGGL(jx_tail_kernel, dim3(n), dim3(JX_TAIL_THREADS), sh, st, d, ctx->d_tfspec, zpart, nblk, ctx->d_cfac, ctx->d_sz0,
                           ctx->d_base, logp_dev, w0, t.row, t.bright, t.chisq, t.parts);
    }
    if (tm) HIPCHK(ctx, hipEventRecord(es.e[5], st));
    if (tm || tm2) ctx->ev_inflight.push_back(es);
    HIPCHK(ctx, hipGetLastError());
    return JX_OK;
}

int jx_eval_device(jx_ctx* ctx, const double* theta_dev, int nwalkers, double* logp_dev) {
    if (!ctx || !theta_dev || !logp_dev || nwalkers < 0) return JX_ERR_INVALID;
    if (!ctx->finalized) { ctx->err = "jx_eval before jx_finalize"; return JX_ERR_STATE; }
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    Taps none;
    const int step = (ctx->route == JX_ROUTE_OPERATOR) ? ctx->op_cap : ctx->chunk;
    for (int w0 = 0; w0 < nwalkers; w0 += step) {
        const int n = std::min(step, nwalkers - w0);
        int rc = run_chunk(ctx, theta_dev, logp_dev, w0, n, none);
        if (rc) return rc;
    }
    return JX_OK;
}

int jx_sample(jx_ctx* ctx, const double* theta0, int nwalkers, int nsteps, double a, uint64_t seed,
              double* chain_out, double* logp_out, int64_t* naccept_out) {
    if (!ctx || !theta0 || nwalkers < 2 || (nwalkers & 1) || nsteps < 0 || !(a > 1.0)) return JX_ERR_INVALID;
    if (!ctx->finalized) { ctx->err = "jx_sample before jx_finalize"; return JX_ERR_STATE; }
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    const int ndim = ctx->cfg.ndim, W = nwalkers, half = W / 2;
    hipStream_t st = ctx->stream;
    double *x = nullptr, *lp = nullptr, *q = nullptr, *lq = nullptr, *zz = nullptr, *chain = nullptr, *lps = nullptr;
    long long* nacc = nullptr;
    auto cleanup = [&]() {};                                     // (the work buffers stay with the context: grow-only, freed by jx_destroy)
#define SCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_); cleanup(); return JX_ERR_HIP; } } while (0)
    // a run re-uses the buffers of the last one when they are large enough: repeated runs (burn-in, then sampling) do not
    // pay for device allocations of tens of megabytes each time
    auto grab = [&](int slot, size_t bytes, void** out) -> hipError_t {
        if (ctx->samp_cap[slot] < bytes) {
            if (ctx->samp_buf[slot]) { (void)hipStreamSynchronize(st); (void)hipFree(ctx->samp_buf[slot]); ctx->samp_buf[slot] = nullptr; ctx->samp_cap[slot] = 0; }
            const hipError_t e = hipMalloc(&ctx->samp_buf[slot], bytes);
            if (e != hipSuccess) return e;
            ctx->samp_cap[slot] = bytes;
        }
        *out = ctx->samp_buf[slot];
        return hipSuccess;
    };
    SCHK(grab(0, sizeof(double) * (size_t)W * ndim, (void**)&x));
    SCHK(grab(1, sizeof(double) * (size_t)W, (void**)&lp));
    SCHK(grab(2, sizeof(double) * (size_t)half * ndim, (void**)&q));
    SCHK(grab(3, sizeof(double) * (size_t)half, (void**)&lq));
    SCHK(grab(4, sizeof(double) * (size_t)half, (void**)&zz));
    SCHK(grab(5, sizeof(long long) * (size_t)W, (void**)&nacc));
    SCHK(hipMemsetAsync(nacc, 0, sizeof(long long) * (size_t)W, st));
    if (chain_out && nsteps) SCHK(grab(6, sizeof(double) * (size_t)nsteps * W * ndim, (void**)&chain));
    if (logp_out && nsteps) SCHK(grab(7, sizeof(double) * (size_t)nsteps * W, (void**)&lps));
    SCHK(hipMemcpyAsync(x, theta0, sizeof(double) * (size_t)W * ndim, hipMemcpyHostToDevice, st));
    int rc = jx_eval_device(ctx, x, W, lp);
    if (rc) { cleanup(); return rc; }
    {
        std::vector<double> l0(W);
        SCHK(hipMemcpyAsync(l0.data(), lp, sizeof(double) * W, hipMemcpyDeviceToHost, st));
        SCHK(hipStreamSynchronize(st));
        for (double v : l0) if (!std::isfinite(v)) { ctx->err = "initial positions must have finite log-posterior"; cleanup(); return JX_ERR_INVALID; }
    }
    const dim3 grid((half + 255) / 256), block(256);
    for (int it = 0; it < nsteps; ++it) {
        for (int hs = 0; hs < 2; ++hs) {
            const int s1 = hs * half, s2 = (1 - hs) * half;
            hipLaunchKernelGGL(jx_sm_propose_kernel, grid, block, 0, st, x, q, zz, ndim, half, s1, s2, 2 * it + hs, a, seed);
            if ((rc = jx_eval_device(ctx, q, half, lq))) { cleanup(); return rc; }
            hipLaunchKernelGGL(jx_sm_accept_kernel, grid, block, 0, st, x, lp, q, lq, zz, nacc, ndim, half, s1, 2 * it + hs, seed);
        }
        if (chain) SCHK(hipMemcpyAsync(chain + (size_t)it * W * ndim, x, sizeof(double) * (size_t)W * ndim, hipMemcpyDeviceToDevice, st));
        if (lps) SCHK(hipMemcpyAsync(lps + (size_t)it * W, lp, sizeof(double) * (size_t)W, hipMemcpyDeviceToDevice, st));
    }
    if (chain) SCHK(hipMemcpyAsync(chain_out, chain, sizeof(double) * (size_t)nsteps * W * ndim, hipMemcpyDeviceToHost, st));
    if (lps) SCHK(hipMemcpyAsync(logp_out, lps, sizeof(double) * (size_t)nsteps * W, hipMemcpyDeviceToHost, st));
    if (naccept_out) SCHK(hipMemcpyAsync(naccept_out, nacc, sizeof(long long) * (size_t)W, hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    SCHK(hipGetLastError());
#undef SCHK
    cleanup();
    return JX_OK;
}

static int ensure_taps(jx_ctx* ctx);

// G by the MAP route's own kernels: the unit profiles e_j go in as injected pressure profiles, their map rows come out
// of the row tap.  (theta only feeds the prep kernel here; the current parameter values keep it on ordinary numbers.)
static int build_operator(jx_ctx* ctx) {
    const jx_config& c = ctx->cfg;
    const int N = c.N, nrow = ctx->nrow, ld = (nrow + 15) & ~15;
    int rc;
    if ((rc = ensure_taps(ctx))) return rc;
    if ((rc = ensure_batch(ctx, ctx->chunk))) return rc;
    double* G = nullptr;
    double* inj = nullptr;
    if ((rc = dev_new(ctx, (size_t)(N + JX_OPM_GPAD) * ld, &G, true))) return rc;   // zero rows behind the last: the kernels read j in pairs / groups of 16, three k-steps ahead
    HIPCHK(ctx, hipMalloc((void**)&inj, sizeof(double) * (size_t)ctx->chunk * N));
    std::vector<double> th((size_t)ctx->chunk * c.ndim);
    {
        const std::vector<double> pv = host_vec<double>(ctx, JX_T_PAR_VALS);
        const std::vector<int32_t> ti = host_vec<int32_t>(ctx, JX_T_THAWED_IDX);
        for (int w = 0; w < ctx->chunk; ++w) for (int k = 0; k < c.ndim; ++k) th[(size_t)w * c.ndim + k] = pv[ti[k]];
    }
    hipStream_t st = ctx->stream;
    const bool tm = ctx->timing_on;
    auto fail = [&](int code) { (void)hipStreamSynchronize(st); ctx->d.inject_pp = nullptr; ctx->timing_on = tm; (void)hipFree(inj); return code; };
    if (hipMemcpyAsync(ctx->d_theta, th.data(), sizeof(double) * th.size(), hipMemcpyHostToDevice, st) != hipSuccess) return fail(JX_ERR_HIP);
    Taps t;
    t.pp = ctx->t_pp; t.ab = ctx->t_ab; t.y = ctx->t_y; t.row = ctx->t_row; t.bright = ctx->t_bright;
    t.chisq = ctx->t_chisq; t.tprof = ctx->t_tprof; t.xprofs = c.sz_only ? nullptr : ctx->t_xprofs; t.parts = ctx->t_parts;
    std::vector<double> eye;
    ctx->timing_on = false;
    for (int j0 = 0; j0 < N; j0 += ctx->chunk) {
        const int n = std::min(ctx->chunk, N - j0);
        eye.assign((size_t)n * N, 0.0);
        for (int w = 0; w < n; ++w) eye[(size_t)w * N + j0 + w] = 1.0;
        if (hipMemcpyAsync(inj, eye.data(), sizeof(double) * eye.size(), hipMemcpyHostToDevice, st) != hipSuccess) return fail(JX_ERR_HIP);
        ctx->d.inject_pp = inj;
        rc = run_chunk(ctx, ctx->d_theta, ctx->d_logp, 0, n, t);
        ctx->d.inject_pp = nullptr;
        if (rc) return fail(rc);
        if (hipMemcpy2DAsync(G + (size_t)j0 * ld, sizeof(double) * ld, ctx->t_row, sizeof(double) * nrow, sizeof(double) * nrow, n,
                             hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(JX_ERR_HIP);
        if (hipStreamSynchronize(st) != hipSuccess) return fail(JX_ERR_HIP);
    }
    ctx->timing_on = tm;
    (void)hipFree(inj);
    ctx->op_cap = std::max(ctx->chunk, ctx->cfg.max_batch > 0 ? ctx->chunk : 16384);      // (an explicit max_batch bounds this route too)
    if ((rc = dev_new(ctx, (size_t)ctx->op_cap * N, &ctx->d_pp))) return rc;
    if ((rc = dev_new(ctx, (size_t)ctx->op_cap, &ctx->d_base_op))) return rc;
    if ((rc = dev_new(ctx, (size_t)ctx->op_cap * nrow, &ctx->d_cfac_op))) return rc;
    if (c.calc_integ && (rc = dev_new(ctx, (size_t)ctx->op_cap, &ctx->d_sz0_op, true))) return rc;
    if ((rc = dev_new(ctx, ((size_t)ctx->op_cap + 32) * nrow, &ctx->d_rows))) return rc;
    ctx->d_G = G;
    ctx->g_ld = ld;
    return JX_OK;
}

int jx_set_route(jx_ctx* ctx, int route) {
    if (!ctx || (route != JX_ROUTE_MAP && route != JX_ROUTE_OPERATOR)) return JX_ERR_INVALID;
    if (!ctx->finalized) { ctx->err = "jx_set_route before jx_finalize"; return JX_ERR_STATE; }
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    if (route == JX_ROUTE_OPERATOR && !ctx->d_G) {
        const int rc = build_operator(ctx);
        if (rc) return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->route = route;
    return JX_OK;
}

int jx_get_route(jx_ctx* ctx) { return ctx ? ctx->route : JX_ERR_INVALID; }

int jx_get_operator(jx_ctx* ctx, double* out, size_t nbytes) {
    if (!ctx || !out) return JX_ERR_INVALID;
    if (!ctx->d_G) { ctx->err = "jx_get_operator: the operator route was never selected"; return JX_ERR_STATE; }
    if (nbytes != sizeof(double) * (size_t)ctx->cfg.N * ctx->nrow) { ctx->err = "jx_get_operator: wrong output size"; return JX_ERR_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    HIPCHK(ctx, hipMemcpy2DAsync(out, sizeof(double) * ctx->nrow, ctx->d_G, sizeof(double) * ctx->g_ld, sizeof(double) * ctx->nrow,
                                 ctx->cfg.N, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return JX_OK;
}

int jx_set_stream(jx_ctx* ctx, void* hip_stream) {
    if (!ctx) return JX_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (ctx->info) FFTCHK(ctx, rocfft_execution_info_set_stream(ctx->info, ctx->stream));
    return JX_OK;
}

int jx_sync(jx_ctx* ctx) {
    if (!ctx) return JX_ERR_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return JX_OK;
}

int jx_eval(jx_ctx* ctx, const double* theta, int nwalkers, double* logp) {
    if (!ctx || nwalkers < 0 || (nwalkers && (!theta || !logp))) return JX_ERR_INVALID;
    if (!ctx->finalized) { ctx->err = "jx_eval before jx_finalize"; return JX_ERR_STATE; }
    if (nwalkers == 0) return JX_OK;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    int rc = ensure_batch(ctx, nwalkers);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_theta, theta, sizeof(double) * (size_t)nwalkers * ctx->cfg.ndim, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = jx_eval_device(ctx, ctx->d_theta, nwalkers, ctx->d_logp))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(logp, ctx->d_logp, sizeof(double) * (size_t)nwalkers, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return JX_OK;
}

static int ensure_taps(jx_ctx* ctx) {
    if (ctx->t_pp) return JX_OK;
    const jx_config& c = ctx->cfg;
    const size_t C = ctx->chunk;
    int rc;
    if ((rc = dev_new(ctx, C * c.N, &ctx->t_pp))) return rc;
    if ((rc = dev_new(ctx, C * c.N, &ctx->t_ab))) return rc;
    if ((rc = dev_new(ctx, C * c.N, &ctx->t_y))) return rc;
    if ((rc = dev_new(ctx, C * ctx->nrow, &ctx->t_row))) return rc;
    if ((rc = dev_new(ctx, C * ctx->nrow, &ctx->t_bright))) return rc;
    if ((rc = dev_new(ctx, C * ctx->nrow, &ctx->t_tprof))) return rc;
    if ((rc = dev_new(ctx, C, &ctx->t_chisq))) return rc;
    if ((rc = dev_new(ctx, C * std::max(1, c.nband * c.nann), &ctx->t_xprofs, true))) return rc;
    if ((rc = dev_new(ctx, C * 4, &ctx->t_parts, true))) return rc;
    if ((rc = dev_new(ctx, C, &ctx->t_integ, true))) return rc;
    return JX_OK;
}

int jx_eval_stage(jx_ctx* ctx, const double* theta, int nwalkers, int stage, double* out, size_t nbytes) {
    if (!ctx || !theta || !out || nwalkers <= 0) return JX_ERR_INVALID;
    if (!ctx->finalized) { ctx->err = "jx_eval_stage before jx_finalize"; return JX_ERR_STATE; }
    if (stage < 0 || stage >= JX_STAGE_COUNT) return JX_ERR_INVALID;
    const jx_config& c = ctx->cfg;
    const size_t S = c.S, P = ctx->P;
    size_t per = 0;
    switch (stage) {
        case JX_STAGE_PP: case JX_STAGE_AB: case JX_STAGE_Y: per = c.N; break;
        case JX_STAGE_Y2D: case JX_STAGE_CONV2D: per = S * S; break;
        case JX_STAGE_MAPROW: case JX_STAGE_BRIGHT: case JX_STAGE_TPROF: per = ctx->nrow; break;
        case JX_STAGE_CHISQ: per = 1; break;
        case JX_STAGE_XPROFS: per = (size_t)c.nband * c.nann; break;
        case JX_STAGE_PARTS: per = 4; break;
        case JX_STAGE_INTEG: per = 1; break;
    }
    if (stage == JX_STAGE_INTEG && !c.calc_integ) { ctx->err = "the 'integ' output needs calc_integ"; return JX_ERR_INVALID; }
    if (stage == JX_STAGE_XPROFS && c.sz_only) { ctx->err = "no X-ray profiles in sz_only mode"; return JX_ERR_INVALID; }
    if (nbytes != per * sizeof(double) * (size_t)nwalkers) { ctx->err = "jx_eval_stage: wrong output size"; return JX_ERR_INVALID; }
    HIPCHK(ctx, hipSetDevice(c.device));
    int rc;
    if ((rc = ensure_taps(ctx))) return rc;
    if (stage == JX_STAGE_CONV2D && ctx->conv_mode == 2 && !ctx->t_conv) {          // the big ones only when asked for
        if ((rc = dev_new(ctx, (size_t)ctx->chunk * S * S, &ctx->t_conv))) return rc;
        if ((rc = dev_new(ctx, (size_t)ctx->chunk * ctx->cv.NJ * S, &ctx->t_convjobs))) return rc;
    }
    const bool y2d_quad = (stage == JX_STAGE_Y2D && ctx->d.quad);
    if (y2d_quad && !ctx->t_y2d && (rc = dev_new(ctx, (size_t)ctx->chunk * S * S, &ctx->t_y2d))) return rc;
    if ((rc = ensure_batch(ctx, nwalkers))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_theta, theta, sizeof(double) * (size_t)nwalkers * c.ndim, hipMemcpyHostToDevice, ctx->stream));
    Taps t;
    t.pp = ctx->t_pp; t.ab = ctx->t_ab; t.y = ctx->t_y; t.row = ctx->t_row; t.bright = ctx->t_bright;
    t.conv = (stage == JX_STAGE_CONV2D) ? ctx->t_convjobs : nullptr;
    t.need_img = (stage == JX_STAGE_Y2D);
    t.chisq = ctx->t_chisq; t.tprof = ctx->t_tprof; t.xprofs = c.sz_only ? nullptr : ctx->t_xprofs; t.parts = ctx->t_parts;
    t.integ = ctx->t_integ;
    for (int w0 = 0; w0 < nwalkers; w0 += ctx->chunk) {
        const int n = std::min(ctx->chunk, nwalkers - w0);
        if ((rc = run_chunk(ctx, ctx->d_theta, ctx->d_logp, w0, n, t))) return rc;
        double* dst = out + per * (size_t)w0;
        const double* src = nullptr;
        switch (stage) {
            case JX_STAGE_PP: src = ctx->t_pp; break;
            case JX_STAGE_AB: src = ctx->t_ab; break;
            case JX_STAGE_Y: src = ctx->t_y; break;
            case JX_STAGE_MAPROW: src = ctx->t_row; break;
            case JX_STAGE_BRIGHT: src = ctx->t_bright; break;
            case JX_STAGE_TPROF: src = ctx->t_tprof; break;
            case JX_STAGE_CHISQ: src = ctx->t_chisq; break;
            case JX_STAGE_XPROFS: src = ctx->t_xprofs; break;
            case JX_STAGE_PARTS: src = ctx->t_parts; break;
            case JX_STAGE_INTEG: src = ctx->t_integ; break;
            default: break;
        }
        if (src) {
            HIPCHK(ctx, hipMemcpyAsync(dst, src, per * sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
        } else {
            const bool dense = (stage == JX_STAGE_CONV2D && ctx->conv_mode == 2);
            if (y2d_quad)
                hipLaunchKernelGGL(jx_expand_quad_kernel, dim3((unsigned)S, n), dim3(256), 0, ctx->stream, ctx->d_img, (size_t)ctx->d.img_ld,
                                   (size_t)ctx->d.img_ws, (int)S, ctx->t_y2d);
            const double* base = (stage == JX_STAGE_Y2D) ? (y2d_quad ? ctx->t_y2d : ctx->d_img) : (dense ? ctx->t_conv : ctx->d_conv);
            const size_t ld = (stage == JX_STAGE_Y2D) ? (y2d_quad ? (size_t)S : (size_t)ctx->d.img_ld) : (dense ? S : P);
            const size_t ws = (stage == JX_STAGE_Y2D) ? (y2d_quad ? (size_t)S * S : (size_t)ctx->d.img_ws) : (dense ? S * S : P * P);
            for (int w = 0; w < n; ++w)
                HIPCHK(ctx, hipMemcpy2DAsync(dst + (size_t)w * S * S, S * sizeof(double), base + (size_t)w * ws,
                                             ld * sizeof(double), S * sizeof(double), S, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return JX_OK;
}

int jx_set_par_vals(jx_ctx* ctx, const double* v, int npar) {
    if (!ctx || !v || npar != ctx->cfg.npar) return JX_ERR_INVALID;
    if (!ctx->finalized) return jx_upload(ctx, JX_T_PAR_VALS, v, sizeof(double) * npar);
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_par_vals, v, sizeof(double) * npar, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return JX_OK;
}

int jx_dev_alloc(jx_ctx* ctx, size_t nbytes, void** out) {
    if (!ctx || !out) return JX_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    HIPCHK(ctx, hipMalloc(out, std::max<size_t>(nbytes, 8)));
    return JX_OK;
}

int jx_dev_free(jx_ctx* ctx, void* p) {
    if (!ctx) return JX_ERR_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipFree(p));
    return JX_OK;
}

int jx_memcpy_h2d(jx_ctx* ctx, void* dev, const void* host, size_t n) {
    if (!ctx || !dev || !host) return JX_ERR_INVALID;
    HIPCHK(ctx, hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return JX_OK;
}

int jx_memcpy_d2h(jx_ctx* ctx, void* host, const void* dev, size_t n) {
    if (!ctx || !dev || !host) return JX_ERR_INVALID;
    HIPCHK(ctx, hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return JX_OK;
}

int jx_timing_enable(jx_ctx* ctx, int on) {
    if (!ctx) return JX_ERR_INVALID;
    ctx->timing_on = on != 0;
    ctx->timing_mode = (on == 2) ? 2 : (on ? 1 : 0);
    return JX_OK;
}

int jx_timing_reset(jx_ctx* ctx) {
    if (!ctx) return JX_ERR_INVALID;
    int rc = drain_events(ctx);
    if (rc) return rc;
    memset(&ctx->acc, 0, sizeof(ctx->acc));
    return JX_OK;
}

int jx_timing_get(jx_ctx* ctx, jx_timing* out) {
    if (!ctx || !out) return JX_ERR_INVALID;
    int rc = drain_events(ctx);
    if (rc) return rc;
    *out = ctx->acc;
    return JX_OK;
}

int jx_get_info(jx_ctx* ctx, int32_t* fft_pad, int32_t* chunk, int32_t* band, int32_t* nrow, int64_t* bytes) {
    if (!ctx || !ctx->finalized) return JX_ERR_STATE;
    if (fft_pad) *fft_pad = ctx->P;
    if (chunk) *chunk = ctx->chunk;
    if (band) *band = ctx->K;
    if (nrow) *nrow = ctx->nrow;
    if (bytes) *bytes = ctx->device_bytes;
    return JX_OK;
}

int jx_get_conv_mode(jx_ctx* ctx) {
    if (!ctx || !ctx->finalized) return JX_ERR_STATE;
    return ctx->conv_mode;
}

int jx_get_conv_layout(jx_ctx* ctx, int32_t out[12]) {
    if (!ctx || !ctx->finalized || !out) return JX_ERR_STATE;
    if (ctx->conv_mode != 2) { ctx->err = "layout of the hand-written convolution only"; return JX_ERR_UNSUPPORTED; }
    const JxConv& cv = ctx->cv;
    out[0] = cv.xsym; out[1] = ctx->d.quad; out[2] = cv.NU; out[3] = cv.NJ; out[4] = cv.fir_ld;
    out[5] = ctx->d.quad ? ctx->d.q_nb : cv.S; out[6] = (int32_t)ctx->d.img_ld; out[7] = cv.P;
    out[8] = ctx->lr.r; out[9] = ctx->lrf.r > 0 ? 1 : 0; out[10] = ctx->lrf.r > 0 ? ctx->kact : 0; out[11] = 0;
    return JX_OK;
}

int jx_debug_workspace(jx_ctx* ctx, int which, void** dev, int32_t geom[4]) {
    if (!ctx || !ctx->finalized || !dev || !geom) return JX_ERR_STATE;
    if (ctx->conv_mode != 2) { ctx->err = "work buffers of the hand-written convolution only"; return JX_ERR_UNSUPPORTED; }
    const JxConv& cv = ctx->cv;
    geom[0] = ctx->chunk; geom[3] = cv.xsym;
    if (ctx->odd && which != 0 && which != 6 && which != 12) { ctx->err = "this work buffer does not exist on the odd-side route"; return JX_ERR_UNSUPPORTED; }
    switch (which) {
        case 0: *dev = ctx->d_img; geom[1] = ctx->d.quad ? ctx->d.q_nb : cv.S; geom[2] = (int)ctx->d.img_ld; geom[3] = ctx->d.quad; break;
        case 1: *dev = ctx->d_Y; geom[1] = cv.NU; geom[2] = cv.fir_ld; break;
        case 2: *dev = ctx->d_C; geom[1] = cv.CROWS; geom[2] = cv.fir_ld; break;
        case 3: if (!cv.xsym) { ctx->err = "no column-0 terms in this mode"; return JX_ERR_UNSUPPORTED; }
                *dev = cv.col0; geom[1] = cv.o + 1; geom[2] = cv.NJ; break;
        case 4: *dev = const_cast<int*>(cv.jrow); geom[0] = 1; geom[1] = cv.NJ; geom[2] = 1; break;
        case 5: *dev = const_cast<int*>(cv.umap); geom[0] = 1; geom[1] = cv.S; geom[2] = 1; break;
        case 10: case 11:
            if (ctx->lr.r == 0) { ctx->err = "no combined rows in this mode"; return JX_ERR_UNSUPPORTED; }
            geom[3] = ctx->lr.r;
            if (which == 10) { *dev = ctx->d_Clr; geom[1] = ctx->lr.r; geom[2] = cv.fir_ld; }
            else { *dev = ctx->d_col0lr; geom[1] = cv.o + 1; geom[2] = ctx->lr.r; }
            break;
        case 6: case 7: case 8: case 9:
            if (ctx->lrf.r == 0) { ctx->err = "no fused buffers in this mode"; return JX_ERR_UNSUPPORTED; }
            geom[3] = ctx->lrf.r;
            if (which == 6) { *dev = ctx->d_Rt; geom[0] = cv.Ph; geom[1] = ctx->tKU; geom[2] = ctx->tW; }
            if (which == 7) { *dev = ctx->d_Ct; geom[0] = ctx->tW; geom[1] = cv.Ph; geom[2] = 64; }
            if (which == 8) { *dev = ctx->d_Ct0; geom[0] = ctx->tW; geom[1] = JX_CT0_X; geom[2] = 64; }
            if (which == 9) { *dev = ctx->d_x0t; geom[0] = 1; geom[1] = ctx->tKU; geom[2] = ctx->tW; }
            break;
        case 12:
            if (!ctx->dct_ok) { ctx->err = "no spline arrays in this mode"; return JX_ERR_UNSUPPORTED; }
            *dev = ctx->d_cf; geom[1] = 1; geom[2] = (int)ctx->dct.cf_ws; break;
        default: ctx->err = "unknown work buffer"; return JX_ERR_INVALID;
    }
    return JX_OK;
}

// The default route of even map sides drops the small singular values of the transfer-function weights and the columns
// past the beam's band limit.  What that costs is measured here, once per context, on the current parameter values: the
// extracted row (joxsz_funcs.py:472) through the truncated route against the route with every job and every column (the
// one the beam-convolved-map tap uses).  Returns the largest difference relative to the row's largest entry.
static int truncation_probe(jx_ctx* ctx, double* est) {
    *est = -1.0;
    if (ctx->conv_mode != 2 || ctx->odd || ctx->lrf.r == 0 || ctx->f32) return JX_OK;      // nothing truncated / no exact route beside it
    const jx_config& c = ctx->cfg;
    const int S = c.S, nrow = ctx->nrow;
    int rc;
    if ((rc = ensure_taps(ctx))) return rc;
    if ((rc = ensure_batch(ctx, 1))) return rc;
    std::vector<double> th(c.ndim);
    {
        const std::vector<double> pv = host_vec<double>(ctx, JX_T_PAR_VALS);
        const std::vector<int32_t> ti = host_vec<int32_t>(ctx, JX_T_THAWED_IDX);
        for (int k = 0; k < c.ndim; ++k) th[k] = pv[ti[k]];
    }
    double *tj = nullptr, *tc = nullptr;
    HIPCHK(ctx, hipMalloc((void**)&tj, sizeof(double) * (size_t)ctx->cv.NJ * S));
    HIPCHK(ctx, hipMalloc((void**)&tc, sizeof(double) * (size_t)S * S));
    double* keep_conv = ctx->t_conv;
    ctx->t_conv = tc;
    auto done = [&](int code) { ctx->t_conv = keep_conv; (void)hipStreamSynchronize(ctx->stream); (void)hipFree(tj); (void)hipFree(tc); return code; };
    if (hipMemcpyAsync(ctx->d_theta, th.data(), sizeof(double) * th.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return done(JX_ERR_HIP);
    Taps t;
    t.pp = ctx->t_pp; t.ab = ctx->t_ab; t.y = ctx->t_y; t.row = ctx->t_row; t.bright = ctx->t_bright;
    t.chisq = ctx->t_chisq; t.tprof = ctx->t_tprof; t.xprofs = c.sz_only ? nullptr : ctx->t_xprofs; t.parts = ctx->t_parts;
    std::vector<double> ra(nrow), rb(nrow);
    if ((rc = run_chunk(ctx, ctx->d_theta, ctx->d_logp, 0, 1, t))) return done(rc);
    if (hipMemcpyAsync(ra.data(), ctx->t_row, sizeof(double) * nrow, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return done(JX_ERR_HIP);
    t.conv = tj;
    if ((rc = run_chunk(ctx, ctx->d_theta, ctx->d_logp, 0, 1, t))) return done(rc);
    if (hipMemcpyAsync(rb.data(), ctx->t_row, sizeof(double) * nrow, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return done(JX_ERR_HIP);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return done(JX_ERR_HIP);
    double mx = 0.0, df = 0.0;
    for (int k = 0; k < nrow; ++k) { mx = std::max(mx, std::fabs(rb[k])); df = std::max(df, std::fabs(ra[k] - rb[k])); }
    if (mx > 0.0 && std::isfinite(mx) && std::isfinite(df)) *est = df / mx;
    return done(JX_OK);
}

// the extracted row (joxsz_funcs.py:472) of one walker at the current parameter values, through the context's default route
static int probe_row(jx_ctx* ctx, std::vector<double>& row) {
    const jx_config& c = ctx->cfg;
    int rc;
    if ((rc = ensure_taps(ctx))) return rc;
    if ((rc = ensure_batch(ctx, 1))) return rc;
    std::vector<double> th(c.ndim);
    const std::vector<double> pv = host_vec<double>(ctx, JX_T_PAR_VALS);
    const std::vector<int32_t> ti = host_vec<int32_t>(ctx, JX_T_THAWED_IDX);
    for (int k = 0; k < c.ndim; ++k) th[k] = pv[ti[k]];
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_theta, th.data(), sizeof(double) * th.size(), hipMemcpyHostToDevice, ctx->stream));
    Taps t;
    t.row = ctx->t_row; t.bright = ctx->t_bright; t.chisq = ctx->t_chisq; t.tprof = ctx->t_tprof;
    t.xprofs = c.sz_only ? nullptr : ctx->t_xprofs; t.parts = ctx->t_parts;
    if ((rc = run_chunk(ctx, ctx->d_theta, ctx->d_logp, 0, 1, t))) return rc;
    row.resize(ctx->nrow);
    HIPCHK(ctx, hipMemcpyAsync(row.data(), ctx->t_row, sizeof(double) * ctx->nrow, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return JX_OK;
}

// Odd map sides have no untruncated route inside the same context: the reference row comes from a second, small context
// built with every singular value above rounding kept.
static int odd_reference_row(jx_ctx* ctx, std::vector<double>& row) {
    jx_config cfg = ctx->cfg;
    cfg.max_batch = 16;
    jx_ctx* ref = nullptr;
    int rc = jx_create(&cfg, &ref);
    if (rc) return rc;
    ref->host = ctx->host; ref->have = ctx->have;
    ref->lr_tol_override = 1e-13;
    rc = finalize_impl(ref);
    if (!rc) rc = probe_row(ref, row);
    if (rc) ctx->err = "reference context of the truncation probe: " + ref->err;
    jx_destroy(ref);
    return rc;
}

static int measure_truncation(jx_ctx* ctx, const std::vector<double>& odd_ref, double* est) {
    if (!ctx->odd) return truncation_probe(ctx, est);
    *est = -1.0;
    if (odd_ref.empty()) return JX_OK;
    std::vector<double> row;
    int rc = probe_row(ctx, row);
    if (rc) return rc;
    double mx = 0.0, df = 0.0;
    for (size_t k = 0; k < row.size(); ++k) { mx = std::max(mx, std::fabs(odd_ref[k])); df = std::max(df, std::fabs(row[k] - odd_ref[k])); }
    if (mx > 0.0 && std::isfinite(mx) && std::isfinite(df)) *est = df / mx;
    return JX_OK;
}

int jx_finalize(jx_ctx* ctx) {
    int rc = finalize_impl(ctx);
    if (rc) return rc;
    if (const char* e = getenv("JOXSZ_TRUNC_PROBE")) { if (atoi(e) == 0) return JX_OK; }
    // The default route drops small singular values of the transfer-function weights (and the columns past the beam's
    // band limit).  What that costs is measured on the caller's own beam / transfer function / parameter values; beyond
    // JX_TRUNC_BOUND of the row's largest entry the context is rebuilt with a cut a hundred times tighter, until the bound
    // holds or every term above rounding is kept.
    double bound = JX_TRUNC_BOUND;
    if (const char* e = getenv("JOXSZ_TRUNC_BOUND")) { const double v = atof(e); if (v > 0.0) bound = v; }
    std::vector<double> odd_ref;
    if (ctx->conv_mode == 2 && ctx->odd && !ctx->f32 && ctx->lr_tol > 2e-13 && (rc = odd_reference_row(ctx, odd_ref))) return rc;
    if ((rc = measure_truncation(ctx, odd_ref, &ctx->trunc_est))) return rc;
    while (ctx->trunc_est > bound && ctx->lr_tol > 2e-13 && !getenv("JOXSZ_LOWRANK_TOL")) {
        jx_ctx* fresh = nullptr;
        if ((rc = jx_create(&ctx->cfg, &fresh))) return rc;
        fresh->host = ctx->host; fresh->have = ctx->have;
        fresh->lr_tol_override = std::max(1e-13, ctx->lr_tol * 1e-2);
        fresh->trunc_retried = ctx->trunc_retried + 1;
        rc = finalize_impl(fresh);
        if (!rc) rc = measure_truncation(fresh, odd_ref, &fresh->trunc_est);
        if (rc) { ctx->err = "finalize pass with a tighter singular-value cut: " + fresh->err; jx_destroy(fresh); return rc; }
        std::swap(*ctx, *fresh);                              // the caller's handle now owns the tighter build
        jx_destroy(fresh);
    }
    return JX_OK;
}

int jx_get_truncation(jx_ctx* ctx, double out[4]) {
    if (!ctx || !ctx->finalized || !out) return JX_ERR_STATE;
    out[0] = ctx->lr_tol; out[1] = ctx->trunc_est; out[2] = (double)ctx->lr.r; out[3] = (double)ctx->trunc_retried;
    return JX_OK;
}

void jx_destroy(jx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) (void)jx_comm_destroy(ctx);
    for (auto& kv : ctx->plans) {
        if (kv.second.beam_fwd) rocfft_plan_destroy(kv.second.beam_fwd);
        if (kv.second.beam_inv) rocfft_plan_destroy(kv.second.beam_inv);
        if (kv.second.tf_fwd) rocfft_plan_destroy(kv.second.tf_fwd);
    }
    if (ctx->info) rocfft_execution_info_destroy(ctx->info);
    for (auto& es : ctx->ev_inflight) for (int k = 0; k < 7; ++k) (void)hipEventDestroy(es.e[k]);
    for (auto& es : ctx->ev_free) for (int k = 0; k < 7; ++k) (void)hipEventDestroy(es.e[k]);
    for (void* p : ctx->dev_allocs) (void)hipFree(p);
    for (void* p : ctx->samp_buf) if (p) (void)hipFree(p);
    if (ctx->d_work) (void)hipFree(ctx->d_work);
    if (ctx->d_theta) (void)hipFree(ctx->d_theta);
    if (ctx->d_logp) (void)hipFree(ctx->d_logp);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (--g_rocfft_refs == 0) rocfft_cleanup();
    delete ctx;
}

}  // extern "C"
