// C-ABI implementation (include/joxsz_hip.h) for gfx950: context, uploads, table building and the per-chunk launch
// sequence.  The SZ side of the log-posterior (joxsz_funcs.py:457-472) behind the Compton-y profile:
//   * the exact form (jx_exact.hpp; default since round 5): the spline, the map, the beam convolution, the transfer function and the row
//     extraction as ONE constant operator on the spline ordinates, built on the host at jx_finalize (jx_tables.hpp: every pixel, every
//     radius, nothing truncated) -- per step: jx_walker2_kernel -> jx_ordrow_kernel -> jx_rowsum_tail_kernel;
//   * the contracted forms of rounds 3-4 (jx_mix.hpp; option JOXSZ_MIX_FORM=legacy|lowrank|full, kept for one round): low-rank (stage 1 on
//     the vector units + stage 2 on the matrix cores) or full (one matrix-core product fed by the sample evaluation), with their
//     truncation guard;
//   * the literal sequence, "rocFFT sequence" below (forward transforms -> beam multiply -> inverse -> window -> forward transforms, on the
//     hand-written LDS transforms of jx_fft.hpp or on rocFFT's plans): the reference's lines executed pass by pass -- the
//     independent cross-check, the fallback for inputs without the mirror structure, and, as a small reference facility inside every other
//     context, the source of the beam-convolved-map tap and of jx_audit.
// Every switch is an option of the context (jx_set_option; the process environment is the default of each name), read once, in
// jx_finalize; nothing on the launch path calls getenv.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/joxsz_hip.h"
#include "jx_kernels.hpp"
#include "jx_mix.hpp"
#include "jx_exact.hpp"
#include "jx_fft.hpp"
#include "jx_tables.hpp"

namespace {

struct Plan3 {
    rocfft_plan beam_fwd = nullptr, beam_inv = nullptr, tf_fwd = nullptr;         // 2-D plans (JOXSZ_FFT_COLUMNS=rocfft, or a side that is not 2^a 3^b 5^c)
    rocfft_plan row_fwd = nullptr, row_inv = nullptr, row_tf = nullptr;           // batched 1-D row plans beside the column kernels of jx_fft.hpp
};

struct EvSet {
    hipEvent_t e[6];
    int walkers;
    bool op;                           // operator route: only e[0], e[1], e[5] were recorded
    bool p1only = false;               // timing modes 2..4: only e[2], e[3] (around one kernel of the step) were recorded
    int p1stage = 2;                   //   which one: 2 stage 1 (contracted forms) / the ordinate product (exact form), 3 the per-walker kernel, 4 the row product + tail (exact form)
    bool exact = false;                // exact form: e[0], e[1], e[2], e[5] were recorded (per-walker kernel, ordinate product, row product + tail)
};

// rocFFT sequence: constants in the padded full-image layout, work buffers for `cap` walkers, plans per batch size
struct FftBack {
    bool ready = false;
    int cap = 0, P = 0, Ph = 0;
    JxDev d;
    int map_threads = 512;
    size_t map_lds = 0;
    double *img = nullptr, *conv = nullptr;
    double2 *spec = nullptr, *tfspec = nullptr;
    // hand-written column passes (jx_fft.hpp): rows of the padded image kept (S: the zero rows are never stored), leading dimensions of the
    // row spectra (multiples of 8 complex), columns per block, the two transforms, the tables in those leading dimensions, Z of the tail
    bool cols = false, rows_custom = false;
    int rows = 0, ldc = 0, ldt = 0;
    JxFft fP{}, fS{};
    double2 *bhat_p = nullptr, *htab_p = nullptr;
    double* zbuf = nullptr;
    std::map<int, Plan3> plans;
    rocfft_execution_info info = nullptr;
    void* work = nullptr;
    size_t work_cap = 0;
    std::vector<void*> allocs;
};

// contracted route: tables and work buffers (freed and rebuilt when the truncation probe asks for a tighter cut)
struct MixBack {
    bool ready = false;
    int form = 0;                      // 0 low-rank (stage 1 + stage 2), 1 full operator on the samples (round 3-4, JOXSZ_MIX_FORM=legacy|lowrank|full); 2 exact (default)
    // exact form (jx_exact.hpp): the ordinate product's operator Ty and first k-steps, the row operator Opk, the ordinates y
    int Nk = 0, Nkp = 0, x_nxt = 0, x_ng = 0, x_ng_use = 0, x_npair = 0, x_nSj = 0, x_ldpp = 0, x_nfold = 0;
    bool x_lean = true;                // JOXSZ_X_FOLD=0 clears it: the timed path then stores the ordinates and forms every tile, like a call with taps
    double *x_Wfk = nullptr;           // odd number of ordinate tiles: the last one's share of the row as an operator on the profile (timed path; jxt::exact_fold_layout)
    double *x_Typ = nullptr, *x_Opk = nullptr, *x_y = nullptr, *x_cf = nullptr, *x_P = nullptr, *x_ppi = nullptr;
    bool x_pairwise = true;            // JOXSZ_X_PAIRWISE=0: the ordinates read back by one block per 16 walkers (jx_rowop_tail_kernel)
    JxMix mx{};
    JxOpg og{};
    JxOpg og_u{};                      // the product restricted to the outputs the tail reads (nxt_u tiles per block, ksplit_u slices); og: every output (taps)
    int nxt_u = 0, ksplit_u = 0, ksplit_u_force = 0;
    bool has_u = false, last_was_u = false;
    int NU_full = 0;                   // distinct rows of the quadrant (mx.NU: the ones stage 1 evaluates)
    std::vector<int> sub;              // indices of the rows (= columns) stage 1 evaluates; empty: all
    int RT = 0, nxt = 0, r = 0, ns = 0, ksteps = 0, wpb = 4, wpb_force = 0, ksplit_force = 0, last_ksplit = 1, dbg = 0;
    bool mfma = false;                 // stage 1 on the fp64 matrix cores (R <= 16: jx_rowmix_mfma_kernel)
    int r_tol = 0;                     // terms above the singular-value cut (r < r_tol: capped to one 16-row tile)
    long long tW = 0;
    int ncol = 0;
    double tol = 0.0;
    double *cft = nullptr, *Dt = nullptr, *Pt = nullptr;
    std::vector<void*> allocs;
    int64_t bytes = 0;
};

}  // namespace

struct jx_ctx {
    jx_config cfg;
    bool finalized = false;
    std::string err;
    std::string devname;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;  // created by jx_create; `stream` may point at a caller's stream instead (jx_set_stream)

    std::map<std::string, std::string> opts;   // jx_set_option (the process environment is the default of every name)

    // host copies of the uploaded tensors
    std::vector<std::vector<unsigned char>> host;
    std::vector<bool> have;

    // derived sizes
    int nrow = 0, nt = 0, Sh = 0, K = 0, chunk = 0, num_cu = 256;
    int nrow_use = 0;                  // outputs of the extracted row the data-radii spline reads with a weight above JX_PRUNE_TOL of its largest (<= nrow)
    bool subsample = true;             // JOXSZ_MIX_SUBSAMPLE=0: stage 1 evaluates every distinct map sample (default: a tensor sub-grid, the rest by interpolation folded into the operators)
    int sub_u0 = 40, sub_u1 = 160, sub_npts = 14;   // full resolution below u0 pixels from the axis, every second row up to u1, every fourth up to 2 u1, every eighth beyond; interpolation points
    bool prune = true;                 // JOXSZ_PRUNE_OUTPUTS=0: the matrix-core product computes every output of the row, read or not
    int64_t device_bytes = 0;
    int conv_mode = 1;                 // 1 rocFFT sequence, 2 contracted route

    // device constants
    std::vector<void*> dev_allocs;
    JxDev d;
    double* d_par_vals = nullptr;
    int map_threads = 512;
    size_t map_lds = 0;
    bool dmat_mirror = false;
    bool map_ok = true;                // the Abel + map kernel fits its spline in LDS (radial grids up to ~1690 points); without it: no map taps, no rocFFT
                                       // reference facility (guard), profile taps from the matrix product's own arrays
    std::vector<double> h_Qtab;        // [qn][qn] pixel radii of the quadrant (host; table builds)
    int qn = 0;
    std::vector<double> h_G;           // [N][N] moment operator of the mirrored spline (host; table builds)

    // per-walker scalars (chunk capacity)
    double *d_base = nullptr, *d_cfac = nullptr, *d_sz0 = nullptr;
    double* d_xr = nullptr;            // [chunk][2] Cash log-likelihood and reject flag of the two-block form of the per-walker kernel
    bool prep_split = true;            // JOXSZ_PREP_SPLIT=0: one block per walker in the per-walker kernel (as in every call with taps)
    bool prep_lean = true;             // JOXSZ_PREP_LEAN=0: the two-block form runs in jx_prep_kernel itself instead of jx_walker2_kernel (same bits)
    double* d_img = nullptr;           // contracted route: quadrant of the Compton-y map (y_2d tap only; allocated on first use)
    // spline arrays as one matrix product (jx_abel_gemm_kernel)
    double* d_Tm = nullptr; int tm_ld = 0, tm_ntile = 0, tm_npair = 0;
    double* d_ppc = nullptr;           // [chunk][N] prep kernel -> jx_abel_gemm_kernel
    // radial sub-grid of the spline-array product (DESIGN of round 4, 6.3): Tm through the interpolation from the kept radii, their indices, the first k-step of every column tile
    bool ag_sub = true;                // JOXSZ_AG_SUBSAMPLE=0: every radius of the profile; "u0,u1,npts": another sub-grid
    int ag_u0 = 64, ag_u1 = 256, ag_npts = 18;
    bool ag_sub_on = false;            // in use (set at the end of finalize_impl; the guard of jx_finalize may take it away)
    int ag_ns = 0, ag_removed = 0;
    double* d_Tm_s = nullptr; int* d_rsub = nullptr; int* d_tks = nullptr;
    std::vector<int> h_rsub;
    bool abel_gemm = true;
    bool prep_fastmath = true;         // JOXSZ_PREP_FASTMATH=0: the per-walker kernel's exp / log from the device library instead of jx_fastmath.hpp's tables
    bool ag_single = false;            // JOXSZ_AG_SINGLE=1: one column tile per wave (twice the blocks) in the spline-array product
    int ag_narrow = -1;                // JOXSZ_AG_NARROW: 16-walker blocks of the spline-array product for every launch (1) / never (0)
    bool f32 = false;                  // jx_config.dtype >= 1: fp32 spline arrays, fp32 evaluation of the map samples
    bool f32c = false;                 // jx_config.dtype == 2: fp32 arithmetic in stage 1 and stage 2 as well (packed fp32 FMAs, fp32 matrix cores)

    MixBack mix;
    FftBack fft;                       // conv_mode 1: the back end (cap = chunk); conv_mode 2: reference facility (built on first use)

    // truncation guard
    double trunc_est[3] = {-1.0, -1.0, -1.0};   // measure_truncation: row at the current values, row over the probe points, SZ log-likelihood
    double trunc_bound = 1e-9, trunc_bound_ll = 1e-8;
    int trunc_retried = 0, trunc_points = 0, trunc_uncapped = 0, trunc_unsub = 0;   // rebuilds in all; of which: the cap on the rank taken away
    bool tol_pinned = false;           // JOXSZ_LOWRANK_TOL given: the guard measures but never overrides
    int form_force = 2;                // JOXSZ_MIX_FORM: 2 exact (default); the contracted forms of rounds 3-4: -1 legacy (the cheaper of the two), 0 lowrank, 1 full
    int map_pair = 1;                  // full-map kernel: two walkers per block (JOXSZ_MAP_PAIR=0: one)
    int usplit = 2;                    // pieces a map column is walked in by stage 1 (JOXSZ_MIX_USPLIT: 1..4; a setting, never a function of the launch)
    bool usplit_forced = false;        // (otherwise plan_mix picks it from the columns stage 1 walks and the chunk)
    int rank_cap = 16;                 // low-rank form: a rank within JX_MIX_CAP_REACH terms above this is cut to it, so that stage 1 fits one 16-row matrix-core
                                       // tile (the guard measures the result and takes the cap away when it costs accuracy; JOXSZ_MIX_RANKCAP=0: never)
    bool mix_mfma = false;             // JOXSZ_MIX_MFMA=1: stage 1 on the fp64 matrix cores when R <= 16 (measured slower than the vector-unit kernel: DESIGN of round 4, 6.1)

    // batch staging for the host-pointer API
    double *d_theta = nullptr, *d_logp = nullptr;
    double *h_theta = nullptr, *h_logp = nullptr;   // pinned host staging of jx_eval (a pageable hipMemcpyAsync stages and synchronises by itself)
    double *hd_theta = nullptr, *hd_logp = nullptr; // the same two buffers as the device sees them (mapped, coherent)
    int eval_direct = 3;                            // JOXSZ_EVAL_DIRECT bit 0: the tail writes the log-probabilities into the host buffer itself; bit 1: the per-walker kernel reads theta from it
    int batch_cap = 0;
    // taps (chunk capacity, allocated on first use)
    double *t_pp = nullptr, *t_ab = nullptr, *t_y = nullptr, *t_row = nullptr, *t_bright = nullptr,
           *t_chisq = nullptr, *t_tprof = nullptr, *t_xprofs = nullptr, *t_parts = nullptr, *t_integ = nullptr, *t_y2d = nullptr;
    void* samp_buf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // jx_sample work buffers (grow-only)
    size_t samp_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    // collapsed route (jx_set_route): Gt [N][g_ld], row j = map row of the unit pressure profile e_j
    int route = JX_ROUTE_MAP;
    double* d_G = nullptr;
    double* d_pp = nullptr;            // [op_cap][N] pressure profiles, prep kernel -> operator kernel
    double *d_base_op = nullptr, *d_cfac_op = nullptr, *d_sz0_op = nullptr;
    double* d_rows = nullptr;          // [op_cap / 32][nrow][32] G pp of large launches (jx_operator_mfma_kernel)
    int op_cap = 0, g_ld = 0;
    bool op_narrow = false;            // JOXSZ_OP_NARROW: the small-launch operator kernel for every launch size

    ncclComm_t comm = nullptr;         // RCCL communicator of this rank (jx_comm_init_rank)
    int comm_rank = 0, comm_size = 1;
    // the per-walker kernel beside the SZ chain (contracted route): second stream, joined in front of the tail
    bool side_on = false;              // JOXSZ_SIDE_STREAM=1: the per-walker kernel on a second stream beside the SZ chain (measured: no gain, profiles/r04_side_stream_timeline.log)
    int side_fork = 0;                 // where the side stream forks: 0 behind the profile kernel, 1 behind stage 1 (JOXSZ_SIDE_FORK)
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_side = nullptr, ev_tail = nullptr;
    // overlapped gather (jx_comm_set_overlap): the collectives of the communicator run on a stream of their own, ordered
    // behind the evaluation by an event; a send buffer still being gathered holds back the next evaluation that writes it
    bool comm_overlap = false;
    hipStream_t comm_stream = nullptr;
    struct GatherSlot { const char* send = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; bool busy = false; };
    GatherSlot gslot[4];
    hipEvent_t ev_fork = nullptr;      // compute stream -> collective stream
    std::vector<std::pair<hipEvent_t, hipEvent_t>> gt_inflight, gt_free;   // events around every all-gather (its own duration)
    double gather_ms = 0.0; long long gather_calls = 0;

    // timing
    bool timing_on = false;
    int timing_mode = 0;               // jx_timing_enable: 1 = every stage, 2 = only the events around the time-dominant kernel
    std::vector<EvSet> ev_inflight, ev_free;
    jx_timing acc{};
};

static int g_rocfft_refs = 0;

static const char* opt_str(const jx_ctx* ctx, const char* name) {
    if (ctx) { const auto it = ctx->opts.find(name); if (it != ctx->opts.end()) return it->second.empty() ? nullptr : it->second.c_str(); }
    const char* e = getenv(name);
    return (e && e[0]) ? e : nullptr;
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

// allocations: `list` = the owner's list (context, or a back end that can be torn down on its own)
template <typename T>
static int dev_put_l(jx_ctx* ctx, std::vector<void*>& list, const T* src, size_t count, T** out) {
    void* p = nullptr;
    HIPCHK(ctx, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    list.push_back(p);
    ctx->device_bytes += (int64_t)(count * sizeof(T));
    if (count) HIPCHK(ctx, hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *out = (T*)p;
    return JX_OK;
}
template <typename T>
static int dev_new_l(jx_ctx* ctx, std::vector<void*>& list, size_t count, T** out, bool zero = false) {
    void* p = nullptr;
    HIPCHK(ctx, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    list.push_back(p);
    ctx->device_bytes += (int64_t)(count * sizeof(T));
    if (zero) HIPCHK(ctx, hipMemset(p, 0, count * sizeof(T)));
    *out = (T*)p;
    return JX_OK;
}
template <typename T> static int dev_put(jx_ctx* ctx, const T* src, size_t count, T** out) { return dev_put_l(ctx, ctx->dev_allocs, src, count, out); }
template <typename T> static int dev_new(jx_ctx* ctx, size_t count, T** out, bool zero = false) { return dev_new_l(ctx, ctx->dev_allocs, count, out, zero); }

template <typename T>
static std::vector<T> host_vec(jx_ctx* ctx, int id) {
    const auto& b = ctx->host[id];
    std::vector<T> v(b.size() / sizeof(T));
    if (!v.empty()) memcpy(v.data(), b.data(), b.size());
    return v;
}

// Every switch of the library: set per context with jx_set_option (before jx_finalize for those it reads, which is all but the two
// SAMPLE_ ones), with the process environment as the default of each.  The list is the one include/joxsz_hip.h documents
// (tests/test_abi.py holds the two and the uses in this file together).
static const char* const kOptions[] = {
    "JOXSZ_CONV", "JOXSZ_MIX_FORM", "JOXSZ_X_PAIRWISE", "JOXSZ_X_FOLD", "JOXSZ_PRUNE_OUTPUTS", "JOXSZ_CHUNK", "JOXSZ_FFT_PAD", "JOXSZ_FFT_COLUMNS", "JOXSZ_FFT_ROWS", "JOXSZ_MAP_SPLIT", "JOXSZ_MAP_PAIR",
    "JOXSZ_EVAL_DIRECT", "JOXSZ_PREP_SPLIT", "JOXSZ_PREP_LEAN", "JOXSZ_PREP_POW", "JOXSZ_PREP_FASTMATH", "JOXSZ_OP_NARROW", "JOXSZ_SAMPLE_FUSED", "JOXSZ_SAMPLE_VIRTUAL_RANKS",
    // the contracted forms of rounds 3-4 (JOXSZ_MIX_FORM=legacy|lowrank|full)
    "JOXSZ_LOWRANK_TOL", "JOXSZ_TRUNC_PROBE", "JOXSZ_TRUNC_BOUND", "JOXSZ_MIX_SUBSAMPLE", "JOXSZ_MIX_RANKCAP", "JOXSZ_MIX_MFMA", "JOXSZ_MIX_USPLIT", "JOXSZ_MIX_WPB",
    "JOXSZ_MIX_KSPLIT", "JOXSZ_MIX_KSPLIT_USE", "JOXSZ_ABEL_GEMM", "JOXSZ_AG_NARROW", "JOXSZ_AG_SINGLE", "JOXSZ_AG_SUBSAMPLE", "JOXSZ_SIDE_STREAM", "JOXSZ_SIDE_FORK",
    // diagnostic build only (make ABLATIONS=1): timing experiments, results are wrong
    "JOXSZ_DBG", "JOXSZ_MIX_DBG", "JOXSZ_X_STAMPS", "JOXSZ_P_STAMPS",
};
static const char* opt_str(const jx_ctx* ctx, const char* name);

// librccl, loaded on first use (dlopen): the functions of the all-gather path only
namespace {
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
RcclApi g_rccl;
std::string g_last_global_error;       // failures of calls that have no context (jx_comm_unique_id)

bool rccl_load() {
    if (g_rccl.h) return true;
    void* h = nullptr;
    std::string why;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        const char* e = dlerror();                              // (one call: it clears the error it returns)
        why = std::string(name) + ": " + (e ? e : "unknown error");
    }
    if (!h) { g_rccl.err = "dlopen(librccl): " + why; return false; }
    RcclApi api;
    bool ok = true;
#define JX_SYM(field, sym) if (ok) { *(void**)(&api.field) = dlsym(h, sym); if (!api.field) { g_rccl.err = std::string("librccl lacks ") + sym; ok = false; } }
    JX_SYM(GetUniqueId, "ncclGetUniqueId") JX_SYM(CommInitRank, "ncclCommInitRank") JX_SYM(AllGather, "ncclAllGather")
    JX_SYM(AllReduce, "ncclAllReduce") JX_SYM(CommCount, "ncclCommCount") JX_SYM(CommDestroy, "ncclCommDestroy") JX_SYM(GetErrorString, "ncclGetErrorString")
#undef JX_SYM
    if (!ok) { dlclose(h); return false; }
    api.h = h;
    g_rccl = api;
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

// ---------------------------------------------------------------------------------------------------------------------
// Abel + map kernel launches (jx_abel_map_sym_kernel / jx_abel_map_kernel): full map, quadrant, or phases 1-3 only
// ---------------------------------------------------------------------------------------------------------------------
static size_t map_lds_need(const JxDev& d, int threads) {
    const int N = d.N, S = d.S;
    size_t scratch = (d.pairw == 2) ? JX_MAP_SCRATCH2_DOUBLES(N) : JX_MAP_SCRATCH_DOUBLES(N);
    if (d.fast_map) scratch = std::max(scratch, (size_t)(threads / 64) * ((S + 3) & ~1));
    const size_t dbl = JX_MAP_FIXED_DOUBLES(N) + scratch + (d.pairw == 2 ? 4 * JX_MAP_NE(N) + 8 : 0);
    return dbl * sizeof(double);
}

// picks pairw and the block size for `d` so that the kernel's LDS fits; returns false when the radial grid is too long
static bool map_geometry(JxDev& d, int want_threads, int* threads_out, size_t* lds_out, bool pair_full = false) {
    const size_t LDS_MAX = 160 * 1024;
    int threads = want_threads;
    d.pairw = 1;
    if (d.quad) {
        d.pairw = 2;                                           // two walkers per block share every table entry, while two such blocks fit a CU
        if (map_lds_need(d, threads) > (LDS_MAX - 2048) / 2) d.pairw = 1;
    } else if (pair_full && d.fast_map) {
        // full rows: two walkers per block take the Abel weights out of the L2 once (2 MB per walker otherwise: the stream
        // that competes with the map's stores); one such block per CU
        d.pairw = 2;
        if (map_lds_need(d, threads) > LDS_MAX - 1024) d.pairw = 1;
    }
    while (threads > 64 && map_lds_need(d, threads) > LDS_MAX - 1024) threads /= 2;
    const size_t lds = map_lds_need(d, threads);
    if (lds > LDS_MAX - 1024) return false;
    *threads_out = threads; *lds_out = lds;
    return true;
}

static void launch_map(hipStream_t st, const JxDev& dm_in, int threads, size_t lds, const double* theta_dev, int w0, int n, double* img,
                       double* tpp, double* tab, double* ty, bool coef_only) {
    JxDev dm = dm_in;
    dm.nlaunch = n;
    const bool vec2 = (dm.S % 2 == 0) && (dm.img_ld % 2 == 0);
    const int npw = (dm.pairw == 2) ? 2 : 1;
    if (coef_only) dm.map_split = 1;
    const dim3 grid(((n + npw - 1) / npw) * dm.map_split), block(threads);
    if (dm.fast_map) {
        const int nait = (dm.q_na + 63) / 64;
#define JX_SYM_LAUNCH(V, NA) hipLaunchKernelGGL((jx_abel_map_sym_kernel<V, NA>), grid, block, lds, st, dm, theta_dev, w0, img, tpp, tab, ty)
        if (vec2) { if (nait <= 3) JX_SYM_LAUNCH(true, 3); else if (nait <= 5) JX_SYM_LAUNCH(true, 5); else JX_SYM_LAUNCH(true, 9); }
        else      { if (nait <= 3) JX_SYM_LAUNCH(false, 3); else if (nait <= 5) JX_SYM_LAUNCH(false, 5); else JX_SYM_LAUNCH(false, 9); }
#undef JX_SYM_LAUNCH
    } else {
        if (vec2) hipLaunchKernelGGL(jx_abel_map_kernel<true>, grid, block, lds, st, dm, theta_dev, w0, img, tpp, tab, ty);
        else hipLaunchKernelGGL(jx_abel_map_kernel<false>, grid, block, lds, st, dm, theta_dev, w0, img, tpp, tab, ty);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// rocFFT sequence
// ---------------------------------------------------------------------------------------------------------------------
static void fft_teardown(FftBack& fb) {
    for (auto& kv : fb.plans) {
        if (kv.second.beam_fwd) rocfft_plan_destroy(kv.second.beam_fwd);
        if (kv.second.beam_inv) rocfft_plan_destroy(kv.second.beam_inv);
        if (kv.second.tf_fwd) rocfft_plan_destroy(kv.second.tf_fwd);
        if (kv.second.row_fwd) rocfft_plan_destroy(kv.second.row_fwd);
        if (kv.second.row_inv) rocfft_plan_destroy(kv.second.row_inv);
        if (kv.second.row_tf) rocfft_plan_destroy(kv.second.row_tf);
    }
    fb.plans.clear();
    if (fb.info) { rocfft_execution_info_destroy(fb.info); fb.info = nullptr; }
    for (void* p : fb.allocs) (void)hipFree(p);
    fb.allocs.clear();
    if (fb.work) { (void)hipFree(fb.work); fb.work = nullptr; fb.work_cap = 0; }
    fb.ready = false;
}

static int fft_factor(int n, int* radix) { return jxt::fft_radices(n, radix, JX_FFT_MAXPASS); }

static int fft_transform(jx_ctx* ctx, FftBack& fb, int n, JxFft& f) {
    f.n = n;
    f.npass = fft_factor(n, f.radix);
    for (int p = 0, ns = 1; p < f.npass; ns *= f.radix[p++]) {
        f.ns[p] = ns;
        f.tstep[p] = p ? n / (ns * f.radix[p]) : 0;
        f.magic[p] = ns > 1 ? (unsigned)((((unsigned long long)1 << 32) + ns - 1) / ns) : 0u;
    }
    std::vector<double> root((size_t)2 * n);
    for (int m = 0; m < n; ++m) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)n;
        root[2 * m] = (double)cosl(a); root[2 * m + 1] = (double)sinl(a);
    }
    double* p;
    int rc;
    if ((rc = dev_put_l(ctx, fb.allocs, root.data(), root.size(), &p))) return rc;
    f.root = reinterpret_cast<const double2*>(p);
    return JX_OK;
}

// the column kernels come per rows-per-lane bound NU = ceil(length / 64): 8 columns per block up to 640, 4 beyond
#define JX_FFT_NU_OF(n) (((n) + 63) / 64)
#define JX_FFT_DISPATCH(nu, X) do {                                                                                     \
        const int nu_ = (nu);                                                                                           \
        if (nu_ <= 4) { X(8, 4); } else if (nu_ <= 8) { X(8, 8); } else if (nu_ <= 9) { X(8, 9); } else if (nu_ <= 10) { X(8, 10); }   \
        else if (nu_ <= 16) { X(4, 16); } else if (nu_ <= 17) { X(4, 17); } else { X(4, 20); }                         \
    } while (0)
static inline int fft_cb(int n) { return JX_FFT_NU_OF(n) <= 10 ? 8 : 4; }
// the roots of unity go to LDS when two blocks still fit a compute unit with them
static inline int fft_roots_in_lds(int n) { return JX_FFT_SMALL(JX_FFT_NU_OF(n)) ? 1 : 0; }

static void fft_launch_beam(FftBack& fb, int walkers, hipStream_t st, int S) {
    const int P = fb.P, r = fft_roots_in_lds(P);
#define X(CB, NU) hipLaunchKernelGGL((jx_fft_beam_cols_kernel<CB, NU>), dim3(walkers, fb.ldc / CB), dim3(64 * CB), JX_FFT_LDS_BYTES(P, CB, r), st, fb.fP, fb.spec, fb.bhat_p, S, fb.ldc)
    JX_FFT_DISPATCH(JX_FFT_NU_OF(P), X);
#undef X
}

static void fft_launch_tf(FftBack& fb, int walkers, hipStream_t st, int Sh) {
    const int S = fb.fS.n, r = fft_roots_in_lds(S);
#define X(CB, NU) hipLaunchKernelGGL((jx_fft_tf_cols_kernel<CB, NU>), dim3(walkers, fb.ldt / CB), dim3(64 * CB), JX_FFT_LDS_BYTES(S, CB, r), st, fb.fS, fb.tfspec, fb.htab_p, fb.ldt, Sh, fb.zbuf, r)
    JX_FFT_DISPATCH(JX_FFT_NU_OF(S), X);
#undef X
}

// row kernels: 8 waves (pairs of rows) per block for the forward pass, 7 for the fused inverse + window + forward one (its second table of
// roots takes the room of the eighth); 4 where the roots stay in global memory
#define JX_FFT_ROWS_DISPATCH(nu, WS, X) do {                                                                            \
        const int nu_ = (nu);                                                                                           \
        if (nu_ <= 4) { X(WS, 4); } else if (nu_ <= 8) { X(WS, 8); } else if (nu_ <= 9) { X(WS, 9); } else if (nu_ <= 10) { X(WS, 10); }   \
        else if (nu_ <= 16) { X(4, 16); } else if (nu_ <= 17) { X(4, 17); } else { X(4, 20); }                         \
    } while (0)

static void fft_launch_rows_fwd(FftBack& fb, int walkers, hipStream_t st, int S) {
    const int P = fb.P, R = walkers * S, small = fft_roots_in_lds(P);
#define X(WPB, NU) hipLaunchKernelGGL((jx_fft_rows_fwd_kernel<WPB, NU>), dim3(((R + 1) / 2 + WPB - 1) / WPB), dim3(64 * WPB), JX_FFT_ROWS_LDS_BYTES(P, WPB, small ? P : 0), st, \
                                      fb.fP, fb.img, fb.spec, S, fb.ldc, R)
    JX_FFT_ROWS_DISPATCH(JX_FFT_NU_OF(P), 8, X);
#undef X
}

static void fft_launch_rows_inv_tf(FftBack& fb, int walkers, hipStream_t st, int S, bool want_conv) {
    const int P = fb.P, R = walkers * S, small = fft_roots_in_lds(P);
#define X(WPB, NU) hipLaunchKernelGGL((jx_fft_rows_inv_tf_kernel<WPB, NU>), dim3(((R + 1) / 2 + WPB - 1) / WPB), dim3(64 * WPB), JX_FFT_ROWS_LDS_BYTES(P, WPB, small ? P + S : 0), st, \
                                      fb.fP, fb.fS, fb.spec, want_conv ? fb.conv : (double*)nullptr, fb.tfspec, fb.ldc, fb.ldt, R)
    JX_FFT_ROWS_DISPATCH(JX_FFT_NU_OF(P), 7, X);
#undef X
}

static int fft_kernels(jx_ctx* ctx, FftBack& fb) {
    // the attribute belongs to the kernel instance, not to the context: another context of the process may launch the same instance with a
    // longer transform, so every instance is allowed the whole LDS of a compute unit (as the Abel + map kernel is)
    const int S = ctx->cfg.S, lds = 160 * 1024 - 1024;
#define X(CB, NU) HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_fft_beam_cols_kernel<CB, NU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds))
    JX_FFT_DISPATCH(JX_FFT_NU_OF(fb.P), X);
#undef X
#define X(CB, NU) HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_fft_tf_cols_kernel<CB, NU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds))
    JX_FFT_DISPATCH(JX_FFT_NU_OF(S), X);
#undef X
#define X(WPB, NU) HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_fft_rows_fwd_kernel<WPB, NU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds))
    JX_FFT_ROWS_DISPATCH(JX_FFT_NU_OF(fb.P), 8, X);
#undef X
#define X(WPB, NU) HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_fft_rows_inv_tf_kernel<WPB, NU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds))
    JX_FFT_ROWS_DISPATCH(JX_FFT_NU_OF(fb.P), 7, X);
#undef X
    return JX_OK;
}

// tables (beam spectrum, transfer-function row table) and work buffers for `cap` walkers; P = padded side
static int fft_setup(jx_ctx* ctx, FftBack& fb, int cap, int P) {
    const jx_config& c = ctx->cfg;
    const int S = c.S, B = c.B;
    int rc;
    fb.cap = cap; fb.P = P; fb.Ph = P / 2 + 1;
    fb.d = ctx->d;
    JxDev& d = fb.d;
    d.P = P; d.Ph = fb.Ph;
    // column passes: hand-written (jx_fft.hpp) when both sides are 2^a 3^b 5^c and short enough for one column per 32 / 64 threads
    {
        int rx[JX_FFT_MAXPASS];
        const char* e = opt_str(ctx, "JOXSZ_FFT_COLUMNS");
        const bool want = !(e && !strcmp(e, "rocfft"));
        fb.cols = want && fft_factor(P, rx) > 1 && fft_factor(S, rx) > 1 && std::max(P, S) <= JX_FFT_MAX_PER_THREAD * 64 && P % 2 == 0;
        const char* er = opt_str(ctx, "JOXSZ_FFT_ROWS");
        fb.rows_custom = fb.cols && !(er && !strcmp(er, "rocfft"));
        fb.rows = fb.cols ? S : P;
        fb.ldc = (fb.Ph + 7) & ~7; fb.ldt = (ctx->Sh + 7) & ~7;
    }
    d.quad = 0; d.img_ld = P; d.img_ws = (long long)fb.rows * P; d.cf_out = nullptr; d.xcol = nullptr;
    if (!map_geometry(d, 512, &fb.map_threads, &fb.map_lds, ctx->map_pair != 0)) { ctx->err = "radial grid too long for the LDS-resident spline"; return JX_ERR_UNSUPPORTED; }
    {
        std::vector<double> beam = host_vec<double>(ctx, JX_T_BEAM_2D), bh, H;
        jxt::beam_spectrum(beam, B, P, c.step * c.step / ((double)P * (double)P), bh);
        double* p;
        if ((rc = dev_put_l(ctx, fb.allocs, bh.data(), bh.size(), &p))) return rc; d.bhat = p;
        jxt::tf_row_table(host_vec<double>(ctx, JX_T_FILTERING), S, H);
        if ((rc = dev_put_l(ctx, fb.allocs, H.data(), H.size(), &p))) return rc; d.htab = p;
        if (fb.cols) {
            // the two tables one column after the other (the lanes of a column's wave read consecutive entries), zero in the padding columns
            std::vector<double> bp((size_t)2 * P * fb.ldc, 0.0), hp((size_t)2 * S * fb.ldt, 0.0);
            for (int y = 0; y < P; ++y)
                for (int x = 0; x < fb.Ph; ++x) { bp[2 * ((size_t)x * P + y)] = bh[2 * ((size_t)y * fb.Ph + x)]; bp[2 * ((size_t)x * P + y) + 1] = bh[2 * ((size_t)y * fb.Ph + x) + 1]; }
            for (int y = 0; y < S; ++y)
                for (int x = 0; x < ctx->Sh; ++x) { hp[2 * ((size_t)x * S + y)] = H[2 * ((size_t)y * ctx->Sh + x)]; hp[2 * ((size_t)x * S + y) + 1] = H[2 * ((size_t)y * ctx->Sh + x) + 1]; }
            if ((rc = dev_put_l(ctx, fb.allocs, bp.data(), bp.size(), &p))) return rc; fb.bhat_p = reinterpret_cast<double2*>(p);
            if ((rc = dev_put_l(ctx, fb.allocs, hp.data(), hp.size(), &p))) return rc; fb.htab_p = reinterpret_cast<double2*>(p);
            if ((rc = fft_transform(ctx, fb, P, fb.fP))) return rc;
            if ((rc = fft_transform(ctx, fb, S, fb.fS))) return rc;
            if ((rc = dev_new_l(ctx, fb.allocs, (size_t)cap * 2 * ctx->Sh, &fb.zbuf))) return rc;
            if ((rc = fft_kernels(ctx, fb))) return rc;
        }
    }
    const size_t R = fb.rows;
    if ((rc = dev_new_l(ctx, fb.allocs, (size_t)cap * R * P, &fb.img, true))) return rc;       // padding stays zero for ever
    if ((rc = dev_new_l(ctx, fb.allocs, (size_t)cap * R * P, &fb.conv))) return rc;
    if ((rc = dev_new_l(ctx, fb.allocs, fb.cols ? (size_t)cap * S * fb.ldc : (size_t)cap * P * fb.Ph, &fb.spec, fb.cols))) return rc;
    if ((rc = dev_new_l(ctx, fb.allocs, fb.cols ? (size_t)cap * S * fb.ldt : (size_t)cap * S * ctx->Sh, &fb.tfspec, fb.cols))) return rc;
    FFTCHK(ctx, rocfft_execution_info_create(&fb.info));
    FFTCHK(ctx, rocfft_execution_info_set_stream(fb.info, ctx->stream));
    HIPCHK(ctx, hipDeviceSynchronize());                        // (the zero fills ran on the null stream)
    fb.ready = true;
    return JX_OK;
}

static int fft_plans(jx_ctx* ctx, FftBack& fb, int batch, Plan3** out) {
    auto it = fb.plans.find(batch);
    if (it != fb.plans.end()) { *out = &it->second; return JX_OK; }
    Plan3 pl;
    const size_t P = fb.P, S = ctx->cfg.S, Sh = ctx->Sh;
    rocfft_plan p1, p2, p3;
    if (fb.cols) {
        // rows only: img [batch * S][P] -> spec [batch * S][ldc]; spec -> conv [batch * S][P]; conv's first S columns -> tfspec [batch * S][ldt]
        auto row_plan = [&](rocfft_plan* out, bool inverse, size_t len, size_t rdist, size_t cdist) -> int {
            rocfft_plan_description desc = nullptr;
            FFTCHK(ctx, rocfft_plan_description_create(&desc));
            size_t one[1] = {1};
            if (!inverse) FFTCHK(ctx, rocfft_plan_description_set_data_layout(desc, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                                              nullptr, nullptr, 1, one, rdist, 1, one, cdist));
            else FFTCHK(ctx, rocfft_plan_description_set_data_layout(desc, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real,
                                                                     nullptr, nullptr, 1, one, cdist, 1, one, rdist));
            size_t l[1] = {len};
            FFTCHK(ctx, rocfft_plan_create(out, rocfft_placement_notinplace, inverse ? rocfft_transform_type_real_inverse : rocfft_transform_type_real_forward,
                                           rocfft_precision_double, 1, l, (size_t)batch * S, desc));
            rocfft_plan_description_destroy(desc);
            return JX_OK;
        };
        int rc;
        if ((rc = row_plan(&pl.row_fwd, false, P, P, fb.ldc))) return rc;
        if ((rc = row_plan(&pl.row_inv, true, P, P, fb.ldc))) return rc;
        if ((rc = row_plan(&pl.row_tf, false, S, P, fb.ldt))) return rc;
        p1 = pl.row_fwd; p2 = pl.row_inv; p3 = pl.row_tf;
    } else {
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
        p1 = pl.beam_fwd; p2 = pl.beam_inv; p3 = pl.tf_fwd;
    }
    size_t w1 = 0, w2 = 0, w3 = 0;
    FFTCHK(ctx, rocfft_plan_get_work_buffer_size(p1, &w1));
    FFTCHK(ctx, rocfft_plan_get_work_buffer_size(p2, &w2));
    FFTCHK(ctx, rocfft_plan_get_work_buffer_size(p3, &w3));
    const size_t need = std::max(w1, std::max(w2, w3));
    if (need > fb.work_cap) {
        if (fb.work) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(fb.work)); fb.work = nullptr; }
        HIPCHK(ctx, hipMalloc(&fb.work, need));
        ctx->device_bytes += (int64_t)need - (int64_t)fb.work_cap;
        fb.work_cap = need;
    }
    if (fb.work_cap) FFTCHK(ctx, rocfft_execution_info_set_work_buffer(fb.info, fb.work, fb.work_cap));
    fb.plans[batch] = pl;
    *out = &fb.plans[batch];
    return JX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// contracted route: host-side plan.  Built before the route is chosen, so that `auto` can fall back.
// ---------------------------------------------------------------------------------------------------------------------
#define JX_MIX_NS 8
#define JX_MIX_RTS(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(36) X(40) X(44) X(48) X(56) X(64)
#define JX_MIX_NXTS(X) X(1) X(2) X(3) X(4) X(5) X(6)
#define JX_MIX_KSPLIT_MAX 64
#define JX_LR_TOL_DEFAULT 1e-8
#define JX_PRUNE_TOL 1e-22            // weights of the data-radii spline below this fraction of the largest are not read (an fp64 sum cannot see them)
#define JX_MIX_CAP_REACH 4             // ranks up to rank_cap + this many terms are cut to rank_cap (then measured by the guard)
struct MixBuild {
    bool ok = false;
    std::string why;
    int form = 0, NU = 0, r = 0, ns = 0, R = 0, RT = 0, nxt = 0, ntile = 0, nog = 0, ksteps = 0;
    int r_tol = 0;                     // terms above the cut before the cap
    bool mfma = false;
    int nxt_u = 0, ntile_u = 0, nog_u = 0;     // tiling of the outputs in use (0: all of them)
    std::vector<double> Op_u;
    int NU_full = 0;                   // distinct rows of the quadrant; NU below = the rows stage 1 evaluates (sub)
    std::vector<int> sub;              // their indices (empty: all)
    size_t krows = 0;
    jxt::MixColumns cols;
    std::vector<double> Cm, Op;
    std::vector<JxSamp> ent;
    int cld = 0;
    double tol = 0.0, beam_tol = 0.0;
    double cost_lowrank = 0.0, cost_full = 0.0;
    // exact form
    int Nk = 0, Nkp = 0, x_nxt = 0, x_ng = 0, x_ng_use = 0, x_npair = 0, x_nSj = 0, x_nfold = 0;
    bool x_lean = true;
    std::vector<double> xWfk;
    std::vector<double> xTyp, xOpk;
};

static void mix_output_tiling(int nrow, int* nxt, int* nog, int* ntile) {
    const int tiles = (nrow + 15) / 16;
    double best = 1e300;
#define JX_PICK(Xv) { const int og = (tiles + Xv - 1) / Xv; const double cost = (double)og * Xv * (1.0 + 0.5 / Xv); if (cost < best) { best = cost; *nxt = Xv; *nog = og; } }
    JX_MIX_NXTS(JX_PICK)
#undef JX_PICK
    *ntile = *nog * *nxt;
}

// form_force: -1 the cheaper form, 0 low-rank, 1 full.  tW: walker stride of the spline arrays (the full form's sample entries
// carry element offsets into them).
static void plan_mix(jx_ctx* ctx, const std::vector<double>& beam, const std::vector<double>& filt, const std::vector<double>& r,
                     double tol, int form_force, long long tW, MixBuild& mb) {
    if (form_force == 2 && ctx->f32) form_force = -1;          // (the fp32 variants exist on the forms of rounds 3-4)
    const jx_config& c = ctx->cfg;
    const int S = c.S, B = c.B, Sh = S / 2 + 1, nrow = S - S / 2;
    if (!ctx->dmat_mirror) { mb.why = "d_mat lacks the mirror structure of centdistmat"; return; }
    if (c.fft_pad != 0) { mb.why = "fft_pad is a parameter of the rocFFT sequence"; return; }
    const int NU = std::max(S / 2, S - 1 - S / 2) + 1;
    if (ctx->qn != NU) { mb.why = "quadrant table size"; return; }
    if (NU > 9 * 64) { mb.why = "map side beyond the symmetric map kernel's range"; return; }
    mb.NU = NU; mb.tol = tol; mb.beam_tol = 1e-14;          // (NU: replaced by the size of the sub-grid below when stage 1 evaluates one)
    mix_output_tiling(nrow, &mb.nxt, &mb.nog, &mb.ntile);
    if (ctx->prune && ctx->nrow_use > 0 && (ctx->nrow_use + 15) / 16 < (nrow + 15) / 16) mix_output_tiling(ctx->nrow_use, &mb.nxt_u, &mb.nog_u, &mb.ntile_u);
    // transfer-function weights of the extracted row, real for a real filter that is symmetric in each wavenumber
    std::vector<double> hy;
    jxt::tf_hy_table(filt, S, hy);
    std::vector<double> A((size_t)S * Sh);
    double maxre = 0.0, maxim = 0.0;
    for (size_t e = 0; e < A.size(); ++e) { A[e] = hy[2 * e]; maxre = std::max(maxre, std::fabs(hy[2 * e])); maxim = std::max(maxim, std::fabs(hy[2 * e + 1])); }
    if (!(maxim <= 1e-15 * maxre)) { mb.why = "transfer-function weights are not real"; return; }
    if (form_force == 2) {
        // (the row of 16 walkers and the eight waves' partial tiles have to fit the LDS of the row product: data radii reaching
        //  across a very large map go to the contracted forms)
        const int nuse0 = std::min(ctx->prune ? std::max(1, ctx->nrow_use) : nrow, nrow);
        if (sizeof(double) * JX_RST_LDS_DOUBLES((nuse0 + 1) | 1, c.nflux, 0) > (size_t)158 * 1024) form_force = -1;
    }
    if (form_force == 2) {
        // ---- exact form: the row as one constant operator on the spline ordinates (every pixel, every radius, nothing truncated)
        std::vector<double> Wy;
        mb.form = 2;
        mb.Nk = jxt::exact_row_operator(beam, B, c.step * c.step, A, S, NU, ctx->h_Qtab, ctx->qn, r, ctx->h_G, ctx->K, Wy);
        if (!ctx->map_ok) mb.Nk = c.N;                          // (the profile taps are then read off the product's own array: every ordinate)
        mb.Nkp = (mb.Nk + 15) & ~15;
        const int tiles_use = (std::max(1, ctx->prune ? ctx->nrow_use : nrow) + 15) / 16, tiles = (nrow + 15) / 16;
        mb.x_nxt = std::min(6, tiles_use);
        mb.x_ng_use = (tiles_use + mb.x_nxt - 1) / mb.x_nxt;
        mb.x_ng = (tiles + mb.x_nxt - 1) / mb.x_nxt;
        jxt::exact_row_layout(Wy, nrow, c.N, mb.Nkp / 16, mb.x_nxt, mb.x_ng, mb.xOpk);
        mb.x_npair = (mb.Nkp / 16 + 1) / 2;
        mb.x_nSj = (c.N + 15) / 16;
        jxt::abel_ordinate_layout(r, c.kpc_cm * c.sigma_T / c.m_e, mb.Nkp / 16, mb.x_nSj, mb.xTyp);
        {   // an odd number of ordinate tiles: the last one folded into the row product (timed path), the others pair up exactly
            const int nS = mb.Nkp / 16;
            const char* e = opt_str(ctx, "JOXSZ_X_FOLD");
            mb.x_lean = !(e && atoi(e) == 0);
            if ((nS & 1) && nS >= 3 && mb.x_lean) {
                mb.x_nfold = mb.x_nSj - (nS - 1);
                jxt::exact_fold_layout(Wy, nrow, r, c.kpc_cm * c.sigma_T / c.m_e, nS, mb.x_nSj, mb.x_nxt, mb.x_ng, mb.xWfk);
            }
        }
        mb.r = 0; mb.ns = 0; mb.R = 0; mb.RT = 0; mb.NU_full = NU; mb.krows = 0;
        mb.ok = true;
        return;
    }
    // ---- low-rank form: separable terms of the beam x singular terms of the weights
    std::vector<double> U, V, by, bx;
    // Which rows (= columns) of the quadrant stage 1 evaluates: all of them, or -- when the quadrant lies inside the radial grid
    // (no fill values, no NaN radius: the map is then smooth away from the core) -- the sub-grid of jxt::mix_row_subset, the
    // others entering through the interpolation folded into both operators below.  jx_finalize measures the result with the
    // truncation (the guard) and takes the sub-grid away first when a bound is exceeded.
    mb.NU_full = NU;
    std::vector<double> Lint;                                    // [NU][ns]
    std::vector<double> Qsub;
    int NUs = NU;
    if (ctx->subsample) {
        bool inside = true;
        for (double v : ctx->h_Qtab) if (!(v <= r.back())) { inside = false; break; }
        jxt::mix_row_subset(NU, ctx->sub_u0, ctx->sub_u1, mb.sub);
        if (!inside || (int)mb.sub.size() > (3 * NU) / 4 || (int)mb.sub.size() < 2 * ctx->sub_npts) mb.sub.clear();
        else {
            NUs = (int)mb.sub.size();
            jxt::mix_interp_matrix(NU, mb.sub, ctx->sub_npts, Lint);
            Qsub.resize((size_t)NUs * NUs);
            for (int a = 0; a < NUs; ++a)
                for (int b2 = 0; b2 < NUs; ++b2) Qsub[(size_t)a * NUs + b2] = ctx->h_Qtab[(size_t)mb.sub[a] * ctx->qn + mb.sub[b2]];
        }
    }
    bool lowrank_ok = mb.sub.empty() ? jxt::mix_column_tables(ctx->h_Qtab, ctx->qn, NU, r, mb.cols, ctx->usplit)
                                     : jxt::mix_column_tables(Qsub, NUs, NUs, r, mb.cols, ctx->usplit);
    std::string why_lr = lowrank_ok ? "" : "pixel radii do not grow along the columns of d_mat";
    if (lowrank_ok && form_force != 1) {
        mb.r = jxt::lowrank_factor_qr(A.data(), S, Sh, tol, U, V);
        mb.ns = jxt::beam_separable_terms(beam, B, c.step * c.step, mb.beam_tol, by, bx);
        mb.r_tol = mb.r;
        // A rank a few terms above 16 is cut to 16 (terms are sorted by singular value: the first rows of U and V): the stage-1
        // kernel instance with 16 rows per column, stage 2 shorter by as much (and one 16-row tile of the matrix cores carries
        // stage 1 where that kernel is asked for).  jx_finalize measures what the cut costs on the caller's data and rebuilds
        // without the cap when it is beyond the bounds.
        if (!ctx->tol_pinned && ctx->rank_cap > 0 && mb.ns > 0 && mb.r * mb.ns > ctx->rank_cap && ctx->rank_cap / mb.ns >= 1 &&
            mb.r <= ctx->rank_cap / mb.ns + JX_MIX_CAP_REACH) {
            mb.r = ctx->rank_cap / mb.ns;
            U.resize((size_t)mb.r * S); V.resize((size_t)mb.r * Sh);
        }
        mb.R = mb.r * mb.ns;
        mb.mfma = ctx->mix_mfma && mb.R <= 16;
        if (mb.r <= 0 || mb.ns <= 0) { lowrank_ok = false; why_lr = "transfer-function weights or beam image vanish"; }
        else if (mb.mfma) mb.RT = 16;
        else {
#define JX_PICK(Rv) if (!mb.RT && mb.R <= Rv) mb.RT = Rv;
            JX_MIX_RTS(JX_PICK)
#undef JX_PICK
            if (!mb.RT) { lowrank_ok = false; why_lr = "rank of the separable form beyond the stage-1 kernel (" + std::to_string(mb.r) + " x " + std::to_string(mb.ns) + " terms)"; }
        }
    }
    if (!ctx->usplit_forced && lowrank_ok && mb.RT) {
        // pieces per column: a chunk of 1024 walkers should fill the chip in as few rounds of waves as possible (24 resident waves per CU:
        // scripts/ubench/occ_rowmix.hip), each piece paying its ring fill once more -- rounds x (1 / pieces + 0.08), measured
        // against JOXSZ_MIX_USPLIT = 1..4 at 512^2 with and without the sub-grid (profiles/r04_subsample_scan.log); the hand-over of
        // the later pieces (RT sums per lane each) has to fit the 64 KB of dynamic LDS of one walker group
        const double cap = 24.0 * std::max(1, ctx->num_cu), ngrp = 16.0;      // (a nominal chunk of 1024 walkers, not this context's: the grouping of a walker's sums must not depend on max_batch)
        double best = 1e300;
        int pick = 1;
        for (int usp = 1; usp <= 4; ++usp) {
            if ((size_t)(usp - 1) * mb.RT * 64 * sizeof(double) > (size_t)64 * 1024) break;
            const double c = std::ceil(NUs * usp * ngrp / cap) * (1.0 / usp + 0.08);
            if (c < best * (1.0 - 1e-9)) { best = c; pick = usp; }
        }
        if (pick != ctx->usplit) {
            ctx->usplit = pick;
            lowrank_ok = mb.sub.empty() ? jxt::mix_column_tables(ctx->h_Qtab, ctx->qn, NU, r, mb.cols, ctx->usplit)
                                        : jxt::mix_column_tables(Qsub, NUs, NUs, r, mb.cols, ctx->usplit);
        }
    }
    // cost of each form in fused multiply-adds per walker; stage 1 runs on the vector units at about 0.6 of the rate the
    // matrix cores reach in the product kernels (measured: 256^2 with 32 terms 0.141 ms low-rank against 0.126 ms full; 257^2 0.155 against 0.132)
    const double nsamp_lr = (double)NUs * NUs;                                   // (samples evaluated: the sub-grid when there is one)
    // (stage 1 on the matrix cores: 16 multiply-adds per sample there, the 4 of the evaluation beside them on the vector units)
    const double nout = mb.ntile_u > 0 ? 16.0 * mb.ntile_u : (double)nrow;      // outputs the timed product computes
    const int nog_t = mb.ntile_u > 0 ? mb.nog_u : mb.nog;
    mb.cost_lowrank = (lowrank_ok && mb.RT) ? (mb.mfma ? nsamp_lr * 18.0 : 1.6 * nsamp_lr * (4.0 + mb.RT)) + nout * NUs * mb.R : 1e300;
    mb.cost_full = nout * nsamp_lr * 0.5 + nsamp_lr * 6.0 * nog_t;                // (the full form evaluates the same sub-grid)
    int form = (mb.cost_lowrank <= mb.cost_full) ? 0 : 1;
    if (form_force == 0) { if (!lowrank_ok || !mb.RT) { mb.why = "low-rank form: " + why_lr; return; } form = 0; }
    if (form_force == 1) form = 1;
    mb.form = form;
    if (form != 0) mb.mfma = false;
    if (form == 0) {
        mb.cld = mb.RT;
        if (mb.sub.empty()) {
            jxt::mix_stage1_operator(U, mb.r, by, mb.ns, S, B, NU, (mb.cols.wld + 4 + 1) & ~1, mb.cld, mb.Cm);   // (zero rows behind the last: the matrix-core kernel reads whole groups of four)
            const size_t K = (size_t)NU * mb.R;
            mb.ksteps = (int)((K + 3) / 4);
            mb.krows = 4 * ((size_t)mb.ksteps + (size_t)JX_MIX_KSPLIT_MAX * JX_OPG_RD + JX_OPG_RD);
            jxt::mix_stage2_operator(V, mb.r, bx, mb.ns, S, B, NU, mb.krows, mb.ntile, mb.Op);
        } else {
            // both operators on every row / column of the quadrant first, then through the interpolation: C_sub = L^T C,
            // Op_sub[(x'_s R + j)] = sum_x' L[x'][x'_s] Op[(x' R + j)]
            std::vector<double> Cf, Of;
            const int rows_f = (NU + 3) & ~3;
            jxt::mix_stage1_operator(U, mb.r, by, mb.ns, S, B, NU, rows_f, mb.cld, Cf);
            const int rows_s = (mb.cols.wld + 4 + 1) & ~1;
            mb.Cm.assign((size_t)rows_s * mb.cld, 0.0);
            for (int u = 0; u < NU; ++u)
                for (int a = 0; a < NUs; ++a) {
                    const double l = Lint[(size_t)u * NUs + a];
                    if (l == 0.0) continue;
                    for (int j = 0; j < mb.cld; ++j) mb.Cm[(size_t)a * mb.cld + j] += l * Cf[(size_t)u * mb.cld + j];
                }
            const size_t Kf = (size_t)NU * mb.R;
            const size_t krows_f = 4 * ((Kf + 3) / 4);
            jxt::mix_stage2_operator(V, mb.r, bx, mb.ns, S, B, NU, krows_f, mb.ntile, Of);
            const size_t K = (size_t)NUs * mb.R;
            mb.ksteps = (int)((K + 3) / 4);
            mb.krows = 4 * ((size_t)mb.ksteps + (size_t)JX_MIX_KSPLIT_MAX * JX_OPG_RD + JX_OPG_RD);
            const size_t rowlen = (size_t)16 * mb.ntile;
            mb.Op.assign(mb.krows * rowlen, 0.0);
            for (int x = 0; x < NU; ++x)
                for (int a = 0; a < NUs; ++a) {
                    const double l = Lint[(size_t)x * NUs + a];
                    if (l == 0.0) continue;
                    for (int j = 0; j < mb.R; ++j) {
                        const double* src = &Of[((size_t)x * mb.R + j) * rowlen];
                        double* dst = &mb.Op[((size_t)a * mb.R + j) * rowlen];
                        for (size_t e = 0; e < rowlen; ++e) dst[e] += l * src[e];
                    }
                }
        }
    } else {
        // ---- full form: every distinct sample (u <= x' when the quadrant is symmetric) is a row of the operator
        mb.r = 0; mb.ns = 0; mb.R = 0; mb.RT = 0;
        std::vector<double> Om;                                  // [nrow][NU][NU]
        jxt::mix_full_operator(beam, B, c.step * c.step, A, S, NU, Om);
        int NQ = NU;                                             // rows = columns of the samples evaluated
        if (!mb.sub.empty()) {
            // the sub-grid: Om_sub[x][a][b] = sum_{u, x'} L[u][a] L[x'][b] Om[x][u][x'] (L: at most sub_npts entries per row)
            std::vector<std::vector<std::pair<int, double>>> Ls(NU);
            for (int u = 0; u < NU; ++u)
                for (int a = 0; a < NUs; ++a) { const double l = Lint[(size_t)u * NUs + a]; if (l != 0.0) Ls[u].emplace_back(a, l); }
            std::vector<double> Os((size_t)nrow * NUs * NUs, 0.0), T((size_t)NU * NUs);
            for (int x = 0; x < nrow; ++x) {
                std::fill(T.begin(), T.end(), 0.0);
                const double* ox = &Om[(size_t)x * NU * NU];
                for (int u = 0; u < NU; ++u)
                    for (int xq = 0; xq < NU; ++xq) {
                        const double v = ox[(size_t)u * NU + xq];
                        if (v == 0.0) continue;
                        for (const auto& e : Ls[xq]) T[(size_t)u * NUs + e.first] += e.second * v;
                    }
                double* os = &Os[(size_t)x * NUs * NUs];
                for (int u = 0; u < NU; ++u)
                    for (const auto& e : Ls[u]) {
                        const double* tu = &T[(size_t)u * NUs];
                        double* oa = &os[(size_t)e.first * NUs];
                        for (int b2 = 0; b2 < NUs; ++b2) oa[b2] += e.second * tu[b2];
                    }
            }
            Om.swap(Os);
            NQ = NUs;
        }
        const std::vector<double>& Qt = mb.sub.empty() ? ctx->h_Qtab : Qsub;
        const int qld = mb.sub.empty() ? ctx->qn : NUs;
        const bool tri = jxt::quadrant_is_symmetric(Qt, qld, NQ);
        std::vector<int> ku, kx;
        for (int u = 0; u < NQ; ++u)
            for (int x = tri ? u : 0; x < NQ; ++x) { ku.push_back(u); kx.push_back(x); }
        const size_t K = ku.size();
        mb.ksteps = (int)((K + 3) / 4);
        mb.krows = 4 * ((size_t)mb.ksteps + (size_t)JX_MIX_KSPLIT_MAX * JX_OPG_RD + JX_OPG_RD + JX_OPG_ECH);
        mb.Op.assign(mb.krows * 16 * (size_t)mb.ntile, 0.0);
        JxSamp zero{};
        mb.ent.assign(mb.krows, zero);
        for (size_t k = 0; k < K; ++k) {
            const int u = ku[k], xq = kx[k];
            int k16; double w[4];
            jxt::spline_sample_weights(r, Qt[(size_t)u * qld + xq], &k16, w);
            JxSamp& e = mb.ent[k];
            e.off = (long long)(k16 / 16) * tW;
            e.a = w[0]; e.b = w[1]; e.c = w[2]; e.d = w[3];
            for (int x = 0; x < nrow; ++x) {
                double v = Om[((size_t)x * NQ + u) * NQ + xq];
                if (tri && xq != u) v += Om[((size_t)x * NQ + xq) * NQ + u];
                mb.Op[(k * 16 + (x & 15)) * mb.ntile + (x >> 4)] = v;
            }
        }
    }
    if (form == 0 && !mb.sub.empty()) mb.NU = NUs;               // (stage 1 walks the sub-grid's columns; the full form keeps NU and lists its samples)
    if (mb.ntile_u > 0 && mb.ntile_u < mb.ntile) {
        // the same operator for the outputs in use alone: tiles [0, ntile_u) of every row, compact
        const size_t rows = mb.Op.size() / ((size_t)16 * mb.ntile);
        mb.Op_u.assign(rows * 16 * (size_t)mb.ntile_u, 0.0);
        for (size_t kx = 0; kx < rows * 16; ++kx)
            for (int t = 0; t < mb.ntile_u; ++t) mb.Op_u[kx * mb.ntile_u + t] = mb.Op[kx * mb.ntile + t];
    } else { mb.ntile_u = 0; mb.nxt_u = 0; mb.nog_u = 0; }
    mb.ok = true;
}

static void mix_teardown(jx_ctx* ctx, MixBack& m) {
    for (void* p : m.allocs) (void)hipFree(p);
    m.allocs.clear();
    ctx->device_bytes -= m.bytes;
    m.bytes = 0;
    m.ready = false;
}

static void mix_kslices(const MixBack& m, bool used, int* ksplit_out, int* kper_out);

static int mix_setup(jx_ctx* ctx, const MixBuild& mb, long long tW) {
    MixBack& m = ctx->mix;
    const int N = ctx->cfg.N;
    const int64_t before = ctx->device_bytes;
    int rc;
    m.form = mb.form; m.RT = mb.RT; m.nxt = mb.nxt; m.r = mb.r; m.ns = mb.ns; m.ksteps = mb.ksteps; m.tW = tW; m.tol = mb.tol;
    m.mfma = mb.mfma; m.r_tol = mb.r_tol;
    m.NU_full = mb.NU_full; m.sub = mb.sub;
    int* qi; double* qd;
    if (mb.form == 2) {
        // exact form: two constant operators and the ordinates of a chunk; none of the work buffers of the contracted forms
        m.Nk = mb.Nk; m.Nkp = mb.Nkp; m.x_nxt = mb.x_nxt; m.x_ng = mb.x_ng; m.x_ng_use = mb.x_ng_use;
        m.x_npair = mb.x_npair; m.x_nSj = mb.x_nSj; m.x_ldpp = 16 * mb.x_nSj; m.x_nfold = mb.x_nfold; m.x_lean = mb.x_lean;
        if (mb.x_nfold && (rc = dev_put_l(ctx, m.allocs, mb.xWfk.data(), mb.xWfk.size(), &m.x_Wfk))) return rc;
        m.cft = nullptr; m.Dt = nullptr; m.Pt = nullptr; m.x_cf = nullptr; m.x_ppi = nullptr; m.has_u = false; m.ncol = 2 * N;
        if ((rc = dev_put_l(ctx, m.allocs, mb.xTyp.data(), mb.xTyp.size(), &m.x_Typ))) return rc;
        if ((rc = dev_put_l(ctx, m.allocs, mb.xOpk.data(), mb.xOpk.size(), &m.x_Opk))) return rc;
        if ((rc = dev_new_l(ctx, m.allocs, (size_t)tW * m.Nkp, &m.x_y, true))) return rc;
        if ((rc = dev_new_l(ctx, m.allocs, (size_t)(tW / 16) * m.x_npair * m.x_ng * 256 * m.x_nxt, &m.x_P, true))) return rc;
        m.bytes = ctx->device_bytes - before;
        m.ready = true;
        return JX_OK;
    }
    const size_t esz = ctx->f32 ? sizeof(float) : sizeof(double);
    const size_t cft_rows = (size_t)N + 2 * JX_MIX_NS + 2;
    if (2 * esz * cft_rows * tW >= ((size_t)1 << 32)) { ctx->err = "contracted route: launch too large for 32-bit knot offsets (lower max_batch)"; return JX_ERR_UNSUPPORTED; }
    JxMix& mx = m.mx;
    memset(&mx, 0, sizeof(mx));
    mx.dbg = m.dbg;
    if (mb.form == 0) {
        mx.NU = mb.NU; mx.R = mb.R; mx.tW = tW; mx.segld = mb.cols.segld; mx.wld = mb.cols.wld; mx.cld = mb.cld;
        mx.cft_bytes = (unsigned)(2 * esz * cft_rows * tW);
        mx.usplit = mb.cols.usplit;
        if ((rc = dev_put_l(ctx, m.allocs, mb.cols.urange.data(), mb.cols.urange.size(), &qi))) return rc; mx.urange = qi;
        if ((rc = dev_put_l(ctx, m.allocs, mb.cols.seg0.data(), mb.cols.seg0.size(), &qi))) return rc; mx.seg0 = qi;
        if ((rc = dev_put_l(ctx, m.allocs, mb.cols.nseg.data(), mb.cols.nseg.size(), &qi))) return rc; mx.nseg = qi;
        if ((rc = dev_put_l(ctx, m.allocs, mb.cols.seg.data(), mb.cols.seg.size(), &qi))) return rc; mx.seg = qi;
        if ((rc = dev_put_l(ctx, m.allocs, mb.cols.w4.data(), mb.cols.w4.size(), &qd))) return rc; mx.w4 = qd;
        if ((rc = dev_put_l(ctx, m.allocs, mb.Cm.data(), mb.Cm.size(), &qd))) return rc; mx.Cm = qd;
    }
    if (ctx->f32c && mb.form == 0) {                              // fp32 arithmetic: the stage-1 tables once more, rounded to fp32
        std::vector<float> wf(mb.cols.w4.begin(), mb.cols.w4.end()), cf(mb.Cm.begin(), mb.Cm.end());
        float* qf;
        if ((rc = dev_put_l(ctx, m.allocs, wf.data(), wf.size(), &qf))) return rc; mx.w4f = qf;
        if ((rc = dev_put_l(ctx, m.allocs, cf.data(), cf.size(), &qf))) return rc; mx.Cmf = qf;
    }
    JxOpg& og = m.og;
    memset(&og, 0, sizeof(og));
    og.tW = tW; og.ntile = mb.ntile; og.nog = mb.nog; og.ldx = 16 * mb.ntile;
    if ((rc = dev_put_l(ctx, m.allocs, mb.Op.data(), mb.Op.size(), &qd))) return rc; og.Op = qd;
    if (ctx->f32c && mb.form == 0) {
        std::vector<float> of(mb.Op.begin(), mb.Op.end());
        float* qf;
        if ((rc = dev_put_l(ctx, m.allocs, of.data(), of.size(), &qf))) return rc; og.Opf = qf;
    }
    m.has_u = mb.ntile_u > 0;
    m.og_u = og;
    if (m.has_u) {
        JxOpg& ou = m.og_u;
        m.nxt_u = mb.nxt_u;
        ou.ntile = mb.ntile_u; ou.nog = mb.nog_u; ou.ldx = 16 * mb.ntile_u;
        if ((rc = dev_put_l(ctx, m.allocs, mb.Op_u.data(), mb.Op_u.size(), &qd))) return rc; ou.Op = qd;
        if (ctx->f32c && mb.form == 0) {
            std::vector<float> of(mb.Op_u.begin(), mb.Op_u.end());
            float* qf;
            if ((rc = dev_put_l(ctx, m.allocs, of.data(), of.size(), &qf))) return rc; ou.Opf = qf;
        }
    }
    m.ncol = 2 * N;
    if ((rc = dev_new_l(ctx, m.allocs, 2 * esz * cft_rows * tW / sizeof(double) + 1, &m.cft, true))) return rc;
    if (mb.form == 0) {
        if ((rc = dev_new_l(ctx, m.allocs, mb.krows * (size_t)tW, &m.Dt, true))) return rc;    // (fp32 arithmetic: the same buffer holds floats)
        og.Dt = m.Dt; og.Dtf = reinterpret_cast<const float*>(m.Dt);
        m.og_u.Dt = og.Dt; m.og_u.Dtf = og.Dtf;
    } else {
        JxSamp* qe;
        if ((rc = dev_put_l(ctx, m.allocs, mb.ent.data(), mb.ent.size(), &qe))) return rc; og.ent = qe; m.og_u.ent = qe;
    }
    og.pstride = (long long)tW * og.ldx + 272;                // (not a power of two: the tail reads all slices of a walker at once)
    m.og_u.pstride = (long long)tW * m.og_u.ldx + 272;
    if ((rc = dev_new_l(ctx, m.allocs, (size_t)JX_MIX_KSPLIT_MAX * og.pstride, &m.Pt))) return rc;
    { int kp; mix_kslices(m, false, &m.last_ksplit, &kp); if (m.has_u) mix_kslices(m, true, &m.ksplit_u, &kp); }   // (reported before the first launch too)
    m.bytes = ctx->device_bytes - before;
    m.ready = true;
    return JX_OK;
}

// K slices of the matrix-core product: a function of the problem alone (about 64 k-steps each, whole multiples of 8 slices so
// that a full chunk fills whole rounds of blocks), never of the launch -- a walker's sums are grouped the same way wherever it
// sits in whatever batch, so its result is bitwise independent of both
static void mix_kslices(const MixBack& m, bool used, int* ksplit_out, int* kper_out) {
    int ksplit = m.ksplit_force > 0 ? m.ksplit_force : (m.ksteps + 32) / 64;
    if (ksplit >= 8 && m.ksplit_force <= 0) ksplit = (ksplit + 4) / 8 * 8;
    // few output groups per walker block (the outputs in use alone; small sides): more K slices, of about 16 k-steps each, so
    // that the launch still fills the chip (measured: profiles/r04_subsample_scan.log)
    if (used && m.ksplit_u_force > 0) ksplit = m.ksplit_u_force;
    else if (used || m.ksplit_force <= 0) ksplit = std::max(ksplit, std::min((m.ksteps + 15) / 16, std::max(1, 32 / std::max(1, used ? m.og_u.nog : m.og.nog))));
    ksplit = std::max(1, std::min(ksplit, JX_MIX_KSPLIT_MAX));
    int kper = (m.ksteps + ksplit - 1) / ksplit;
    kper = (kper + JX_OPG_RD - 1) / JX_OPG_RD * JX_OPG_RD;
    ksplit = (m.ksteps + kper - 1) / kper;
    *ksplit_out = ksplit; *kper_out = kper;
}

// stage 1 + stage 2 (low-rank form) or the one product of the full form; es: the launch's event set or null
template <typename F>
static int launch_mix(jx_ctx* ctx, int n, EvSet* es, bool used /* the product for the outputs the tail reads alone */, F&& between /* called between stage 1 and the matrix-core product */) {
    MixBack& m = ctx->mix;
    hipStream_t st = ctx->stream;
    if (m.form == 0) {
        JxMix mx = m.mx;
        mx.n = n;
        // block = gpb walker groups x usplit pieces of one column (usplit: a setting of the context, never of the launch)
        const int usp = mx.usplit, ngrp = (n + 63) / 64;
        int wcap = m.wpb;
        size_t lds_c = 0;
        const int crows = (mx.wld + 4 + 1) & ~1;
        if (m.mfma) {
            // the matrix-core kernel keeps the operator in LDS: blocks of 8 waves, two per CU -- or of 16, one per CU, when the
            // operator is too large for two copies (sides beyond ~600)
            lds_c = sizeof(double) * 16 * (size_t)crows;
            wcap = (2 * (lds_c + sizeof(double) * JX_MXM_REGION(usp) * (8 / usp)) <= 158 * 1024) ? 8 : 16;
            if (m.wpb_force > 0) wcap = m.wpb_force;
        }
        if (!m.mfma && usp > 1) {                                   // walker groups per block: their hand-over (RT sums per lane and later piece) within 64 KB
            const int fit = (int)((size_t)64 * 1024 / (sizeof(double) * (usp - 1) * m.RT * 64));
            wcap = std::min(wcap, std::max(fit, 0) * usp);
            if (wcap < usp) { ctx->err = "stage 1: JOXSZ_MIX_USPLIT = " + std::to_string(usp) + " pieces of " + std::to_string(m.RT) + " rows per column do not fit the LDS hand-over (lower it)"; return JX_ERR_UNSUPPORTED; }
        }
        const int gpb = std::max(1, std::min(wcap / usp, ngrp)), wpb = gpb * usp;
        const int nq = (ngrp + gpb - 1) / gpb;
        mx.cper = (nq <= 8 && 8 % nq == 0) ? 8 / nq : 0;
        const dim3 g1((unsigned)(mx.cper ? 8 * ((mx.NU + mx.cper - 1) / mx.cper) : nq * mx.NU));
        const size_t lds1 = m.mfma ? lds_c + sizeof(double) * JX_MXM_REGION(usp) * gpb : sizeof(double) * (size_t)gpb * (usp - 1) * m.RT * 64;
        // (the vector-unit kernel keeps the default 64 KB limit of dynamic LDS: mix_setup clamps usplit so that a block's hand-over fits)
        if (lds1 > (m.mfma ? (size_t)158 * 1024 : (size_t)64 * 1024)) { ctx->err = "stage 1: the hand-over of the column pieces does not fit the LDS (lower JOXSZ_MIX_USPLIT or JOXSZ_MIX_WPB)"; return JX_ERR_UNSUPPORTED; }
        bool done = false;
        if (ctx->f32c) {
            const size_t ldsf = sizeof(float) * (size_t)gpb * (usp - 1) * m.RT * 64;
#define JX_MIXF_GO(Rv) if (!done && m.RT == Rv) { \
            hipLaunchKernelGGL((jx_rowmix_f32_kernel<Rv, JX_MIX_NS>), g1, dim3(64 * wpb), ldsf, st, mx, reinterpret_cast<const float2*>(m.cft), reinterpret_cast<float*>(m.Dt)); \
            done = true; }
            JX_MIX_RTS(JX_MIXF_GO)
#undef JX_MIXF_GO
        }
        if (!done && m.mfma) {
            if (ctx->f32) hipLaunchKernelGGL((jx_rowmix_mfma_kernel<JX_MIX_NS, float2>), g1, dim3(64 * wpb), lds1, st, mx, crows, reinterpret_cast<const float2*>(m.cft), m.Dt);
            else hipLaunchKernelGGL((jx_rowmix_mfma_kernel<JX_MIX_NS, double2>), g1, dim3(64 * wpb), lds1, st, mx, crows, reinterpret_cast<const double2*>(m.cft), m.Dt);
            done = true;
        }
#define JX_MIX_GO(Rv) if (!done && m.RT == Rv) { \
            if (ctx->f32) hipLaunchKernelGGL((jx_rowmix_kernel<Rv, JX_MIX_NS, float2>), g1, dim3(64 * wpb), lds1, st, mx, reinterpret_cast<const float2*>(m.cft), m.Dt); \
            else hipLaunchKernelGGL((jx_rowmix_kernel<Rv, JX_MIX_NS, double2>), g1, dim3(64 * wpb), lds1, st, mx, reinterpret_cast<const double2*>(m.cft), m.Dt); \
            done = true; }
        JX_MIX_RTS(JX_MIX_GO)
#undef JX_MIX_GO
        if (!done) { ctx->err = "no stage-1 kernel for this rank"; return JX_ERR_UNSUPPORTED; }
    }
    if (es) HIPCHK(ctx, hipEventRecord(es->e[3], st));
    { const int rcb = between(); if (rcb) return rcb; }
    {
        used = used && m.has_u;
        JxOpg og = used ? m.og_u : m.og;
        const int nxt = used ? m.nxt_u : m.nxt;
        og.n = n;
        const int nwb = (n + 127) / 128;
        int ksplit, kper;
        mix_kslices(m, used, &ksplit, &kper);
        og.ksplit = ksplit; og.kper = kper;
        if (used) m.ksplit_u = ksplit; else m.last_ksplit = ksplit;   // (the tail sums this many partials)
        m.last_was_u = used;
        const int nunit = ksplit * og.nog;
        og.kmajor = ksplit >= 8 ? 1 : 0;
        const dim3 g2((unsigned)(og.kmajor ? 8 * nwb * og.nog * ((ksplit + 7) / 8) : 8 * nwb * ((nunit + 7) / 8)));
        const size_t lds = m.form == 1 ? (size_t)JX_OPG_ECH * 4 * sizeof(JxSamp) : 0;
        bool done = false;
#define JX_OPGF_GO(Xv) if (!done && ctx->f32c && nxt == Xv) { \
            hipLaunchKernelGGL((jx_opgemm_f32_kernel<Xv>), g2, dim3(256), 0, st, og, reinterpret_cast<float*>(m.Pt)); done = true; }
        JX_MIX_NXTS(JX_OPGF_GO)
#undef JX_OPGF_GO
#define JX_OPG_GO(Xv) if (!done && nxt == Xv) { \
            if (m.form == 1 && ctx->f32) hipLaunchKernelGGL((jx_opgemm_kernel<1, Xv, float2>), g2, dim3(256), lds, st, og, reinterpret_cast<const float2*>(m.cft), m.Pt); \
            else if (m.form == 1) hipLaunchKernelGGL((jx_opgemm_kernel<1, Xv, double2>), g2, dim3(256), lds, st, og, reinterpret_cast<const double2*>(m.cft), m.Pt); \
            else hipLaunchKernelGGL((jx_opgemm_kernel<0, Xv, double2>), g2, dim3(256), lds, st, og, reinterpret_cast<const double2*>(m.cft), m.Pt); \
            done = true; }
        JX_MIX_NXTS(JX_OPG_GO)
#undef JX_OPG_GO
        if (!done) { ctx->err = "no stage-2 kernel for this output tiling"; return JX_ERR_UNSUPPORTED; }
    }
    if (es && !es->p1only) HIPCHK(ctx, hipEventRecord(es->e[4], st));
    return JX_OK;
}

extern "C" {

int jx_comm_unique_id(void* id_out) {
    if (!id_out) return JX_ERR_INVALID;
    if (!rccl_load()) { g_last_global_error = g_rccl.err; return JX_ERR_COMM; }
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { g_last_global_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return JX_ERR_COMM; }
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

int jx_comm_count(jx_ctx* ctx) {
    if (!ctx) return JX_ERR_INVALID;
    if (!ctx->comm) { ctx->err = "jx_comm_count before jx_comm_init_rank"; return JX_ERR_STATE; }
    int n = 0;
    NCCLCHK(ctx, g_rccl.CommCount(ctx->comm, &n));
    return n;
}

static int gather_drain(jx_ctx* ctx) {
    for (auto& pr : ctx->gt_inflight) {
        HIPCHK(ctx, hipEventSynchronize(pr.second));
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
        ctx->gather_ms += ms; ctx->gather_calls += 1;
        ctx->gt_free.push_back(pr);
    }
    ctx->gt_inflight.clear();
    return JX_OK;
}

// the stream the communicator's collectives run on; in overlap mode it first waits for everything enqueued on the compute stream so far
static int comm_fork(jx_ctx* ctx, hipStream_t* out) {
    if (!ctx->comm_overlap) { *out = ctx->stream; return JX_OK; }
    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_fork, 0));
    *out = ctx->comm_stream;
    return JX_OK;
}

int jx_comm_set_overlap(jx_ctx* ctx, int on) {
    if (!ctx) return JX_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->comm_stream));
    if (on && !ctx->comm_stream) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        for (auto& g : ctx->gslot) HIPCHK(ctx, hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
    }
    for (auto& g : ctx->gslot) g.busy = false;
    ctx->comm_overlap = on != 0;
    return JX_OK;
}

int jx_allgather_logp(jx_ctx* ctx, const double* send_dev, double* recv_dev, int count) {
    if (!ctx || !send_dev || !recv_dev || count < 0) return JX_ERR_INVALID;
    if (!ctx->comm) { ctx->err = "jx_allgather_logp before jx_comm_init_rank"; return JX_ERR_STATE; }
    if (count == 0) return JX_OK;
    int rc;
    hipStream_t cs;
    if ((rc = comm_fork(ctx, &cs))) return rc;
    std::pair<hipEvent_t, hipEvent_t> pr{nullptr, nullptr};
    const bool timed = ctx->timing_on && ctx->timing_mode == 1;    // (the stage pass; the timed region of bench.py carries no events of the gather)
    if (timed) {
        if (ctx->gt_inflight.size() > 4096 && (rc = gather_drain(ctx))) return rc;
        if (!ctx->gt_free.empty()) { pr = ctx->gt_free.back(); ctx->gt_free.pop_back(); }
        else { HIPCHK(ctx, hipEventCreate(&pr.first)); HIPCHK(ctx, hipEventCreate(&pr.second)); }
        HIPCHK(ctx, hipEventRecord(pr.first, cs));
    }
    NCCLCHK(ctx, g_rccl.AllGather(send_dev, recv_dev, (size_t)count, ncclDouble, ctx->comm, cs));
    if (timed) { HIPCHK(ctx, hipEventRecord(pr.second, cs)); ctx->gt_inflight.push_back(pr); }
    if (ctx->comm_overlap) {
        // remember the send buffer: the next evaluation that writes it waits for this gather (run_chunk), nothing else does
        jx_ctx::GatherSlot* slot = nullptr;
        for (auto& g : ctx->gslot) if (g.busy && g.send == (const char*)send_dev) slot = &g;
        if (!slot) for (auto& g : ctx->gslot) if (!g.busy) { slot = &g; break; }
        if (!slot) {                                                 // every slot taken: retire the oldest by waiting for it on the compute stream
            slot = &ctx->gslot[0];
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, slot->done, 0));
        }
        slot->send = (const char*)send_dev; slot->bytes = sizeof(double) * (size_t)count; slot->busy = true;
        HIPCHK(ctx, hipEventRecord(slot->done, cs));
    }
    return JX_OK;
}

int jx_comm_allreduce_max(jx_ctx* ctx, double* inout_dev, int count) {
    if (!ctx || !inout_dev || count < 1) return JX_ERR_INVALID;
    if (!ctx->comm) { ctx->err = "jx_comm_allreduce_max before jx_comm_init_rank"; return JX_ERR_STATE; }
    int rc;
    hipStream_t cs;
    if ((rc = comm_fork(ctx, &cs))) return rc;
    NCCLCHK(ctx, g_rccl.AllReduce(inout_dev, inout_dev, (size_t)count, ncclDouble, ncclMax, ctx->comm, cs));
    if (ctx->comm_overlap) {                                         // the result is awaited on the compute stream, as in strict mode
        HIPCHK(ctx, hipEventRecord(ctx->ev_fork, cs));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_fork, 0));
    }
    return JX_OK;
}

int jx_comm_gather_time(jx_ctx* ctx, double* ms_total, int64_t* calls) {
    if (!ctx || !ms_total || !calls) return JX_ERR_INVALID;
    int rc = gather_drain(ctx);
    if (rc) return rc;
    *ms_total = ctx->gather_ms; *calls = ctx->gather_calls;
    ctx->gather_ms = 0.0; ctx->gather_calls = 0;
    return JX_OK;
}

int jx_comm_destroy(jx_ctx* ctx) {
    if (!ctx) return JX_ERR_INVALID;
    if (!ctx->comm) return JX_OK;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
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

const char* jx_last_error(jx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_global_error.c_str(); }

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
    if (c.dtype < 0 || c.dtype > 2) return JX_ERR_INVALID;
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

}  // extern "C"

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
    for (int i = 0; i < N; ++i) {
        if (!(r[i] > 0) || (i && !(r[i] > r[i - 1]))) { ctx->err = "r_pp must be positive and increasing"; return JX_ERR_INVALID; }
    }
    std::vector<int32_t> thawed = host_vec<int32_t>(ctx, JX_T_THAWED_IDX);
    for (int k = 0; k < c.ndim; ++k)
        if (thawed[k] < 0 || thawed[k] >= c.npar) { ctx->err = "thawed_idx out of range"; return JX_ERR_INVALID; }

    // ---- does d_mat have the mirror structure d_mat[iy][ix] = Q[|iy-c|][|ix-c|] (what centdistmat builds)?
    std::vector<double> dm_h = host_vec<double>(ctx, JX_T_D_MAT);
    const int cc0 = S / 2, qn = std::max(cc0, S - 1 - cc0) + 1;
    std::vector<double>& Qtab = ctx->h_Qtab;
    Qtab.assign((size_t)qn * qn, 0.0);
    ctx->qn = qn;
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
    ctx->dmat_mirror = dmat_mirror;

    // ---- environment, read here and nowhere else
    int want = c.conv_mode;
    if (const char* e = opt_str(ctx, "JOXSZ_CONV")) {
        if (!strcmp(e, "rocfft")) want = 1; else if (!strcmp(e, "custom") || !strcmp(e, "mix")) want = 2; else if (!strcmp(e, "auto")) want = 0;
    }
    if (want < 0 || want > 2) { ctx->err = "conv_mode must be 0, 1 or 2"; return JX_ERR_INVALID; }
    double lr_tol0 = (S < 400) ? 1e-13 : JX_LR_TOL_DEFAULT;
    // (singular-value cut of the transfer-function weights: small maps -- a beam image comparable with the map, the
    //  log-posterior a small difference of large terms -- keep every term above rounding, where it costs next to nothing)
    if (const char* e = opt_str(ctx, "JOXSZ_LOWRANK_TOL")) { const double v2 = atof(e); if (v2 > 0.0 && v2 < 1e-6) { lr_tol0 = v2; ctx->tol_pinned = true; } }
    if (const char* e = opt_str(ctx, "JOXSZ_TRUNC_BOUND")) { const double v = atof(e); if (v > 0.0) { ctx->trunc_bound = v; ctx->trunc_bound_ll = std::min(ctx->trunc_bound_ll, 10.0 * v); } }
    if (const char* e = opt_str(ctx, "JOXSZ_MIX_FORM")) {
        if (!strcmp(e, "lowrank")) ctx->form_force = 0; else if (!strcmp(e, "full")) ctx->form_force = 1;
        else if (!strcmp(e, "legacy")) ctx->form_force = -1; else if (!strcmp(e, "exact")) ctx->form_force = 2;
    }
    if (const char* e = opt_str(ctx, "JOXSZ_MAP_PAIR")) ctx->map_pair = atoi(e) != 0;
    if (const char* e = opt_str(ctx, "JOXSZ_MIX_USPLIT")) { const int v = atoi(e); if (v >= 1 && v <= 4) { ctx->usplit = v; ctx->usplit_forced = true; } }
    if (const char* e = opt_str(ctx, "JOXSZ_MIX_SUBSAMPLE")) {
        // 0: every distinct sample; "u0,u1,npts": full resolution below u0 pixels from the axis, every second row up to u1, every fourth up to 2 u1, every eighth beyond
        int a0 = 0, a1 = 0, a2 = 0;
        const int got = sscanf(e, "%d,%d,%d", &a0, &a1, &a2);
        if (got == 1 && a0 == 0) ctx->subsample = false;
        else if (got == 3 && a0 >= 8 && a1 >= a0 && a2 >= 4 && a2 <= 16) { ctx->sub_u0 = a0; ctx->sub_u1 = a1; ctx->sub_npts = a2; }
    }
    if (const char* e = opt_str(ctx, "JOXSZ_EVAL_DIRECT")) { const int v = atoi(e); if (v >= 0 && v <= 3) ctx->eval_direct = v; }
    if (const char* e = opt_str(ctx, "JOXSZ_MIX_RANKCAP")) { const int v = atoi(e); if (v >= 0 && v <= 16) ctx->rank_cap = v; }
    if (const char* e = opt_str(ctx, "JOXSZ_MIX_MFMA")) ctx->mix_mfma = atoi(e) != 0;
    if (const char* e = opt_str(ctx, "JOXSZ_MIX_WPB")) { const int v = atoi(e); if (v >= 1 && v <= 16) { ctx->mix.wpb_force = v; ctx->mix.wpb = std::min(v, 4); } }
    if (const char* e = opt_str(ctx, "JOXSZ_MIX_KSPLIT")) { const int v = atoi(e); if (v >= 1) ctx->mix.ksplit_force = v; }
    if (const char* e = opt_str(ctx, "JOXSZ_MIX_KSPLIT_USE")) { const int v = atoi(e); if (v >= 1) ctx->mix.ksplit_u_force = v; }
    if (const char* e = opt_str(ctx, "JOXSZ_ABEL_GEMM")) ctx->abel_gemm = atoi(e) != 0;
    if (const char* e = opt_str(ctx, "JOXSZ_AG_NARROW")) ctx->ag_narrow = atoi(e) != 0;
    if (const char* e = opt_str(ctx, "JOXSZ_AG_SINGLE")) ctx->ag_single = atoi(e) != 0;
    if (const char* e = opt_str(ctx, "JOXSZ_X_PAIRWISE")) ctx->mix.x_pairwise = atoi(e) != 0;
    if (const char* e = opt_str(ctx, "JOXSZ_PREP_SPLIT")) ctx->prep_split = atoi(e) != 0;
    if (const char* e = opt_str(ctx, "JOXSZ_PREP_LEAN")) ctx->prep_lean = atoi(e) != 0;
    if (const char* e = opt_str(ctx, "JOXSZ_AG_SUBSAMPLE")) {
        int a0 = 0, a1 = 0, a2 = 0;
        const int got = sscanf(e, "%d,%d,%d", &a0, &a1, &a2);
        if (got == 1 && a0 == 0) ctx->ag_sub = false;
        else if (got == 3 && a0 >= 8 && a1 >= a0 && a2 >= 4 && a2 <= 24) { ctx->ag_u0 = a0; ctx->ag_u1 = a1; ctx->ag_npts = a2; }
    }
    if (const char* e = opt_str(ctx, "JOXSZ_PREP_FASTMATH")) ctx->prep_fastmath = atoi(e) != 0;
    ctx->op_narrow = opt_str(ctx, "JOXSZ_OP_NARROW") != nullptr;
    if (const char* e = opt_str(ctx, "JOXSZ_SIDE_STREAM")) ctx->side_on = atoi(e) != 0;
    if (const char* e = opt_str(ctx, "JOXSZ_SIDE_FORK")) ctx->side_fork = atoi(e) != 0;
    ctx->f32 = c.dtype >= 1;
    ctx->f32c = c.dtype == 2;
    if (ctx->f32 && !ctx->abel_gemm) { ctx->err = "dtype f32 takes its spline arrays from the matrix product only (JOXSZ_ABEL_GEMM=0 is an f64 setting)"; return JX_ERR_UNSUPPORTED; }

    int chunk = c.max_batch > 0 ? c.max_batch : 1024;
    if (const char* e = opt_str(ctx, "JOXSZ_CHUNK")) { int v = atoi(e); if (v > 0) chunk = v; }

    JxDev& d = ctx->d;
    memset(&d, 0, sizeof(d));
    d.S = S; d.N = N; d.B = B; d.Sh = ctx->Sh; d.nrow = ctx->nrow; d.nt = ctx->nt;
    d.nflux = c.nflux; d.nconv = c.nconv; d.nann = c.nann; d.nband = c.nband; d.ntab = c.ntab;
    d.npar = c.npar; d.ndim = c.ndim; d.ne_mode = c.ne_mode; d.exclude_unphy_mass = c.exclude_unphy_mass;
    d.sz_only = c.sz_only;
#ifdef JOXSZ_ABLATIONS
    if (const char* e = opt_str(ctx, "JOXSZ_DBG")) d.dbg = atoi(e);      // diagnostic build only: timing experiments, results are wrong
    if (const char* e = opt_str(ctx, "JOXSZ_MIX_DBG")) ctx->mix.dbg = atoi(e);
#endif
    d.y_scale = c.kpc_cm * c.sigma_T / c.m_e;
    d.inv_h_mean = (double)(N - 1) / (r[N - 1] - r[0]);
    d.prep_pow = opt_str(ctx, "JOXSZ_PREP_POW") && atoi(opt_str(ctx, "JOXSZ_PREP_POW")) ? 1 : 0;
    if (!c.sz_only) {
        const std::vector<double> lt = host_vec<double>(ctx, JX_T_LNT);
        for (int i = 1; i < c.ntab; ++i)
            if (!(lt[i] > lt[i - 1])) { ctx->err = "lnT must be increasing"; return JX_ERR_INVALID; }
        d.inv_dlnT = (double)(c.ntab - 1) / (lt[c.ntab - 1] - lt[0]);
    }

    int rc;
    // ---- evaluation matrix of g at the data radii (joxsz_funcs.py:476), and which outputs of the extracted row it reads at all:
    //      the cardinal functions of a cubic spline decay by 2 - sqrt(3) per knot, so data radii inside r_max give the row beyond
    //      r_max + ~35 pixels weights below 1e-22 of the largest -- nothing an fp64 sum can see.  The matrix-core product computes the
    //      outputs in use (whole tiles of 16); the row and brightness taps still get every output.
    std::vector<double> h_E, h_flux;
    {
        std::vector<double> radius = host_vec<double>(ctx, JX_T_RADIUS);
        std::vector<double> xk(radius.begin() + S / 2, radius.end());
        for (size_t i = 1; i < xk.size(); ++i)
            if (!(xk[i] > xk[i - 1])) { ctx->err = "radius[S//2:] must be increasing"; return JX_ERR_INVALID; }
        h_flux = host_vec<double>(ctx, JX_T_FLUX_DATA);
        std::vector<double> q(h_flux.begin(), h_flux.begin() + c.nflux);
        if (!jxt::nak_eval_matrix(xk, q, h_E)) { ctx->err = "profile spline: singular system"; return JX_ERR_INVALID; }
        if (const char* e = opt_str(ctx, "JOXSZ_PRUNE_OUTPUTS")) ctx->prune = atoi(e) != 0;
        double emax = 0.0;
        for (double v : h_E) if (std::isfinite(v)) emax = std::max(emax, std::fabs(v));
        int kuse = 0;
        bool finite = true;
        for (int k = 0; k < ctx->nrow; ++k)
            for (int dd = 0; dd < c.nflux; ++dd) {
                const double v = h_E[(size_t)dd * ctx->nrow + k];
                if (!std::isfinite(v)) finite = false;
                else if (std::fabs(v) > JX_PRUNE_TOL * emax) kuse = k + 1;
            }
        ctx->nrow_use = (finite && emax > 0.0) ? std::max(1, kuse) : ctx->nrow;
    }
    // ---- does the Abel + map kernel (taps, rocFFT route, the guard's reference facility) fit this radial grid in LDS?  Without it
    //      the contracted route still runs -- its timed kernels never needed it -- but nothing can measure a truncation, so there
    //      is none: every singular term above rounding is kept and no cap applies.
    {
        JxDev dt = d;
        dt.quad = 1; dt.fast_map = (dmat_mirror && qn <= 9 * 64) ? 1 : 0; dt.q_na = dt.q_nb = qn;
        int th; size_t ld;
        ctx->map_ok = map_geometry(dt, 512, &th, &ld);
        if (!ctx->map_ok) { lr_tol0 = std::min(lr_tol0, 1e-13); ctx->rank_cap = 0; ctx->tol_pinned = true; ctx->subsample = false; }   // (nothing could measure a truncation or a sub-grid: neither is taken)
    }
    // ---- spline moment operator of the mirrored grid (joxsz_funcs.py:460) and the half-width of its band
    if (!jxt::mirrored_spline_op(r, ctx->h_G)) { ctx->err = "spline operator: singular system"; return JX_ERR_INVALID; }
    ctx->K = d.K = jxt::band_halfwidth(ctx->h_G, N, 1e-20);
    // ---- which back end: contracted route or the rocFFT sequence
    std::vector<double> beam_h = host_vec<double>(ctx, JX_T_BEAM_2D);
    MixBuild mixb;
    const long long tW = ((long long)chunk + 127) & ~127LL;
    if (want != 1) {
        plan_mix(ctx, beam_h, host_vec<double>(ctx, JX_T_FILTERING), r, lr_tol0, ctx->form_force, tW, mixb);
        if (want == 2 && !mixb.ok) { ctx->err = "contracted route: " + mixb.why; return JX_ERR_UNSUPPORTED; }
    }
    ctx->conv_mode = mixb.ok ? 2 : 1;
    if (ctx->f32 && ctx->conv_mode != 2) {                       // never silently fall back to the fp64 arithmetic
        ctx->err = "dtype f32 is available on the contracted route only" + (mixb.why.empty() ? std::string() : " (" + mixb.why + ")");
        return JX_ERR_UNSUPPORTED;
    }
    if (ctx->f32c && mixb.form != 0) {
        ctx->err = "dtype 2 (fp32 arithmetic) exists on the low-rank form of the contracted route only; this problem takes the full form (dtype 1 keeps the arithmetic in fp64)";
        return JX_ERR_UNSUPPORTED;
    }
    int P = c.fft_pad > 0 ? c.fft_pad : jxt::next_smooth_even(S + o);
    if (const char* e = opt_str(ctx, "JOXSZ_FFT_PAD")) { int v = atoi(e); if (v > 0 && ctx->conv_mode == 1) P = v; }
    if (P < S + o) { ctx->err = "fft_pad smaller than S + (B-1)/2"; return JX_ERR_INVALID; }
    d.P = P; d.Ph = P / 2 + 1;

    // ---- Abel weights in on-the-fly form (per-source factor, diagonal, first off-diagonal)
    {
        std::vector<double> cj, dg, sp;
        jxt::abel_onfly_tables(r, cj, dg, sp);
        std::vector<double> tab((size_t)N * 4);
        for (int j = 0; j < N; ++j) { tab[4 * j] = r[j]; tab[4 * j + 1] = cj[j]; tab[4 * j + 2] = dg[j]; tab[4 * j + 3] = sp[j]; }
        double* p;
        if ((rc = dev_put(ctx, tab.data(), tab.size(), &p))) return rc; d.abel_tab = p;
    }
    // ---- spline moment operator of the mirrored grid, stored as a band; operator of the spline-array matrix product
    std::vector<double> h_Tm;
    {
        const std::vector<double>& G = ctx->h_G;
        const int K = ctx->K;
        std::vector<double> band((size_t)(2 * K + 1) * N, 0.0);
        for (int i = 0; i < N; ++i)
            for (int k = -K; k <= K; ++k) {
                const int j = i + k;
                if (j >= 0 && j < N) band[(size_t)(k + K) * N + i] = G[(size_t)i * N + j];
            }
        double* p; if ((rc = dev_put(ctx, band.data(), band.size(), &p))) return rc; d.gband = p;
        if (ctx->conv_mode == 2 && mixb.form != 2) {
            ctx->tm_ntile = (2 * N + 15) / 16;
            ctx->tm_npair = (ctx->tm_ntile + 1) / 2;
            ctx->tm_ld = 32 * ctx->tm_npair;
            jxt::abel_spline_operator(r, G, K, d.y_scale, JX_AG_ROWS(N), ctx->tm_ld, h_Tm);
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
    std::vector<double> h_hw;
    {
        std::vector<double> rt(r.begin(), r.begin() + ctx->nt), G;
        if (!jxt::mirrored_spline_op(rt, G)) { ctx->err = "h(0) operator: singular system"; return JX_ERR_INVALID; }
        std::vector<double>& hw = h_hw;
        hw.assign(ctx->nt, 0.0);
        for (int j = 0; j < ctx->nt; ++j) hw[j] = -0.5 * rt[0] * rt[0] * G[j];
        hw[0] += 1.0;
        double* p; if ((rc = dev_put(ctx, hw.data(), hw.size(), &p))) return rc; d.hw = p;
    }
    {
        double* p; if ((rc = dev_put(ctx, h_E.data(), h_E.size(), &p))) return rc; d.emat = p;
        if ((rc = dev_put(ctx, h_flux.data(), h_flux.size(), &p))) return rc; d.flux = p;
    }
    // ---- twiddles of the final inverse transform of the extracted row (rocFFT sequence's tail)
    {
        double* p;
        std::vector<double> tw((size_t)S * 2);
        for (int m = 0; m < S; ++m) { tw[2 * m] = std::cos(2.0 * jxt::kPi * m / S); tw[2 * m + 1] = std::sin(2.0 * jxt::kPi * m / S); }
        if ((rc = dev_put(ctx, tw.data(), tw.size(), &p))) return rc; d.twid = p;
    }
    // ---- plain copies
    {
        double* p; int* q;
#define PUTD(field, id) { std::vector<double> v = host_vec<double>(ctx, id); if ((rc = dev_put(ctx, v.data(), v.size(), &p))) return rc; d.field = p; }
#define PUTI(field, id) { std::vector<int32_t> v = host_vec<int32_t>(ctx, id); if ((rc = dev_put(ctx, v.data(), v.size(), &q))) return rc; d.field = q; }
        {
            std::vector<double> lr = r;
            for (double& v : lr) v = std::log(v);
            if ((rc = dev_put(ctx, lr.data(), lr.size(), &p))) return rc;
            d.lr_pp = p;
        }
        {
            std::vector<double> fm;
            jxt::fastmath_tables(fm);                             // exp / log tables of the per-walker kernel (jx_fastmath.hpp)
            if ((rc = dev_put(ctx, fm.data(), fm.size(), &p))) return rc;
            d.fm_tab = p;
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
        {   // exp / log tables, radii and their logarithms, h(0) weights, conversion table in one run: what an SZ-side block of jx_walker2_kernel copies into LDS
            std::vector<double> pack;
            jxt::fastmath_tables(pack);
            pack.insert(pack.end(), r.begin(), r.end());
            for (double v : r) pack.push_back(std::log(v));
            pack.insert(pack.end(), h_hw.begin(), h_hw.end());
            for (int id : {JX_T_CONV_T, JX_T_CONV_V}) { const std::vector<double> v = host_vec<double>(ctx, id); pack.insert(pack.end(), v.begin(), v.end()); }
            if (pack.size() != JX_SZ_PACK_DOUBLES(d)) { ctx->err = "per-walker table pack: size"; return JX_ERR_INVALID; }
            if ((rc = dev_put(ctx, pack.data(), pack.size(), &p))) return rc;
            d.sz_pack = p;
        }
        if (!c.sz_only) {
            PUTD(x_r_ne, JX_T_X_R_NE) PUTD(x_r_T, JX_T_X_R_T) PUTD(projvols, JX_T_PROJVOLS) PUTD(cts, JX_T_CTS)
            PUTD(areascales, JX_T_AREASCALES) PUTD(exposures, JX_T_EXPOSURES) PUTD(backrates, JX_T_BACKRATES)
            PUTD(geomarea, JX_T_GEOMAREA) PUTD(lnT, JX_T_LNT) PUTD(lnrate, JX_T_LNRATE)
            // exp / log tables and the small tables of the X-ray side in one run: what an X-ray block of jx_walker2_kernel copies into LDS
            std::vector<double> pack;
            jxt::fastmath_tables(pack);
            for (int id : {JX_T_X_R_NE, JX_T_X_R_T, JX_T_GEOMAREA, JX_T_LNT, JX_T_PROJVOLS, JX_T_AREASCALES, JX_T_EXPOSURES, JX_T_BACKRATES, JX_T_CTS}) {
                const std::vector<double> v = host_vec<double>(ctx, id);
                pack.insert(pack.end(), v.begin(), v.end());
            }
            if (pack.size() != JX_XR_PACK_DOUBLES(d)) { ctx->err = "X-ray table pack: size"; return JX_ERR_INVALID; }
            if ((rc = dev_put(ctx, pack.data(), pack.size(), &p))) return rc;
            d.xr_pack = p;
        }
#undef PUTD
#undef PUTI
    }
    // ---- symmetric-map tables of the Abel + map kernel: coefficient slot and local abscissa of every quadrant radius
    {
        const int na = qn;
        const bool sym = dmat_mirror && na <= 9 * 64;       // register-resident half row: |ix-c| < 576
        d.fast_map = sym ? 1 : 0;
        d.q_na = d.q_nb = na;
        if (sym) {
            std::vector<int32_t> qk((size_t)na * na);
            std::vector<double> qt((size_t)na * na);
            for (size_t e = 0; e < Qtab.size(); ++e) {
                const double dd = Qtab[e];
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
    int split = c.map_split > 0 ? c.map_split : 1;
    if (const char* e = opt_str(ctx, "JOXSZ_MAP_SPLIT")) { int v = atoi(e); if (v > 0) split = v; }
    d.map_split = std::min(split, S);

    // ---- chunk capacity: the work buffers of a launch stay under 24 GB
    const size_t budget = (size_t)24 << 30;
    if (ctx->conv_mode == 2) {
        // quadrant image of the Compton-y map (y_2d tap; only the profile taps and the full-map measurement use the kernel)
        d.quad = 1;
        d.img_ld = (d.q_na + 16) & ~15; d.img_ws = (long long)d.q_nb * d.img_ld;          // rows start on cache lines, >= 1 spare column
        const size_t per_walker = sizeof(double) * ((size_t)mixb.krows + (size_t)(JX_MIX_KSPLIT_MAX + 1) * 16 * mixb.ntile + 8 * (size_t)N + 64);
        if (c.max_batch <= 0 && !opt_str(ctx, "JOXSZ_CHUNK")) while (chunk > 128 && per_walker * chunk > budget) chunk /= 2;
    } else {
        d.quad = 0; d.img_ld = P; d.img_ws = (long long)P * P;
        const size_t per_walker = sizeof(double) * ((size_t)P * P * 2 + (size_t)P * d.Ph * 2 + (size_t)S * ctx->Sh * 2);
        while (chunk > 1 && per_walker * chunk > budget) chunk /= 2;
    }
    ctx->chunk = chunk;
    if (!map_geometry(d, 512, &ctx->map_threads, &ctx->map_lds)) {
        if (ctx->conv_mode != 2) { ctx->err = "radial grid too long for the LDS-resident spline of the Abel + map kernel (the rocFFT sequence needs it)"; return JX_ERR_UNSUPPORTED; }
        // the contracted forms decided their truncation and sub-grids on the earlier check of the same geometry: had that one passed and this one not,
        // those approximations would run with nothing to measure them -- refuse rather than run unguarded; the exact form takes none
        if (ctx->map_ok && mixb.form != 2) { ctx->err = "radial grid too long for the Abel + map kernel after the contracted tables were planned with it: use the exact form (JOXSZ_MIX_FORM=exact)"; return JX_ERR_UNSUPPORTED; }
        ctx->map_ok = false;
    }
    {
        const int lds = 160 * 1024 - 1024;
#define JX_ATTR(V, NA) HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_abel_map_sym_kernel<V, NA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds))
        JX_ATTR(true, 3); JX_ATTR(true, 5); JX_ATTR(true, 9); JX_ATTR(false, 3); JX_ATTR(false, 5); JX_ATTR(false, 9);
#undef JX_ATTR
        HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_abel_map_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_abel_map_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    if ((rc = dev_new(ctx, (size_t)chunk, &ctx->d_base))) return rc;
    if ((rc = dev_new(ctx, (size_t)chunk * 2, &ctx->d_xr))) return rc;
    if ((rc = dev_new(ctx, (size_t)chunk * ctx->nrow, &ctx->d_cfac))) return rc;
    if (c.calc_integ && (rc = dev_new(ctx, (size_t)chunk, &ctx->d_sz0, true))) return rc;
    if (ctx->conv_mode == 2) {
        const long long tW2 = ((long long)chunk + 127) & ~127LL;
        if (tW2 != tW && mixb.form == 1) {                       // (the chunk shrank: the full form's sample offsets carry the walker stride)
            mixb = MixBuild();
            plan_mix(ctx, beam_h, host_vec<double>(ctx, JX_T_FILTERING), r, lr_tol0, ctx->form_force, tW2, mixb);
            if (!mixb.ok) { ctx->err = "contracted route: " + mixb.why; return JX_ERR_UNSUPPORTED; }
        }
        if ((rc = dev_new(ctx, (size_t)chunk * d.q_nb, &d.xcol))) return rc;
        if ((rc = mix_setup(ctx, mixb, tW2))) return rc;
        if (ctx->side_on) {
            HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_side, hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_tail, hipEventDisableTiming));
        }
        if ((rc = dev_put(ctx, h_Tm.data(), h_Tm.size(), &ctx->d_Tm))) return rc;
        if ((rc = dev_new(ctx, (size_t)chunk * (mixb.form == 2 ? 16 * (size_t)mixb.x_nSj : (size_t)N), &ctx->d_ppc, true))) return rc;   // (exact form: rows padded with zeros to whole 16-radius steps)
        // The pressure profile is smooth away from the core: its values on a sub-grid of the radial grid carry the others by high-order
        // interpolation, pp ~ L pp_sub, and the product needs only L^T Tm -- K shrinks from N to the kept radii (222 of 500, 285 of 1000)
        // and with it the k loop that bounds the kernel.  Measured by the guard of jx_finalize like the sub-grid of map samples.
        if (ctx->ag_sub && ctx->map_ok && mixb.form != 2) {
            std::vector<int> rs;
            jxt::mix_row_subset(N, ctx->ag_u0, ctx->ag_u1, rs);
            const int ns = (int)rs.size();
            if (ns <= (3 * N) / 4 && ns >= 2 * ctx->ag_npts && ctx->ag_u0 >= ctx->ag_npts) {
                std::vector<double> L;
                jxt::mix_interp_matrix(N, rs, ctx->ag_npts, L);
                const int ld = ctx->tm_ld, rows = JX_AG_ROWS(ns);
                std::vector<double> Ts((size_t)rows * ld, 0.0);
                for (int j = 0; j < N; ++j)
                    for (int a = 0; a < ns; ++a) {
                        const double l = L[(size_t)j * ns + a];
                        if (l == 0.0) continue;
                        const double* src = &h_Tm[(size_t)j * ld];
                        double* dst = &Ts[(size_t)a * ld];
                        for (int e = 0; e < ld; ++e) dst[e] += l * src[e];
                    }
                // first k-step (four rows) of every column tile with an entry; made non-decreasing in the tile (the kernel's blocks start at their lowest tile's)
                std::vector<int> tks(ctx->tm_ntile, 0);
                for (int t = 0; t < ctx->tm_ntile; ++t) {
                    int first = ns;
                    for (int a = 0; a < ns && first == ns; ++a)
                        for (int e = 16 * t; e < std::min(16 * t + 16, ld); ++e) if (Ts[(size_t)a * ld + e] != 0.0) { first = a; break; }
                    tks[t] = std::min(first, ns - 1) / 4;
                }
                for (int t = ctx->tm_ntile - 2; t >= 0; --t) tks[t] = std::min(tks[t], tks[t + 1]);
                if ((rc = dev_put(ctx, Ts.data(), Ts.size(), &ctx->d_Tm_s))) return rc;
                if ((rc = dev_put(ctx, rs.data(), rs.size(), &ctx->d_rsub))) return rc;
                if ((rc = dev_put(ctx, tks.data(), tks.size(), &ctx->d_tks))) return rc;
                ctx->h_rsub = rs; ctx->ag_ns = ns; ctx->ag_sub_on = true;
            }
        }
        HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_rowmix_mfma_kernel<JX_MIX_NS, double2>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
        HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_rowmix_mfma_kernel<JX_MIX_NS, float2>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
#define JX_OPG_ATTR(Xv) HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_opgemm_kernel<1, Xv, double2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); \
        HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_opgemm_kernel<1, Xv, float2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        JX_MIX_NXTS(JX_OPG_ATTR)
#undef JX_OPG_ATTR
#define JX_ROP_ATTR(Xv) HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_rowop_tail_kernel<Xv>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024)); \
        HIPCHK(ctx, hipFuncSetAttribute((const void*)jx_rowsum_tail_kernel<Xv>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
        JX_MIX_NXTS(JX_ROP_ATTR)
#undef JX_ROP_ATTR
    } else {
        if ((rc = fft_setup(ctx, ctx->fft, chunk, P))) return rc;
    }
    HIPCHK(ctx, hipDeviceSynchronize());                          // (the zero fills ran on the null stream)
    ctx->finalized = true;
    return JX_OK;
}

static int ensure_batch(jx_ctx* ctx, int n) {
    if (n <= ctx->batch_cap) return JX_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_theta) { (void)hipFree(ctx->d_theta); (void)hipFree(ctx->d_logp); ctx->d_theta = ctx->d_logp = nullptr; ctx->batch_cap = 0; }
    if (ctx->h_theta) { (void)hipHostFree(ctx->h_theta); (void)hipHostFree(ctx->h_logp); ctx->h_theta = ctx->h_logp = nullptr; }
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_theta, sizeof(double) * (size_t)n * ctx->cfg.ndim));
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_logp, sizeof(double) * (size_t)n));
    HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_theta, sizeof(double) * (size_t)n * ctx->cfg.ndim, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_logp, sizeof(double) * (size_t)n, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(ctx, hipHostGetDevicePointer((void**)&ctx->hd_theta, ctx->h_theta, 0));
    HIPCHK(ctx, hipHostGetDevicePointer((void**)&ctx->hd_logp, ctx->h_logp, 0));
    ctx->batch_cap = n;
    return JX_OK;
}

static int get_evset(jx_ctx* ctx, EvSet* out) {
    if (!ctx->ev_free.empty()) { *out = ctx->ev_free.back(); ctx->ev_free.pop_back(); return JX_OK; }
    for (int k = 0; k < 6; ++k) HIPCHK(ctx, hipEventCreate(&out->e[k]));
    return JX_OK;
}

static int drain_events(jx_ctx* ctx) {
    if (ctx->ev_inflight.empty()) return JX_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& es : ctx->ev_inflight) {
        float ms[5] = {0, 0, 0, 0, 0}, tot;
        if (es.p1only) {                               // two events per launch sequence: one kernel of the step alone
            HIPCHK(ctx, hipEventElapsedTime(&ms[2], es.e[2], es.e[3]));
            if (es.p1stage == 3) ctx->acc.prep_ms += ms[2];
            else if (es.p1stage == 4) ctx->acc.tail_ms += ms[2];
            else if (es.exact) ctx->acc.abel_map_ms += ms[2];
            else ctx->acc.beam_fft_ms += ms[2];
            ctx->acc.launches += 1; ctx->acc.walkers += es.walkers;
            ctx->ev_free.push_back(es);
            continue;
        }
        if (es.op) {                                   // collapsed route: prep, then one kernel
            HIPCHK(ctx, hipEventElapsedTime(&ms[0], es.e[0], es.e[1]));
            HIPCHK(ctx, hipEventElapsedTime(&ms[4], es.e[1], es.e[5]));
        } else if (es.exact) {                         // exact form: per-walker kernel, ordinate product, row product + tail
            HIPCHK(ctx, hipEventElapsedTime(&ms[0], es.e[0], es.e[1]));
            HIPCHK(ctx, hipEventElapsedTime(&ms[1], es.e[1], es.e[2]));
            HIPCHK(ctx, hipEventElapsedTime(&ms[4], es.e[2], es.e[5]));
        } else {
            for (int k = 0; k < 5; ++k) HIPCHK(ctx, hipEventElapsedTime(&ms[k], es.e[k], es.e[k + 1]));
        }
        HIPCHK(ctx, hipEventElapsedTime(&tot, es.e[0], es.e[5]));
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
           *tprof = nullptr, *xprofs = nullptr, *parts = nullptr, *integ = nullptr;
    bool need_img = false;            // the Compton-y map itself is wanted (y_2d tap)
    bool need_conv = false;           // the beam-convolved map is wanted (conv_2d tap): the hand-written row kernels write it only then
};

// the reference facility of a contracted-route context: the rocFFT sequence for up to 16 walkers at a time
static int ensure_ref(jx_ctx* ctx) {
    if (ctx->fft.ready) return JX_OK;
    if (!ctx->map_ok) { ctx->err = "radial grid too long for the Abel + map kernel: the rocFFT reference facility (beam-convolved map tap, truncation probe) does not exist on this problem"; return JX_ERR_UNSUPPORTED; }
    return fft_setup(ctx, ctx->fft, std::min(16, ctx->chunk), ctx->d.P);
}

// One chunk: walkers [w0, w0+n) of the batch whose thetas live at theta_dev.  use_ref: through the reference facility
// (rocFFT sequence) of a contracted-route context instead of its own back end.
static int run_chunk(jx_ctx* ctx, const double* theta_dev, double* logp_dev, int w0, int n, const Taps& t, bool use_ref = false, const JxSm* smp = nullptr /* contracted route: the stretch move inside the per-walker kernel and the tail */) {
    JxSm smv;
    memset(&smv, 0, sizeof(smv));
    if (smp) smv = *smp;
    const bool fftb = ctx->conv_mode == 1 || use_ref;
    const JxDev& d = fftb ? ctx->fft.d : ctx->d;
    const bool op_route = ctx->route == JX_ROUTE_OPERATOR && !t.pp && !ctx->d.inject_pp && !use_ref;   // stage taps and the operator build: map route
    int rc;
    hipStream_t st = ctx->stream;
    if (ctx->comm_overlap)                                         // a gather still reading this output buffer: the evaluation waits for it, and only for it
        for (auto& g : ctx->gslot)
            if (g.busy && g.send < (const char*)(logp_dev + w0 + n) && (const char*)(logp_dev + w0) < g.send + g.bytes) {     // (byte ranges overlap)
                // (a gather that has already finished costs a query on the host, not a barrier in the queue)
                if (hipEventQuery(g.done) != hipSuccess) HIPCHK(ctx, hipStreamWaitEvent(st, g.done, 0));
                g.busy = false;
            }
    EvSet es;
    // timing mode 2 records the two events around the time-dominant kernel only (stage 1 of the contracted route)
    const bool tm = ctx->timing_on && ctx->timing_mode == 1 && !use_ref;
    const bool tm2 = ctx->timing_on && ctx->timing_mode >= 2 && !fftb && !op_route;
    // the operator route has no per-walker work buffers beyond these three, so its launches can be much larger than a chunk
    double* base_buf = op_route ? ctx->d_base_op : ctx->d_base;
    double* cfac_buf = op_route ? ctx->d_cfac_op : ctx->d_cfac;
    double* sz0_buf = op_route ? ctx->d_sz0_op : ctx->d_sz0;           // null unless calc_integ
    const bool mix = !fftb && !op_route;
    // contracted route: the spline arrays come from one matrix product, unless the profile taps are asked for (they live in
    // the Abel kernel, which then writes the arrays itself) or the matrix product is switched off
    const bool want_abel_taps = t.pp || t.ab || t.y;
    // (a grid too long for the Abel kernel: always the matrix product; the profile taps are then read off its own arrays)
    const bool exact = mix && ctx->mix.form == 2;                   // the exact form: ordinate product + row product with the tail as its epilogue
    const bool ag = exact || (mix && ctx->abel_gemm && (ctx->f32 || !ctx->map_ok || (!want_abel_taps && !ctx->d.inject_pp)));
    if (tm || tm2) {
        if (ctx->ev_inflight.size() > 2048 && (rc = drain_events(ctx))) return rc;
        if ((rc = get_evset(ctx, &es))) return rc;
        es.walkers = n;
        es.op = false;
        es.p1only = tm2;
        es.exact = exact;
        es.p1stage = (exact && ctx->timing_mode >= 3) ? ctx->timing_mode : 2;
        if (tm) HIPCHK(ctx, hipEventRecord(es.e[0], st));
    }
    // Contracted route: only the pressure profile is in the way of the SZ chain (spline arrays -> stage 1 -> stage 2); priors,
    // vetoes, X-ray side and conversion factors feed the tail alone.  So a small kernel evaluates the profile in line, and the
    // per-walker kernel runs on a second stream BESIDE THE MATRIX-CORE PRODUCT: forked by an event behind stage 1 (stage 1
    // fills every wave slot of the chip, the product leaves more than half of the registers free), joined by one in front of
    // the tail.  Everything the fork event follows on the compute stream -- the previous tail, which reads the buffers this
    // kernel writes, and whatever produced theta -- is therefore complete when it starts.  Calls with taps stay in line.
    const bool any_tap = t.pp || t.ab || t.y || t.row || t.bright || t.chisq || t.tprof || t.xprofs || t.parts || t.integ || t.need_img;
    const bool side = mix && ag && !exact && ctx->side_stream && !ctx->d.inject_pp && !any_tap;
    // two blocks per walker in the per-walker kernel (the X-ray side beside the rest): the timed sequence of the contracted route only
    // (up to ~640 radii: beyond, the grid pass on half the threads is the longer of the two chains by more than the split saves --
    //  measured at N = 1000: 32.5 -> 34.7 us; at N = 500: 27.6 -> 23.4, at N = 313: 26.8 -> 22.3)
    const bool xr_split = mix && ctx->prep_split && !any_tap && !side && !d.sz_only && !d.prep_pow && !d.calc_integ /* (its sums are grouped by the block's size) */ && 3 * d.nann <= 128 && d.nband * d.nann <= 4096 && d.N <= 640;
    auto launch_prep = [&](hipStream_t ps, double* pp_buf) {
        JxDev dp = d;
        dp.inject_pp = ctx->d.inject_pp;
        dp.xr_split = xr_split ? 1 : 0;
        dp.xr_out = ctx->d_xr;
        dp.pp_ld = (exact && pp_buf == ctx->d_ppc) ? ctx->mix.x_ldpp : 0;
#ifdef JOXSZ_ABLATIONS
        if (opt_str(ctx, "JOXSZ_P_STAMPS")) {                        // (diagnostic build only)
            static long long* stamps = nullptr;
            if (!stamps) { (void)hipMalloc((void**)&stamps, sizeof(long long) * 8 * 65536); (void)hipMemset(stamps, 0, sizeof(long long) * 8 * 65536); }
            dp.stamps = stamps;
            static int calls = 0;
            if (++calls == 40) {
                (void)hipStreamSynchronize(ps);
                const int nb = xr_split ? 2 * n : n;
                std::vector<long long> h((size_t)nb * 8);
                (void)hipMemcpy(h.data(), stamps, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
                long long t0 = h[0];
                for (int b = 0; b < nb; ++b) t0 = std::min(t0, h[(size_t)b * 8]);
                for (int half = 0; half < (xr_split ? 2 : 1); ++half) {
                    const int b0 = half * n, b1 = b0 + n;
                    double mx[7] = {0}, sm[7] = {0};
                    for (int b = b0; b < b1; ++b)
                        for (int k = 0; k < 7; ++k) { const double v = h[(size_t)b * 8 + k] ? (h[(size_t)b * 8 + k] - t0) * 0.01 : 0.0; mx[k] = std::max(mx[k], v); sm[k] += v / n; }
                    fprintf(stderr, "prep stamps, %s blocks (us after the first start; mean/max): start %.2f/%.2f params %.2f/%.2f priors %.2f/%.2f grid pass %.2f/%.2f veto+integ %.2f/%.2f factors %.2f/%.2f end %.2f/%.2f\n",
                            half ? "X-ray" : (xr_split ? "SZ-side" : "all"), sm[0], mx[0], sm[1], mx[1], sm[2], mx[2], sm[3], mx[3], sm[4], mx[4], sm[5], mx[5], sm[6], mx[6]);
                }
            }
        }
#endif
        const unsigned pgrid = xr_split ? 2u * (unsigned)n : (unsigned)n, pthr = xr_split ? 128u : (unsigned)JX_PREP_THREADS;
        if (xr_split && ctx->prep_lean && ctx->prep_fastmath && pp_buf && sizeof(double) * JX_W2_LDS_DOUBLES(d) <= (size_t)19 * 1024 + 512) {
            // the two-block form as two lean roles (same arithmetic, same bits; every table of a role in one batched copy into LDS)
            hipLaunchKernelGGL(jx_walker2_kernel, dim3(pgrid), dim3(JX_W2_THREADS), sizeof(double) * JX_W2_LDS_DOUBLES(d), ps, dp, theta_dev, w0, base_buf, cfac_buf, pp_buf, smv);
            return;
        }
        const size_t sh = sizeof(double) * (JX_LDS_HDR + (size_t)2 * d.N + 2 * d.nann + 2 * (size_t)d.nband * d.nann + 2 * (size_t)d.nconv + 8 + JX_FM_TABLE_DOUBLES);
        if (d.prep_pow) hipLaunchKernelGGL((jx_prep_kernel<true, false>), dim3(pgrid), dim3(pthr), sh, ps, dp, theta_dev, w0,
                                           base_buf, cfac_buf, pp_buf, sz0_buf, t.tprof, t.xprofs, t.parts, t.integ, smv);
        else if (ctx->prep_fastmath) hipLaunchKernelGGL((jx_prep_kernel<false, true>), dim3(pgrid), dim3(pthr), sh, ps, dp, theta_dev, w0,
                                                        base_buf, cfac_buf, pp_buf, sz0_buf, t.tprof, t.xprofs, t.parts, t.integ, smv);
        else hipLaunchKernelGGL((jx_prep_kernel<false, false>), dim3(pgrid), dim3(pthr), sh, ps, dp, theta_dev, w0,
                                base_buf, cfac_buf, pp_buf, sz0_buf, t.tprof, t.xprofs, t.parts, t.integ, smv);
    };
    {
        double* pp_buf = op_route ? ctx->d_pp : ((ag && !ctx->d.inject_pp) ? ctx->d_ppc : (double*)nullptr);
        if (side) {
            JxDev dp = d;
            const size_t shp = sizeof(double) * (JX_LDS_HDR + 8);
            if (d.prep_pow) hipLaunchKernelGGL(jx_pp_kernel<true>, dim3(n), dim3(128), shp, st, dp, theta_dev, w0, pp_buf);
            else hipLaunchKernelGGL(jx_pp_kernel<false>, dim3(n), dim3(128), shp, st, dp, theta_dev, w0, pp_buf);
            if (ctx->side_fork == 0) {                                      // fork behind the profile kernel: beside the spline-array product
                HIPCHK(ctx, hipEventRecord(ctx->ev_tail, st));
                HIPCHK(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_tail, 0));
                launch_prep(ctx->side_stream, nullptr);
                HIPCHK(ctx, hipEventRecord(ctx->ev_side, ctx->side_stream));
            }
        } else {
            if (tm2 && es.p1stage == 3) HIPCHK(ctx, hipEventRecord(es.e[2], st));
            launch_prep(st, pp_buf);
            if (tm2 && es.p1stage == 3) HIPCHK(ctx, hipEventRecord(es.e[3], st));
        }
    }
    if (tm) HIPCHK(ctx, hipEventRecord(es.e[1], st));
    if (op_route) {
        // collapsed route: the SZ side is one kernel (no map, no transforms)
        es.op = true;
        const int nrow = d.nrow, Re = (nrow + 1) & ~1;
        // large launches: G pp on the fp64 matrix cores (G fetched once per 32 walkers), the rest in the kernel behind it
        const int mt = (nrow + 63) / 64;                                     // row tiles per wave
        const bool wide = n >= 4096 && (mt == 1 || mt == 2 || mt == 4 || mt == 8) && !ctx->op_narrow;
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
    if (mix) {
        MixBack& m = ctx->mix;
        if (ctx->f32 && (want_abel_taps || t.need_img)) { ctx->err = "dtype f32: the profile and map taps exist in the f64 build of the context only"; return JX_ERR_UNSUPPORTED; }
        if (ctx->f32 && !ag) { ctx->err = "dtype f32 takes its spline arrays from the matrix product only (JOXSZ_ABEL_GEMM=0 is an f64 setting)"; return JX_ERR_UNSUPPORTED; }
        if (!ctx->map_ok && (!ag || t.need_img)) { ctx->err = "radial grid too long for the Abel + map kernel: no Compton-y map tap on this problem (and JOXSZ_ABEL_GEMM=0 is not available)"; return JX_ERR_UNSUPPORTED; }
        if (exact) {
            // ---- exact form: y = y_scale A pp on the matrix cores and each column-tile pair's share of out = Wy y (jx_ordrow_kernel), then the
            //      partial rows added in pair order with the tail behind them (jx_rowsum_tail_kernel)
            const int N = d.N;
            const double* pp_src = ctx->d_ppc;
            if (ctx->d.inject_pp) {                              // (operator build: the unit profiles into the padded rows of the product)
                if (!m.x_ppi && (rc = dev_new_l(ctx, m.allocs, (size_t)ctx->chunk * m.x_ldpp, &m.x_ppi, true))) return rc;
                HIPCHK(ctx, hipMemcpy2DAsync(m.x_ppi, sizeof(double) * m.x_ldpp, ctx->d.inject_pp, sizeof(double) * N, sizeof(double) * N, n, hipMemcpyDeviceToDevice, st));
                pp_src = m.x_ppi;
            }
            JxRowOp ro;
            memset(&ro, 0, sizeof(ro));
            const bool all = t.row || t.bright;                  // the row and brightness taps get every output; the data-radii matrix reads the same ones either way
            ro.n = n; ro.nS = m.Nkp / 16; ro.ldy = m.Nkp; ro.ng = all ? m.x_ng : m.x_ng_use;
            ro.nuse = std::min(ctx->prune ? ctx->nrow_use : d.nrow, d.nrow);
            ro.ldr = (ro.nuse + 1) | 1;
            ro.lde = ro.nuse;
            ro.ldpp = m.x_ldpp; ro.nSj = m.x_nSj; ro.npair = m.x_npair;
            ro.nfold = 0; ro.s0f = 0; ro.Wfk = nullptr;
            const bool lean = m.x_lean && m.x_pairwise && !all && !want_abel_taps;     // the timed path: nothing reads the ordinates
            if (m.x_nfold && lean) {                             // folded form: the last ordinate tile is not computed, one pair fewer
                ro.nfold = m.x_nfold; ro.s0f = ro.nS - 1; ro.Wfk = m.x_Wfk; ro.npair = (ro.nS - 1) / 2;
            }
#ifdef JOXSZ_ABLATIONS
            ro.dbg = m.dbg;
            if (opt_str(ctx, "JOXSZ_X_STAMPS")) {                    // (diagnostic build only: the environment is read on the launch path here)
                static long long* stamps = nullptr;
                if (!stamps) { HIPCHK(ctx, hipMalloc((void**)&stamps, sizeof(long long) * 8 * 65536)); }
                ro.stamps = stamps;
                static int calls = 0;
                if (++calls == 40) {
                    HIPCHK(ctx, hipStreamSynchronize(st));
                    const int nb = 8 * ((((n + 15) / 16) + 7) / 8) * ro.npair;
                    std::vector<long long> h((size_t)nb * 8);
                    HIPCHK(ctx, hipMemcpy(h.data(), stamps, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
                    long long t0 = h[0];
                    for (int b = 0; b < nb; ++b) t0 = std::min(t0, h[(size_t)b * 8]);
                    double mx[5] = {0, 0, 0, 0, 0}, sm[5] = {0, 0, 0, 0, 0};
                    for (int b = 0; b < nb; ++b)
                        for (int k = 0; k < 5; ++k) { const double v = (h[(size_t)b * 8 + k] - t0) * 0.01; mx[k] = std::max(mx[k], v); sm[k] += v / nb; }
                    fprintf(stderr, "ordrow stamps (us after the first block's start; mean / max over %d blocks): start %.2f/%.2f  first loads %.2f/%.2f  k loop %.2f/%.2f  ordinates %.2f/%.2f  end %.2f/%.2f\n",
                            nb, sm[0], mx[0], sm[1], mx[1], sm[2], mx[2], sm[3], mx[3], sm[4], mx[4]);
                    {                                                            // where the late first operands are: by XCD, by eighth of the grid, and the ten latest blocks
                        double bx[8] = {0}, bo[8] = {0}; int cx[8] = {0}, co[8] = {0}, late = 0;
                        std::vector<std::pair<double, int>> fl;
                        for (int b = 0; b < nb; ++b) {
                            const double v = (h[(size_t)b * 8 + 1] - h[(size_t)b * 8]) * 0.01;
                            bx[b & 7] += v; ++cx[b & 7]; bo[b * 8 / nb] += v; ++co[b * 8 / nb]; if (v > 3.0) ++late;
                            fl.push_back({v, b});
                        }
                        std::sort(fl.rbegin(), fl.rend());
                        fprintf(stderr, "  first operands after the block's own start, mean by XCD:");
                        for (int k = 0; k < 8; ++k) fprintf(stderr, " %.2f", bx[k] / std::max(1, cx[k]));
                        fprintf(stderr, " | by eighth of the grid:");
                        for (int k = 0; k < 8; ++k) fprintf(stderr, " %.2f", bo[k] / std::max(1, co[k]));
                        fprintf(stderr, " | %d blocks beyond 3 us; latest:", late);
                        for (int k = 0; k < 10 && k < (int)fl.size(); ++k) fprintf(stderr, " %d(%.1f, start %.1f)", fl[k].second, fl[k].first, (h[(size_t)fl[k].second * 8] - t0) * 0.01);
                        fprintf(stderr, "\n");
                    }
                    for (int pp_ = 0; pp_ < ro.npair; ++pp_) {                  // per pair: mean first loads | k loop | end
                        double a[3] = {0, 0, 0}, e_max = 0; int cnt = 0;
                        for (int b = 0; b < nb; ++b) if ((b >> 3) % ro.npair == pp_) {
                            a[0] += (h[(size_t)b * 8 + 1] - t0) * 0.01; a[1] += (h[(size_t)b * 8 + 2] - t0) * 0.01; a[2] += (h[(size_t)b * 8 + 4] - t0) * 0.01;
                            e_max = std::max(e_max, (h[(size_t)b * 8 + 4] - t0) * 0.01); ++cnt;
                        }
                        fprintf(stderr, "  pair %2d: first loads %.2f  k loop %.2f  end %.2f (max %.2f)\n", pp_, a[0] / cnt, a[1] / cnt, a[2] / cnt, e_max);
                    }
                }
            }
#endif
            ro.Opk = m.x_Opk; ro.Typ = m.x_Typ; ro.pp = pp_src; ro.y = lean ? (double*)nullptr : m.x_y; ro.P = m.x_P;      // (lean: the ordinates stay in LDS)
            const size_t lds_max = (size_t)158 * 1024;
            const bool pairwise = m.x_pairwise;
            {
                JxRowOp r1 = ro;
                if (!pairwise) r1.ng = 0;                        // (the ordinates alone: the row product reads them back)
                const int ntw = (n + 15) / 16;
                const dim3 grid((unsigned)(8 * ((ntw + 7) / 8) * r1.npair));
                const size_t sh1 = sizeof(double) * JX_ORD_LDS_DOUBLES;
                if (tm2 && es.p1stage == 2) HIPCHK(ctx, hipEventRecord(es.e[2], st));
                bool done = false;
#define JX_ORD_GO(Xv) if (!done && m.x_nxt == Xv) { hipLaunchKernelGGL((jx_ordrow_kernel<Xv>), grid, dim3(256), sh1, st, r1); done = true; }
                JX_MIX_NXTS(JX_ORD_GO)
#undef JX_ORD_GO
                if (!done) { ctx->err = "no ordinate-product kernel for this output tiling"; return JX_ERR_UNSUPPORTED; }
                if (tm2 && es.p1stage == 2) HIPCHK(ctx, hipEventRecord(es.e[3], st));
            }
            if (want_abel_taps) {
                if (ctx->map_ok) {
                    // the profile taps come from the Abel kernel's own phases 1-3 (an independent evaluation of the same three lines)
                    if (!m.x_cf && (rc = dev_new_l(ctx, m.allocs, (size_t)ctx->chunk * 2 * N, &m.x_cf, true))) return rc;
                    JxDev dm = ctx->d;
                    dm.cf_out = m.x_cf; dm.cf_ws = 2LL * N; dm.cf_tr = 0;
                    launch_map(st, dm, ctx->map_threads, ctx->map_lds, theta_dev, w0, n, nullptr, t.pp, t.ab, t.y, true);
                } else {
                    if (t.pp) HIPCHK(ctx, hipMemcpy2DAsync(t.pp, sizeof(double) * N, pp_src, sizeof(double) * m.x_ldpp, sizeof(double) * N, n, hipMemcpyDeviceToDevice, st));
                    if (t.y || t.ab) hipLaunchKernelGGL(jx_unpack_ordinates_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)n), dim3(256), 0, st,
                                                        m.x_y, m.Nkp, N, d.y_scale, t.y, t.ab);
                }
            }
            if (t.need_img) {
                if (!ctx->d_img && (rc = dev_new(ctx, (size_t)ctx->chunk * d.img_ws, &ctx->d_img, true))) return rc;
                launch_map(st, ctx->d, ctx->map_threads, ctx->map_lds, theta_dev, w0, n, ctx->d_img, nullptr, nullptr, nullptr, false);
            }
            if (tm) HIPCHK(ctx, hipEventRecord(es.e[2], st));
            {
                JxDev dtl = d;
                dtl.xr_split = xr_split ? 1 : 0;
                dtl.xr_out = ctx->d_xr;
                const size_t extra = pairwise ? 0 : sizeof(double) * (size_t)JX_ROP_NW * 16 * JX_ROP_LDP(m.x_nxt);
                if (extra + sizeof(double) * JX_RST_LDS_DOUBLES(ro.ldr, d.nflux, ro.lde) > lds_max) ro.lde = 0;   // (the data-radii matrix from memory then)
                const size_t shr = extra + sizeof(double) * JX_RST_LDS_DOUBLES(ro.ldr, d.nflux, ro.lde);
                if (shr > lds_max) { ctx->err = "exact form: the outputs the data-radii spline reads do not fit the tail's LDS (JOXSZ_MIX_FORM=legacy runs them)"; return JX_ERR_UNSUPPORTED; }
                if (tm2 && es.p1stage == 4) HIPCHK(ctx, hipEventRecord(es.e[2], st));
                bool done = false;
                const dim3 gridt((unsigned)((n + 15) / 16));
#define JX_ROP_GO(Xv) if (!done && m.x_nxt == Xv) { \
                    if (pairwise) hipLaunchKernelGGL((jx_rowsum_tail_kernel<Xv>), gridt, dim3(JX_RST_THREADS), shr, st, dtl, ro, ctx->d_cfac, ctx->d_sz0, ctx->d_base, logp_dev, w0, t.row, t.bright, t.chisq, t.parts, smv); \
                    else hipLaunchKernelGGL((jx_rowop_tail_kernel<Xv>), gridt, dim3(64 * JX_ROP_NW), shr, st, dtl, ro, ctx->d_cfac, ctx->d_sz0, ctx->d_base, logp_dev, w0, t.row, t.bright, t.chisq, t.parts, smv); \
                    done = true; }
                JX_MIX_NXTS(JX_ROP_GO)
#undef JX_ROP_GO
                if (!done) { ctx->err = "no row-product kernel for this output tiling"; return JX_ERR_UNSUPPORTED; }
                if (tm2 && es.p1stage == 4) HIPCHK(ctx, hipEventRecord(es.e[3], st));
            }
            if (tm) HIPCHK(ctx, hipEventRecord(es.e[5], st));
            if (tm || tm2) ctx->ev_inflight.push_back(es);
            HIPCHK(ctx, hipGetLastError());
            return JX_OK;
        }
        if (ag) {
            // 32 walkers per block, or 16 when that would leave SIMDs without a wave (a walker's sums do not depend on it)
            const int gy = (ctx->tm_npair + 3) / 4;
            const bool narrow = ctx->ag_narrow >= 0 ? ctx->ag_narrow != 0 : (size_t)((n + 31) / 32) * gy * 4 < 1024;
            const dim3 grid(narrow ? (n + 15) / 16 : (n + 31) / 32, gy);
            const double* pp_src = ctx->d.inject_pp ? ctx->d.inject_pp : ctx->d_ppc;
            const size_t shg = sizeof(double) * JX_OPM_JC * 33;
            const bool rsg = ctx->ag_sub_on;                        // the radial sub-grid: K = the kept radii, gathered from the profiles
            const int agN = rsg ? ctx->ag_ns : d.N;
            const double* agT = rsg ? ctx->d_Tm_s : ctx->d_Tm;
            const int* agR = rsg ? ctx->d_rsub : nullptr;
            const int* agK = rsg ? ctx->d_tks : nullptr;
#define JX_AG_GO(TOv, NWTv) do { if (rsg) hipLaunchKernelGGL((jx_abel_gemm_kernel<1, TOv, 1, NWTv, 0, true>), grid, dim3(256), shg, st, pp_src, n, agN, agT, ctx->tm_ld, \
                                               d.K, ctx->tm_ntile, ctx->tm_npair, reinterpret_cast<TOv*>(m.cft), m.tW, (long long)m.ncol, agR, d.N, agK); \
                                  else hipLaunchKernelGGL((jx_abel_gemm_kernel<1, TOv, 1, NWTv, 0, false>), grid, dim3(256), shg, st, pp_src, n, agN, agT, ctx->tm_ld, \
                                               d.K, ctx->tm_ntile, ctx->tm_npair, reinterpret_cast<TOv*>(m.cft), m.tW, (long long)m.ncol, agR, d.N, agK); } while (0)
#define JX_AG_GO1(TOv) hipLaunchKernelGGL((jx_abel_gemm_kernel<1, TOv, 1, 2, 1>), dim3(grid.x, (unsigned)((ctx->tm_ntile + 3) / 4)), dim3(256), shg, st, pp_src, n, agN, agT, ctx->tm_ld, \
                                               d.K, ctx->tm_ntile, ctx->tm_npair, reinterpret_cast<TOv*>(m.cft), m.tW, (long long)m.ncol, agR, d.N, agK)
            if (ctx->ag_single && !narrow && !rsg) { if (ctx->f32) JX_AG_GO1(float); else JX_AG_GO1(double); }
            else if (ctx->f32) { if (narrow) JX_AG_GO(float, 1); else JX_AG_GO(float, 2); }
            else { if (narrow) JX_AG_GO(double, 1); else JX_AG_GO(double, 2); }
#undef JX_AG_GO
#undef JX_AG_GO1
            if (!ctx->map_ok && want_abel_taps && !ctx->f32) {
                // the profile taps without the Abel kernel: the pressure profile as the per-walker kernel wrote it, the spline
                // ordinates y_k of the matrix product, the Abel integral as y / y_scale
                if (t.pp) HIPCHK(ctx, hipMemcpyAsync(t.pp, pp_src, sizeof(double) * (size_t)n * d.N, hipMemcpyDeviceToDevice, st));
                if (t.y || t.ab) hipLaunchKernelGGL(jx_unpack_splines_kernel, dim3((unsigned)((d.N + 255) / 256), (unsigned)n), dim3(256), 0, st,
                                                    reinterpret_cast<const double2*>(m.cft), m.tW, d.N, d.y_scale, t.y, t.ab);
            }
        } else {
            // phases 1-3 of the Abel kernel (profile, Abel integral, Compton y, spline moments): taps out, arrays walker-minor
            JxDev dm = ctx->d;
            dm.cf_out = m.cft; dm.cf_ws = m.tW; dm.cf_tr = 1;
            launch_map(st, dm, ctx->map_threads, ctx->map_lds, theta_dev, w0, n, nullptr, t.pp, t.ab, t.y, true);
        }
        if (t.need_img) {                                          // the Compton-y map tap: the quadrant of distinct pixels
            if (!ctx->d_img && (rc = dev_new(ctx, (size_t)ctx->chunk * d.img_ws, &ctx->d_img, true))) return rc;
            launch_map(st, ctx->d, ctx->map_threads, ctx->map_lds, theta_dev, w0, n, ctx->d_img, nullptr, nullptr, nullptr, false);
        }
        if (tm || tm2) HIPCHK(ctx, hipEventRecord(es.e[2], st));
        auto between = [&]() -> int {
            if (!side || ctx->side_fork != 1) return JX_OK;
            HIPCHK(ctx, hipEventRecord(ctx->ev_tail, st));                  // ("fork": behind stage 1)
            HIPCHK(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_tail, 0));
            launch_prep(ctx->side_stream, nullptr);
            HIPCHK(ctx, hipEventRecord(ctx->ev_side, ctx->side_stream));
            return JX_OK;
        };
        // the matrix-core product computes the outputs the data-radii spline of the tail reads (nrow_use of nrow); the whole row when
        // the row or the brightness profile is tapped
        const bool used = m.has_u && !t.row && !t.bright;
        if ((rc = launch_mix(ctx, n, (tm || tm2) ? &es : nullptr, used, between))) return rc;
        const int nks = used ? m.ksplit_u : m.last_ksplit, nuse = used ? std::min(ctx->nrow_use, d.nrow) : d.nrow;
        const JxOpg& ogt = used ? m.og_u : m.og;
        const size_t sh = sizeof(double) * (JX_LDS_HDR + (size_t)d.nrow + 8);
        if (side) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_side, 0));
        JxDev dtl = d;
        dtl.xr_split = xr_split ? 1 : 0;
        dtl.xr_out = ctx->d_xr;
        if (ctx->f32c) hipLaunchKernelGGL(jx_tail_row_kernel<float>, dim3(n), dim3(JX_TAIL_THREADS), sh, st, dtl, reinterpret_cast<const float*>(m.Pt), nks,
                                          ogt.pstride, ogt.ldx, nuse, ctx->d_cfac, ctx->d_sz0, ctx->d_base, logp_dev, w0, t.row, t.bright, t.chisq, t.parts, smv);
        else hipLaunchKernelGGL(jx_tail_row_kernel<double>, dim3(n), dim3(JX_TAIL_THREADS), sh, st, dtl, m.Pt, nks, ogt.pstride, ogt.ldx, nuse,
                                ctx->d_cfac, ctx->d_sz0, ctx->d_base, logp_dev, w0, t.row, t.bright, t.chisq, t.parts, smv);

        if (tm) HIPCHK(ctx, hipEventRecord(es.e[5], st));
        if (tm || tm2) ctx->ev_inflight.push_back(es);
        HIPCHK(ctx, hipGetLastError());
        return JX_OK;
    }
    // ---- rocFFT sequence
    FftBack& fb = ctx->fft;
    if (n > fb.cap) { ctx->err = "rocFFT sequence: launch beyond its work buffers"; return JX_ERR_INVALID; }
    Plan3* pl = nullptr;
    if (!fb.rows_custom && (rc = fft_plans(ctx, fb, n, &pl))) return rc;          // (no rocFFT plan where rows and columns are hand-written)
    {
        JxDev dm = fb.d;
        dm.inject_pp = ctx->d.inject_pp;
        launch_map(st, dm, fb.map_threads, fb.map_lds, theta_dev, w0, n, fb.img, t.pp, t.ab, t.y, false);
    }
    if (tm) HIPCHK(ctx, hipEventRecord(es.e[2], st));
    const double* zin = nullptr;
    if (fb.cols) {
        // rows by rocFFT, columns by jx_fft.hpp: forward, times the beam spectrum, inverse in one pass over the row spectra; then the
        // window's row spectra and its column pass, which leaves the spectrum of the extracted row
        const int S = d.S;
        if (fb.rows_custom) {
            fft_launch_rows_fwd(fb, n, st, S);
            fft_launch_beam(fb, n, st, S);
            if (tm) HIPCHK(ctx, hipEventRecord(es.e[3], st));       // (the inverse row pass of the convolution is inside the next kernel)
            fft_launch_rows_inv_tf(fb, n, st, S, t.need_conv);
        } else {
            void* a[1] = {fb.img};   void* b[1] = {fb.spec};
            FFTCHK(ctx, rocfft_execute(pl->row_fwd, a, b, fb.info));
            fft_launch_beam(fb, n, st, S);
            void* c2[1] = {fb.conv};
            FFTCHK(ctx, rocfft_execute(pl->row_inv, b, c2, fb.info));
            if (tm) HIPCHK(ctx, hipEventRecord(es.e[3], st));
            void* t2[1] = {fb.tfspec};
            FFTCHK(ctx, rocfft_execute(pl->row_tf, c2, t2, fb.info));
        }
        fft_launch_tf(fb, n, st, d.Sh);
        if (tm) HIPCHK(ctx, hipEventRecord(es.e[4], st));
        zin = fb.zbuf;
    } else {
    {
        void* in[1] = {fb.img};
        void* out[1] = {fb.spec};
        FFTCHK(ctx, rocfft_execute(pl->beam_fwd, in, out, fb.info));
        const size_t per = (size_t)fb.P * fb.Ph, total = per * n;
        const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
        hipLaunchKernelGGL(jx_beam_mul_kernel, dim3(blocks), dim3(256), 0, st, fb.spec, (const double2*)fb.d.bhat, per, total);
        void* in2[1] = {fb.spec};
        void* out2[1] = {fb.conv};
        FFTCHK(ctx, rocfft_execute(pl->beam_inv, in2, out2, fb.info));
    }
    if (tm) HIPCHK(ctx, hipEventRecord(es.e[3], st));
    {
        void* in[1] = {fb.conv};
        void* out[1] = {fb.tfspec};
        FFTCHK(ctx, rocfft_execute(pl->tf_fwd, in, out, fb.info));
    }
    if (tm) HIPCHK(ctx, hipEventRecord(es.e[4], st));
    }
    {
        const size_t sh = sizeof(double) * (JX_LDS_HDR + (size_t)2 * d.Sh + d.nrow + 8 + (size_t)2 * d.S);      // (+ the roots of unity of the row)
        hipLaunchKernelGGL(jx_tail_kernel, dim3(n), dim3(JX_TAIL_THREADS), sh, st, fb.d, fb.tfspec, zin, ctx->d_cfac, ctx->d_sz0,
                           ctx->d_base, logp_dev, w0, t.row, t.bright, t.chisq, t.parts);
    }
    if (tm) { HIPCHK(ctx, hipEventRecord(es.e[5], st)); ctx->ev_inflight.push_back(es); }
    HIPCHK(ctx, hipGetLastError());
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

static Taps all_taps(jx_ctx* ctx, bool profiles) {
    Taps t;
    if (profiles) { t.pp = ctx->t_pp; t.ab = ctx->t_ab; t.y = ctx->t_y; }
    t.row = ctx->t_row; t.bright = ctx->t_bright; t.chisq = ctx->t_chisq; t.tprof = ctx->t_tprof;
    t.xprofs = ctx->cfg.sz_only ? nullptr : ctx->t_xprofs; t.parts = ctx->t_parts; t.integ = ctx->t_integ;
    return t;
}

// ---------------------------------------------------------------------------------------------------------------------
// Truncation guard of the low-rank form.  The extracted row (joxsz_funcs.py:472) through the contracted route against the
// rocFFT sequence (exact, independent) on parameter vectors spread over the prior box: the current values and the corners
// of the box in the pressure-profile shape parameters (a, b, r_p) that are thawed.  est = largest difference relative to
// the row's largest entry.
// ---------------------------------------------------------------------------------------------------------------------
static void probe_vectors(jx_ctx* ctx, std::vector<double>& th, int* npts) {
    const jx_config& c = ctx->cfg;
    const std::vector<double> pv = host_vec<double>(ctx, JX_T_PAR_VALS), pmin = host_vec<double>(ctx, JX_T_PAR_MIN), pmax = host_vec<double>(ctx, JX_T_PAR_MAX);
    const std::vector<int32_t> ti = host_vec<int32_t>(ctx, JX_T_THAWED_IDX);
    std::vector<double> base(c.ndim);
    for (int k = 0; k < c.ndim; ++k) base[k] = pv[ti[k]];
    const int shape[3] = {P_A, P_B, P_RP};
    int slot[3] = {-1, -1, -1}, nv = 0;
    for (int s = 0; s < 3; ++s)
        for (int k = 0; k < c.ndim; ++k)
            if (ti[k] == shape[s] && std::isfinite(pmin[shape[s]]) && std::isfinite(pmax[shape[s]]) && pmax[shape[s]] > pmin[shape[s]]) slot[nv++] = k;
    th.assign(base.begin(), base.end());
    int n = 1;
    for (int corner = 0; nv > 0 && corner < (1 << nv); ++corner) {
        std::vector<double> v = base;
        for (int s = 0; s < nv; ++s) {
            const int par = ti[slot[s]];
            // (just inside the box: on the bound itself a profile can degenerate, and the sampler never sits there)
            const double lo = pmin[par] + 0.02 * (pmax[par] - pmin[par]), hi = pmax[par] - 0.02 * (pmax[par] - pmin[par]);
            v[slot[s]] = ((corner >> s) & 1) ? hi : lo;
        }
        th.insert(th.end(), v.begin(), v.end());
        ++n;
    }
    *npts = n;
}

// est[0] = largest row difference relative to the row's largest entry at the current parameter values, est[1] = the same
// over all probe points, est[2] = largest |difference of the SZ log-likelihood| relative to max(1, |SZ log-likelihood|) over
// all probe points (-1 each where nothing finite came back)
static int measure_truncation(jx_ctx* ctx, double est[3], int* used) {
    est[0] = est[1] = est[2] = -1.0; *used = 0;
    if (ctx->conv_mode != 2 || ctx->mix.form == 2 || (ctx->mix.form != 0 && ctx->mix.sub.empty() && !ctx->ag_sub_on)) return JX_OK;   // nothing truncated, every distinct sample and radius evaluated
    if (!ctx->map_ok) return JX_OK;                                         // nothing to measure against (and every term above rounding is kept)
    const jx_config& c = ctx->cfg;
    int rc, npts = 0;
    std::vector<double> th;
    probe_vectors(ctx, th, &npts);
    if ((rc = ensure_taps(ctx))) return rc;
    if ((rc = ensure_ref(ctx))) return rc;
    npts = std::min(npts, ctx->fft.cap);
    if ((rc = ensure_batch(ctx, npts))) return rc;
    const int nrow = ctx->nrow;
    std::vector<double> ra((size_t)npts * nrow), rb((size_t)npts * nrow), la((size_t)npts * 4), lb((size_t)npts * 4);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_theta, th.data(), sizeof(double) * (size_t)npts * c.ndim, hipMemcpyHostToDevice, ctx->stream));
    const bool tm = ctx->timing_on;
    ctx->timing_on = false;
    Taps t = all_taps(ctx, false);
    auto fetch = [&](std::vector<double>& rows, std::vector<double>& parts) {
        if (hipMemcpyAsync(rows.data(), ctx->t_row, sizeof(double) * rows.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return (int)JX_ERR_HIP;
        if (hipMemcpyAsync(parts.data(), ctx->t_parts, sizeof(double) * parts.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return (int)JX_ERR_HIP;
        return (hipStreamSynchronize(ctx->stream) == hipSuccess) ? (int)JX_OK : (int)JX_ERR_HIP;
    };
    rc = run_chunk(ctx, ctx->d_theta, ctx->d_logp, 0, npts, t);
    if (!rc) rc = fetch(ra, la);
    if (!rc) rc = run_chunk(ctx, ctx->d_theta, ctx->d_logp, 0, npts, t, true);
    if (!rc) rc = fetch(rb, lb);
    ctx->timing_on = tm;
    if (rc) return rc;
    for (int p = 0; p < npts; ++p) {
        double mx = 0.0, df = 0.0;
        bool fin = true;
        for (int k = 0; k < nrow; ++k) {
            const double a = ra[(size_t)p * nrow + k], b = rb[(size_t)p * nrow + k];
            if (!std::isfinite(a) || !std::isfinite(b)) { fin = false; break; }
            mx = std::max(mx, std::fabs(b)); df = std::max(df, std::fabs(a - b));
        }
        const double sa = la[(size_t)p * 4 + 1], sb = lb[(size_t)p * 4 + 1];   // SZ log-likelihood (-chi^2/2 [+ integrated-Compton term])
        if (!fin || !(mx > 0.0) || !std::isfinite(sa) || !std::isfinite(sb)) continue;
        if (p == 0) est[0] = df / mx;
        est[1] = std::max(est[1], df / mx);
        est[2] = std::max(est[2], std::fabs(sa - sb) / std::max(1.0, std::fabs(sb)));
        *used += 1;
    }
    return JX_OK;
}

extern "C" {

int jx_finalize(jx_ctx* ctx) {
    int rc = finalize_impl(ctx);
    if (rc) return rc;
    if (const char* e = opt_str(ctx, "JOXSZ_TRUNC_PROBE")) { if (atoi(e) == 0) return JX_OK; }
    if (ctx->conv_mode != 2 || ctx->mix.form == 2 || (ctx->mix.form != 0 && ctx->mix.sub.empty() && !ctx->ag_sub_on)) return JX_OK;
    // The low-rank form drops the small singular values of the transfer-function weights.  What that costs is measured on the
    // caller's own beam / transfer function / prior box (measure_truncation); beyond the bounds the tables are rebuilt with
    // a cut ten times tighter -- in place: stream, communicator and every other piece of the context stay -- until the
    // bounds hold or every term above rounding is kept.
    if ((rc = measure_truncation(ctx, ctx->trunc_est, &ctx->trunc_points))) return rc;
    auto too_large = [&]() {
        // the row at the current parameter values (where the chain lives) within 1e-9 of its largest entry; the SZ
        // log-likelihood at every probe point -- corners of the prior box included -- within 1e-8 relative, a hundred times
        // inside the 1e-6 the log-posterior is held to.  (An estimate that could not be taken counts as too large.)
        return !(ctx->trunc_est[0] >= 0.0 && ctx->trunc_est[0] <= ctx->trunc_bound && ctx->trunc_est[2] >= 0.0 && ctx->trunc_est[2] <= ctx->trunc_bound_ll);
    };
    auto rebuild = [&](double tol) -> int {
        MixBuild mb;
        plan_mix(ctx, host_vec<double>(ctx, JX_T_BEAM_2D), host_vec<double>(ctx, JX_T_FILTERING), host_vec<double>(ctx, JX_T_R_PP), tol, ctx->form_force, ctx->mix.tW, mb);
        if (!mb.ok) { ctx->err = "contracted route, tables rebuilt by the guard: " + mb.why; return JX_ERR_UNSUPPORTED; }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        const long long tW = ctx->mix.tW;
        mix_teardown(ctx, ctx->mix);
        int rc2 = mix_setup(ctx, mb, tW);
        if (rc2) return rc2;
        HIPCHK(ctx, hipDeviceSynchronize());
        return measure_truncation(ctx, ctx->trunc_est, &ctx->trunc_points);
    };
    // (an f32 context is measured and reported but never rebuilt: the rounding of its spline arrays is of the bounds' size)
    const bool sub_at_start = !ctx->mix.sub.empty(), rad_at_start = ctx->ag_sub_on;
    double tol_now = ctx->mix.tol;
    if (!ctx->f32 && ctx->ag_sub_on && too_large()) {
        // the radial sub-grid of the spline-array product goes first: nothing to rebuild, the full operator is resident
        ctx->ag_sub_on = false; ctx->ag_removed += 1;
        if ((rc = measure_truncation(ctx, ctx->trunc_est, &ctx->trunc_points))) return rc;
    }
    while (!ctx->f32 && (!ctx->mix.sub.empty() || (!ctx->tol_pinned && ctx->mix.form == 0 && (ctx->mix.tol > 2e-13 || ctx->mix.r < ctx->mix.r_tol))) && too_large()) {
        // first the sub-grid of stage 1 goes (every distinct sample evaluated), then the cap on the rank (the same cut, every
        // term above it kept), then the cut tightens
        const bool subbed = !ctx->mix.sub.empty();
        const bool capped = !subbed && ctx->mix.r < ctx->mix.r_tol;
        if (subbed) { ctx->subsample = false; ctx->trunc_unsub += 1; }
        else ctx->trunc_retried += 1;
        if (capped) { ctx->rank_cap = 0; ctx->trunc_uncapped += 1; }
        tol_now = (subbed || capped) ? ctx->mix.tol : std::max(1e-13, ctx->mix.tol * 1e-1);
        if ((rc = rebuild(tol_now))) return rc;
    }
    if (sub_at_start && ctx->trunc_unsub > 0 && ctx->trunc_retried == 0 && too_large()) {
        // every distinct sample evaluated and still outside (a cut set by hand, which is measured but never tightened): the
        // sub-grid was not what the bounds object to
        ctx->subsample = true;
        if ((rc = rebuild(tol_now))) return rc;
        ctx->trunc_unsub = 0;
    } else if (sub_at_start && ctx->trunc_unsub > 0 && ctx->trunc_retried > 0) {
        // the truncation was (also) at fault: the sub-grid once more on the tables the loop ended at, kept when the bounds hold
        ctx->subsample = true;
        if ((rc = rebuild(tol_now))) return rc;
        if (!ctx->mix.sub.empty() && !too_large()) ctx->trunc_unsub = 0;             // (either form: the full form's figures are those of the sub-grid alone)
        else { ctx->subsample = false; if ((rc = rebuild(tol_now))) return rc; }
    }
    if (rad_at_start && !ctx->ag_sub_on && (ctx->trunc_unsub > 0 || ctx->trunc_retried > 0 || too_large())) {
        // something else was (also) at fault: the radial sub-grid once more on what the guard ended at, kept when it changes nothing for the worse
        const double e0 = ctx->trunc_est[0], e2 = ctx->trunc_est[2];
        const bool was_ok = !too_large();
        ctx->ag_sub_on = true;
        if ((rc = measure_truncation(ctx, ctx->trunc_est, &ctx->trunc_points))) return rc;
        const bool keep = was_ok ? !too_large() : (ctx->trunc_est[0] <= 1.5 * e0 + 1e-13 && ctx->trunc_est[2] <= 1.5 * e2 + 1e-13);
        if (keep) ctx->ag_removed = 0;
        else { ctx->ag_sub_on = false; if ((rc = measure_truncation(ctx, ctx->trunc_est, &ctx->trunc_points))) return rc; }
    }
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
#define SCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_); return JX_ERR_HIP; } } while (0)
    // a run re-uses the buffers of the last one when they are large enough: repeated runs (burn-in, then sampling) do not
    // pay for device allocations of tens of megabytes each time (the buffers stay with the context, freed by jx_destroy)
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
    if (rc) return rc;
    {
        std::vector<double> l0(W);
        SCHK(hipMemcpyAsync(l0.data(), lp, sizeof(double) * W, hipMemcpyDeviceToHost, st));
        SCHK(hipStreamSynchronize(st));
        for (double v : l0) if (!std::isfinite(v)) { ctx->err = "initial positions must have finite log-posterior"; return JX_ERR_INVALID; }
    }
    const dim3 grid((half + 255) / 256), block(256);
    const bool fused = ctx->conv_mode == 2 && ctx->route != JX_ROUTE_OPERATOR && !ctx->side_stream && !(opt_str(ctx, "JOXSZ_SAMPLE_FUSED") && atoi(opt_str(ctx, "JOXSZ_SAMPLE_FUSED")) == 0);
    // shares of a half step: one per rank of the communicator, or JOXSZ_SAMPLE_VIRTUAL_RANKS of them run in turn by this process
    const bool dist = ctx->comm != nullptr;                                  // (a communicator of one rank runs the same collectives: in place, on its own share = everything)
    int nshare = dist ? ctx->comm_size : 1;
    if (!dist) if (const char* e = opt_str(ctx, "JOXSZ_SAMPLE_VIRTUAL_RANKS")) nshare = std::max(1, atoi(e));
    if (nshare > 1 && (!fused || half % nshare)) { ctx->err = "jx_sample over " + std::to_string(nshare) + " ranks: the contracted route and a half ensemble divisible by the ranks"; return JX_ERR_INVALID; }
    const int r_first = dist ? ctx->comm_rank : 0, r_last = dist ? ctx->comm_rank + 1 : nshare;
    for (int it = 0; it < nsteps; ++it) {
        for (int hs = 0; hs < 2; ++hs) {
            const int s1 = hs * half, s2 = (1 - hs) * half;
            if (fused) {
                // contracted route: the proposal is drawn by the per-walker kernel, the tail accepts or rejects -- five launches per half step, not seven
                JxSm sm;
                memset(&sm, 0, sizeof(sm));
                sm.on = 1; sm.ndim = ndim; sm.half = half; sm.s1 = s1; sm.s2 = s2; sm.iter2 = 2 * it + hs; sm.a = a; sm.seed = seed;
                sm.x = x; sm.q = q; sm.zz = zz; sm.lp = lp; sm.nacc = reinterpret_cast<long long*>(nacc);
                Taps none;
                // Walkers shard over the ranks of the context's communicator (jx_comm_init_rank): the proposals of a half step read the OTHER half
                // and the walker's own position only, so a rank moves its contiguous share of the half alone, and one in-place all-gather of the
                // share's positions and log-posteriors brings every rank's copy of the ensemble up to date before the next half step -- the RCCL
                // exchange of a real sampling run.  (JOXSZ_SAMPLE_VIRTUAL_RANKS=R, no communicator: the R shares one after the other in this
                // process -- the same chain by construction; the test of the share arithmetic where there is one GPU.)
                for (int r = r_first; r < r_last; ++r) {
                    const int lo = (int)((long long)r * half / nshare), hi = (int)((long long)(r + 1) * half / nshare);
                    for (int w0 = lo; w0 < hi; w0 += ctx->chunk)
                        if ((rc = run_chunk(ctx, q, lq, w0, std::min(ctx->chunk, hi - w0), none, false, &sm))) return rc;
                }
                if (dist) {
                    const size_t share = (size_t)half / nshare;
                    NCCLCHK(ctx, g_rccl.AllGather(x + ((size_t)s1 + ctx->comm_rank * share) * ndim, x + (size_t)s1 * ndim, share * ndim, ncclDouble, ctx->comm, st));
                    NCCLCHK(ctx, g_rccl.AllGather(lp + (size_t)s1 + ctx->comm_rank * share, lp + (size_t)s1, share, ncclDouble, ctx->comm, st));
                }
                continue;
            }
            hipLaunchKernelGGL(jx_sm_propose_kernel, grid, block, 0, st, x, q, zz, ndim, half, s1, s2, 2 * it + hs, a, seed);
            if ((rc = jx_eval_device(ctx, q, half, lq))) return rc;
            hipLaunchKernelGGL(jx_sm_accept_kernel, grid, block, 0, st, x, lp, q, lq, zz, nacc, ndim, half, s1, 2 * it + hs, seed);
        }
        if (chain) SCHK(hipMemcpyAsync(chain + (size_t)it * W * ndim, x, sizeof(double) * (size_t)W * ndim, hipMemcpyDeviceToDevice, st));
        if (lps) SCHK(hipMemcpyAsync(lps + (size_t)it * W, lp, sizeof(double) * (size_t)W, hipMemcpyDeviceToDevice, st));
    }
    if (chain) SCHK(hipMemcpyAsync(chain_out, chain, sizeof(double) * (size_t)nsteps * W * ndim, hipMemcpyDeviceToHost, st));
    if (lps) SCHK(hipMemcpyAsync(logp_out, lps, sizeof(double) * (size_t)nsteps * W, hipMemcpyDeviceToHost, st));
    if (dist) NCCLCHK(ctx, g_rccl.AllReduce(nacc, nacc, (size_t)W, ncclInt64, ncclSum, ctx->comm, st));      // (a rank counted its own shares only)
    if (naccept_out) SCHK(hipMemcpyAsync(naccept_out, nacc, sizeof(long long) * (size_t)W, hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    SCHK(hipGetLastError());
#undef SCHK
    return JX_OK;
}

}  // extern "C"

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
    // (fp32 contexts take their spline arrays from the matrix product of the injected profiles: no profile taps there)
    Taps t = all_taps(ctx, !ctx->f32 && ctx->map_ok);
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

extern "C" {

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
    if (ctx->fft.info) FFTCHK(ctx, rocfft_execution_info_set_stream(ctx->fft.info, ctx->stream));
    return JX_OK;
}

int jx_sync(jx_ctx* ctx) {
    if (!ctx) return JX_ERR_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->comm_stream));
    return JX_OK;
}

int jx_eval(jx_ctx* ctx, const double* theta, int nwalkers, double* logp) {
    if (!ctx || nwalkers < 0 || (nwalkers && (!theta || !logp))) return JX_ERR_INVALID;
    if (!ctx->finalized) { ctx->err = "jx_eval before jx_finalize"; return JX_ERR_STATE; }
    if (nwalkers == 0) return JX_OK;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    int rc = ensure_batch(ctx, nwalkers);
    if (rc) return rc;
    // through pinned staging: the copies are then plain DMA enqueued on the stream, and the call waits once, at its end
    // The staging buffers are mapped into the device's address space: the per-walker kernel reads its 13 parameters straight
    // from the host buffer and the tail stores the log-probability there (posted writes over PCIe, visible at the
    // synchronisation) -- no copy command on either side of the five kernels.  JOXSZ_EVAL_DIRECT=0: both as DMA copies.
    memcpy(ctx->h_theta, theta, sizeof(double) * (size_t)nwalkers * ctx->cfg.ndim);
    const bool th_direct = (ctx->eval_direct & 2) != 0, lp_direct = (ctx->eval_direct & 1) != 0;
    if (!th_direct) HIPCHK(ctx, hipMemcpyAsync(ctx->d_theta, ctx->h_theta, sizeof(double) * (size_t)nwalkers * ctx->cfg.ndim, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = jx_eval_device(ctx, th_direct ? ctx->hd_theta : ctx->d_theta, nwalkers, lp_direct ? ctx->hd_logp : ctx->d_logp))) return rc;
    if (!lp_direct) HIPCHK(ctx, hipMemcpyAsync(ctx->h_logp, ctx->d_logp, sizeof(double) * (size_t)nwalkers, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(logp, ctx->h_logp, sizeof(double) * (size_t)nwalkers);
    return JX_OK;
}

int jx_eval_stage(jx_ctx* ctx, const double* theta, int nwalkers, int stage, double* out, size_t nbytes) {
    if (!ctx || !theta || !out || nwalkers <= 0) return JX_ERR_INVALID;
    if (!ctx->finalized) { ctx->err = "jx_eval_stage before jx_finalize"; return JX_ERR_STATE; }
    if (stage < 0 || stage >= JX_STAGE_COUNT) return JX_ERR_INVALID;
    const jx_config& c = ctx->cfg;
    const size_t S = c.S;
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
    // the beam-convolved map exists in the rocFFT sequence only: contracted-route contexts send this tap through their
    // reference facility (16 walkers at a time)
    const bool via_ref = stage == JX_STAGE_CONV2D && ctx->conv_mode == 2;
    if (via_ref && (rc = ensure_ref(ctx))) return rc;
    const bool y2d_quad = (stage == JX_STAGE_Y2D && ctx->conv_mode == 2);
    if (y2d_quad && !ctx->t_y2d && (rc = dev_new(ctx, (size_t)ctx->chunk * S * S, &ctx->t_y2d))) return rc;
    if ((rc = ensure_batch(ctx, nwalkers))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_theta, theta, sizeof(double) * (size_t)nwalkers * c.ndim, hipMemcpyHostToDevice, ctx->stream));
    const bool profiles = !ctx->f32 || stage == JX_STAGE_PP || stage == JX_STAGE_AB || stage == JX_STAGE_Y;
    Taps t = all_taps(ctx, profiles);
    t.need_img = (stage == JX_STAGE_Y2D);
    t.need_conv = (stage == JX_STAGE_CONV2D);
    const int step = via_ref ? ctx->fft.cap : ctx->chunk;
    for (int w0 = 0; w0 < nwalkers; w0 += step) {
        const int n = std::min(step, nwalkers - w0);
        if ((rc = run_chunk(ctx, ctx->d_theta, ctx->d_logp, w0, n, t, via_ref))) return rc;
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
            const size_t P = ctx->fft.P;
            if (y2d_quad)
                hipLaunchKernelGGL(jx_expand_quad_kernel, dim3((unsigned)S, n), dim3(256), 0, ctx->stream, ctx->d_img, (size_t)ctx->d.img_ld,
                                   (size_t)ctx->d.img_ws, (int)S, ctx->t_y2d);
            const double* base = (stage == JX_STAGE_Y2D) ? (y2d_quad ? ctx->t_y2d : ctx->fft.img) : ctx->fft.conv;
            const size_t ld = (stage == JX_STAGE_Y2D && y2d_quad) ? S : P;
            const size_t ws = (stage == JX_STAGE_Y2D && y2d_quad) ? S * S : P * (size_t)ctx->fft.rows;
            for (int w = 0; w < n; ++w)
                HIPCHK(ctx, hipMemcpy2DAsync(dst + (size_t)w * S * S, S * sizeof(double), base + (size_t)w * ws,
                                             ld * sizeof(double), S * sizeof(double), S, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return JX_OK;
}

int jx_set_option(jx_ctx* ctx, const char* name, const char* value) {
    if (!ctx || !name) return JX_ERR_INVALID;
    std::string key = name;
    for (char& ch : key) ch = (char)toupper((unsigned char)ch);
    if (key.rfind("JOXSZ_", 0) != 0) key = "JOXSZ_" + key;
    bool known = false;
    for (const char* k : kOptions) if (key == k) known = true;
    if (!known) { ctx->err = "jx_set_option: unknown option " + key; return JX_ERR_INVALID; }
    const bool at_call = key == "JOXSZ_SAMPLE_FUSED" || key == "JOXSZ_SAMPLE_VIRTUAL_RANKS";
    if (ctx->finalized && !at_call) { ctx->err = "jx_set_option(" + key + ") after jx_finalize: read there and nowhere else"; return JX_ERR_STATE; }
    ctx->opts[key] = value ? value : "";                         // (an empty value: the option unset, whatever the environment says)
    return JX_OK;
}

// Run-time assurance: the given walkers through the context's own route (as jx_eval runs it) AND through the rocFFT sequence held inside
// the same context (joxsz_funcs.py:460-467 executed literally, 16 walkers at a time: an independent implementation, not the same tables).
int jx_audit(jx_ctx* ctx, const double* theta, int n, double out[4]) {
    if (!ctx || !theta || n < 1 || !out) return JX_ERR_INVALID;
    if (!ctx->finalized) { ctx->err = "jx_audit before jx_finalize"; return JX_ERR_STATE; }
    out[0] = out[1] = 0.0; out[2] = -1.0; out[3] = 0.0;
    if (ctx->conv_mode != 2) return JX_OK;                       // (the rocFFT sequence IS this context's route: nothing to compare)
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    int rc;
    if ((rc = ensure_taps(ctx))) return rc;
    if ((rc = ensure_ref(ctx))) return rc;
    const int cap = ctx->fft.cap, nrow = ctx->nrow, ndim = ctx->cfg.ndim;
    if ((rc = ensure_batch(ctx, cap))) return rc;
    std::vector<double> ra((size_t)cap * nrow), rb((size_t)cap * nrow), la((size_t)cap * 4), lb((size_t)cap * 4);
    const bool tm = ctx->timing_on;
    const int route = ctx->route;
    ctx->timing_on = false;
    ctx->route = JX_ROUTE_MAP;                                   // (the taps exist on this route; the collapsed route is built from it)
    Taps t = all_taps(ctx, false);
    auto fetch = [&](std::vector<double>& rows, std::vector<double>& parts, int m) {
        if (hipMemcpyAsync(rows.data(), ctx->t_row, sizeof(double) * (size_t)m * nrow, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return (int)JX_ERR_HIP;
        if (hipMemcpyAsync(parts.data(), ctx->t_parts, sizeof(double) * (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return (int)JX_ERR_HIP;
        return (hipStreamSynchronize(ctx->stream) == hipSuccess) ? (int)JX_OK : (int)JX_ERR_HIP;
    };
    rc = JX_OK;
    for (int w0 = 0; w0 < n && !rc; w0 += cap) {
        const int m = std::min(cap, n - w0);
        if (hipMemcpyAsync(ctx->d_theta, theta + (size_t)w0 * ndim, sizeof(double) * (size_t)m * ndim, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = JX_ERR_HIP; break; }
        rc = run_chunk(ctx, ctx->d_theta, ctx->d_logp, 0, m, t);
        if (!rc) rc = fetch(ra, la, m);
        if (!rc) rc = run_chunk(ctx, ctx->d_theta, ctx->d_logp, 0, m, t, true);
        if (!rc) rc = fetch(rb, lb, m);
        if (rc) break;
        for (int p = 0; p < m; ++p) {
            double mx = 0.0, df = 0.0;
            bool fin = true;
            for (int k = 0; k < nrow; ++k) {
                const double a = ra[(size_t)p * nrow + k], b = rb[(size_t)p * nrow + k];
                if (!std::isfinite(a) || !std::isfinite(b)) { fin = false; break; }
                mx = std::max(mx, std::fabs(b)); df = std::max(df, std::fabs(a - b));
            }
            const double sa = la[(size_t)p * 4 + 1], sb = lb[(size_t)p * 4 + 1];     // SZ log-likelihood (-chi^2/2 [+ integrated-Compton term])
            if (!fin || !(mx > 0.0) || !std::isfinite(sa) || !std::isfinite(sb)) continue;
            if (la[(size_t)p * 4 + 3] != 0.0) continue;               // (a rejected walker -- outside the prior box, vetoed -- is no place a chain lives)
            out[3] += 1.0;
            out[1] = std::max(out[1], df / mx);
            if (std::fabs(sa - sb) >= out[0]) { out[0] = std::fabs(sa - sb); out[2] = (double)(w0 + p); }
        }
    }
    ctx->timing_on = tm;
    ctx->route = route;
    return rc;
}

int jx_set_par_vals(jx_ctx* ctx, const double* v, int npar) {
    if (!ctx || !v || npar != ctx->cfg.npar) return JX_ERR_INVALID;
    if (!ctx->finalized) return jx_upload(ctx, JX_T_PAR_VALS, v, sizeof(double) * npar);
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_par_vals, v, sizeof(double) * npar, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->host[JX_T_PAR_VALS].assign((const unsigned char*)v, (const unsigned char*)v + sizeof(double) * npar);
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
    ctx->timing_mode = (on >= 2 && on <= 4) ? on : (on ? 1 : 0);
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
    if (fft_pad) *fft_pad = ctx->d.P;
    if (chunk) *chunk = ctx->chunk;
    if (band) *band = ctx->K;
    if (nrow) *nrow = ctx->nrow;
    if (bytes) *bytes = ctx->device_bytes;
    return JX_OK;
}

int jx_get_fft_info(jx_ctx* ctx, int32_t out[34]) {
    if (!ctx || !ctx->finalized || !out) return JX_ERR_STATE;
    const FftBack& fb = ctx->fft;
    for (int i = 0; i < 34; ++i) out[i] = 0;
    if (!fb.ready) return JX_OK;
    out[0] = 1; out[1] = fb.cols ? 1 : 0; out[2] = fb.rows_custom ? 1 : 0; out[3] = fb.P;
    if (!fb.cols) return JX_OK;
    out[4] = fb.ldc; out[5] = fb.ldt; out[6] = fft_cb(fb.P);
    out[7] = fb.fP.npass;
    for (int p = 0; p < fb.fP.npass && p < JX_FFT_MAXPASS; ++p) out[8 + p] = fb.fP.radix[p];
    out[20] = fb.fS.npass;
    for (int p = 0; p < fb.fS.npass && p < JX_FFT_MAXPASS; ++p) out[21 + p] = fb.fS.radix[p];
    return JX_OK;
}

int jx_get_conv_mode(jx_ctx* ctx) {
    if (!ctx || !ctx->finalized) return JX_ERR_STATE;
    return ctx->conv_mode;
}

int jx_get_conv_layout(jx_ctx* ctx, int32_t out[12]) {
    if (!ctx || !ctx->finalized || !out) return JX_ERR_STATE;
    if (ctx->conv_mode != 2) { ctx->err = "layout of the contracted route only"; return JX_ERR_UNSUPPORTED; }
    const MixBack& m = ctx->mix;
    if (m.form == 2) {
        out[0] = 2; out[1] = ctx->qn; out[2] = m.Nk; out[3] = m.Nkp; out[4] = 0; out[5] = 0; out[6] = m.x_nxt; out[7] = m.x_nxt * m.x_ng;
        out[8] = m.Nkp / 4; out[9] = (int32_t)m.tW; out[10] = 16 * m.x_nxt * m.x_ng_use; out[11] = JX_ROP_NW;
        return JX_OK;
    }
    out[0] = m.form; out[1] = ctx->qn; out[2] = m.r; out[3] = m.ns; out[4] = m.form == 0 ? m.mx.R : 0; out[5] = m.RT;
    out[6] = m.nxt; out[7] = m.og.ntile; out[8] = m.ksteps; out[9] = (int32_t)m.tW; out[10] = m.og.ldx; out[11] = m.last_ksplit;
    return JX_OK;
}

int jx_debug_workspace(jx_ctx* ctx, int which, void** dev, int32_t geom[4]) {
    if (!ctx || !ctx->finalized || !dev || !geom) return JX_ERR_STATE;
    if (ctx->conv_mode != 2) { ctx->err = "work buffers of the contracted route only"; return JX_ERR_UNSUPPORTED; }
    const MixBack& m = ctx->mix;
    geom[0] = geom[1] = geom[2] = geom[3] = 0;
    if (m.form == 2) {
        if (which == 6) { *dev = m.x_y; geom[0] = 1; geom[1] = (int)m.tW; geom[2] = m.Nkp; geom[3] = 8; return JX_OK; }
        if (which == 7) { *dev = m.x_Opk; geom[0] = m.x_ng * (m.Nkp / 16) * 4; geom[1] = 64; geom[2] = m.x_nxt; geom[3] = 8; return JX_OK; }
        if (which == 0 && ctx->d_img) { *dev = ctx->d_img; geom[0] = ctx->chunk; geom[1] = ctx->d.q_nb; geom[2] = (int)ctx->d.img_ld; geom[3] = 8; return JX_OK; }
        ctx->err = "exact form: work buffers 6 (ordinates [1][tW][Nkp]) and 7 (row operator) -- and 0 after a y_2d tap -- exist; the others belong to the contracted forms";
        return JX_ERR_UNSUPPORTED;
    }
    if (which >= 6) { ctx->err = "work buffers 6 and 7 belong to the exact form"; return JX_ERR_UNSUPPORTED; }
    switch (which) {
        case 0: if (!ctx->d_img) { ctx->err = "the map quadrant exists after the first y_2d tap"; return JX_ERR_STATE; }
                *dev = ctx->d_img; geom[0] = ctx->chunk; geom[1] = ctx->d.q_nb; geom[2] = (int)ctx->d.img_ld; geom[3] = 8; break;
        case 1: *dev = m.cft; geom[0] = ctx->cfg.N; geom[1] = (int)m.tW; geom[2] = 2; geom[3] = ctx->f32 ? 4 : 8; break;
        case 2: if (m.form != 0) { ctx->err = "no stage-1 rows in the full form"; return JX_ERR_UNSUPPORTED; }
                *dev = m.Dt; geom[0] = m.mx.NU; geom[1] = m.mx.R; geom[2] = (int)m.tW; geom[3] = ctx->f32c ? 4 : 8; break;
        case 3: *dev = m.Pt; geom[0] = m.last_was_u ? m.ksplit_u : m.last_ksplit; geom[1] = (int)m.tW; geom[2] = m.last_was_u ? m.og_u.ldx : m.og.ldx; geom[3] = ctx->f32c ? 4 : 8; break;
        case 4: if (m.form != 0) { ctx->err = "no stage-1 operator in the full form"; return JX_ERR_UNSUPPORTED; }
                *dev = const_cast<double*>(m.mx.Cm); geom[0] = 1; geom[1] = m.mx.wld; geom[2] = m.mx.cld; geom[3] = 8; break;
        case 5: *dev = const_cast<double*>(m.og.Op); geom[0] = 4 * m.ksteps; geom[1] = 16; geom[2] = m.og.ntile; geom[3] = 8; break;
        default: ctx->err = "unknown work buffer"; return JX_ERR_INVALID;
    }
    return JX_OK;
}

int jx_get_sampling(jx_ctx* ctx, int32_t out[8], int32_t* rows, int nrows_cap) {
    if (!ctx || !ctx->finalized || !out) return JX_ERR_STATE;
    const MixBack& m = ctx->mix;
    const bool on = ctx->conv_mode == 2 && !m.sub.empty();
    out[0] = on ? m.NU_full : (ctx->conv_mode == 2 ? ctx->qn : 0); out[1] = on ? (int)m.sub.size() : out[0];
    out[2] = ctx->sub_u0; out[3] = ctx->sub_u1; out[4] = ctx->sub_npts; out[5] = on ? 1 : 0; out[6] = ctx->trunc_unsub; out[7] = 0;
    if (rows) for (int i = 0; i < nrows_cap && i < out[1]; ++i) rows[i] = on ? m.sub[i] : i;
    return JX_OK;
}

int jx_get_radial_sampling(jx_ctx* ctx, int32_t out[6], int32_t* rows, int nrows_cap) {
    if (!ctx || !ctx->finalized || !out) return JX_ERR_STATE;
    const bool on = ctx->ag_sub_on;
    out[0] = ctx->cfg.N; out[1] = on ? ctx->ag_ns : ctx->cfg.N; out[2] = ctx->ag_u0; out[3] = ctx->ag_u1; out[4] = ctx->ag_npts;
    out[5] = on ? 1 : 0;
    if (rows) for (int i = 0; i < nrows_cap && i < out[1]; ++i) rows[i] = on ? ctx->h_rsub[i] : i;
    return ctx->ag_removed;
}

int jx_get_output_pruning(jx_ctx* ctx, int32_t out[6]) {
    if (!ctx || !ctx->finalized || !out) return JX_ERR_STATE;
    const MixBack& m = ctx->mix;
    if (ctx->conv_mode == 2 && m.form == 2) {
        out[0] = ctx->nrow; out[1] = ctx->prune ? ctx->nrow_use : ctx->nrow; out[2] = std::min(16 * m.x_nxt * m.x_ng_use, 16 * m.x_nxt * m.x_ng);
        out[3] = m.x_nxt; out[4] = JX_ROP_NW; out[5] = m.x_ng_use < m.x_ng ? 1 : 0;
        return JX_OK;
    }
    const bool u = ctx->conv_mode == 2 && m.has_u;
    out[0] = ctx->nrow; out[1] = ctx->nrow_use; out[2] = u ? 16 * m.og_u.ntile : ctx->nrow; out[3] = u ? m.nxt_u : 0; out[4] = u ? m.ksplit_u : 0;
    out[5] = u ? 1 : 0;
    return JX_OK;
}

int jx_get_truncation(jx_ctx* ctx, double out[12]) {
    if (!ctx || !ctx->finalized || !out) return JX_ERR_STATE;
    const bool lr = ctx->conv_mode == 2 && ctx->mix.form == 0;
    out[0] = lr ? ctx->mix.tol : 0.0; out[1] = ctx->trunc_est[0]; out[2] = lr ? (double)ctx->mix.r : 0.0; out[3] = (double)ctx->trunc_retried;
    out[4] = (double)ctx->trunc_points; out[5] = ctx->trunc_bound; out[6] = ctx->trunc_est[1]; out[7] = ctx->trunc_est[2];
    out[8] = lr ? (double)ctx->mix.r_tol : 0.0; out[9] = ctx->trunc_bound_ll; out[10] = (lr && ctx->mix.mfma) ? 1.0 : 0.0;
    out[11] = (double)ctx->trunc_uncapped;
    return JX_OK;
}

// Duration of the Abel + map kernel writing the full S x S Compton-y map of `nwalkers` walkers (the kernel BASELINE's metric
// is worded around: profile -> Abel integral -> spline -> map, S^2 * 8 B per walker), averaged over `repeats` launches
// between two HIP events on the context's stream.  Scratch image allocated for the call.
namespace {
// scratch of the measurement calls below: device buffers and events released on every path out of the call
struct Scratch {
    std::vector<void*> mem;
    std::vector<hipEvent_t> ev;
    ~Scratch() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); for (void* p : mem) (void)hipFree(p); }
    hipError_t alloc(void** p, size_t n) { const hipError_t e = hipMalloc(p, n); if (e == hipSuccess) mem.push_back(*p); return e; }
    hipError_t event(hipEvent_t* e) { const hipError_t r = hipEventCreate(e); if (r == hipSuccess) ev.push_back(*e); return r; }
};
}  // namespace

int jx_map_kernel_time(jx_ctx* ctx, const double* theta_dev, int nwalkers, int repeats, double* ms_out) {
    if (!ctx || !theta_dev || nwalkers < 1 || repeats < 1 || !ms_out) return JX_ERR_INVALID;
    if (!ctx->finalized) { ctx->err = "jx_map_kernel_time before jx_finalize"; return JX_ERR_STATE; }
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    const size_t S = ctx->cfg.S;
    JxDev dm = ctx->d;
    dm.quad = 0; dm.img_ld = (long long)S; dm.img_ws = (long long)(S * S); dm.cf_out = nullptr; dm.xcol = nullptr;
    dm.map_split = std::max(dm.map_split, 2);                  // (two row slabs per walker: launches desynchronise, stores overlap compute)
    int threads; size_t lds;
    if (!map_geometry(dm, 512, &threads, &lds, ctx->map_pair != 0)) { ctx->err = "radial grid too long for the LDS-resident spline"; return JX_ERR_UNSUPPORTED; }
    Scratch sc;
    double* img = nullptr;
    HIPCHK(ctx, sc.alloc((void**)&img, sizeof(double) * (size_t)nwalkers * S * S));
    hipEvent_t e0, e1;
    HIPCHK(ctx, sc.event(&e0)); HIPCHK(ctx, sc.event(&e1));
    hipStream_t st = ctx->stream;
    launch_map(st, dm, threads, lds, theta_dev, 0, nwalkers, img, nullptr, nullptr, nullptr, false);       // warm-up
    HIPCHK(ctx, hipEventRecord(e0, st));
    for (int i = 0; i < repeats; ++i) launch_map(st, dm, threads, lds, theta_dev, 0, nwalkers, img, nullptr, nullptr, nullptr, false);
    HIPCHK(ctx, hipEventRecord(e1, st));
    HIPCHK(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
    HIPCHK(ctx, hipGetLastError());
    *ms_out = (double)ms / repeats;
    return JX_OK;
}

// Test hook: the table-driven exp and log of the per-walker kernel (jx_fastmath.hpp) on n host values, through the same tables the
// context uploaded -- what tests/test_gpu_fastmath.py holds to 2 ulp against long double.
__global__ void __launch_bounds__(256) jx_fastmath_kernel(const double* __restrict__ tab, const double* __restrict__ x, double* __restrict__ oe, double* __restrict__ ol, int n) {
    __shared__ double st[JX_FM_TABLE_DOUBLES];
    for (int i = threadIdx.x; i < JX_FM_TABLE_DOUBLES; i += blockDim.x) st[i] = tab[i];
    __syncthreads();
    const JxFm t{st, st + JX_FM_EXP_N};
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { oe[i] = jx_fm_exp(t, x[i]); ol[i] = jx_fm_log(t, x[i]); }
}

int jx_fastmath_eval(jx_ctx* ctx, const double* x, int n, double* exp_out, double* log_out) {
    if (!ctx || !x || n < 1 || !exp_out || !log_out) return JX_ERR_INVALID;
    if (!ctx->finalized) { ctx->err = "jx_fastmath_eval before jx_finalize"; return JX_ERR_STATE; }
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    Scratch sc;
    double *dx = nullptr, *de = nullptr, *dl = nullptr;
    HIPCHK(ctx, sc.alloc((void**)&dx, sizeof(double) * (size_t)n)); HIPCHK(ctx, sc.alloc((void**)&de, sizeof(double) * (size_t)n)); HIPCHK(ctx, sc.alloc((void**)&dl, sizeof(double) * (size_t)n));
    HIPCHK(ctx, hipMemcpyAsync(dx, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(jx_fastmath_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d.fm_tab, dx, de, dl, n);
    HIPCHK(ctx, hipMemcpyAsync(exp_out, de, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(log_out, dl, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return JX_OK;
}

// What a pair of HIP events around ONE kernel of a dependent chain reads when that kernel does nothing: a one-wave kernel that
// returns at once, between two events, with another such kernel in front and behind (the situation of the stage events of
// jx_timing_enable).  The figure (a few microseconds: the command processor's hand-over between dependent dispatches, which
// rocprofv3's kernel trace does not count into a kernel's duration) is what the HIP-event duration of a short kernel carries on
// top of the kernel itself; bench.py reports it beside `roofline.launch_ms`.
__global__ void jx_null_kernel(int* p) { if (p && threadIdx.x == 1024) *p = 0; }

int jx_event_bracket_time(jx_ctx* ctx, int repeats, double* ms_out) {
    if (!ctx || repeats < 1 || repeats > 256 || !ms_out) return JX_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    Scratch sc;
    std::vector<hipEvent_t> ev(2 * (size_t)repeats);
    for (auto& e : ev) HIPCHK(ctx, sc.event(&e));
    hipStream_t st = ctx->stream;
    for (int i = 0; i < repeats; ++i) {
        hipLaunchKernelGGL(jx_null_kernel, dim3(1), dim3(64), 0, st, (int*)nullptr);
        HIPCHK(ctx, hipEventRecord(ev[2 * i], st));
        hipLaunchKernelGGL(jx_null_kernel, dim3(1), dim3(64), 0, st, (int*)nullptr);
        HIPCHK(ctx, hipEventRecord(ev[2 * i + 1], st));
        hipLaunchKernelGGL(jx_null_kernel, dim3(1), dim3(64), 0, st, (int*)nullptr);
    }
    HIPCHK(ctx, hipStreamSynchronize(st));
    HIPCHK(ctx, hipGetLastError());
    double sum = 0.0;
    for (int i = 0; i < repeats; ++i) { float ms = 0.f; HIPCHK(ctx, hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); sum += ms; }
    *ms_out = sum / repeats;
    return JX_OK;
}

// Device-to-device copy bandwidth of this GPU (read + write bytes per second, in GB/s): `nbytes` copied `repeats` times
// between two scratch buffers by a plain grid-stride kernel (16 B per lane), timed with HIP events on the context's
// stream.  The practical HBM roofline to quote beside the nominal one.
__global__ void __launch_bounds__(256) jx_copy_kernel(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) jx_fill_kernel(double2* __restrict__ dst, size_t n, double v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = double2{v, v};
}
__global__ void __launch_bounds__(256) jx_readsum_kernel(const double2* __restrict__ src, double* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const double2 v = src[i]; s += v.x + v.y; }
    if (s == 1.2345e300) out[0] = s;                       // (never: keeps the loads)
}

// mode 0: copy (bytes read + written counted), 1: read stream, 2: write stream.  Launch shapes: the best of
// scripts/ubench/hbm_rates.hip on this chip (a write stream wants FEW blocks: one per CU sweeps memory as one compact window).
int jx_stream_bandwidth(jx_ctx* ctx, int mode, size_t nbytes, int repeats, double* gbps_out) {
    if (!ctx || mode < 0 || mode > 2 || nbytes < 4096 || repeats < 1 || !gbps_out) return JX_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    Scratch sc;
    void *a = nullptr, *b = nullptr;
    HIPCHK(ctx, sc.alloc(&a, nbytes));
    HIPCHK(ctx, sc.alloc(&b, nbytes));
    hipStream_t st = ctx->stream;
    hipEvent_t e0, e1;
    HIPCHK(ctx, sc.event(&e0)); HIPCHK(ctx, sc.event(&e1));
    HIPCHK(ctx, hipMemsetAsync(a, 0, nbytes, st));
    HIPCHK(ctx, hipMemsetAsync(b, 0, nbytes, st));
    const size_t n = nbytes / sizeof(double2);
    const int per_cu = mode == 2 ? 1 : 4;
    const dim3 grid((unsigned)(ctx->num_cu * per_cu));
    auto go = [&]() {
        if (mode == 0) hipLaunchKernelGGL(jx_copy_kernel, grid, dim3(256), 0, st, (const double2*)a, (double2*)b, n);
        else if (mode == 1) hipLaunchKernelGGL(jx_readsum_kernel, grid, dim3(256), 0, st, (const double2*)a, (double*)b, n);
        else hipLaunchKernelGGL(jx_fill_kernel, grid, dim3(256), 0, st, (double2*)b, n, 1.0);
    };
    go();
    HIPCHK(ctx, hipEventRecord(e0, st));
    for (int i = 0; i < repeats; ++i) go();
    HIPCHK(ctx, hipEventRecord(e1, st));
    HIPCHK(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
    HIPCHK(ctx, hipGetLastError());
    *gbps_out = (mode == 0 ? 2.0 : 1.0) * (double)(n * sizeof(double2)) * repeats / ((double)ms * 1e-3) / 1e9;
    return JX_OK;
}

int jx_copy_bandwidth(jx_ctx* ctx, size_t nbytes, int repeats, double* gbps_out) {
    if (!ctx || nbytes < 4096 || repeats < 1 || !gbps_out) return JX_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    Scratch sc;
    void *a = nullptr, *b = nullptr;
    HIPCHK(ctx, sc.alloc(&a, nbytes));
    HIPCHK(ctx, sc.alloc(&b, nbytes));
    hipStream_t st = ctx->stream;
    hipEvent_t e0, e1;
    HIPCHK(ctx, sc.event(&e0)); HIPCHK(ctx, sc.event(&e1));
    HIPCHK(ctx, hipMemsetAsync(a, 0, nbytes, st));
    const size_t n = nbytes / sizeof(double2);
    const dim3 grid((unsigned)(ctx->num_cu * 16));
    hipLaunchKernelGGL(jx_copy_kernel, grid, dim3(256), 0, st, (const double2*)a, (double2*)b, n);
    HIPCHK(ctx, hipEventRecord(e0, st));
    for (int i = 0; i < repeats; ++i) hipLaunchKernelGGL(jx_copy_kernel, grid, dim3(256), 0, st, (const double2*)a, (double2*)b, n);
    HIPCHK(ctx, hipEventRecord(e1, st));
    HIPCHK(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
    HIPCHK(ctx, hipGetLastError());
    *gbps_out = 2.0 * (double)(n * sizeof(double2)) * repeats / ((double)ms * 1e-3) / 1e9;
    return JX_OK;
}

void jx_destroy(jx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) (void)jx_comm_destroy(ctx);
    fft_teardown(ctx->fft);
    mix_teardown(ctx, ctx->mix);
    for (auto& es : ctx->ev_inflight) for (int k = 0; k < 6; ++k) (void)hipEventDestroy(es.e[k]);
    for (auto& es : ctx->ev_free) for (int k = 0; k < 6; ++k) (void)hipEventDestroy(es.e[k]);
    for (auto& pr : ctx->gt_inflight) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto& pr : ctx->gt_free) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
    if (ctx->ev_side) (void)hipEventDestroy(ctx->ev_side);
    if (ctx->ev_tail) (void)hipEventDestroy(ctx->ev_tail);
    for (auto& g : ctx->gslot) if (g.done) (void)hipEventDestroy(g.done);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    for (void* p : ctx->dev_allocs) (void)hipFree(p);
    for (void* p : ctx->samp_buf) if (p) (void)hipFree(p);
    if (ctx->d_theta) (void)hipFree(ctx->d_theta);
    if (ctx->d_logp) (void)hipFree(ctx->d_logp);
    if (ctx->h_theta) (void)hipHostFree(ctx->h_theta);
    if (ctx->h_logp) (void)hipHostFree(ctx->h_logp);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (--g_rocfft_refs == 0) rocfft_cleanup();
    delete ctx;
}

}  // extern "C"
