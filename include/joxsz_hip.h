/*
 * joxsz_hip.h -- C-ABI of the MI355X (gfx950) JoXSZ log-posterior library.
 *
 * The reference (fcastagna/JoXSZ) is pure Python and has no FFI of its own;
 * its "plugin API" for this path is the bound callable installed on the Fit
 * object at joxsz_main.py:186-188 and handed to emcee at joxsz_main.py:206:
 *
 *     fit.getLikelihood(vals) -> float        (joxsz_funcs.py:507-546)
 *
 * This header is what a ctypes binding of that callable binds (see
 * INTEGRATION.md).  Plain pointers and sizes only; no C++ or torch types.
 * Every function returns 0 on success or a negative jx_status; nothing throws
 * across the boundary.  Physics rejections (parameter outside its prior box,
 * r_c > r_s, non-monotone hydrostatic mass, non-positive X-ray model) are NOT
 * errors: they are -inf entries in the output, exactly as the reference
 * returns them (joxsz_funcs.py:519-520, 524-525, 532, 397-407).
 *
 * Ownership: the caller owns all host buffers; the context owns all device
 * memory, the rocFFT plans and the HIP stream; jx_upload copies.
 * Threading: one context per device per host thread; a context is not
 * thread-safe.  Multi-GPU = one process, one context per device.
 */
#ifndef JOXSZ_HIP_H
#define JOXSZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JX_ABI_VERSION 5

typedef struct jx_ctx jx_ctx;

typedef enum jx_status {
    JX_OK = 0,
    JX_ERR_INVALID = -1,      /* bad argument / inconsistent sizes            */
    JX_ERR_STATE = -2,        /* call order (upload after finalize, ...)      */
    JX_ERR_MISSING = -3,      /* a required tensor was never uploaded         */
    JX_ERR_HIP = -4,          /* HIP runtime error (jx_last_error has detail) */
    JX_ERR_ROCFFT = -5,       /* rocFFT error                                 */
    JX_ERR_NOMEM = -6,        /* host or device allocation failed             */
    JX_ERR_NODEVICE = -7,     /* no usable gfx950 device                      */
    JX_ERR_UNSUPPORTED = -8,  /* size outside what the kernels support        */
    JX_ERR_COMM = -9          /* RCCL error or library not loadable (jx_last_error has detail) */
} jx_status;

/* Sizes and scalars of one problem: the shapes of SZ_data (joxsz_funcs.py:157-170)
 * and of the X-ray Data/Annuli/Band objects (joxsz_main.py:116-125). */
typedef struct jx_config {
    int32_t abi_version;      /* = JX_ABI_VERSION                                              */
    int32_t S;                /* SZ map side: d_mat is S x S              (joxsz_main.py:105)  */
    int32_t N;                /* radial grid points: len(r_pp)            (joxsz_main.py:104)  */
    int32_t B;                /* beam image side (odd)                    (joxsz_funcs.py:64-68)*/
    int32_t nflux;            /* SZ flux data points: flux_data is 3 x nflux (joxsz_main.py:97)*/
    int32_t nconv;            /* Compton->Jy/beam table points            (joxsz_main.py:108)  */
    int32_t nann;             /* X-ray annuli                             (joxsz_main.py:116)  */
    int32_t nband;            /* X-ray energy bands                       (joxsz_main.py:73)   */
    int32_t ntab;             /* count-rate table length (CountRate.Tlogvals, joxsz_funcs.py:669)*/
    int32_t npar;             /* parameter table length: 16 (single) or 19 (double beta)       */
    int32_t ndim;             /* thawed parameters = len(fit.thawed)      (joxsz_main.py:179)  */
    int32_t ne_mode;          /* 0 = 'single', 1 = 'double'               (joxsz_main.py:135)  */
    int32_t exclude_unphy_mass; /* joxsz_main.py:88                                            */
    int32_t sz_only;          /* 1 = skip the X-ray term (build extension, BASELINE configs[1])*/
    int32_t device;           /* HIP device ordinal                                            */
    int32_t max_batch;        /* walkers processed per internal chunk (0 = library default)    */
    int32_t fft_pad;          /* padded side of the beam convolution (0 = library default)     */
    int32_t map_split;        /* row slabs per walker in the Abel+map kernel (0 = default)     */
    int32_t conv_mode;        /* beam + transfer-function step: 0 auto, 1 the literal FFT sequence (joxsz_funcs.py:460-467 executed pass by pass; called "rocFFT sequence" below: its transforms are hand-written
                               * LDS transforms, csrc/jx_fft.hpp, wherever both sides are 2^a 3^b 5^c, and rocFFT plans elsewhere or on request, see jx_get_fft_info), 2 hand-written kernels:
                               * the exact form (default: the row as one constant operator on the spline ordinates, see jx_get_conv_layout) or, behind the option
                               * JOXSZ_MIX_FORM, the contracted forms of rounds 3-4 */
    int32_t dtype;            /* arithmetic of the SZ stages: 0 = f64 (the reference's; the exact form).  The fp32 variants run on the contracted low-rank form of
                               * round 4 (16 singular terms, a sub-grid of map samples; measured by its guard, jx_get_truncation): 1 = fp32 spline arrays, every sum in
                               * fp64 (|delta chi^2/2| ~1e-6); 2 = fp32 arithmetic in stage 1 (packed fp32 FMAs) and stage 2 (fp32 matrix cores), K slices added in fp64:
                               * |delta chi^2/2| 2e-5 at 512^2, 1e-4 at 1024^2 ABSOLUTE -- inside north_star's relative 1e-6 only because the log-posterior is ~1e4;
                               * jx_audit measures it on the caller's walkers */
    int32_t calc_integ;       /* SZ_data.calc_integ: integrated-Compton term (joxsz_funcs.py:480-484, joxsz_main.py:65) */
    int32_t reserved1;        /* keeps the doubles 8-byte aligned; must be 0                   */
    double step;              /* arcsec                                   (joxsz_main.py:21)   */
    double kpc_as;            /* kpc per arcsec                           (joxsz_main.py:96)   */
    double m_e;               /* keV                                      (joxsz_main.py:22)   */
    double sigma_T;           /* cm^2                                     (joxsz_main.py:23)   */
    double kpc_cm;            /* mbproj2.physconstants.kpc_cm             (joxsz_funcs.py:6)   */
    double integ_mu;          /* SZ_data.integ_mu                         (joxsz_main.py:66)   */
    double integ_sig;         /* SZ_data.integ_sig                        (joxsz_main.py:67)   */
} jx_config;

/* Constant tensors, all float64 row-major unless noted. */
typedef enum jx_tensor {
    JX_T_R_PP = 0,        /* [N]            kpc                 SZ_data.r_pp      */
    JX_T_D_MAT,           /* [S,S]          kpc                 SZ_data.d_mat     */
    JX_T_BEAM_2D,         /* [B,B]                              SZ_data.beam_2d   */
    JX_T_FILTERING,       /* [S,S]          FFT layout          SZ_data.filtering */
    JX_T_RADIUS,          /* [S]            arcsec              SZ_data.radius    */
    JX_T_FLUX_DATA,       /* [3,nflux]      r, flux, err        SZ_data.flux_data */
    JX_T_CONV_T,          /* [nconv]        keV                 joxsz_main.py:108 */
    JX_T_CONV_V,          /* [nconv]        1e3*Jy/beam         joxsz_main.py:109 */
    JX_T_PAR_VALS,        /* [npar]   current value of every parameter (frozen ones matter) */
    JX_T_PAR_MIN,         /* [npar]   Param.minval                                          */
    JX_T_PAR_MAX,         /* [npar]   Param.maxval                                          */
    JX_T_PAR_KIND,        /* [npar]   int32: 0 = Param (box), 1 = ParamGaussian             */
    JX_T_PAR_MU,          /* [npar]   ParamGaussian.prior_mu                                */
    JX_T_PAR_SIGMA,       /* [npar]   ParamGaussian.prior_sigma                             */
    JX_T_THAWED_IDX,      /* [ndim]   int32: slot of each thawed value, order of fit.thawed */
    JX_T_X_R_NE,          /* [nann]   kpc: radii where n_e is evaluated                     */
    JX_T_X_R_T,           /* [nann]   kpc: annuli.midpt_kpc (joxsz_funcs.py:339)            */
    JX_T_PROJVOLS,        /* [nann,nann] cm^3: annuli.projvols                              */
    JX_T_CTS,             /* [nband,nann]  band.cts (NaN = missing, joxsz_funcs.py:504)     */
    JX_T_AREASCALES,      /* [nband,nann]  joxsz_funcs.py:204                               */
    JX_T_EXPOSURES,       /* [nband,nann]  joxsz_funcs.py:201                               */
    JX_T_BACKRATES,       /* [nband,nann]  joxsz_funcs.py:207                               */
    JX_T_GEOMAREA,        /* [nann]   arcmin^2: annuli.geomarea_arcmin2                     */
    JX_T_LNT,             /* [ntab]   CountRate.Tlogvals                                    */
    JX_T_LNRATE,          /* [nband,2,ntab] ln(rate) at Z=0 and Z=1 (joxsz_funcs.py:680)    */
    JX_T_INTEG_W,         /* [N+1] only with calc_integ: weights w of  cint = sum_j w_j v_j,  v = [f(0), y_0 .. y_{N-1}]  --
                             the Simpson rule of joxsz_funcs.py:481-483 on its arcmin grid, times the radius, times 2 pi */
    JX_T_COUNT
} jx_tensor;

/* Intermediate quantities that can be read back per walker (parity taps).
 * The names follow get_sz_like's `output` argument (joxsz_funcs.py:439-493). */
typedef enum jx_stage {
    JX_STAGE_PP = 0,      /* [W,N]      pressure profile            joxsz_funcs.py:453 */
    JX_STAGE_AB,          /* [W,N]      Abel transform              joxsz_funcs.py:457 */
    JX_STAGE_Y,           /* [W,N]      Compton y(r)                joxsz_funcs.py:459 */
    JX_STAGE_Y2D,         /* [W,S,S]    Compton-y map               joxsz_funcs.py:462 */
    JX_STAGE_CONV2D,      /* [W,S,S]    beam-convolved map          joxsz_funcs.py:464 */
    JX_STAGE_MAPROW,      /* [W,nrow]   map_out[S//2, S//2:]        joxsz_funcs.py:472 */
    JX_STAGE_BRIGHT,      /* [W,nrow]   output='bright'             joxsz_funcs.py:473 */
    JX_STAGE_CHISQ,       /* [W]        output='chisq'              joxsz_funcs.py:478 */
    JX_STAGE_TPROF,       /* [W,nrow]   [h(0), t_prof]              joxsz_funcs.py:469-473 */
    JX_STAGE_XPROFS,      /* [W,nband,nann] calcProfiles()          joxsz_funcs.py:527 */
    JX_STAGE_PARTS,       /* [W,4]      xray like, sz like, prior, reject mask (bit0 box, bit1 mass, bit2 r_c>r_s, bit3 xray<=0) */
    JX_STAGE_INTEG,       /* [W]        output='integ' (calc_integ only)  joxsz_funcs.py:481-487 */
    JX_STAGE_COUNT
} jx_stage;

/* GPU time per stage, accumulated over launches since the last reset, measured
 * with HIP events on the context's stream. */
typedef struct jx_timing {
    double prep_ms;       /* priors, mass veto, T profile, X-ray Cash      */
    double abel_map_ms;   /* exact form: ordinate product (Abel transform + Compton-y scale) and row product on the matrix cores (jx_ordrow_kernel); contracted forms:
                           * Abel integral + spline arrays as one matrix product; rocFFT sequence: profile -> Abel -> spline -> y map */
    double beam_fft_ms;   /* exact form: 0; contracted forms: stage 1 (jx_rowmix_kernel); rocFFT sequence: forward rows + column kernel (hand-written transforms; the inverse rows are
                           * inside the next kernel), or R2C + multiply + C2R (rocFFT plans) */
    double tf_fft_ms;     /* exact form: 0; contracted forms: stage 2 / the full form's product (jx_opgemm_kernel); rocFFT sequence: inverse rows + window + forward rows, then the
                           * transfer-function column kernel (hand-written), or R2C of the S x S window (rocFFT plans) */
    double tail_ms;       /* exact form: partial rows added, conversion, data radii, chi^2, total (jx_rowsum_tail_kernel); otherwise filter + central row + conversion + chi^2 */
    double total_ms;      /* first event to last event of each launch      */
    int64_t launches;     /* internal chunks timed                         */
    int64_t walkers;      /* walkers those chunks processed                */
    double gemm_ms;       /* unused (0); kept so that the struct keeps its size */
} jx_timing;

int  jx_create(const jx_config* cfg, jx_ctx** out);
int  jx_upload(jx_ctx* ctx, int tensor_id, const void* host, size_t nbytes);
/* Build the walker-independent tables (Abel weights, spline operators, the row operator of the exact form -- or the operators of a
 * contracted form, or the beam spectrum and transfer-function table of the rocFFT sequence) and the work buffers.  Must be called once
 * after all uploads and before jx_eval.  The options below are read here and nowhere else (the two SAMPLE_ ones: by jx_sample). */
int  jx_finalize(jx_ctx* ctx);

/* One switch of the library, per context.  `name` with or without its JOXSZ_ prefix, any case; `value` as text (NULL or "": unset, whatever
 * the environment says).  The process environment variable of the same name is the DEFAULT of every option, so nothing has to depend on
 * it.  JX_ERR_INVALID for a name that is not in this table, JX_ERR_STATE after jx_finalize for an option that is read there.
 *
 *   name                         values (default)          effect
 *   JOXSZ_CONV                   auto|rocfft|custom (auto) overrides jx_config.conv_mode
 *   JOXSZ_MIX_FORM               exact|legacy|lowrank|full (exact)  form of the hand-written route: the exact form, or the contracted forms of rounds 3-4
 *                                                          (legacy: the cheaper of low-rank and full, as round 4 picked; they carry their own options below)
 *   JOXSZ_X_PAIRWISE             1|0 (1)                   exact form: 0 = its reference kernels (one block per 16 walkers reads the ordinates back; jx_rowop_tail_kernel)
 *   JOXSZ_X_FOLD                 1|0 (1)                   exact form, timed path: the ordinates stay in LDS (nothing reads them), and with an odd number of 16-ordinate tiles the last tile's
 *                                                          share of the row is an operator on the profile inside the row product, the others in exact pairs (one block fewer per 16
 *                                                          walkers: 768 instead of 832 at 512^2 / 1024 walkers); 0: every tile through the ordinate product, ordinates stored (as calls with taps do)
 *   JOXSZ_PRUNE_OUTPUTS          1|0 (1)                   0: the row product computes every output of the row, read by the data-radii spline or not
 *   JOXSZ_CHUNK                  walkers                   overrides jx_config.max_batch
 *   JOXSZ_FFT_PAD, JOXSZ_MAP_SPLIT, JOXSZ_MAP_PAIR         rocFFT sequence / Abel + map kernel: padded side, row slabs per walker, two walkers per block (1)
 *   JOXSZ_FFT_COLUMNS            custom|rocfft (custom)    rocFFT sequence: its column passes hand-written (jx_fft.hpp: rocFFT transforms the rows, one kernel per column group does
 *                                                          forward, times the beam spectrum, inverse) wherever both sides are 2^a 3^b 5^c and <= 1280; rocfft: rocFFT's own 2-D plans
 *   JOXSZ_FFT_ROWS               custom|rocfft (custom)    beside the hand-written columns: the row transforms hand-written as well (two real rows per complex transform; the
 *                                                          inverse rows of the convolution and the forward rows of the transfer function in one kernel) or batched 1-D rocFFT plans
 *   JOXSZ_EVAL_DIRECT            0..3 (3)                  jx_eval: bit 0 the tail stores into the caller-visible host buffer, bit 1 the per-walker kernel reads theta from it
 *   JOXSZ_PREP_SPLIT             1|0 (1)                   per-walker kernel as two blocks per walker (X-ray side beside the rest); same bits
 *   JOXSZ_PREP_LEAN              1|0 (1)                   the two-block form as jx_walker2_kernel (every table of a block in one batched copy into LDS); 0: inside jx_prep_kernel; same bits
 *   JOXSZ_PREP_POW               0|1 (0)                   per-walker kernel: profiles with pow() as written in the reference instead of through their exponents
 *   JOXSZ_PREP_FASTMATH          1|0 (1)                   per-walker kernel: table-driven exp / log (2 ulp) instead of the device library's
 *   JOXSZ_OP_NARROW              0|1 (0)                   collapsed route: the small-launch kernel at every launch size
 *   JOXSZ_SAMPLE_FUSED           1|0 (1)                   jx_sample: proposal and acceptance inside the likelihood's kernels (0: kernels of their own)
 *   JOXSZ_SAMPLE_VIRTUAL_RANKS   R                         jx_sample without a communicator: R shares of each half step in turn (a test of the share arithmetic)
 *   -- contracted forms only (JOXSZ_MIX_FORM=legacy|lowrank|full) --
 *   JOXSZ_LOWRANK_TOL            cut (1e-8 / 1e-13)        singular-value cut of the transfer-function weights, pinned (measured, never tightened)
 *   JOXSZ_TRUNC_PROBE, JOXSZ_TRUNC_BOUND                   truncation guard: 0 = off; bound on the row estimate (1e-9)
 *   JOXSZ_MIX_SUBSAMPLE          0 | u0,u1,npts            sub-grid of map samples: 0 = every distinct sample
 *   JOXSZ_MIX_RANKCAP            0..16 (16)                cap on the rank of the low-rank form
 *   JOXSZ_AG_SUBSAMPLE           0 | u0,u1,npts            radial sub-grid of the spline-array product: 0 = every radius
 *   JOXSZ_MIX_MFMA, JOXSZ_MIX_USPLIT, JOXSZ_MIX_WPB, JOXSZ_MIX_KSPLIT, JOXSZ_MIX_KSPLIT_USE, JOXSZ_ABEL_GEMM, JOXSZ_AG_NARROW, JOXSZ_AG_SINGLE,
 *   JOXSZ_SIDE_STREAM, JOXSZ_SIDE_FORK                     tuning experiments of round 4
 *   -- diagnostic build only (make ABLATIONS=1; results are wrong or the launch path reads the clock) --
 *   JOXSZ_DBG, JOXSZ_MIX_DBG, JOXSZ_X_STAMPS, JOXSZ_P_STAMPS
 */
int  jx_set_option(jx_ctx* ctx, const char* name, const char* value);

/* theta: [nwalkers, ndim] row-major float64; logp: [nwalkers] float64.
 * The call returns after logp is complete (synchronous on the context's stream). */
int  jx_eval(jx_ctx* ctx, const double* theta_host, int nwalkers, double* logp_host);
/* Same with DEVICE pointers (hipMalloc'ed by jx_dev_alloc or by any other HIP
 * allocator of this process on the context's device).  Asynchronous: returns
 * once the work is enqueued on the context's stream; pair with jx_sync. */
int  jx_eval_device(jx_ctx* ctx, const double* theta_dev, int nwalkers, double* logp_dev);
int  jx_sync(jx_ctx* ctx);
/* Enqueue everything that follows on the caller's HIP stream (a hipStream_t, e.g. the one the caller's collective
 * library orders its work against) instead of the context's own; NULL goes back to the own stream.  The context never
 * destroys a stream it was handed.  Pending work on the previous stream is waited for first.  (NULL is NOT the legacy
 * default stream here: a caller whose work sits on that stream, e.g. torch's default stream, creates a stream for it.) */
int  jx_set_stream(jx_ctx* ctx, void* hip_stream);
/* Device-resident affine-invariant ensemble sampler (Goodman & Weare stretch move in emcee's red/blue form: what
 * mcmc.sample does with the reference's callable, joxsz_funcs.py:593-622): nsteps iterations of two half steps, with
 * proposals, log-posterior evaluation, accept/reject and the chain all on the device and one copy back at the end.
 * theta0 [nwalkers][ndim] (finite log-posterior required, nwalkers even); chain_out [nsteps][nwalkers][ndim],
 * logp_out [nsteps][nwalkers], naccept_out [nwalkers] are host arrays and may be NULL.  Random numbers are
 * Philox4x32-10 keyed by seed with counter (walker slot, 2 iteration + half, draw, 0): a run is reproducible and can
 * be replayed on the host (joxsz_amd/sampler.py).
 *
 * Over several GPUs: with a communicator on the context (jx_comm_init_rank) every rank calls jx_sample with the same arguments; a rank
 * moves its contiguous share of each half step (the half ensemble must divide by the ranks), then one in-place RCCL all-gather of
 * the share's positions and log-posteriors brings every rank's copy of the ensemble up to date; chains, log-posteriors and summed
 * acceptance counts come back identical on every rank, and identical to the single-rank run (the shares of a half step do not read
 * one another).  On the contracted route the proposal is drawn inside the per-walker kernel and accepted or rejected inside the tail
 * (JOXSZ_SAMPLE_FUSED=0: kernels of their own, single rank only).  JOXSZ_SAMPLE_VIRTUAL_RANKS=R runs R shares in turn in one process
 * (a test of the share arithmetic).
 */
int  jx_sample(jx_ctx* ctx, const double* theta0_host, int nwalkers, int nsteps, double a, uint64_t seed,
               double* chain_out, double* logp_out, int64_t* naccept_out);
/* Route of jx_eval / jx_eval_device / jx_sample for the SZ side of the log-posterior.
 *   JX_ROUTE_MAP (default): per walker, every call: pressure profile (joxsz_funcs.py:453), forward Abel transform and Compton-y scale
 *       (:457-459), then everything that is linear with constant coefficients behind them -- the mirrored cubic spline, the S x S map, the
 *       beam convolution, the transfer function, the central row (:460-472) -- as ONE constant operator on the spline ordinates, built
 *       at jx_finalize from the caller's d_mat, beam image and filter with nothing truncated (the exact form; conv_mode 1 executes the
 *       same lines literally through rocFFT).
 *   JX_ROUTE_OPERATOR: those steps are linear in the pressure profile with constant coefficients, so the row is
 *       G pp with one constant nrow x N matrix.  Switching to this route builds G once by sending the N unit profiles
 *       through the MAP route's kernels (a few launches); afterwards a walker costs press_fun + one nrow x N
 *       matrix-vector product on the SZ side (launches of 4096 walkers and more: one product on the fp64 matrix cores;
 *       JOXSZ_OP_NARROW=1 keeps the small-launch kernel).  Same results to rounding (the operator inherits the MAP route's own
 *       truncation, see jx_get_conv_layout); works for every map size, odd sides included.
 * jx_eval_stage always runs the MAP route (the intermediate stages do not exist on the other one).  The data tensors
 * G depends on (r_pp, d_mat, beam_2d, filtering, step, constants) are fixed at jx_finalize, so G never goes stale. */
enum { JX_ROUTE_MAP = 0, JX_ROUTE_OPERATOR = 1 };
int  jx_set_route(jx_ctx* ctx, int route);
int  jx_get_route(jx_ctx* ctx);
/* Copy of G as the library holds it, out_host[N][nrow] (column j = row response of the unit profile e_j);
 * JX_ERR_STATE before the operator route was first selected. */
int  jx_get_operator(jx_ctx* ctx, double* out_host, size_t nbytes);
/* Parity/debug tap: evaluates and copies one intermediate quantity to host. */
int  jx_eval_stage(jx_ctx* ctx, const double* theta_host, int nwalkers, int stage_id,
                   double* out_host, size_t nbytes);

/* Replace the stored value of one parameter-table entry (keeps a frozen
 * parameter, or the "current parameters" of getLikelihood(None), in step with
 * the host-side fit object; joxsz_funcs.py:515-516). */
int  jx_set_par_vals(jx_ctx* ctx, const double* par_vals, int npar);

/* Run-time assurance: the given walkers (host parameter vectors [n][ndim]) through this context's own route AND through the rocFFT sequence
 * held inside the same context (joxsz_funcs.py:460-467 executed literally, 16 walkers at a time: an independent implementation on the same
 * inputs), compared where a chain lives.  out = {largest |difference of the SZ log-likelihood| (absolute), largest difference of the extracted
 * row relative to the row's largest entry, index of the walker with the largest log-likelihood difference (-1: none), walkers compared: those
 * that gave finite numbers on both sides and are not rejected (prior box, mass veto, r_c > r_s, X-ray model)}.  The exact form reads ~1e-11 / 1e-14; a contracted form or an fp32 variant reads what its approximations cost on
 * exactly these walkers (joxsz_amd/chain.py::mcmc_run calls it every few hundred steps and warns once).  Costs ~1 ms per 16 walkers.
 * All zeros with conv_mode 1 (the rocFFT sequence is then the route itself). */
int  jx_audit(jx_ctx* ctx, const double* theta_host, int n, double out[4]);

/* Device memory helpers so that the host side needs no other GPU runtime. */
int  jx_dev_alloc(jx_ctx* ctx, size_t nbytes, void** dev_out);
int  jx_dev_free(jx_ctx* ctx, void* dev);
int  jx_memcpy_h2d(jx_ctx* ctx, void* dev, const void* host, size_t nbytes);
int  jx_memcpy_d2h(jx_ctx* ctx, void* host, const void* dev, size_t nbytes);

/* ---- multi-GPU: one process and one context per device, walkers sharded with no data-path collective; the only exchange
 * is the all-gather of the log-probabilities (RCCL over xGMI), enqueued on the context's stream behind the evaluation.
 * The reference's counterpart is multiprocessing.Pool handed to emcee (joxsz_main.py:203-208).
 *   jx_comm_unique_id   rank 0 creates the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means
 *                       (joxsz_amd/dist.py: a file under the launcher's run directory; the C-ABI does not care);
 *   jx_comm_init_rank   every rank, collectively (ncclCommInitRank on the context's device);
 *   jx_allgather_logp   recv_dev[rank * count .. ] <- send_dev[0 .. count) of every rank (float64), asynchronous on the
 *                       context's stream like jx_eval_device; send and receive buffers are device memory.  In overlap mode the gather
 *                       runs on the second stream: read recv_dev only behind jx_sync (jx_memcpy_d2h and the caller's own kernels on
 *                       the compute stream are NOT ordered behind it);
 *   jx_comm_allreduce_max  in-place maximum over the ranks of `count` float64 on the device (timing, barriers);
 *   jx_comm_set_overlap on != 0: the communicator's collectives run on a second stream of the context, each ordered behind
 *                       what the compute stream holds at the moment of the call; the next evaluation does NOT wait for the
 *                       gather unless it writes the very buffer being sent (alternate two output buffers and step n+1's
 *                       kernels overlap gather n); jx_sync waits for both streams.  0 (default): strict -- collectives on the
 *                       compute stream, in order with everything else (what a sampler needs that proposes from the gathered
 *                       values);
 *   jx_comm_gather_time sum of the all-gathers' own durations (HIP events on their stream, recorded while jx_timing_enable
 *                       is on) and their number since the last call;
 *   jx_comm_destroy     collective teardown (jx_destroy does it too).
 * librccl is loaded when the first of these is called, not before: single-GPU use never touches it. */
#define JX_COMM_ID_BYTES 128
int  jx_comm_unique_id(void* id_out /* JX_COMM_ID_BYTES */);
int  jx_comm_init_rank(jx_ctx* ctx, const void* id /* JX_COMM_ID_BYTES */, int nranks, int rank);
int  jx_allgather_logp(jx_ctx* ctx, const double* send_dev, double* recv_dev, int count);
int  jx_comm_allreduce_max(jx_ctx* ctx, double* inout_dev, int count);
int  jx_comm_set_overlap(jx_ctx* ctx, int on);
int  jx_comm_gather_time(jx_ctx* ctx, double* ms_total, int64_t* calls);
int  jx_comm_destroy(jx_ctx* ctx);
/* ranks of the communicator as RCCL counts them (ncclCommCount); < 0 on error */
int  jx_comm_count(jx_ctx* ctx);

int  jx_timing_reset(jx_ctx* ctx);
int  jx_timing_enable(jx_ctx* ctx, int on);        /* 0 off; 1 an event behind every kernel of the step; 2..4 only the two events around ONE kernel -- exact form:
                                                     * 2 the ordinate + row product (abel_map_ms), 3 the per-walker kernel (prep_ms), 4 the tail (tail_ms);
                                                     * contracted forms: 2 = stage 1 (beam_fft_ms); launches and walkers are filled */
int  jx_timing_get(jx_ctx* ctx, jx_timing* out);   /* synchronises the stream */

/* Introspection: derived sizes chosen by the library. */
int  jx_get_info(jx_ctx* ctx, int32_t* fft_pad, int32_t* chunk, int32_t* spline_band,
                 int32_t* nrow, int64_t* device_bytes);
/* 1 = rocFFT sequence, 2 = contracted route (what `conv_mode` resolved to); <0 on error */
int  jx_get_conv_mode(jx_ctx* ctx);
/* The transforms of the rocFFT sequence of this context (its own route with conv_mode 1, or the reference facility of a contracted-route
 * context once jx_audit / a conv_2d tap / the truncation probe has built it): out = {built (0: nothing else is filled), columns hand-written
 * (jx_fft.hpp) 0|1, rows hand-written 0|1, padded side P, leading dimension of the row spectra of the padded image, of the window,
 * columns per block, passes of the length-P transform, then its radices (up to 12), passes of the length-S transform, its radices}.
 * columns = 0: rocFFT's own 2-D plans (a side with a prime factor beyond 5, or beyond 1280, or JOXSZ_FFT_COLUMNS=rocfft). */
int  jx_get_fft_info(jx_ctx* ctx, int32_t out[34]);
/* What the hand-written route looks like on this problem: out = {form, NU, rank, beam_terms, R, RT, nxt, ntile, ksteps, tW,
 * ldx, ksplit}.  form 2 = exact (default): out[2] = ordinates the row operator reads (the radial grid beyond the map's corner plus the
 * band of the spline's moment operator does not reach the row: 400 of 500 at 512^2), out[3] = the same in whole tiles of 16, out[6] = output
 * tiles per group, out[7] = output tiles of the whole row, out[8] = k-steps of the row product, out[10] = outputs the timed launch computes,
 * out[11] = 8; nothing is truncated or sub-sampled.  form 0 = low-rank: the beam image in `beam_terms` separable terms, the transfer-function weights of the
 * extracted row in `rank` singular terms (cut JOXSZ_LOWRANK_TOL relative to the largest, default 1e-8 at sides >= 400 and
 * 1e-13 below, see jx_get_truncation), R = rank * beam_terms rows kept per map column by stage 1 (kernel instance RT >= R),
 * stage 2 one matrix-core product with K = NU * R in `ksteps` steps of 4; form 1 = full: one operator row per distinct map
 * sample (NU (NU + 1) / 2), no truncation, chosen when it is the cheaper one (measured beam / rough transfer function;
 * JOXSZ_MIX_FORM=lowrank|full forces a form).  NU distinct map rows = distinct columns, nxt output tiles per block of
 * `ntile`, tW walker stride of the work buffers, ldx doubles per partial row, ksplit K slices of the last launch.
 * JX_ERR_UNSUPPORTED with the rocFFT back end. */
int  jx_get_conv_layout(jx_ctx* ctx, int32_t out[12]);
/* Which map samples stage 1 of the low-rank form evaluates: out = {distinct rows (= columns) of the map quadrant NU; rows stage 1
 * evaluates; full resolution below this many pixels from the axis; every second row up to here (u1), every fourth up to 2 u1, every eighth beyond; interpolation
 * points; 1 when a sub-grid is in use; rebuilds in which the guard took it away; 0}; rows (optional, nrows_cap entries): the kept
 * indices.  Away from the cluster core the Compton-y map varies on the scale of the radius, far above the pixel, so the quadrant
 * is recoverable from a tensor sub-grid of its rows and columns by local polynomial interpolation, Q ~ L Q_sub L^T, and the
 * contraction needs only the transformed operators L^T C and G (L x I): measured error 1e-13 of the row's maximum over the outputs in use
 * at the corners of the prior box in (a, b, r_p), 1e-8 on the log-posterior over the whole box (profiles/r04_subsample_check.log,
 * r04_box_parity.log), measured again on the caller's data by the guard of jx_get_truncation, which takes the sub-grid away first.  JOXSZ_MIX_SUBSAMPLE=0: every distinct sample; "u0,u1,npts": another sub-grid.  Not used when the quadrant
 * reaches beyond the radial grid (fill values: the map is not smooth there), nor where nothing could measure it (radial grids beyond
 * the Abel kernel's LDS).  Both forms of the contracted route use it. */
int  jx_get_sampling(jx_ctx* ctx, int32_t out[8], int32_t* rows, int nrows_cap);
/* Which radii of the pressure profile the spline-array product (jx_abel_gemm_kernel) multiplies: out = {radii of the grid N; radii in
 * use; every radius below this index; every second up to here (u1), every fourth up to 2 u1, every eighth beyond; interpolation points;
 * 1 when the sub-grid is in use}; rows (optional): the kept indices; returns how often the guard took the sub-grid away (>= 0).  The
 * profile is smooth away from the core, so its values on a sub-grid of the radial grid carry the others by high-order interpolation,
 * pp ~ L pp_sub, and the product needs only L^T Tm: K shrinks from N to the kept radii (222 of 500), and the k loop that bounds that
 * kernel with it.  Measured on the caller's data by the guard of jx_get_truncation, which takes it away first.
 * JOXSZ_AG_SUBSAMPLE=0: every radius; "u0,u1,npts": another sub-grid.  Not used where nothing could measure it. */
int  jx_get_radial_sampling(jx_ctx* ctx, int32_t out[6], int32_t* rows, int nrows_cap);
/* Which outputs of the extracted row the matrix-core product computes when no tap asks for the row: out = {nrow; outputs the
 * data-radii spline of the tail (joxsz_funcs.py:476) reads with a weight above 1e-22 of its largest -- the cardinal functions
 * of a cubic spline decay by 2 - sqrt(3) per knot, so the row beyond the last data radius + ~35 pixels does not reach an fp64
 * sum; outputs computed (whole tiles of 16); output tiles per block; K slices; 1 when the restriction is in use (contracted
 * route, fewer tiles than the whole row; JOXSZ_PRUNE_OUTPUTS=0: never)}.  The row and brightness taps always get every output. */
int  jx_get_output_pruning(jx_ctx* ctx, int32_t out[6]);
/* What the truncation of the low-rank form costs on this problem, measured in jx_finalize against the rocFFT sequence
 * (exact, independent kernels) at the probe points: the current parameter values and the corners of the prior box in the
 * thawed shape parameters of the pressure profile (a, b, r_p).  out = {singular-value cut in use; largest difference of the
 * extracted row at the current parameter values, relative to the row's largest entry; rank; number of times jx_finalize
 * found an estimate above its bound and rebuilt the tables -- in place -- with a cut ten times tighter (a rebuild may end
 * in the full form, where the growing rank has made it the cheaper one: rank 0, nothing truncated); probe points that
 * gave finite numbers; bound on out[1] (1e-9; JOXSZ_TRUNC_BOUND); the row difference over ALL probe points; largest
 * difference of the SZ log-likelihood over all probe points relative to max(1, |SZ log-likelihood|); terms above the cut
 * (rank < this: the rank was cut to 16 so that stage 1 fits one 16-row tile of the fp64 matrix cores -- ranks up to 20 are,
 * and the guard takes the cap away before it tightens the cut; JOXSZ_MIX_RANKCAP=0: never cut); bound on out[7] (1e-8);
 * 1 when stage 1 runs on the matrix cores (R <= 16; JOXSZ_MIX_MFMA=0: never); how many of the rebuilds took the cap away}.
 * -1 where nothing is truncated (full form, rocFFT) or nothing was measured (JOXSZ_TRUNC_PROBE=0); an explicit
 * JOXSZ_LOWRANK_TOL is measured but never overridden, and neither is an f32 context (its rounding is of the bounds' size). */
int  jx_get_truncation(jx_ctx* ctx, double out[12]);
/* Test hook (contracted route): device address and geometry of a work buffer holding the last evaluated chunk.
 *   0 quadrant of the Compton-y map [chunk][NU][ld] (exists after the first y_2d tap): geom = {chunk, NU, ld, 8}
 *   1 spline arrays, walker-minor [N][tW][2] = (y_k, M_k) of walker w at ((k tW + w) 2): geom = {N, tW, 2, element bytes}
 *   2 stage-1 rows [NU][R][tW] (low-rank form): geom = {NU, R, tW, 8}
 *   3 partial rows of the last launch, slice ks of walker w at ks (tW ldx + 272) + w ldx: geom = {ksplit, tW, ldx, 8}
 *   4 stage-1 operator C[u][j] (low-rank form), rows of `cld` doubles: geom = {1, wld, cld, 8}
 *   5 operator of the matrix-core product, Op[(kappa * 16 + (x & 15)) * ntile + (x >> 4)]: geom = {4 ksteps, 16, ntile, 8}
 * exact form (the others belong to the contracted forms):
 *   6 Compton-y ordinates of the last chunk, walker-major y[w][k]: geom = {1, tW, Nkp, 8} (written by calls with taps, by the reference kernels and with JOXSZ_X_FOLD=0: the timed path keeps them in LDS)
 *   7 row operator as the matrix cores read it, Opk[(((g nS + s) 4 + e) 64 + lane) nxt + t] = Wy[16 (g nxt + t) + (lane & 15)][16 s + 4 (lane >> 4) + e]: geom = {ng nS 4, 64, nxt, 8} */
int  jx_debug_workspace(jx_ctx* ctx, int which, void** dev, int32_t geom[4]);
/* Duration (ms, HIP events on the context's stream, mean of `repeats` launches) of the Abel + map kernel writing the full
 * S x S Compton-y map of `nwalkers` walkers whose parameter vectors are at theta_dev: the kernel BASELINE's metric is worded
 * around (profile -> Abel integral -> spline -> map; S^2 * 8 bytes per walker).  The evaluation path never stores the map;
 * this call exists for that measurement and allocates its own scratch image. */
int  jx_map_kernel_time(jx_ctx* ctx, const double* theta_dev, int nwalkers, int repeats, double* ms_out);
/* Test hook: the table-driven exp and log the per-walker kernel evaluates its profile chains with (csrc/jx_fastmath.hpp; 16 and 22
 * vector instructions against the device library's 38 and 95; JOXSZ_PREP_FASTMATH=0 selects the library's), on n host values
 * through the tables of this context.  Held to 2 ulp against long double by tests/test_gpu_fastmath.py. */
int  jx_fastmath_eval(jx_ctx* ctx, const double* x, int n, double* exp_out, double* log_out);
/* What a pair of HIP events around one kernel of a dependent chain reads when the kernel does nothing (a one-wave kernel that
 * returns at once, another in front and behind): the part of a stage's HIP-event duration that is not the kernel -- rocprofv3's
 * kernel trace does not count it.  Mean over `repeats` (<= 256) brackets, in ms. */
int  jx_event_bracket_time(jx_ctx* ctx, int repeats, double* ms_out);
/* Device-to-device copy bandwidth of this GPU in GB/s (bytes read + bytes written per second): `nbytes` copied `repeats` times
 * between two scratch buffers, HIP events on the context's stream.  The practical HBM roofline beside the nominal 8 TB/s. */
int  jx_copy_bandwidth(jx_ctx* ctx, size_t nbytes, int repeats, double* gbps_out);
/* The same for one kind of stream: mode 0 copy (bytes read + written), 1 read only, 2 write only (the full-map kernel of
 * jx_map_kernel_time is a write stream).  Hand-written grid-stride kernels in the launch shapes that do best on this chip
 * (scripts/ubench/hbm_rates.hip). */
int  jx_stream_bandwidth(jx_ctx* ctx, int mode, size_t nbytes, int repeats, double* gbps_out);
int  jx_device_count(void);
const char* jx_device_name(jx_ctx* ctx);

const char* jx_strerror(int status);
const char* jx_last_error(jx_ctx* ctx);            /* detail of the last failure on this context; NULL context: of the last context-free call (jx_comm_unique_id) */
void jx_destroy(jx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* JOXSZ_HIP_H */
