#!/usr/bin/env python3
"""Headline benchmark: walker-likelihoods per second of the joint X-ray + SZ
log-posterior at a 512 x 512 map and a 500-point radial grid (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A "step" is one evaluation of the log-posterior of every walker of the ensemble
(what one emcee iteration costs: joxsz_funcs.py:593,600,622 -> getLikelihood
per walker).  Workload at N=1: BASELINE.json configs[2] -- 1024 walkers, S=512,
N=500, joint likelihood, synthetic CL J1226.9+3332-shaped inputs, parameter
vectors already resident in HBM when the timed region starts.  For N>1 every
rank evaluates its own 1024 walkers (weak scaling, walkers are independent) and
the log-probabilities are all-gathered over RCCL inside the timed region --
through the library's own C-ABI (jx_comm_*, jx_allgather_logp): this program
imports no torch.

Ranks: under a launcher (torch.distributed.run sets RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_PORT) each process is one rank.  Without one, `--gpus N`
with N > 1 makes THIS process the launcher: it starts N fresh child processes
(one per GPU) before anything touches a GPU, never initialises HIP itself,
relays rank 0's single JSON line and exits non-zero if any rank fails.

Prints ONE JSON line (rank 0).  `ms_per_step` is the MEDIAN over at least 25 timed regions of `--steps` steps each (each
bracketed by barrier + synchronisation; `ms_per_step_min` / `_max`, `timed_regions`, `timed_ms_total`); beside the contract's fields:
  `roofline`            the longest kernel of the step (picked from a stage pass in the warm-up): the matrix-core kernel on its
                        algorithmic flops over its HIP-event duration against the fp64 peak, or -- when the per-walker kernel is the
                        longest -- `bound: "latency"`, `frac: null`, its shares of wave cycles, and the matrix-core kernel beside it;
  `roofline_step`       the whole step against the HBM roofline (rocprofv3 PMC bytes of the committed profile / ms_per_step);
  `step_kernels`        the three kernels of the step: duration, bound, PMC bytes, shares of wave cycles;
  `north_star_abel_map_kernel`  the fused profile -> Abel -> spline -> full S x S map kernel the north_star's ">= 60 % of
                        the HBM roofline" is about (jx_map_kernel_time), against the nominal and the measured copy roofline;
  `north_star_route`    the whole step of north_star's literal design (that kernel, then the rocFFT sequence) on the same walkers;
  `legacy_contracted_route`, `collapsed_route`   the contracted forms of round 4 and the collapsed (operator) route of the same library
                        on the same walkers: rate, stage times, difference to the default, error against the oracle sample;
  `host_pointer`        jx_eval with host pointers, one synchronisation per call, at 15 / 128 / 1024 walkers per call: the path emcee exercises;
  `device_sampler`      jx_sample (with the communicator on the context in the N > 1 rehearsal: `exchange_ms_per_half_step`);
  `gather_ms_per_step`  (N > 1, or JOXSZ_BENCH_FORCE_DIST=1) the all-gather's own duration on its stream (strict mode: the headline);
  `cpu_baseline`        the numpy/scipy oracle on this box's host cores (process pool, single process, per-stage ms);
  `other_configs`       strong-scaling rows of BASELINE configs[3] and configs[4] (this rank's shard).
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy there; measured here per run)
FP64_PEAK_TFLOPS = 78.6          # MI355X: fp64 vector = fp64 matrix = 256 CUs x 128 flop/clk x 2.4 GHz
FP64_FMA_MEASURED_TFLOPS = 62.8  # scripts/ubench/fma_sgpr.hip on this chip: v_fmac_f64 with register operands, 8 waves/SIMD (profiles/r03_fma_sgpr.log)


def pmc_file(S):
    """Latest committed rocprofv3 PMC measurement (scripts/measure_traffic.py -> profiles/*_pmc_traffic.json) of this map
    side, or None.  The counters cannot be read from inside the benchmark process."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')), reverse=True):
        try:
            j = json.load(open(f))
            if j.get('S', 512) == S:
                j['file'] = os.path.relpath(f, ROOT)
                return j
        except Exception:
            pass
    return None


def sq_counters_file():
    """The committed SQ counter pass of the newest round (profiles/rNN_sq_counters.csv): {'file', 'rows'} or {}."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_sq_counters.csv')))
    if not files:
        return {}
    try:
        lines = open(files[-1]).read().splitlines()
        cols = lines[0].split(',')
        rows = []
        for ln in lines[1:]:                                  # (kernel names carry commas of their own: the counters are the last fields)
            parts = ln.rsplit(',', len(cols) - 1)
            rows.append(dict(zip(cols, parts)))
        return {'file': os.path.relpath(files[-1], ROOT), 'rows': rows}
    except Exception:
        return {}


def pmc_kernel_entry(j, kernel):
    """The PMC entry of the full-size launches of `kernel` (base name, template arguments dropped): instances of the same kernel
    template that only ran on a handful of walkers (set-up, guard, taps) are in the file too; the full-size one has the largest
    grid (the most bytes where an older file carries no grid)."""
    if not j:
        return None
    best = None
    for name, d in j['kernels'].items():
        if name.split('<')[0].replace('void ', '').strip() == kernel:
            key = (d.get('grid_size', 0), d['total_bytes'])
            if best is None or key > best[0]:
                best = (key, d)
    return best[1] if best else None


def pmc_traffic(j, kernel, walkers_per_launch):
    """HBM bytes per launch of `kernel` from that file, rescaled to this launch size."""
    d = pmc_kernel_entry(j, kernel)
    return d['total_bytes'] / j.get('walkers_per_launch', 1024) * walkers_per_launch if d else None


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle): process pool as the reference runs it, one process, and where one call's time goes
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    pb, th = args
    from oracle import joxsz_oracle as orc
    return orc.log_posterior_batch(pb, th)


def host_cores():
    """Cores this process may actually use: the scheduler affinity mask capped by the
    cgroup CPU quota (a GPU box exposes all host CPUs but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    if os.environ.get('JOXSZ_CPU_CORES'):
        n = int(os.environ['JOXSZ_CPU_CORES'])
    return max(1, n)


def cpu_stage_ms(pb, theta, repeats=3):
    """Milliseconds of one oracle call by stage: the same scipy/numpy primitives the reference calls,
    timed one after the other on one parameter vector."""
    from scipy.interpolate import interp1d
    from scipy.signal import fftconvolve
    from scipy.fftpack import fft2, ifft2
    from oracle import joxsz_oracle as orc, pyabel_direct
    p = orc.pars_dict(pb, theta)
    acc = {k: 0.0 for k in ('profile', 'abel', 'spline_build', 'map', 'beam_conv', 'tf_filter', 'tail', 'xray_and_priors')}
    S = pb.d_mat.shape[0]
    for _ in range(repeats):
        t = time.perf_counter(); pp = orc.press_fun(p, pb.r_pp); acc['profile'] += time.perf_counter() - t
        t = time.perf_counter(); ab = pyabel_direct.direct_transform_forward(pp, pb.r_pp); acc['abel'] += time.perf_counter() - t
        y = pb.kpc_cm * pb.sigma_T / pb.m_e * ab
        t = time.perf_counter()
        f = interp1d(np.append(-pb.r_pp, pb.r_pp), np.append(y, y), 'cubic', bounds_error=False, fill_value=(0., 0.))
        acc['spline_build'] += time.perf_counter() - t
        t = time.perf_counter(); y2d = f(pb.d_mat); acc['map'] += time.perf_counter() - t
        t = time.perf_counter(); conv = fftconvolve(y2d, pb.beam_2d, 'same') * pb.step ** 2; acc['beam_conv'] += time.perf_counter() - t
        t = time.perf_counter(); row = np.real(ifft2(fft2(conv) * pb.filtering))[S // 2, S // 2:]; acc['tf_filter'] += time.perf_counter() - t
        t = time.perf_counter(); orc.sz_stages(pb, p); full = time.perf_counter() - t
        t = time.perf_counter(); orc.get_likelihood(pb, theta); tot = time.perf_counter() - t
        acc['tail'] += 0.0
        acc['xray_and_priors'] += max(0.0, tot - full)
        del row
    out = {k: 1e3 * v / repeats for k, v in acc.items()}
    out['tail'] = None                                   # (temperature profile, conversion, spline to the data radii, chi^2: inside the remainder)
    return out


def cpu_baseline(pb, thetas, target_s=12.0):
    """The oracle (numpy/scipy restatement of the reference path) (a) mapped over walkers with multiprocessing.Pool on all
    host cores, exactly the reference's parallelism (joxsz_main.py:203-206), (b) in one process, on bounded samples of the
    workload; plus the per-stage milliseconds of one call."""
    import multiprocessing as mp
    from oracle import joxsz_oracle as orc
    cores = host_cores()
    t = time.perf_counter()
    orc.log_posterior_batch(pb, thetas[:1])
    one = time.perf_counter() - t
    n1 = max(4, min(64, int(3.0 / max(one, 1e-3))))
    t = time.perf_counter()
    orc.log_posterior_batch(pb, thetas[:n1])
    single = n1 / (time.perf_counter() - t)
    stages = cpu_stage_ms(pb, thetas[0])
    per_core = max(1, int(target_s / max(one, 1e-3)))
    n = min(len(thetas), per_core * cores)
    n = max(cores, (n // cores) * cores)
    sample = thetas[:n]
    chunks = [(pb, c) for c in np.array_split(sample, cores)]
    ctxm = mp.get_context('fork')
    with ctxm.Pool(cores) as pool:
        pool.map(_cpu_worker, [(pb, thetas[:1])] * cores)          # warm the workers
        t = time.perf_counter()
        res = pool.map(_cpu_worker, chunks)
        dt = time.perf_counter() - t
    logp = np.concatenate(res)
    return dict(value=n / dt, unit='walker-likelihoods/s', cores=cores, kind='port',
                sample='%d walkers of the same workload, oracle/joxsz_oracle.py over multiprocessing.Pool(%d), %.1f s'
                       % (n, cores, dt),
                single_process={'value': single, 'unit': 'walker-likelihoods/s', 'cores': 1, 'sample': '%d walkers, one process' % n1},
                stage_ms_per_call=stages), sample, logp


# ---------------------------------------------------------------------------------------------------------------------
# launcher mode: this process starts the ranks and never touches a GPU
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv, timeout_s=3000.0, python=None, script=None):
    """Start n child processes (RANK 0..n-1, one GPU each), wait for them, return (exit code, rank 0's last stdout line).
    Children that are still running when one fails or the timeout expires are terminated by PID."""
    import random
    python = python or sys.executable
    script = script or os.path.abspath(__file__)
    port = int(os.environ.get('MASTER_PORT', 0)) or random.randint(20000, 45000)
    tag = 'bench_%d_%d' % (os.getpid(), int(time.time() * 1e3))
    t_launch = repr(time.time())                          # one launch start for all ranks (a reader rejects an id file older than it)
    import tempfile
    out0 = tempfile.TemporaryFile(mode='w+')              # rank 0's stdout: a file, so that no amount of output can block it on a full pipe
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), JOXSZ_RDZV_TAG=tag, JOXSZ_RDZV_T0=t_launch,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([python, script] + argv, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=None))
    t0 = time.time()
    rc = 0
    pending = set(range(n))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
        if rc != 0 or time.time() - t0 > timeout_s:
            if pending and rc == 0:
                rc = 124
            for r in pending:                             # exact PIDs only
                procs[r].terminate()
            t1 = time.time()
            while any(procs[r].poll() is None for r in pending) and time.time() - t1 < 10.0:
                time.sleep(0.1)
            for r in pending:
                if procs[r].poll() is None:
                    procs[r].kill()
            break
        time.sleep(0.05)
    out0.seek(0)
    out = out0.read()
    out0.close()
    lines = [l for l in out.strip().splitlines() if l.strip()]
    return rc, (lines[-1] if lines else '')


def agree_on_side_times(comm, dts):
    """The one collective of the side measurements: element-wise maximum over the ranks of this rank's times per step (+inf where
    the measurement failed here).  Called OUTSIDE every exception handler, after rank-local measurements that cannot raise:
    every rank reaches it whatever happened before, so no rank is left waiting in it (a collective inside
    try/except is a deadlock when one rank throws).  A rank that dies outright takes the launch down through the launcher."""
    dts = [float(d) if d is not None and np.isfinite(d) else np.inf for d in dts]
    if comm is None:
        return dts
    return [float(v) for v in comm.max_over_ranks(dts)]


def time_steps(ctx, th_ptr, W, lp_ptr, steps, warmup):
    for _ in range(warmup):
        ctx.eval_device(th_ptr, W, lp_ptr)
    ctx.sync()
    t = time.perf_counter()
    for _ in range(steps):
        ctx.eval_device(th_ptr, W, lp_ptr)
    ctx.sync()
    return (time.perf_counter() - t) / steps


def with_env(env, fn):
    """fn() with os.environ temporarily extended (jx_finalize reads its switches from the environment of the calling process)."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# what the kernels of a step are called and what bounds them, by form of the SZ side (jx_get_conv_layout)
STEP_KERNELS = {
    'exact': [('prep_ms', 'jx_walker2_kernel', 'per-walker work as two lean roles in one launch: parameters, priors, grid pass (pressure, mass veto, T_SZ), h(0), conversion factors | X-ray profiles, count rates, projection, Cash sum',
               'latency: a chain of dependent phases and fp64 exp/log chains; no flop or byte count prices it (its share of wave cycles is listed)'),
              ('abel_map_ms', 'jx_ordrow_kernel', 'ordinate product y = y_scale A pp (forward Abel transform + Compton-y scale) and each column-tile pair\'s share of the row product out = Wy y, fp64 matrix cores',
               'mfma (v_mfma_f64_16x16x4)'),
              ('tail_ms', 'jx_rowsum_tail_kernel', 'partial rows added in pair order, conversion, not-a-knot spline to the data radii, chi^2, total (+ the stretch move\'s acceptance inside jx_sample)',
               'latency: one trip to memory for its operands, then three short dependent phases')],
    'lowrank': [('prep_ms', 'jx_prep_kernel', 'per-walker kernel', 'latency'), ('abel_map_ms', 'jx_abel_gemm_kernel', 'spline-array product', 'mfma'),
                ('beam_fft_ms', 'jx_rowmix_kernel', 'stage 1: map samples evaluated and mixed per column', 'valu'),
                ('tf_fft_ms', 'jx_opgemm_kernel', 'stage 2: one fp64 matrix-core product', 'mfma'), ('tail_ms', 'jx_tail_row_kernel', 'tail', 'latency')],
    'full': [('prep_ms', 'jx_prep_kernel', 'per-walker kernel', 'latency'), ('abel_map_ms', 'jx_abel_gemm_kernel', 'spline-array product', 'mfma'),
             ('tf_fft_ms', 'jx_opgemm_kernel', 'full form: samples evaluated by the lanes that feed the matrix cores', 'mfma'), ('tail_ms', 'jx_tail_row_kernel', 'tail', 'latency')],
}
TIMING_MODE_OF_STAGE = {'abel_map_ms': 2, 'prep_ms': 3, 'tail_ms': 4}      # jx_timing_enable: the two events around ONE kernel of the step (exact form)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--walkers', type=int, default=1024, help='walkers per GPU')
    ap.add_argument('--S', type=int, default=512)
    ap.add_argument('--N', type=int, default=500)
    ap.add_argument('--sz-only', action='store_true')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--regions', type=int, default=0, help='timed regions of --steps steps each (0: at least 25, and as many as 50 ms of timed work take)')
    ap.add_argument('--no-full-map', action='store_true', help='skip the side measurement of the full-map Abel kernel, of the stream rates and of the literal route')
    ap.add_argument('--route', choices=('map', 'operator'), default='map',
                    help="'map': profile, Abel transform and row operator per walker (the BASELINE metric); 'operator': the collapsed route (jx_set_route)")
    ap.add_argument('--fwhm', type=float, default=18.5, help='beam FWHM in arcsec (B = 2*floor(3*fwhm/step)+1)')
    ap.add_argument('--dtype', choices=('f64', 'f32', 'f32c'), default='f64',
                    help="'f64': the reference's arithmetic (the metric); 'f32' / 'f32c': the fp32 variants (contracted forms of round 4)")
    ap.add_argument('--no-f32', action='store_true', help='skip the side measurement of the fp32 variants')
    ap.add_argument('--no-host-pointer', action='store_true', help='skip the side measurements of jx_eval with host pointers and of jx_sample (profiling runs: keeps the trace to the timed steps)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the strong-scaling rows of BASELINE configs[3] and configs[4]')
    ap.add_argument('--no-other-routes', action='store_true', help='skip the collapsed and legacy-contracted rows')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # no launcher: be one.  Nothing above or below this line in this process touches a GPU.
        rc, line = spawn_ranks(args.gpus, sys.argv[1:])
        if line:
            print(line)
        raise SystemExit(rc if rc else (0 if line else 1))

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    args.gpus = world

    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=args.S, N=args.N, seed=0, sz_only=args.sz_only, fwhm=args.fwhm)
    W = args.walkers

    # ---- CPU baseline first: it forks, which must happen before HIP is initialised ----
    cpu = None
    cpu_sample = cpu_logp = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu:
        th_cpu = datasets.walker_ball(pb, 16384, spread=0.02, seed=11)
        # the data the walkers are scored against do not change the cost; use the placeholder data
        cpu, cpu_sample, cpu_logp = cpu_baseline(pb, th_cpu, args.cpu_seconds)

    multi = args.gpus > 1 or bool(os.environ.get('JOXSZ_BENCH_FORCE_DIST'))      # the env switch rehearses the N>1 plumbing at N=1
    from joxsz_amd.posterior import JoxszPosterior
    post = JoxszPosterior(pb, device=local_rank)
    ctx = post.ctx

    def oracle_parity(c):
        """max relative error of the CPU sample's log-posteriors on context c (None without a sample); refuses different rejections"""
        if cpu_sample is None:
            return None
        got = c.eval(cpu_sample)
        fin = np.isfinite(cpu_logp)
        if not np.array_equal(np.isfinite(got), fin):
            raise SystemExit('bench: GPU/oracle disagree on which walkers are rejected')
        return float(np.max(np.abs(got[fin] - cpu_logp[fin]) / np.abs(cpu_logp[fin]))) if fin.any() else 0.0

    # parity of the CPU sample on the very same problem tensors: the bench refuses to report a number for results that differ
    if args.route != 'map':
        ctx.set_route(args.route)
    parity = oracle_parity(ctx)
    if parity is not None and parity > 1e-6:
        raise SystemExit('bench: parity %.3e exceeds 1e-6' % parity)
    # the other routes of the same library on the same sample, while the placeholder data are in place (rows of the line, further down)
    parity_other = {}
    if rank == 0 and args.route == 'map' and not args.no_other_routes and cpu_sample is not None:
        for name, env, route in (('collapsed_route', {}, 'operator'), ('legacy_contracted_route', {'JOXSZ_MIX_FORM': 'legacy', 'JOXSZ_QUIET': '1'}, None)):
            try:
                po = with_env(env, lambda: JoxszPosterior(pb, device=local_rank))
                if route:
                    po.ctx.set_route(route)
                parity_other[name] = oracle_parity(po.ctx)
                po.close()
            except BaseException as exc:                     # (a row of the line, not the headline)
                parity_other[name] = 'error: %s' % exc

    # ---- synthetic observations from the model itself at the fiducial vector, then the walker ball ----
    t0 = datasets.fiducial_theta(pb)
    t0w = np.repeat(t0[None, :], 8, axis=0)
    bright = ctx.eval_stage(t0w, 'bright')[0]
    xprofs = None if pb.sz_only else ctx.eval_stage(t0w, 'xprofs')[0]
    post.close()
    datasets.fill_data(pb, bright, xprofs, seed=0)
    post = JoxszPosterior(pb, device=local_rank, dtype=args.dtype)
    ctx = post.ctx
    if args.route != 'map':
        ctx.set_route(args.route)
    cand = datasets.walker_ball(pb, 4 * W, spread=0.02, seed=100 + rank)
    lp = ctx.eval(cand)
    good = cand[np.isfinite(lp)]
    if len(good) < W:
        raise SystemExit('bench: only %d finite walkers of %d' % (len(good), len(cand)))
    theta = np.ascontiguousarray(good[:W])

    # ---- device-resident inputs ----
    th_ptr = ctx.dev_alloc(theta.nbytes)
    lp_ptr = ctx.dev_alloc(8 * W)
    ctx.h2d(th_ptr, theta)
    comm = None
    lp_ptrs, all_ptrs = [lp_ptr], [None]
    # N > 1: the headline runs the gather STRICTLY in order on the compute stream -- what a sampler needs that proposes from the gathered
    # values; the overlapped mode (second stream, two output buffers) is probed in the warm-up and reported beside it.
    overlap = multi and bool(os.environ.get('JOXSZ_BENCH_OVERLAP_GATHER'))
    if multi:
        # RCCL through the library's own C-ABI (jx_comm_*), no torch in this process.
        from joxsz_amd.dist import RcclGather
        comm = RcclGather(ctx, rank=rank, world=world, overlap=overlap)
        lp_ptrs = [lp_ptr, ctx.dev_alloc(8 * W)]
        all_ptrs = [ctx.dev_alloc(8 * W * world), ctx.dev_alloc(8 * W * world)]
    nstep = [0]

    def step():
        k = nstep[0] % len(lp_ptrs)
        ctx.eval_device(th_ptr, W, lp_ptrs[k])
        if comm is not None:
            comm.all_gather(lp_ptrs[k], all_ptrs[k], W)
        nstep[0] += 1

    def fence():
        if comm is not None:
            comm.barrier()
        ctx.sync()                                           # (both streams of the context)

    for _ in range(args.warmup):
        step()
    fence()
    gather_probe = None
    if comm is not None and not os.environ.get('JOXSZ_BENCH_STRICT_GATHER') and not os.environ.get('JOXSZ_BENCH_OVERLAP_GATHER'):
        # still inside the warm-up: the same short run in both gather modes (reported; the timed regions stay strict)
        gather_probe = {}
        nprobe = max(10, min(50, args.steps))
        for mode in (True, False):
            ctx.comm_set_overlap(mode)
            for _ in range(3):
                step()
            fence()
            t_probe = time.perf_counter()
            for _ in range(nprobe):
                step()
            fence()
            gather_probe['overlapped' if mode else 'strict'] = 1e3 * comm.max_over_ranks(time.perf_counter() - t_probe) / nprobe
        ctx.comm_set_overlap(False)
        fence()
    # ---- which kernel of the step is the longest: one stage pass inside the warm-up (HIP events behind every kernel) ----
    lay = ctx.conv_layout or {}
    form = lay.get('form')
    mixed = ctx.conv == 'custom' and args.route == 'map'
    ctx.timing_enable(1)
    ctx.timing_reset()
    fence()
    for _ in range(max(10, min(50, args.steps))):
        step()
    fence()
    tm_w = ctx.timing()
    ctx.timing_enable(False)
    kernels = STEP_KERNELS.get(form, []) if mixed else []
    dom = max(kernels, key=lambda k: tm_w[k[0]]) if kernels else None
    # Timed regions: `--steps` steps each, bracketed by barrier + synchronisation on both sides, at least 25 of them and at least 50 ms in
    # all; the MEDIAN region is the figure (a single 2 ms region has no spread estimate).  Inside them: the two HIP
    # events around the longest kernel of the step only (what `roofline` prices); events behind every kernel cost a few per cent of a
    # step, so the stage breakdown comes from one more pass right after.
    p1_mode = TIMING_MODE_OF_STAGE.get(dom[0], 2) if (dom and form == 'exact') else (2 if mixed else 1)
    ctx.timing_enable(False)
    fence()
    est = max(1e-6, tm_w['total_ms'] / max(1, tm_w['launches']) * 1e-3 * (W / max(1.0, tm_w['walkers'] / max(1, tm_w['launches']))) * args.steps)
    if comm is not None:
        est = comm.max_over_ranks(est)                       # (every rank runs the same number of regions: each carries a barrier)
    nreg = args.regions if args.regions > 0 else int(min(2000, max(25, np.ceil(0.05 / est))))
    regions = []
    for _ in range(nreg):
        fence()
        t_start = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        regions.append(time.perf_counter() - t_start)
    if comm is not None:
        regions = [float(v) for v in comm.max_over_ranks(regions)]
    # the same regions once more with the two HIP events around the longest kernel of the step (a pair of events between dependent
    # kernels costs the step 3-6 us of command-processor work: the headline regions above carry none)
    ctx.timing_enable(p1_mode)
    ctx.timing_reset()
    regions_ev = []
    for _ in range(max(5, min(nreg, 25))):
        fence()
        t_start = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        regions_ev.append(time.perf_counter() - t_start)
    if comm is not None:
        regions_ev = [float(v) for v in comm.max_over_ranks(regions_ev)]
    tm_timed = ctx.timing()
    ctx.timing_enable(1)
    ctx.timing_reset()
    fence()
    for _ in range(args.steps):
        step()
    fence()
    tm = ctx.timing()
    gather_ms = None
    if comm is not None:
        gms, gn = ctx.comm_gather_time()                     # the all-gathers' own durations (events on their stream) of the stage pass
        gather_ms = gms / max(1, gn)
    ctx.timing_enable(False)
    elapsed = float(np.median(regions))

    last = (nstep[0] - 1) % len(lp_ptrs)
    if comm is not None:
        final = np.empty(W * world)
        ctx.d2h(final, all_ptrs[last])
    else:
        final = np.empty(W)
        ctx.d2h(final, lp_ptr)
    if not np.all(np.isfinite(final)):
        raise SystemExit('bench: non-finite log-probabilities in the timed batch')
    lp_own = final[rank * W:(rank + 1) * W] if comm is not None else final
    # ---- every figure of the headline is fixed from here on; nothing below can change or lose it.  The side measurements run
    #      rank-locally inside try/except WITHOUT collectives; the ranks meet once more, in one all-reduce outside any
    #      exception handler, to agree on the strong-scaling rows (a rank that failed contributes +inf).

    # the path emcee calls (joxsz_main.py:206, vectorize=True): host theta in, host log-probabilities out, one sync per call --
    # at the reference's own ensemble (30 walkers = 15 per half step, joxsz_main.py:44), at 128 and at the launch size
    host_ptr = None
    if rank == 0 and not args.no_host_pointer:
        try:
            host_ptr = {'unit': 'walker-likelihoods/s', 'calls': {},
                        'note': 'jx_eval: host parameter vectors in (%d B per walker over PCIe), host log-probabilities out, one stream synchronisation per call -- '
                                'what emcee.EnsembleSampler(..., vectorize=True) exercises (joxsz_main.py:206); 15 walkers per call = the half steps of the '
                                'reference\'s default ensemble of 30 (joxsz_main.py:44)' % (8 * theta.shape[1])}
            for nw in sorted({15, 128, W}):
                nw = min(nw, W)
                for _ in range(3):
                    ctx.eval(theta[:nw])
                kk = max(20, min(args.steps, 100))
                t = time.perf_counter()
                for _ in range(kk):
                    got_h = ctx.eval(theta[:nw])
                dt = (time.perf_counter() - t) / kk
                host_ptr['calls'][str(nw)] = {'walkers_per_call': nw, 'ms_per_call': 1e3 * dt, 'value': nw / dt, 'calls': kk,
                                              'max_abs_diff_vs_device_resident': float(np.max(np.abs(got_h - lp_own[:nw])))}
            host_ptr['value'] = host_ptr['calls'][str(W)]['value']
            host_ptr['ms_per_call'] = host_ptr['calls'][str(W)]['ms_per_call']
            host_ptr['max_abs_diff_vs_device_resident'] = max(c['max_abs_diff_vs_device_resident'] for c in host_ptr['calls'].values())
        except Exception as exc:
            host_ptr = {'error': str(exc)}

    # the caller of the path: the device-resident stretch-move loop on an ensemble of 2 W walkers -- each half step one
    # evaluation of W proposals, drawn in the per-walker kernel and accepted or rejected in the tail
    samp = None
    # (with a communicator jx_sample is a collective of its own: rehearsed at world size 1; on more ranks only with JOXSZ_BENCH_SAMPLER_DIST=1,
    #  every rank calling it -- a side measurement must never be able to take the scaling line down)
    samp_dist = comm is not None and (world == 1 or bool(os.environ.get('JOXSZ_BENCH_SAMPLER_DIST')))
    if (rank == 0 or samp_dist) and (comm is None or samp_dist) and args.route == 'map' and not args.no_host_pointer:
        try:
            x0 = np.ascontiguousarray(np.concatenate((theta, good[W:2 * W] if len(good) >= 2 * W else theta[::-1] * (1 + 1e-9))))
            ns1, ns = 20, 120
            ctx.sample(x0, ns)                                     # (warm-up at the longer length: the call's device buffers are sized once)
            t1 = t2 = float('inf')
            for _ in range(3):                                     # (best of three: the host side -- fresh numpy pages for the returned chain -- adds milliseconds now and then)
                t = time.perf_counter()
                ctx.sample(x0, ns1, seed=3)
                t1 = min(t1, time.perf_counter() - t)
                t = time.perf_counter()
                _, lps_s, nacc_s = ctx.sample(x0, ns, seed=3)      # (two run lengths: the slope is the step, the intercept the call's set-up, initial evaluation and copy back)
                t2 = min(t2, time.perf_counter() - t)
            dt = (t2 - t1) / (ns - ns1)
            samp = {'value': 2 * W / dt, 'unit': 'walker-updates/s', 'ms_per_step': 1e3 * dt, 'walkers': 2 * W, 'steps': ns, 'ms_per_call_besides_the_steps': 1e3 * (t1 - ns1 * dt),
                    'acceptance': float(nacc_s.sum()) / (2 * W * ns), 'finite': bool(np.isfinite(lps_s).all()),
                    'exchange_ms_per_half_step': None,
                    'note': 'jx_sample: proposals, evaluation, accept/reject and chain on the device, one copy back at the end; a step = two half steps of %d walkers' % W}
            if comm is not None:
                # (N > 1 or the rehearsal at N = 1: the same run with the communicator on the context shards each half step over the ranks and
                #  all-gathers positions and log-posteriors in place -- two collectives per half step; the difference to the run above is their cost)
                samp['exchange_ms_per_half_step'] = 0.5 * samp['ms_per_step'] - 1e3 * (elapsed / args.steps)
                samp['exchange_note'] = ('with a communicator on the context jx_sample moves this rank\'s share of each half step and exchanges positions and log-posteriors by two '
                                         'in-place RCCL all-gathers: half a sampler step (one half step) minus one evaluation step of the same %d walkers (which carries the log-probability gather)' % W)
        except Exception as exc:
            samp = {'error': str(exc)}

    # the fp32 variants on the same walkers (BASELINE configs[4]'s tolerance sweep): beside the f64 metric, never instead of it
    f32 = None
    if rank == 0 and comm is None and args.route == 'map' and args.dtype == 'f64' and not args.no_f32:
        f32 = {}
        notes = {'f32': 'jx_config.dtype = 1 (contracted form of round 4, JOXSZ_MIX_FORM=legacy semantics): spline arrays (y_k, M_k) rounded to fp32 once; sums, '
                        'matrix-core product, tail and everything per-walker in fp64',
                 'f32c': 'jx_config.dtype = 2 (contracted form of round 4): fp32 arithmetic -- stage 1 in packed fp32 FMAs, stage 2 on v_mfma_f32_16x16x4_f32, K slices added '
                         'in fp64 by the tail; everything per-walker in fp64.  |delta chi^2/2| is 1e-5 ... 1e-4 absolute (inside north_star\'s RELATIVE 1e-6 because the '
                         'log-posterior is ~1e4): outside the 1e-6 absolute bar the fp64 tests hold'}
        ch64 = ctx.eval_stage(theta[:256], 'chisq')
        for dt_name in ('f32', 'f32c'):
            try:
                p3 = with_env({'JOXSZ_QUIET': '1'}, lambda: JoxszPosterior(pb, device=local_rank, dtype=dt_name))
                c3 = p3.ctx
                t3, l3 = c3.dev_alloc(theta.nbytes), c3.dev_alloc(8 * W)
                c3.h2d(t3, theta)
                dt = time_steps(c3, t3, W, l3, max(20, args.steps // 4), 3)
                lp32 = np.empty(W)
                c3.d2h(lp32, l3)
                ch32 = c3.eval_stage(theta[:256], 'chisq')
                rel = np.abs(lp32 - final) / np.abs(final)
                f32[dt_name] = {'dtype': dt_name, 'form': (c3.conv_layout or {}).get('form'), 'value': W / dt, 'unit': 'walker-likelihoods/s', 'ms_per_step': 1e3 * dt,
                                'speedup_vs_f64': (elapsed / args.steps) / dt,
                                'rel_dlogp_vs_f64': {'max': float(rel.max()), 'median': float(np.median(rel))},
                                'abs_dchisq_half_vs_f64': {'max': float(np.abs(ch32 - ch64).max() / 2), 'median': float(np.median(np.abs(ch32 - ch64)) / 2)},
                                'note': notes[dt_name]}
                p3.close()
            except Exception as exc:
                f32[dt_name] = {'dtype': dt_name, 'error': str(exc)}

    # the kernel north_star's ">= 60 % of the HBM roofline in the Abel+map kernel" is about: profile -> Abel -> spline -> full
    # S x S map, and the measured copy bandwidth of this card as the practical roofline beside the nominal one
    full_map = None
    copy_gbs = None
    streams = None
    if rank == 0 and comm is None and args.route == 'map' and not args.no_full_map:
        try:
            streams = {k: ctx.stream_bandwidth(k, 2 << 30, 10) for k in ('read', 'write', 'copy')}
            copy_gbs = streams['copy']
            ms = ctx.map_kernel_time(th_ptr, W, 10)
            b = W * args.S * args.S * 8.0
            full_map = {'kernel': 'jx_abel_map_sym_kernel (profile -> Abel -> spline -> full S x S map; jx_map_kernel_time)',
                        'launch_ms': ms, 'bytes_per_launch': b, 'achieved_GBps': b / (ms * 1e-3) / 1e9,
                        'frac_of_hbm_peak': b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        'frac_of_measured_write_stream': b / (ms * 1e-3) / 1e9 / streams['write'],
                        'frac_of_measured_copy_roofline': b / (ms * 1e-3) / 1e9 / copy_gbs,
                        'note': 'a store stream: S^2 * 8 B written per walker (SURVEY 8(d)), priced against the nominal peak, against what a plain '
                                'write stream of the same 2 GiB gets on this card, and against a copy (bytes read + written)'}
        except Exception as exc:
            full_map = {'error': str(exc)}

    def side_route(make, label, note):
        """one more context of the same library on the same walkers: rate, stage times, difference to the default route"""
        pr = make()
        cr = pr.ctx
        tr_, lr_ = cr.dev_alloc(theta.nbytes), cr.dev_alloc(8 * W)
        cr.h2d(tr_, theta)
        nst = 10 if label == 'literal' else max(20, min(args.steps, 100))
        dts_ = [time_steps(cr, tr_, W, lr_, nst, 3) for _ in range(3 if label == 'literal' else 5)]
        dt = float(np.median(dts_))
        cr.timing_enable(1); cr.timing_reset()
        for _ in range(3):
            cr.eval_device(tr_, W, lr_)
        tmr = cr.timing()
        cr.timing_enable(False)
        lpr = np.empty(W)
        cr.d2h(lpr, lr_)
        out_ = {'value': W / dt, 'unit': 'walker-likelihoods/s', 'ms_per_step': 1e3 * dt, 'walkers_per_launch': min(W, cr.chunk) if label != 'collapsed' else W,
                'form': (cr.conv_layout or {}).get('form') if cr.conv == 'custom' else None,
                'stage_ms_per_step': {k: tmr[k] / 3 for k in ('prep_ms', 'abel_map_ms', 'beam_fft_ms', 'tf_fft_ms', 'tail_ms')},
                'max_rel_diff_vs_default_route': float(np.max(np.abs(lpr - lp_own) / np.abs(lp_own))),
                'ms_per_step_of_the_default_route_over_this': None, 'note': note}
        return pr, cr, out_

    # north_star's literal design as a whole step: fused profile -> Abel -> map kernel, then the rocFFT sequence for the beam
    # convolution and the transfer function (joxsz_funcs.py:460-467 executed step by step), same walkers, same problem
    ns_route = None
    if rank == 0 and comm is None and args.route == 'map' and not args.no_full_map and ctx.conv == 'custom':
        try:
            pr, cr, ns_route = side_route(lambda: JoxszPosterior(pb, device=local_rank, conv='rocfft', max_batch=W), 'literal',
                                          'the same library with conv = rocfft: every map and its spectra go through HBM (map written; row spectra written, read and written by the column '
                                          'pass, read by the fused inverse-row / window / forward-row pass; window spectra written and read once)')
            fi = cr.fft_info()
            ns_route['transforms'] = fi
            if fi.get('columns') == 'custom' and fi.get('rows') == 'custom':
                ns_route['route'] = ('jx_abel_map_sym_kernel -> jx_fft_rows_fwd_kernel -> jx_fft_beam_cols_kernel (forward, x beam spectrum, inverse) -> jx_fft_rows_inv_tf_kernel '
                                     '(inverse rows, S x S window, forward rows) -> jx_fft_tf_cols_kernel (forward, x transfer function, column sum) -> jx_tail_kernel')
            else:
                ns_route['route'] = 'jx_abel_map_sym_kernel -> rocFFT R2C -> jx_beam_mul_kernel -> rocFFT C2R -> rocFFT R2C of the S x S window -> jx_tail_kernel'
            ns_route['fft_pad'] = cr.fft_pad
            ns_route['survey_8d_bytes_per_step'] = 2.0 * W * args.S * args.S * 8.0
            ns_route['frac_of_hbm_peak_on_survey_8d_bytes'] = ns_route['survey_8d_bytes_per_step'] / (ns_route['ms_per_step'] * 1e-3) / 1e9 / HBM_PEAK_GBS
            # HBM bytes of the route from the latest committed PMC pass over it (scripts/measure_traffic.py <tag> literal), same shape and launch size only
            for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic_literal.json')), reverse=True):
                pj = json.load(open(f))
                if pj.get('S') == args.S and pj.get('N') == args.N and pj.get('walkers_per_launch') == W:
                    ns_route['pmc_traffic'] = {'file': os.path.relpath(f, ROOT), 'total_bytes': pj['total_bytes'], 'over_survey_8d_bytes': pj['total_over_survey_8d_bytes'],
                                               'per_kernel_bytes': {k: v['total_bytes'] for k, v in pj['kernels'].items()},
                                               'frac_of_hbm_peak_on_pmc_bytes': pj['total_bytes'] / (ns_route['ms_per_step'] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                    break
            pr.close()
        except Exception as exc:
            ns_route = {'error': str(exc)}
    # literal | contracted (round 4) | collapsed beside the default: the same walkers through the other routes of the same library
    other_routes = {}
    if rank == 0 and comm is None and args.route == 'map' and not args.no_other_routes and ctx.conv == 'custom':
        for name, make, note in (
                ('collapsed_route', lambda: JoxszPosterior(pb, device=local_rank, route='operator'),
                 'jx_set_route(JX_ROUTE_OPERATOR): row = G pp with one constant nrow x N matrix built by sending the N unit profiles through the default route\'s own '
                 '(exact) kernels -- profile, then ONE matrix-vector product per walker; the Abel transform is folded into G and runs N times at set-up, never per walker'),
                ('legacy_contracted_route', lambda: with_env({'JOXSZ_MIX_FORM': 'legacy', 'JOXSZ_QUIET': '1'}, lambda: JoxszPosterior(pb, device=local_rank)),
                 'JOXSZ_MIX_FORM=legacy: the contracted forms of rounds 3-4 (rank-16 low-rank tables, a sub-grid of map samples, a radial sub-grid of the profile, '
                 'measured by the truncation guard) that were the default until round 4; kept for one round')):
            try:
                pr, cr, row = side_route(make, 'collapsed' if name == 'collapsed_route' else 'legacy', note)
                row['max_rel_err_vs_oracle_sample'] = parity_other.get(name)
                if name == 'legacy_contracted_route':
                    row['truncation'] = {k: v for k, v in cr.truncation.items() if k != 'warning'}
                    row['sampling'] = {k: v for k, v in cr.sampling.items() if k != 'rows'}
                    row['radial_sampling'] = {k: v for k, v in cr.radial_sampling.items() if k != 'rows'}
                pr.close()
                other_routes[name] = row
            except Exception as exc:
                other_routes[name] = {'error': str(exc)}

    # strong-scaling rows of the other BASELINE configs: this rank's shard of configs[3] (4096 walkers, 512^2) and of
    # configs[4] (8192 walkers, 1024^2 / 1000-pt, fp64 and fp32), a few steps each, outside the timed region
    other = None
    want_other = args.route == 'map' and not args.no_other_configs and (args.S, args.N, pb.sz_only) == (512, 500, False)
    dts = [np.inf, np.inf, np.inf]                            # this rank's time per step: configs[3], configs[4] f64, configs[4] f32c
    if want_other:
        other = {}
        try:
            from joxsz_amd.dist import shard_bounds
            lo, hi = shard_bounds(4096, world, rank)
            n3 = hi - lo
            th3 = np.ascontiguousarray(np.resize(theta, (n3, theta.shape[1])))
            # (a context sized for the shard: one launch sequence for all of it instead of one per 1024 walkers)
            p3c = JoxszPosterior(pb, device=local_rank, max_batch=n3) if n3 > ctx.chunk else None
            c3 = p3c.ctx if p3c is not None else ctx
            p3t, l3t = c3.dev_alloc(th3.nbytes), c3.dev_alloc(8 * n3)
            c3.h2d(p3t, th3)
            dts[0] = float(np.median([time_steps(c3, p3t, n3, l3t, 10, 2) for _ in range(5)]))
            other['configs[3]'] = {'workload': '4096 walkers, 512x512 map, 500-pt grid, joint; %d walkers on this rank' % n3,
                                   'unit': 'walker-likelihoods/s', 'scaling': 'strong', 'dtype': 'f64', 'walkers_per_launch': c3.chunk}
            if p3c is not None:
                p3c.close()
            lo, hi = shard_bounds(8192, world, rank)
            n4 = hi - lo
            pb4 = datasets.synthetic_problem(S=1024, N=1000, seed=0)
            for i4, dt_name in enumerate(('f64', 'f32c')):
                p4 = with_env({'JOXSZ_QUIET': '1'}, lambda: JoxszPosterior(pb4, device=local_rank, dtype=dt_name, max_batch=n4))
                c4 = p4.ctx
                cand4 = datasets.walker_ball(pb4, 256, spread=0.02, seed=5)
                ok4 = cand4[np.isfinite(c4.eval(cand4))]
                th4 = np.ascontiguousarray(np.resize(ok4, (n4, ok4.shape[1])))
                a4, b4 = c4.dev_alloc(th4.nbytes), c4.dev_alloc(8 * n4)
                c4.h2d(a4, th4)
                dts[1 + i4] = float(np.median([time_steps(c4, a4, n4, b4, 5, 1) for _ in range(3)]))
                other['configs[4] ' + dt_name] = {'workload': '8192 walkers, 1024x1024 map, 1000-pt grid, joint; %d walkers on this rank' % n4,
                                                  'unit': 'walker-likelihoods/s', 'scaling': 'strong',
                                                  'dtype': dt_name, 'conv_layout': c4.conv_layout, 'walkers_per_launch': c4.chunk}
                p4.close()
        except Exception as exc:
            other['error'] = '%s: %s' % (type(exc).__name__, exc)
    if want_other:
        dts = agree_on_side_times(comm, dts)
    if other is not None:
        for key, tot, dt in (('configs[3]', 4096, dts[0]), ('configs[4] f64', 8192, dts[1]), ('configs[4] f32c', 8192, dts[2])):
            if key in other:
                if np.isfinite(dt):
                    other[key].update(value=tot / dt, ms_per_step=1e3 * dt)
                else:
                    other[key] = {'error': 'not measured on every rank'}

    if rank == 0:
        S = args.S
        launches = max(1, tm['launches'])
        walkers_per_launch = tm['walkers'] / launches
        stage_keys = ('prep_ms', 'abel_map_ms', 'beam_fft_ms', 'tf_fft_ms', 'tail_ms')
        stage_ms = {k: tm[k] / launches for k in stage_keys}
        try:
            ev_null_ms = ctx.event_bracket_time(64)
        except Exception:
            ev_null_ms = None
        value = W * world * args.steps / elapsed
        ms_step = 1e3 * elapsed / args.steps
        pj = pmc_file(S)
        nrow = ctx.nrow
        prn = ctx.output_pruning
        sq = sq_counters_file()

        def wave_shares(kernel):
            row = next((r for r in sq.get('rows', []) if kernel in r['kernel'] and float(r['SQ_WAVE_CYCLES']) > 1e6), None)
            if not row:
                return None
            wc = float(row['SQ_WAVE_CYCLES'])
            return {'source': sq['file'], 'waiting': float(row['SQ_WAIT_ANY']) / wc, 'issue_stalled': float(row['SQ_WAIT_INST_ANY']) / wc,
                    'issuing': float(row['SQ_ACTIVE_INST_ANY']) / wc, 'issuing_valu': float(row['SQ_ACTIVE_INST_VALU']) / wc}

        step_kernels = None
        roof = None
        if mixed and kernels:
            tot_stage = sum(stage_ms[k[0]] for k in kernels)
            step_kernels = [{'kernel': k[1], 'what': k[2], 'bound': k[3], 'ms_hip_events_stage_pass': stage_ms[k[0]], 'share_of_stage_sum': stage_ms[k[0]] / max(1e-12, tot_stage),
                             'traffic_bytes_per_launch': pmc_traffic(pj, k[1], walkers_per_launch), 'share_of_wave_cycles': wave_shares(k[1])} for k in kernels]
            key, kname, kwhat, kbound = dom
            k_ms = tm_timed[key] / max(1, tm_timed['launches'])
            common = {'kernel': kname, 'what': kwhat, 'launch_ms': k_ms,
                      'launch_ms_source': 'HIP events around this kernel alone over %d timed regions of `steps` steps run right behind the headline regions (jx_timing_enable(%d)); the kernel '
                                          'picked is the longest of the stage pass run in the warm-up' % (len(regions_ev), p1_mode),
                      'ms_per_step_of_the_regions_with_these_events': 1e3 * float(np.median(regions_ev)) / args.steps,
                      'launch_ms_of_an_empty_kernel': ev_null_ms,
                      'launch_ms_note': 'a pair of HIP events around one kernel of a dependent chain also spans the command processor\'s hand-over in front of and behind it -- '
                                        'launch_ms_of_an_empty_kernel is what the same pair reads around a kernel that does nothing (jx_event_bracket_time); rocprofv3\'s kernel trace '
                                        '(profiles/*_kernel_stats.csv) counts the kernel alone',
                      'share_of_step': k_ms / max(1e-12, ms_step * walkers_per_launch / W),
                      'traffic': pmc_traffic(pj, kname, walkers_per_launch), 'traffic_source': (pj or {}).get('file'), 'traffic_measured_in_this_run': False,
                      'share_of_wave_cycles': wave_shares(kname),
                      'survey_8d_bytes_per_launch': walkers_per_launch * S * S * 8.0,
                      'survey_8d_note': 'SURVEY 8(d) prices the Abel+map kernel at S^2 * 8 B written per walker; this step never writes a map (see north_star_abel_map_kernel and '
                                        'north_star_route for the kernels that do): its own work is counted below'}
            if form == 'exact':
                nk = lay['rank']                                          # ordinates in use
                nout = prn['outputs_computed']
                macs = sum(args.N - k for k in range(nk)) + nout * nk      # triangular Abel product for the ordinates in use + the row product
                fl = 2.0 * macs * walkers_per_launch
                if kbound.startswith('mfma'):
                    ach = fl / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
                    roof = dict(common, bound='mfma', achieved=ach, peak=FP64_PEAK_TFLOPS, unit='TFLOP/s', frac=ach / FP64_PEAK_TFLOPS, flops_per_launch=fl,
                                ordinates=nk, outputs_computed=nout,
                                note='algorithmic flops = 2 * [sum_{k < %d} (N - k) + %d * %d] per walker: the upper-triangular Abel product for the %d ordinates the row reads (of N = %d) '
                                     'and the %d x %d row product, over the kernel\'s HIP-event duration; peak = dense fp64 matrix-core peak (the instruction issues at 47 TFLOP/s '
                                     'chip-wide in scripts/ubench/mfma_f64_rate.hip, 0.60 of it)' % (nk, nout, nk, nk, args.N, nout, nk))
                else:
                    roof = dict(common, bound='latency', achieved=None, peak=None, unit=None, frac=None,
                                note='the longest kernel of the step is a chain of dependent phases (parameters -> priors -> grid pass of fp64 exp/log chains -> veto -> conversion '
                                     'factors | X-ray profiles -> count rates -> projection -> Cash sum): neither a flop nor a byte count prices it; its shares of wave cycles and '
                                     'its HBM bytes are given, and `roofline_product` prices the matrix-core kernel behind it',
                                matrix_core_kernel={'kernel': 'jx_ordrow_kernel', 'launch_ms_stage_pass': stage_ms['abel_map_ms'], 'flops_per_launch': fl,
                                                    'achieved': fl / (stage_ms['abel_map_ms'] * 1e-3) / 1e12 if stage_ms['abel_map_ms'] > 0 else None, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                                    'frac': (fl / (stage_ms['abel_map_ms'] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if stage_ms['abel_map_ms'] > 0 else None})
            else:
                roof = dict(common, bound=kbound, achieved=None, peak=None, unit=None, frac=None, note='contracted form of round 4: see profiles/r04_bench.json for its priced kernels')
        # HBM bytes of the step: exactly the kernels of the step, nothing else the PMC file holds
        step_pmc = None
        step_pmc_kernels = None
        if pj and mixed and kernels:
            step_pmc_kernels = {k[1]: pmc_traffic(pj, k[1], W) for k in kernels if pmc_kernel_entry(pj, k[1])}
            step_pmc = sum(step_pmc_kernels.values()) if step_pmc_kernels else None
        for r_ in [ns_route] + list(other_routes.values()):
            if r_ and 'ms_per_step' in r_:
                r_['ms_per_step_of_the_default_route_over_this'] = ms_step / r_['ms_per_step']
        if ns_route and 'ms_per_step' in ns_route:
            ns_route['speedup_of_default_route'] = ns_route['ms_per_step'] / ms_step
        none_or = lambda d, active: ({k: v for k, v in d.items() if k != 'rows'} if active else 'none')
        tr = ctx.truncation
        out = {
            'metric': 'walker-likelihoods/sec at 512^2 map, 500-pt grid' if (S, args.N) == (512, 500)
                      else 'walker-likelihoods/sec at %d^2 map, %d-pt grid' % (S, args.N),
            'value': value, 'unit': 'walker-likelihoods/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_step, 'ms_per_step_min': 1e3 * min(regions) / args.steps, 'ms_per_step_max': 1e3 * max(regions) / args.steps,
            'timed_regions': len(regions), 'timed_ms_total': 1e3 * sum(regions),
            'ms_per_step_note': 'median over the timed regions of `steps` steps each (every region bracketed by barrier + synchronisation; value = walkers x steps / median region; '
                                'no HIP event inside them: `roofline.launch_ms` comes from the same regions run once more with two events around the longest kernel)',
            'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': '%d walkers/GPU, %dx%d SZ map, %d-pt radial grid, %s likelihood, '
                                   'synthetic CL J1226.9+3332-shaped inputs%s'
                                   % (W, S, S, args.N, 'SZ-only' if pb.sz_only else 'joint X-ray+SZ',
                                      ' (BASELINE configs[2])' if (W, S, args.N, pb.sz_only) == (1024, 512, 500, False) else ''),
                       'walkers_per_gpu': W, 'S': S, 'N': args.N, 'B': pb.B, 'chunk': ctx.chunk, 'route': ctx.route, 'conv': ctx.conv, 'form': form,
                       'conv_layout': ctx.conv_layout, 'output_pruning': prn,
                       'sampling': none_or(ctx.sampling, ctx.sampling['active']), 'radial_sampling': none_or(ctx.radial_sampling, ctx.radial_sampling['active']),
                       'truncation': ('none' if (tr['rank'] == 0 and tr['est_rel_row_err'] < 0) else {k: v for k, v in tr.items() if k != 'warning'}),
                       'parallelism': 'walkers sharded x%d' % world, 'device': ctx.device_name,
                       'gather': (('overlapped: second stream, two output buffers' if overlap else 'strict: on the compute stream, in order') if comm is not None else None),
                       'gather_probe_ms_per_step': gather_probe},
            'n_ranks_seen': (comm.n_ranks_seen if comm is not None else 1),
            'gather_ms_per_step': gather_ms,
            'roofline': roof,
            # the whole step against the HBM roofline: measured bytes (rocprofv3 PMC, profiles/*_pmc_traffic.json)
            'roofline_step': {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'peak_measured': copy_gbs, 'unit': 'GB/s', 'ms_per_step': ms_step,
                              'traffic_bytes_per_step': step_pmc, 'traffic_by_kernel': step_pmc_kernels, 'traffic_source': (pj or {}).get('file'),
                              'traffic_measured_in_this_run': False,
                              'achieved': (step_pmc / (ms_step * 1e-3) / 1e9) if step_pmc else None,
                              'frac': (step_pmc / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_pmc else None,
                              'survey_8d_bytes_per_step': 2.0 * W * S * S * 8.0,
                              'note': 'the step is three short dependent launches (latency, then fp64 matrix cores): its HBM traffic is the small per-walker arrays between '
                                      'the kernels, far under SURVEY 8(d)\'s 2 S^2 * 8 B per walker -- no map, spectrum or convolved map is ever stored'},
            'step_kernels': step_kernels,
            'north_star_abel_map_kernel': full_map,
            'north_star_route': ns_route,
            'collapsed_route': other_routes.get('collapsed_route'),
            'legacy_contracted_route': other_routes.get('legacy_contracted_route'),
            'host_pointer': host_ptr,
            'device_sampler': samp,
            'hbm_copy_bandwidth_measured_GBps': copy_gbs,
            'hbm_stream_bandwidth_measured_GBps': streams,
            'fp32_variant': f32,
            'cpu_baseline': cpu,
            'stage_ms_per_step': {k: tm[k] / args.steps for k in stage_keys + ('total_ms',)},
            'stage_ms_note': 'HIP events behind every kernel, from one more pass of `steps` steps right after the timed regions',
            'parity_max_rel_err': parity,
            'other_configs': other,
        }
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    post.close()


if __name__ == '__main__':
    main()
