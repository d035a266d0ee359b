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

Prints ONE JSON line (rank 0) with
  `roofline`            the time-dominant kernel of the step (stage 1 of the contracted route: fp64 vector FMAs) on its
                        algorithmic flops over its HIP-event duration, measured inside the timed region;
  `roofline_product`    the matrix-core product behind it, same way;
  `roofline_step`       the whole step against the HBM roofline (rocprofv3 PMC bytes of the committed profile / ms_per_step);
  `north_star_abel_map_kernel`  the fused profile -> Abel -> spline -> full S x S map kernel the north_star's ">= 60 % of
                        the HBM roofline" is about (jx_map_kernel_time), against the nominal and the measured copy roofline;
  `north_star_route`    the whole step of north_star's literal design (that kernel, then the rocFFT sequence) on the same walkers;
  `host_pointer`        jx_eval with host pointers, one synchronisation per call: the path emcee exercises;
  `gather_ms_per_step`  (N > 1, or JOXSZ_BENCH_FORCE_DIST=1) the all-gather's own duration on its stream;
  `cpu_baseline`        the numpy/scipy oracle on this box's host cores (process pool, single process, per-stage ms);
  `other_configs`       strong-scaling rows of BASELINE configs[3] and configs[4] (this rank's shard).
"""
import argparse
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
    ap.add_argument('--no-full-map', action='store_true', help='skip the side measurement of the full-map Abel kernel and of the copy bandwidth')
    ap.add_argument('--route', choices=('map', 'operator'), default='map',
                    help="'map': the reference's sequence of steps per walker (the BASELINE metric); 'operator': the collapsed route (jx_set_route)")
    ap.add_argument('--fwhm', type=float, default=18.5, help='beam FWHM in arcsec (B = 2*floor(3*fwhm/step)+1)')
    ap.add_argument('--dtype', choices=('f64', 'f32', 'f32c'), default='f64',
                    help="'f64': the reference's arithmetic (the metric); 'f32': fp32 spline arrays; 'f32c': fp32 arithmetic in stages 1 and 2")
    ap.add_argument('--no-f32', action='store_true', help='skip the side measurement of the fp32 variant')
    ap.add_argument('--no-host-pointer', action='store_true', help='skip the side measurement of jx_eval with host pointers (profiling runs: keeps the trace to the timed steps)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the strong-scaling rows of BASELINE configs[3] and configs[4]')
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

    # parity of the CPU sample on the very same problem tensors: the bench refuses to report a number for results that differ
    parity = None
    if cpu_sample is not None:
        if args.route != 'map':
            ctx.set_route(args.route)
        got = ctx.eval(cpu_sample)
        fin = np.isfinite(cpu_logp)
        if not np.array_equal(np.isfinite(got), fin):
            raise SystemExit('bench: GPU/oracle disagree on which walkers are rejected')
        parity = float(np.max(np.abs(got[fin] - cpu_logp[fin]) / np.abs(cpu_logp[fin]))) if fin.any() else 0.0
        if parity > 1e-6:
            raise SystemExit('bench: parity %.3e exceeds 1e-6' % parity)

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
    overlap = multi and not os.environ.get('JOXSZ_BENCH_STRICT_GATHER')      # (JOXSZ_BENCH_OVERLAP_GATHER=1 pins the other mode; neither: the warm-up picks)
    if multi:
        # RCCL through the library's own C-ABI (jx_comm_*), no torch in this process.  The gather of step n runs on a second
        # stream of the context behind an event; step n+1 writes the OTHER of two output buffers, so its kernels overlap the
        # gather (a buffer still being sent holds back only the evaluation that would overwrite it).  The host never waits
        # inside a step.  JOXSZ_BENCH_STRICT_GATHER=1: every collective in order on the compute stream instead.
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
        # Which of the two gather modes this machine wants is not known in advance: the second stream hides the collective's
        # latency (tens of microseconds over 8 ranks) but costs every kernel of the step a little (two active queues); in line it
        # costs its own duration.  Still inside the warm-up: the same short run in both modes, the ranks agree on the faster one
        # (one max-reduction each -- every rank takes the same branch), and the timed region below runs in that mode.
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
        overlap = gather_probe['overlapped'] < gather_probe['strict']
        ctx.comm_set_overlap(overlap)
        fence()
    # Timed region: HIP events around the time-dominant kernel only (stage 1 of the contracted route; jx_timing_enable(2)) --
    # its duration is what `roofline` prices.  Events behind every stage cost a few per cent of a step (markers between
    # dependent kernels), so the full stage breakdown comes from a second, identical pass of the same K steps right after.
    p1_only = (args.route == 'map' and ctx.conv == 'custom')
    ctx.timing_enable(2 if p1_only else 1)
    ctx.timing_reset()
    fence()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t_start
    tm_timed = ctx.timing()
    tm = tm_timed
    if p1_only:
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

    last = (nstep[0] - 1) % len(lp_ptrs)
    if comm is not None:
        elapsed = comm.max_over_ranks(elapsed)
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

    # the path emcee calls (joxsz_main.py:206, vectorize=True): host theta in, host log-probabilities out, one sync per call
    host_ptr = None
    if rank == 0 and not args.no_host_pointer:
        try:
            for _ in range(3):
                ctx.eval(theta)
            kk = max(5, min(args.steps, 50))
            t = time.perf_counter()
            for _ in range(kk):
                got_h = ctx.eval(theta)
            dt = (time.perf_counter() - t) / kk
            host_ptr = {'value': W / dt, 'unit': 'walker-likelihoods/s', 'ms_per_call': 1e3 * dt, 'calls': kk,
                        'max_abs_diff_vs_device_resident': float(np.max(np.abs(got_h - lp_own))),
                        'note': 'jx_eval: host parameter vectors in (%d B per walker over PCIe), host log-probabilities out, one stream '
                                'synchronisation per call -- what emcee.EnsembleSampler(..., vectorize=True) exercises (joxsz_main.py:206)' % (8 * theta.shape[1])}
        except Exception as exc:
            host_ptr = {'error': str(exc)}

    # the caller of the path: the device-resident stretch-move loop on an ensemble of 2 W walkers -- each half step one
    # evaluation of W proposals, drawn in the per-walker kernel and accepted or rejected in the tail
    samp = None
    if rank == 0 and comm is None and args.route == 'map' and not args.no_host_pointer:
        try:
            x0 = np.ascontiguousarray(np.concatenate((theta, good[W:2 * W] if len(good) >= 2 * W else theta[::-1] * (1 + 1e-9))))
            ns1, ns = 10, 60
            ctx.sample(x0, ns)                                     # (warm-up at the longer length: the call's device buffers are sized once)                                       # (two run lengths: the slope is the step, the intercept the call's set-up, initial evaluation and copy back)
            t = time.perf_counter()
            ctx.sample(x0, ns1, seed=3)
            t1 = time.perf_counter() - t
            t = time.perf_counter()
            _, lps_s, nacc_s = ctx.sample(x0, ns, seed=3)
            dt = ((time.perf_counter() - t) - t1) / (ns - ns1)
            samp = {'value': 2 * W / dt, 'unit': 'walker-updates/s', 'ms_per_step': 1e3 * dt, 'walkers': 2 * W, 'steps': ns, 'ms_per_call_besides_the_steps': 1e3 * (t1 - ns1 * dt),
                    'acceptance': float(nacc_s.sum()) / (2 * W * ns), 'finite': bool(np.isfinite(lps_s).all()),
                    'note': 'jx_sample: proposals, evaluation, accept/reject and chain on the device, one copy back at the end; a step = two half steps of %d walkers' % W}
        except Exception as exc:
            samp = {'error': str(exc)}

    # the fp32 variant on the same walkers (BASELINE configs[4]'s tolerance sweep): beside the f64 metric, never instead of it
    f32 = None
    if rank == 0 and comm is None and args.route == 'map' and args.dtype == 'f64' and not args.no_f32:
        f32 = {}
        notes = {'f32': 'jx_config.dtype = 1: spline arrays (y_k, M_k) rounded to fp32 once and read as fp32 by the sample evaluation; sums, matrix-core '
                        'product, tail and everything per-walker in fp64',
                 'f32c': 'jx_config.dtype = 2: fp32 arithmetic -- stage 1 in packed fp32 FMAs (v_pk_fma_f32), stage 2 on v_mfma_f32_16x16x4_f32, stage-1 rows '
                         'and partial rows in fp32, K slices added in fp64 by the tail; everything per-walker (priors, X-ray, conversion, chi^2) in fp64'}
        ch64 = ctx.eval_stage(theta[:256], 'chisq')
        for dt_name in ('f32', 'f32c'):
            try:
                p3 = JoxszPosterior(pb, device=local_rank, dtype=dt_name)
                c3 = p3.ctx
                t3, l3 = c3.dev_alloc(theta.nbytes), c3.dev_alloc(8 * W)
                c3.h2d(t3, theta)
                dt = time_steps(c3, t3, W, l3, max(5, args.steps // 4), 3)
                lp32 = np.empty(W)
                c3.d2h(lp32, l3)
                ch32 = c3.eval_stage(theta[:256], 'chisq')
                rel = np.abs(lp32 - final) / np.abs(final)
                f32[dt_name] = {'dtype': dt_name, 'value': W / dt, 'unit': 'walker-likelihoods/s', 'ms_per_step': 1e3 * dt,
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
                        'note': 'a store stream: S^2 * 8 B written per walker, priced against the nominal peak, against what a plain '
                                'write stream of the same 2 GiB gets on this card, and against a copy (bytes read + written)'}
        except Exception as exc:
            full_map = {'error': str(exc)}

    # north_star's literal design as a whole step: fused profile -> Abel -> map kernel, then the rocFFT sequence for the beam
    # convolution and the transfer function (joxsz_funcs.py:460-467 executed step by step), same walkers, same problem
    ns_route = None
    if rank == 0 and comm is None and args.route == 'map' and not args.no_full_map and ctx.conv == 'custom':
        try:
            pr = JoxszPosterior(pb, device=local_rank, conv='rocfft', max_batch=W)
            cr = pr.ctx
            tr_, lr_ = cr.dev_alloc(theta.nbytes), cr.dev_alloc(8 * W)
            cr.h2d(tr_, theta)
            dt = time_steps(cr, tr_, W, lr_, 5, 2)
            cr.timing_enable(1); cr.timing_reset()
            for _ in range(3):
                cr.eval_device(tr_, W, lr_)
            tmr = cr.timing()
            cr.timing_enable(False)
            lpr = np.empty(W)
            cr.d2h(lpr, lr_)
            ns_route = {'route': 'jx_abel_map_sym_kernel -> rocFFT R2C -> jx_beam_mul_kernel -> rocFFT C2R -> rocFFT R2C of the S x S window -> jx_tail_kernel',
                        'value': W / dt, 'unit': 'walker-likelihoods/s', 'ms_per_step': 1e3 * dt, 'walkers_per_launch': cr.chunk, 'fft_pad': cr.fft_pad,
                        'stage_ms_per_step': {k: tmr[k] / 3 for k in ('prep_ms', 'abel_map_ms', 'beam_fft_ms', 'tf_fft_ms', 'tail_ms')},
                        'max_rel_diff_vs_default_route': float(np.max(np.abs(lpr - lp_own) / np.abs(lp_own))),
                        'speedup_of_default_route': None,
                        'note': 'the same library with conv = rocfft: every map, its padded spectrum and the convolved map go through HBM'}
            pr.close()
        except Exception as exc:
            ns_route = {'error': str(exc)}

    # the same batch on the collapsed route, only when asked for (--route operator): not the BASELINE metric
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
            dts[0] = time_steps(c3, p3t, n3, l3t, 5, 2)
            other['configs[3]'] = {'workload': '4096 walkers, 512x512 map, 500-pt grid, joint; %d walkers on this rank' % n3,
                                   'unit': 'walker-likelihoods/s', 'scaling': 'strong', 'dtype': 'f64', 'walkers_per_launch': c3.chunk}
            if p3c is not None:
                p3c.close()
            lo, hi = shard_bounds(8192, world, rank)
            n4 = hi - lo
            pb4 = datasets.synthetic_problem(S=1024, N=1000, seed=0)
            for i4, dt_name in enumerate(('f64', 'f32c')):
                p4 = JoxszPosterior(pb4, device=local_rank, dtype=dt_name, max_batch=n4)
                c4 = p4.ctx
                cand4 = datasets.walker_ball(pb4, 256, spread=0.02, seed=5)
                ok4 = cand4[np.isfinite(c4.eval(cand4))]
                th4 = np.ascontiguousarray(np.resize(ok4, (n4, ok4.shape[1])))
                a4, b4 = c4.dev_alloc(th4.nbytes), c4.dev_alloc(8 * n4)
                c4.h2d(a4, th4)
                dts[1 + i4] = time_steps(c4, a4, n4, b4, 3, 1)
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
        lay = ctx.conv_layout or {}
        stage_ms = {k: tm[k] / launches for k in ('prep_ms', 'abel_map_ms', 'beam_fft_ms', 'tf_fft_ms', 'tail_ms')}
        mixed = ctx.conv == 'custom' and args.route == 'map'
        lowrank = mixed and lay.get('form') == 'lowrank'
        stage_kernels = None
        if mixed:
            stage_kernels = {'prep_ms': 'jx_prep_kernel (priors, vetoes, X-ray Cash likelihood, pressure and temperature profiles)',
                             'abel_map_ms': 'jx_abel_gemm_kernel (Abel transform, y scale and spline moments of the launch as one fp64 matrix-core product)',
                             'beam_fft_ms': 'jx_rowmix_kernel (stage 1: map samples evaluated from the spline and mixed per column, fp64 vector FMAs)' if lowrank
                                            else '(none: the full form has one kernel)',
                             'tf_fft_ms': 'jx_opgemm_kernel (stage 2: beam along x + circular transfer-function kernels + row extraction as one fp64 matrix-core product)' if lowrank
                                          else 'jx_opgemm_kernel (full form: map samples evaluated by the lanes that feed the fp64 matrix cores)',
                             'tail_ms': 'jx_tail_row_kernel (partial rows summed in fixed order, conversion, chi^2, total)'}
        try:
            ev_null_ms = ctx.event_bracket_time(64)
        except Exception:
            ev_null_ms = None
        value = W * world * args.steps / elapsed
        ms_step = 1e3 * elapsed / args.steps
        pj = pmc_file(S)
        nrow = ctx.nrow
        roof = roof2 = None
        if mixed:
            NU = lay['NU']
            if lowrank:
                # stage 1: per sample and walker 4 FMAs of the spline evaluation + R of the mixing
                k_ms = tm_timed['beam_fft_ms'] / max(1, tm_timed['launches'])
                NUe = ctx.sampling['rows_evaluated']                      # rows (= columns) of the quadrant stage 1 evaluates (the sub-grid)
                fl = 2.0 * NUe * NUe * (4 + lay['R']) * walkers_per_launch
                ach = fl / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
                roof = {'kernel': 'jx_rowmix_kernel', 'bound': 'valu (fp64 vector FMA; the same 78.6 TFLOP/s as the dense fp64 matrix-core peak)',
                        'achieved': ach, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / FP64_PEAK_TFLOPS,
                        'traffic': pmc_traffic(pj, 'jx_rowmix_kernel', walkers_per_launch),
                        'traffic_source': (pj or {}).get('file'), 'traffic_measured_in_this_run': False,
                        'peak_measured': FP64_FMA_MEASURED_TFLOPS, 'frac_of_peak_measured': ach / FP64_FMA_MEASURED_TFLOPS,
                        'launch_ms': k_ms, 'launch_ms_source': 'HIP events around this kernel inside the timed region (jx_timing_enable(2))',
                        'launch_ms_of_an_empty_kernel': ev_null_ms,
                        'launch_ms_note': 'a pair of HIP events around one kernel of a dependent chain also spans the command processor\'s hand-over '
                                          'in front of and behind it -- launch_ms_of_an_empty_kernel is what the same pair reads around a kernel that does nothing '
                                          '(jx_event_bracket_time); rocprofv3\'s kernel trace (profiles/*_kernel_stats.csv) counts the kernel alone, so its average sits '
                                          'between launch_ms - launch_ms_of_an_empty_kernel and launch_ms.  achieved and frac use launch_ms as measured (the lower figure)',
                        'flops_per_launch': fl, 'share_of_step': k_ms / max(1e-12, ms_step * walkers_per_launch / W),
                        'algorithmic_bytes_per_launch': walkers_per_launch * (16.0 * pb.N + 8.0 * NUe * lay['R']),
                        'samples_evaluated_per_walker': NUe * NUe, 'distinct_samples_per_walker': NU * NU,
                        'survey_8d_bytes_per_launch': walkers_per_launch * S * S * 8.0,
                        'note': 'algorithmic flops = 2 * NUe^2 * (4 + R) per walker (4 FMAs evaluate a map sample from the spline, R mix it into '
                                'the rows kept per column; NUe = %d of the quadrant\'s %d distinct rows = columns are evaluated -- the sub-grid of '
                                'jx_get_sampling, the interpolation to the others sits in the operators -- R = %d) / HIP-event duration.  The S x S map of '
                                'SURVEY 8(d) (S^2 * 8 B per walker) is never written: every sample lives in a register.  The kernel is bound by '
                                'the fp64 vector units, not by HBM; peak_measured = sustained v_fmac_f64 rate of this chip (scripts/ubench/fma_sgpr.hip)'
                                % (NUe, NU, lay['R'])}
            p_ms = stage_ms['tf_fft_ms']
            K4 = lay['ksteps'] * 4
            prn = ctx.output_pruning
            nout = prn['outputs_computed'] if prn['active'] else nrow      # the timed product computes the outputs the tail's data-radii spline reads
            fl2 = 2.0 * nout * K4 * walkers_per_launch
            ach2 = fl2 / (p_ms * 1e-3) / 1e12 if p_ms > 0 else 0.0
            roof2 = {'kernel': 'jx_opgemm_kernel', 'bound': 'mfma', 'achieved': ach2, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': ach2 / FP64_PEAK_TFLOPS, 'traffic': pmc_traffic(pj, 'jx_opgemm_kernel', walkers_per_launch),
                     'launch_ms': p_ms, 'launch_ms_source': 'HIP events of the stage pass', 'flops_per_launch': fl2,
                     'outputs_computed': nout, 'outputs_of_the_row': nrow,
                     'note': 'outputs x K x walkers product on v_mfma_f64_16x16x4 (%d of the row\'s %d outputs: the ones the data-radii spline of the tail reads '
                             'with a weight above 1e-22 of its largest, in whole tiles of 16; K = %d rows of the operator)' % (nout, nrow, K4)}
            if not lowrank:
                roof, roof2 = roof2, None
        # HBM bytes of the step: exactly the kernels of the step (stage_kernels), nothing else the PMC file holds
        step_names = ('jx_prep_kernel', 'jx_abel_gemm_kernel', 'jx_rowmix_kernel', 'jx_rowmix_mfma_kernel', 'jx_opgemm_kernel', 'jx_tail_row_kernel')
        step_pmc = None
        step_pmc_kernels = None
        if pj and mixed:
            step_pmc_kernels = {n: pmc_traffic(pj, n, W) for n in step_names if pmc_kernel_entry(pj, n)}
            step_pmc = sum(step_pmc_kernels.values()) if step_pmc_kernels else None
        # every kernel of the step with the duration the stage pass read for it: since the sub-grid of stage 1 no kernel
        # dominates -- the per-walker kernel, stage 1 and the spline-array product are within 15 % of one another
        step_kernels = None
        if mixed and stage_kernels:
            bounds = {'prep_ms': 'latency + fp64 exp/log chains of the grid pass on the vector units (73 % busy; DESIGN 6.3)',
                      'abel_map_ms': 'fp64 matrix cores at one wave per SIMD: operand fetch of the k loop (DESIGN 6.3)',
                      'beam_fft_ms': 'fp64 vector FMA issue (`roofline`)' if lowrank else None,
                      'tf_fft_ms': 'fp64 matrix cores + operand fetch (`roofline_product`)' if lowrank else 'fp64 matrix cores fed by the lanes that evaluate the samples (`roofline`)',
                      'tail_ms': 'L2 reads of the partial rows + latency'}
            step_kernels = [{'kernel': stage_kernels[k].split(' ')[0], 'ms_hip_events_stage_pass': stage_ms[k], 'share_of_stage_sum': stage_ms[k] / max(1e-12, sum(stage_ms[q] for q in bounds)),
                             'bound': bounds[k]} for k in ('prep_ms', 'abel_map_ms', 'beam_fft_ms', 'tf_fft_ms', 'tail_ms') if bounds[k]]
            # where each kernel's wave cycles go, from the committed counter pass of the same command (profiles/*_sq_counters.csv; not measured in this run)
            sq = sq_counters_file()
            for e in step_kernels:
                row = next((r for r in sq.get('rows', []) if e['kernel'] in r['kernel'] and float(r['SQ_WAVE_CYCLES']) > 1e6), None)
                if row:
                    wc = float(row['SQ_WAVE_CYCLES'])
                    e['wave_cycles_source'] = sq['file']
                    e['share_of_wave_cycles'] = {'waiting': float(row['SQ_WAIT_ANY']) / wc, 'issue_stalled': float(row['SQ_WAIT_INST_ANY']) / wc,
                                                 'issuing': float(row['SQ_ACTIVE_INST_ANY']) / wc, 'issuing_valu': float(row['SQ_ACTIVE_INST_VALU']) / wc}
            if roof is not None:
                longest = max(step_kernels, key=lambda e: e['ms_hip_events_stage_pass'])
                roof['longest_kernel_of_the_step'] = {'kernel': longest['kernel'], 'ms_hip_events_stage_pass': longest['ms_hip_events_stage_pass']}
                roof['longest_kernel_note'] = ('no kernel dominates the step: the per-walker kernel (latency + fp64 exp/log chains; no flop or byte count prices it, its '
                                               'vector-issue share is in step_kernels), stage 1 and the spline-array product lie within 15 % of one another; `roofline` prices '
                                               'stage 1, the kernel that holds the arithmetic of the path')
        if ns_route and 'ms_per_step' in ns_route:
            ns_route['speedup_of_default_route'] = ns_route['ms_per_step'] / ms_step
        out = {
            'metric': 'walker-likelihoods/sec at 512^2 map, 500-pt grid' if (S, args.N) == (512, 500)
                      else 'walker-likelihoods/sec at %d^2 map, %d-pt grid' % (S, args.N),
            'value': value, 'unit': 'walker-likelihoods/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_step, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': '%d walkers/GPU, %dx%d SZ map, %d-pt radial grid, %s likelihood, '
                                   'synthetic CL J1226.9+3332-shaped inputs%s'
                                   % (W, S, S, args.N, 'SZ-only' if pb.sz_only else 'joint X-ray+SZ',
                                      ' (BASELINE configs[2])' if (W, S, args.N, pb.sz_only) == (1024, 512, 500, False) else ''),
                       'walkers_per_gpu': W, 'S': S, 'N': args.N, 'B': pb.B, 'chunk': ctx.chunk, 'route': ctx.route, 'conv': ctx.conv,
                       'conv_layout': ctx.conv_layout, 'output_pruning': ctx.output_pruning,
                       'sampling': {k: v for k, v in ctx.sampling.items() if k != 'rows'}, 'parallelism': 'walkers sharded x%d' % world, 'device': ctx.device_name,
                       'gather': (('overlapped: second stream, two output buffers' if overlap else 'strict: on the compute stream') if comm is not None else None),
                       'gather_probe_ms_per_step': gather_probe},
            'n_ranks_seen': (comm.n_ranks_seen if comm is not None else 1),
            'gather_ms_per_step': gather_ms,
            'roofline': roof,
            'roofline_product': roof2,
            # the whole step against the HBM roofline: measured bytes (rocprofv3 PMC, profiles/*_pmc_traffic.json)
            'roofline_step': {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'peak_measured': copy_gbs, 'unit': 'GB/s', 'ms_per_step': ms_step,
                              'traffic_bytes_per_step': step_pmc, 'traffic_by_kernel': step_pmc_kernels, 'traffic_source': (pj or {}).get('file'),
                              'traffic_measured_in_this_run': False,
                              'achieved': (step_pmc / (ms_step * 1e-3) / 1e9) if step_pmc else None,
                              'frac': (step_pmc / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_pmc else None,
                              'survey_8d_bytes_per_step': 2.0 * W * S * S * 8.0,
                              'note': 'the step is compute-bound (fp64 vector units, then fp64 matrix cores): its HBM traffic is the small '
                                      'per-walker arrays between the kernels'},
            'step_kernels': step_kernels,
            'north_star_abel_map_kernel': full_map,
            'north_star_route': ns_route,
            'host_pointer': host_ptr,
            'device_sampler': samp,
            'hbm_copy_bandwidth_measured_GBps': copy_gbs,
            'hbm_stream_bandwidth_measured_GBps': streams,
            'truncation': ctx.truncation,
            'fp32_variant': f32,
            'cpu_baseline': cpu,
            'stage_ms_per_step': {k: tm[k] / args.steps for k in ('prep_ms', 'abel_map_ms', 'beam_fft_ms', 'tf_fft_ms', 'tail_ms', 'total_ms')},
            'stage_ms_note': 'HIP events behind every stage, from a second pass of the same steps right after the timed region '
                             '(the timed region itself carries only the two events around the time-dominant kernel)' if p1_only else None,
            'stage_kernels': stage_kernels,
            'parity_max_rel_err': parity,
            'other_configs': other,
        }
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    post.close()


if __name__ == '__main__':
    main()
