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
imports no torch; the launcher (torch.distributed.run) only sets RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_PORT.

Prints ONE JSON line (rank 0) with
  `roofline`       the time-dominant kernel of the step on the bytes it must move (HIP-event duration on the library's
                   stream), with SURVEY 8(d)'s S*S*8 B/walker figure beside it and the rocprofv3 PMC traffic of the committed
                   profile;
  `roofline_step`  the whole step: measured HBM bytes / ms_per_step against the 8 TB/s peak;
  `north_star_abel_map_kernel`  the fused profile->Abel->spline->map kernel storing the full S x S map (the kernel the
                   north_star's ">= 60 % of the HBM roofline" is about), measured beside the metric;
  `cpu_baseline`   the numpy/scipy oracle on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def pmc_file(S):
    """Latest committed rocprofv3 PMC measurement (scripts/measure_traffic.py -> profiles/*_pmc_traffic.json) of this map
    side, or None.  The counters cannot be read from inside the benchmark process."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')), reverse=True):
        try:
            j = json.load(open(f))
            if j.get('S', 512) == S:
                return j
        except Exception:
            pass
    return None


def pmc_traffic(j, kernel, walkers_per_launch):
    """HBM bytes per launch of `kernel` from that file, rescaled to this launch size."""
    if not j:
        return None
    for name, d in j['kernels'].items():
        if kernel in name:
            return d['total_bytes'] / j.get('walkers_per_launch', 1024) * walkers_per_launch
    return None


def must_move_bytes(ctx, pb, lay, W):
    """Bytes every kernel of the default route has to move per launch of W walkers whatever its implementation (its
    inputs read once + its outputs written once; walker-independent tables excluded), from the layout the library chose.
    The reference's S x S map (S^2 * 8 B per walker written, read once) is never materialised on this route."""
    N, S = pb.N, pb.S
    nrow = ctx.nrow
    if not lay or not lay.get('fused'):
        return None
    NU, kact, r = lay['NU'], lay['kact'], lay['rank']
    P = lay['P']
    coef = 16.0 * N                                   # (y_k, M_k) per knot
    rows_t = 8.0 * kact * NU + 8.0 * NU               # real row spectra below the band limit + column 0
    ct = 8.0 * kact * r + 8.0 * 40 * r                # combined rows + their column-0 terms
    zp = 16.0 * (S // 2 + 1) * ((r + 13) // 14)       # partial Z per pass-3 block
    small = 8.0 * (pb.ndim + nrow + 2)
    per = {
        'jx_prep_kernel': small + 8.0 * N,                # (+ the pressure profile for the spline-array product)
        'jx_abel_gemm_kernel': 8.0 * N + coef,
        'jx_rowdct_kernel': coef + rows_t,
        'jx_lowrank_kernel': rows_t + ct,
        'jx_rowtf2_kernel': ct + zp,
        'jx_tail_fft_kernel': zp + small,
    }
    return {k: v * W for k, v in per.items()}


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


def cpu_baseline(pb, thetas, target_s=15.0):
    """The oracle (numpy/scipy restatement of the reference path) mapped over
    walkers with multiprocessing.Pool on all host cores, exactly the reference's
    parallelism (joxsz_main.py:203-206), on a bounded sample of the workload."""
    import multiprocessing as mp
    from oracle import joxsz_oracle as orc
    cores = host_cores()
    t = time.perf_counter()
    orc.log_posterior_batch(pb, thetas[:1])
    one = time.perf_counter() - t
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
                       % (n, cores, dt)), sample, logp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--walkers', type=int, default=1024, help='walkers per GPU')
    ap.add_argument('--S', type=int, default=512)
    ap.add_argument('--N', type=int, default=500)
    ap.add_argument('--sz-only', action='store_true')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--no-full-map', action='store_true', help='skip the side measurement of the full-map Abel kernel')
    ap.add_argument('--route', choices=('map', 'operator'), default='map',
                    help="'map': the reference's sequence of steps per walker (the BASELINE metric); 'operator': the collapsed route (jx_set_route)")
    ap.add_argument('--fwhm', type=float, default=18.5, help='beam FWHM in arcsec (B = 2*floor(3*fwhm/step)+1)')
    ap.add_argument('--dtype', choices=('f64', 'f32'), default='f64', help="'f64': the reference's arithmetic (the metric); 'f32': the fp32 variant")
    ap.add_argument('--no-f32', action='store_true', help='skip the side measurement of the fp32 variant')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus %d needs a launcher that starts that many ranks (torch.distributed.run sets RANK/WORLD_SIZE/LOCAL_RANK)' % args.gpus)
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

    # parity spot-check of the CPU sample on the very same problem tensors
    parity = None
    if cpu_sample is not None:
        if args.route != 'map':
            ctx.set_route(args.route)
        got = ctx.eval(cpu_sample)
        fin = np.isfinite(cpu_logp)
        if not np.array_equal(np.isfinite(got), fin) and not os.environ.get('JOXSZ_DBG'):
            raise SystemExit('bench: GPU/oracle disagree on which walkers are rejected')
        parity = float(np.max(np.abs(got[fin] - cpu_logp[fin]) / np.abs(cpu_logp[fin]))) if fin.any() else 0.0
        if parity > 1e-6 and not os.environ.get('JOXSZ_DBG'):
            raise SystemExit('bench: parity %.3e exceeds 1e-6' % parity)

    # ---- synthetic observations from the model itself at the fiducial vector, then the walker ball ----
    t0 = datasets.fiducial_theta(pb)
    t0w = np.repeat(t0[None, :], W, axis=0)              # full-size launches only: the rocprof averages stay comparable
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
    if os.environ.get('JOXSZ_DBG'):
        good = cand
    if len(good) < W:
        raise SystemExit('bench: only %d finite walkers of %d' % (len(good), len(cand)))
    theta = np.ascontiguousarray(good[:W])

    # ---- device-resident inputs ----
    th_ptr = ctx.dev_alloc(theta.nbytes)
    lp_ptr = ctx.dev_alloc(8 * W)
    ctx.h2d(th_ptr, theta)
    comm = None
    all_ptr = None
    if multi:
        # RCCL through the library's own C-ABI (jx_comm_*): the gather is enqueued on the context's stream behind the
        # evaluation, the host never waits inside a step, no torch in this process
        from joxsz_amd.dist import RcclGather
        comm = RcclGather(ctx, rank=rank, world=world)
        all_ptr = ctx.dev_alloc(8 * W * world)

    def step():
        ctx.eval_device(th_ptr, W, lp_ptr)
        if comm is not None:
            comm.all_gather(lp_ptr, all_ptr, W)

    def fence():
        if comm is not None:
            comm.barrier()
        ctx.sync()

    for _ in range(args.warmup):
        step()
    fence()
    # Timed region: HIP events around the time-dominant kernel only (pass 1 of the default route; jx_timing_enable(2)) -- its
    # duration is what `roofline` prices.  Events behind every stage cost ~4 % of a step (seven markers between dependent
    # kernels), so the full stage breakdown comes from a second, identical pass of the same K steps right after.
    p1_only = (args.route == 'map')
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
        if not tm_timed['launches'] or tm_timed['beam_fft_ms'] <= 0.0:        # (routes without a pass 1 of their own, e.g. rocFFT)
            tm_timed = tm
    ctx.timing_enable(False)

    if comm is not None:
        elapsed = comm.max_over_ranks(elapsed)
        final = np.empty(W * world)
        ctx.d2h(final, all_ptr)
    else:
        final = np.empty(W)
        ctx.d2h(final, lp_ptr)
    if not np.all(np.isfinite(final)) and not os.environ.get('JOXSZ_DBG'):
        raise SystemExit('bench: non-finite log-probabilities in the timed batch')

    # the same batch on the collapsed route, outside the timed region of the metric: reported beside it
    also = None
    if comm is None and args.route == 'map':
        try:
            lp_map = final
            ctx.set_route('operator')
            for _ in range(args.warmup):
                step()
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            dt = time.perf_counter() - t1
            lp_op = np.empty(W)
            ctx.d2h(lp_op, lp_ptr)
            fin = np.isfinite(lp_map)
            also = {'route': 'operator', 'value': W * args.steps / dt, 'unit': 'walker-likelihoods/s', 'ms_per_step': 1e3 * dt / args.steps,
                    'max_rel_diff_vs_map_route': (float(np.max(np.abs(lp_op[fin] - lp_map[fin]) / np.abs(lp_map[fin]))) if fin.any() else None),
                    'same_rejections': bool(np.array_equal(np.isfinite(lp_op), fin)),
                    'note': 'same walkers, same library, jx_set_route(JX_ROUTE_OPERATOR): the SZ side as one constant nrow x N matrix '
                            'applied to the pressure profile; not the BASELINE metric (no Abel+map kernel runs per step)'}
            ctx.set_route('map')
        except Exception as exc:                                  # never let the side measurement take the metric down
            also = {'route': 'operator', 'error': str(exc)}


    # the fp32 variant on the same walkers (BASELINE configs[4]'s tolerance sweep): beside the f64 metric, never instead of it
    f32 = None
    if rank == 0 and comm is None and args.route == 'map' and args.dtype == 'f64' and not args.no_f32:
        try:
            p3 = JoxszPosterior(pb, device=local_rank, dtype='f32')
            c3 = p3.ctx
            t3, l3 = c3.dev_alloc(theta.nbytes), c3.dev_alloc(8 * W)
            c3.h2d(t3, theta)
            for _ in range(args.warmup):
                c3.eval_device(t3, W, l3)
            c3.sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                c3.eval_device(t3, W, l3)
            c3.sync()
            dt = time.perf_counter() - t1
            lp32 = np.empty(W)
            c3.d2h(lp32, l3)
            ch32, ch64 = c3.eval_stage(theta[:256], 'chisq'), ctx.eval_stage(theta[:256], 'chisq')
            rel = np.abs(lp32 - final) / np.abs(final)
            f32 = {'dtype': 'f32', 'value': W * args.steps / dt, 'unit': 'walker-likelihoods/s', 'ms_per_step': 1e3 * dt / args.steps,
                   'rel_dlogp_vs_f64': {'max': float(rel.max()), 'median': float(np.median(rel))},
                   'abs_dchisq_vs_f64': {'max': float(np.abs(ch32 - ch64).max()), 'median': float(np.median(np.abs(ch32 - ch64)))},
                   'note': 'fp32 evaluation of the map rows + fp32 row transform + fp32 storage of row spectra and combined rows; '
                           'matrix products, inverse transforms, tail and everything per-walker in fp64 (jx_config.dtype = 1)'}
            p3.close()
        except Exception as exc:
            f32 = {'dtype': 'f32', 'error': str(exc)}

    # the kernel that meets north_star's "Abel+map kernel at >= 60 % of the HBM roofline": the same fused kernel storing the
    # reference's full S x S map (JOXSZ_FULL_MAP=1), measured beside the metric (the default route never stores a map)
    full_map = None
    if rank == 0 and comm is None and args.route == 'map' and not args.no_full_map:
        try:
            os.environ['JOXSZ_FULL_MAP'] = '1'
            p2 = JoxszPosterior(pb, device=local_rank)
            os.environ.pop('JOXSZ_FULL_MAP')
            c2 = p2.ctx
            for _ in range(2):
                c2.eval(theta)
            c2.timing_enable(True); c2.timing_reset()
            for _ in range(5):
                c2.eval(theta)
            t2 = c2.timing()
            ms = t2['abel_map_ms'] / max(1, t2['launches'])
            wl = t2['walkers'] / max(1, t2['launches'])
            full_map = {'kernel': 'jx_abel_map_sym_kernel (JOXSZ_FULL_MAP=1: profile -> Abel -> spline -> full S x S map)',
                        'launch_ms': ms, 'bytes_per_launch': wl * args.S * args.S * 8.0,
                        'achieved_GBps': wl * args.S * args.S * 8.0 / (ms * 1e-3) / 1e9,
                        'frac_of_hbm_peak': wl * args.S * args.S * 8.0 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            p2.close()
        except Exception as exc:
            os.environ.pop('JOXSZ_FULL_MAP', None)
            full_map = {'error': str(exc)}

    if rank == 0:
        S = args.S
        launches = max(1, tm['launches'])
        walkers_per_launch = tm['walkers'] / launches
        lay = ctx.conv_layout or {}
        stage_ms = {k: tm[k] / launches for k in ('prep_ms', 'abel_map_ms', 'beam_fft_ms', 'tf_fft_ms', 'tail_ms')}
        if tm.get('gemm_ms', 0.0) > 0.0:                      # the matrix products and pass 3 are timed apart on the default route
            stage_ms['gemm_ms'] = tm['gemm_ms'] / launches
            stage_ms['pass3_ms'] = stage_ms.pop('tf_fft_ms') - stage_ms['gemm_ms']
        dct = bool(lay.get('fused')) and not os.environ.get('JOXSZ_DCT') == '0'
        stage_kernels = {'prep_ms': 'jx_prep_kernel',
                         'abel_map_ms': 'jx_abel_gemm_kernel (Abel transform, y scale and spline moments of the launch as one fp64 MFMA product)' if dct else 'jx_abel_map_sym_kernel',
                         'beam_fft_ms': ('jx_rowdct_kernel (map rows evaluated from the spline + real-even row transform)' if dct
                                         else 'jx_rowfft2_kernel (pass 1)'),
                         'tf_fft_ms': 'jx_lowrank_kernel (fp64 MFMA GEMM) + jx_rowtf2_kernel (pass 3)', 'tail_ms': 'jx_tail_fft_kernel',
                         'gemm_ms': 'jx_lowrank_kernel (FIR + job combination as fp64 MFMA matrix products)',
                         'pass3_ms': 'jx_rowtf2_kernel (inverse row transform, crop, forward transform, transfer-function weights)'}
        dom = max(stage_ms, key=stage_ms.get)
        dom_kernel = stage_kernels[dom].split(' ')[0]
        k_ms = stage_ms[dom]
        if dom == 'beam_fft_ms':                              # measured inside the timed region
            k_ms = tm_timed['beam_fft_ms'] / max(1, tm_timed['launches'])
        mm = must_move_bytes(ctx, pb, lay, walkers_per_launch) if dct else None
        pj = pmc_file(S)
        survey_bytes = walkers_per_launch * S * S * 8.0       # the reference's map, S^2 * 8 B per walker
        dom_bytes = (mm or {}).get(dom_kernel)
        if dom_bytes is None:
            dom_bytes = survey_bytes
        achieved = dom_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        value = W * world * args.steps / elapsed
        ms_step = 1e3 * elapsed / args.steps
        step_must = sum(mm.values()) * (W / walkers_per_launch) if mm else None
        step_pmc = None
        if pj:
            step_pmc = sum(d['total_bytes'] for n, d in pj['kernels'].items() if n.startswith(('jx_', 'void jx_')) and 'operator' not in n) \
                / pj.get('walkers_per_launch', 1024) * W
        onchip = None
        if dct and dom_kernel == 'jx_rowdct_kernel' and k_ms > 0:
            esz = 4.0 if args.dtype == 'f32' else 8.0
            NUr, Q = lay['NU'], lay['P'] // 4
            ns = S // 2 + (S % 2)                                    # samples per distinct row (the unpaired column apart)
            l1_b = walkers_per_launch * NUr * ns * 4.0 * esz        # (y_k, M_k, y_k+1, M_k+1) per (sample, walker) returned to registers
            lds_b = walkers_per_launch * NUr * esz * (2.75 * ns + 12.0 * Q)   # q write, z-build read (7/4) + z write, two FFT levels in place, split read
            n_cu, clk = 256, 2.4e9
            l1_peak, lds_peak = n_cu * 64.0 * clk / 1e9, n_cu * 128.0 * clk / 1e9
            onchip = {'kernel': dom_kernel, 'vector_l1_return': {'bytes_per_launch': l1_b, 'achieved': l1_b / (k_ms * 1e-3) / 1e9, 'peak': l1_peak, 'unit': 'GB/s',
                                                                'frac': l1_b / (k_ms * 1e-3) / 1e9 / l1_peak},
                      'lds': {'bytes_per_launch': lds_b, 'achieved': lds_b / (k_ms * 1e-3) / 1e9, 'peak': lds_peak, 'unit': 'GB/s', 'frac': lds_b / (k_ms * 1e-3) / 1e9 / lds_peak},
                      'note': 'byte model of DESIGN 5.3 (not counters); peaks = 256 CUs x 64 (128) B/clk x 2.4 GHz; the two phases alternate inside a block, '
                              'so the sum of the two fractions is the share of the kernel time either resource is busy at best overlap'}
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
                       'walkers_per_gpu': W, 'S': S, 'N': args.N, 'B': pb.B, 'fft_pad': ctx.fft_pad,
                       'chunk': ctx.chunk, 'route': ctx.route, 'conv': ctx.conv, 'conv_layout': ctx.conv_layout, 'parallelism': 'walkers sharded x%d' % world, 'device': ctx.device_name},
            # the time-dominant kernel of the step, on the bytes it has to move (inputs once + outputs once)
            'roofline': {'kernel': dom_kernel, 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': pmc_traffic(pj, dom_kernel, walkers_per_launch),
                         'launch_ms': k_ms, 'launch_ms_source': 'HIP events around this kernel inside the timed region' if dom == 'beam_fft_ms' else 'HIP events of the stage pass',
                         'bytes_per_launch': dom_bytes, 'share_of_step': k_ms / max(1e-12, ms_step * walkers_per_launch / W),
                         'survey_8d_bytes_per_launch': survey_bytes,
                         'survey_8d_GBps': survey_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0,
                         'note': 'achieved = bytes this kernel must move (walker inputs read once, outputs written once: DESIGN 5) / its '
                                 'HIP-event duration; survey_8d_* = SURVEY 8(d)\'s S^2*8 B per walker for the map stage, which this route '
                                 'never writes (the rows are evaluated from the spline inside this kernel). The kernel is not HBM-bound: '
                                 'its floor is on-chip data movement (vector-L1 return path 64 B/clk, LDS 128 B/clk; DESIGN 5.3)'},
            # what does bound the time-dominant kernel: on-chip data movement, priced at nominal peaks
            'roofline_onchip': onchip,
            # the whole step against the HBM roofline: measured bytes (rocprofv3 PMC, profiles/*_pmc_traffic.json) and compulsory bytes
            'roofline_step': {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'ms_per_step': ms_step,
                              'traffic_bytes_per_step': step_pmc,
                              'achieved': (step_pmc / (ms_step * 1e-3) / 1e9) if step_pmc else None,
                              'frac': (step_pmc / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_pmc else None,
                              'must_move_bytes_per_step': step_must,
                              'survey_8d_bytes_per_step': 2.0 * W * S * S * 8.0,
                              'time_dominant_kernel': dom_kernel},
            'north_star_abel_map_kernel': full_map,
            'fp32_variant': f32,
            'cpu_baseline': cpu,
            'stage_ms_per_step': {k: tm[k] / args.steps for k in
                                  ('prep_ms', 'abel_map_ms', 'beam_fft_ms', 'tf_fft_ms', 'gemm_ms', 'tail_ms', 'total_ms')},
            'stage_ms_note': 'HIP events behind every stage, from a second pass of the same steps right after the timed region '
                             '(the timed region itself carries only the two events around the time-dominant kernel)' if p1_only else None,
            'stage_kernels': stage_kernels if lay.get('fused') else None,
            'parity_max_rel_err': parity,
            'operator_route': also,
        }
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    post.close()


if __name__ == '__main__':
    main()
