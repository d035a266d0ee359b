"""The bench line committed with the round (profiles/r04_bench.json, written by bench.py on an MI355X) carries every field of
the driver's contract and of SURVEY 8(d): metric / config of BASELINE.json, roofline and cpu_baseline objects, figures that
hang together."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, 'profiles', name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_committed_bench_line_follows_the_contract():
    d = _line('r04_bench.json')
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert d['metric'].split(';')[0] == base['metric'].split(';')[0].replace('²', '^2')
    for k in ('value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f64' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert '1024 walkers' in d['config']['workload'] and '512x512' in d['config']['workload']
    # value = walkers per step / time per step
    assert abs(d['value'] - d['config']['walkers_per_gpu'] / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    assert d['value'] >= 10000                                            # north_star's target on one MI355X
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'launch_ms'):
        assert k in r, k
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['frac'] < 1 and r['traffic'] is not None
    # achieved = algorithmic flops of the launch / the kernel's measured duration
    assert abs(r['achieved'] - r['flops_per_launch'] / (r['launch_ms'] * 1e-3) / 1e12) < 1e-6 * r['achieved']
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['unit'] == d['unit'] and c['value'] > 0
    assert 'single_process' in c and 'stage_ms_per_call' in c
    assert d['parity_max_rel_err'] <= 1e-6                                # north_star's tolerance, checked inside the run
    ns = d['north_star_abel_map_kernel']
    assert ns['frac_of_hbm_peak'] >= 0.60                                 # north_star: >= 60 % of the HBM roofline in the Abel+map kernel
    assert abs(ns['achieved_GBps'] - ns['bytes_per_launch'] / (ns['launch_ms'] * 1e-3) / 1e9) < 1e-6 * ns['achieved_GBps']
    t = d['truncation']
    assert t['points'] >= 9 and 0 <= t['est_rel_row_err'] <= t['bound'] and 0 <= t['est_rel_sz_like_err_box'] <= 1e-8
    assert d['n_ranks_seen'] == 1


def _fracs(o, path=''):
    if isinstance(o, dict):
        for k, v in o.items():
            if k == 'frac' or k.startswith('frac_of'):
                yield path + '/' + k, v
            else:
                yield from _fracs(v, path + '/' + k)


def test_every_roofline_figure_of_the_line_follows_from_the_profiles():
    """roofline_step summed the stream micro-benchmarks of the PMC file (frac 2.71).  Every fraction of the line is
    now in (0, 1]; the step's HBM bytes are exactly the five kernels of the step in the committed PMC file of this round; the
    product's flops count the outputs it computes; the two objects the judge asked for are there."""
    d = _line('r04_bench.json')
    fr = dict(_fracs(d))
    assert len(fr) >= 6
    for k, v in fr.items():
        assert v is None or 0 < v <= 1.0, (k, v)
    rs = d['roofline_step']
    assert rs['traffic_source'] == 'profiles/r04_pmc_traffic.json' and rs['traffic_measured_in_this_run'] is False
    pj = json.load(open(os.path.join(ROOT, rs['traffic_source'])))
    names = {'jx_prep_kernel', 'jx_abel_gemm_kernel', 'jx_rowmix_kernel', 'jx_opgemm_kernel', 'jx_tail_row_kernel'}
    assert set(rs['traffic_by_kernel']) == names
    import bench
    want = sum(bench.pmc_kernel_entry(pj, n)['total_bytes'] for n in names)      # (the full-size instance of each kernel template)
    assert abs(rs['traffic_bytes_per_step'] - want) <= 1e-9 * want and abs(sum(rs['traffic_by_kernel'].values()) - want) <= 1e-9 * want
    assert 0.05 < rs['frac'] < 0.5 and abs(rs['frac'] - rs['traffic_bytes_per_step'] / (rs['ms_per_step'] * 1e-3) / 1e9 / rs['peak']) < 1e-9
    rp = d['roofline_product']
    assert rp['outputs_computed'] == d['config']['output_pruning']['outputs_computed'] <= rp['outputs_of_the_row']
    assert abs(rp['achieved'] - rp['flops_per_launch'] / (rp['launch_ms'] * 1e-3) / 1e12) < 1e-6 * rp['achieved']
    ns = d['north_star_route']                                           # north_star's literal design as a whole step
    assert 'rocFFT' in ns['route'] and ns['walkers_per_launch'] == 1024 and ns['ms_per_step'] > d['ms_per_step'] and ns['max_rel_diff_vs_default_route'] < 1e-9
    assert abs(ns['speedup_of_default_route'] - ns['ms_per_step'] / d['ms_per_step']) < 1e-9 * ns['speedup_of_default_route']
    hp = d['host_pointer']                                               # the path emcee calls
    assert hp['max_abs_diff_vs_device_resident'] == 0.0 and 0.5 * d['value'] < hp['value'] <= 1.02 * d['value']
    ds = d['device_sampler']                                            # the caller of the path: the device-resident stretch-move loop
    assert ds['finite'] and 0.2 < ds['acceptance'] < 0.6 and 0.7 * d['value'] < ds['value'] < 1.1 * d['value']
    fv = d['fp32_variant']
    assert fv['f32']['rel_dlogp_vs_f64']['max'] < 1e-8 and fv['f32c']['rel_dlogp_vs_f64']['max'] < 1e-6 and fv['f32c']['speedup_vs_f64'] > 1.05
    oc = d['other_configs']
    assert 'error' not in oc and all(oc[k]['value'] > 0 for k in ('configs[3]', 'configs[4] f64', 'configs[4] f32c'))


def test_the_rehearsed_n_gt_1_lines_carry_the_gather_time():
    for name, mode in (('r04_bench_force_dist_overlap.json', 'overlapped'), ('r04_bench_force_dist_strict.json', 'strict'), ('r04_bench_force_dist.json', None)):
        f = _line(name)
        assert f['n_ranks_seen'] == 1 and 0 < f['gather_ms_per_step'] < 0.05 and f['gather_ms_per_step'] < 0.2 * f['ms_per_step']
        pr = f['config']['gather_probe_ms_per_step']
        if mode is not None:
            assert f['config']['gather'].startswith(mode) and pr is None
        else:
            # no mode pinned: the warm-up ran both and the timed region took the faster (the ranks agree through one max-reduction each)
            assert set(pr) == {'overlapped', 'strict'} and f['config']['gather'].startswith(min(pr, key=pr.get))


def test_rocprof_kernel_statistics_agree_with_the_bench_line():
    """profiles/r04_kernel_stats.csv (rocprofv3 --kernel-trace of the same command, full-size launches only): the dominant
    kernel's average duration within 10 % of the HIP-event duration the roofline is computed from."""
    import csv
    d = _line('r04_bench.json')
    rows = list(csv.DictReader(open(os.path.join(ROOT, 'profiles', 'r04_kernel_stats.csv'))))
    # (since the sub-grid of stage 1 no kernel dominates: the per-walker kernel, stage 1 and the spline-array product lie within 15 %
    # of one another; `roofline` stays with stage 1, the one with the arithmetic, and `step_kernels` lists all five)
    top = [r for r in rows[:3] if d['roofline']['kernel'] in r['Name']][0]
    assert float(top['AverageUs']) >= 0.90 * float(rows[0]['AverageUs'])
    lk = d['roofline']['longest_kernel_of_the_step']
    assert lk['kernel'] in rows[0]['Name'] or abs(lk['ms_hip_events_stage_pass'] - max(k['ms_hip_events_stage_pass'] for k in d['step_kernels'])) < 1e-12
    sk = {k['kernel']: k for k in d['step_kernels']}
    assert len(sk) == 5 and all(0 < k['share_of_wave_cycles']['issuing_valu'] < 1 for k in d['step_kernels'])
    for r in rows[:5]:
        name = [k for k in sk if k in r['Name']][0]
        ev = sk[name]['ms_hip_events_stage_pass']
        assert float(r['AverageUs']) * 1e-3 <= ev * 1.02 and ev <= float(r['AverageUs']) * 1e-3 + d['roofline']['launch_ms_of_an_empty_kernel'] + 0.0015, (name, ev, r['AverageUs'])
    # (a pair of HIP events around a kernel of a dependent chain also spans the hand-over in front of and behind it: what the pair
    # reads around an empty kernel is reported beside launch_ms; rocprofv3 counts the kernel alone)
    roof = d['roofline']
    assert 0 < roof['launch_ms_of_an_empty_kernel'] < 0.010
    assert roof['launch_ms'] - roof['launch_ms_of_an_empty_kernel'] - 0.05 * roof['launch_ms'] <= float(top['AverageUs']) * 1e-3 <= roof['launch_ms'] * 1.02
    # the kernels of the step add up to the step (launch gaps excluded)
    step_us = sum(float(r['AverageUs']) for r in rows if int(r['FullSizeCalls']) >= 100)
    assert 0.85 * d['ms_per_step'] * 1e3 <= step_us <= 1.02 * d['ms_per_step'] * 1e3
    f = _line('r04_bench_force_dist.json')
    assert f['n_ranks_seen'] == 1 and abs(f['value'] - d['value']) <= 0.12 * d['value']      # the N > 1 plumbing at N = 1: the gather and its events
