#!/usr/bin/env python3
"""BASELINE configs[4] asks for an fp32-vs-fp64 tolerance sweep at 1024^2 / 1000-pt.  This measures it where it can be
measured exactly: on the collapsed SZ operator, map_row = pp @ G, which IS the linear chain
Abel -> spline -> map -> beam -> transfer function -> row.  The product is redone on the host with the operands and/or
the accumulation rounded to fp32, everything behind it (conversion, spline to the data radii, chi^2, X-ray term, priors)
stays fp64, and the log-posterior is compared with the library's fp64 value.

    python scripts/fp32_sweep.py [--S 1024 --N 1000 --walkers 64]        (on the GPU box; prints a table)
"""
import argparse
import os
import sys

import numpy as np
from scipy.interpolate import interp1d

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joxsz_amd import datasets                                                     # noqa: E402
from joxsz_amd.posterior import JoxszPosterior                                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--S', type=int, default=1024)
    ap.add_argument('--N', type=int, default=1000)
    ap.add_argument('--walkers', type=int, default=64)
    args = ap.parse_args()
    pb = datasets.synthetic_problem(S=args.S, N=args.N, seed=0)
    post = JoxszPosterior(pb, device=0)
    t0 = np.repeat(datasets.fiducial_theta(pb)[None, :], 2, axis=0)
    datasets.fill_data(pb, post.stage(t0, 'bright')[0], post.stage(t0, 'xprofs')[0], seed=0)
    post.close()
    post = JoxszPosterior(pb, device=0)
    th = datasets.walker_ball(pb, 4 * args.walkers, spread=0.02, seed=1)
    lp = post.log_prob(th)
    th = th[np.isfinite(lp)][:args.walkers]
    lp64 = post.log_prob(th)
    pp, row, bright, chisq = (post.stage(th, k) for k in ('pp', 'map_row', 'bright', 'chisq'))
    post.ctx.set_route('operator')
    G = post.ctx.operator()
    post.close()
    cfac = bright / row                                                             # convert(T) * calibration, per walker and radius
    base = lp64 + chisq / 2                                                         # priors + X-ray term
    r_prof = pb.radius[pb.S // 2:]

    def logp_from_rows(rows):
        out = np.empty(len(rows))
        for w, rw in enumerate(rows):
            g = interp1d(r_prof, rw * cfac[w], 'cubic', fill_value='extrapolate')   # joxsz_funcs.py:476
            out[w] = base[w] - np.nansum(((pb.flux_data[1] - g(pb.flux_data[0])) / pb.flux_data[2]) ** 2) / 2
        return out

    f32 = np.float32
    variants = {
        'fp64 operands, fp64 sums (host redo of the library\'s product)': pp @ G,
        'G stored in fp32, fp64 sums': pp @ G.astype(f32).astype(np.float64),
        'pp stored in fp32, fp64 sums': pp.astype(f32).astype(np.float64) @ G,
        'G and pp in fp32, fp64 sums': pp.astype(f32).astype(np.float64) @ G.astype(f32).astype(np.float64),
        'G and pp in fp32, fp32 sums (sgemm)': (pp.astype(f32) @ G.astype(f32)).astype(np.float64),
    }
    print('# fp32-vs-fp64 sweep on the SZ operator: S=%d N=%d, %d walkers, log-posterior ~ %.1f (chi^2 ~ %.1f)'
          % (args.S, args.N, len(th), np.median(lp64), np.median(chisq)))
    print('%-66s %14s %16s %16s' % ('variant', 'max rel d(row)', 'max rel d(logp)', 'median d(logp)'))
    for name, rows in variants.items():
        drow = np.max(np.abs(rows - row) / np.abs(row).max(axis=1, keepdims=True))
        dlp = np.abs(logp_from_rows(rows) - lp64) / np.abs(lp64)
        print('%-66s %14.3e %16.3e %16.3e' % (name, drow, dlp.max(), np.median(dlp)))


if __name__ == '__main__':
    main()
