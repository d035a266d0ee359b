#!/usr/bin/env python3
"""Soak of the timed path: the device-resident stretch-move loop for many steps (every step = two launches of the three kernels with the proposal
and the acceptance inside), then the live walkers audited against the literal sequence, and the same walkers evaluated twice more for bitwise
repeatability.      python3 scripts/soak.py [steps]      (GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
pb = datasets.synthetic_problem(S=512, N=500, seed=0)
post = JoxszPosterior(pb, device=0)
cand = np.ascontiguousarray(datasets.walker_ball(pb, 4096, spread=0.02, seed=1))
p0 = np.ascontiguousarray(cand[np.isfinite(post.log_prob(cand))][:2048])         # (jx_sample wants start positions with a finite log-posterior)
assert p0.shape[0] == 2048
t = time.perf_counter()
done, x, acc = 0, p0, []
while done < steps:
    k = min(2000, steps - done)
    chain, lps, nacc = post.sample(x, k, seed=done)
    x = np.ascontiguousarray(chain[-1])
    acc.append(float(np.mean(nacc)) / k)
    lp = lps[-1]
    assert np.all(np.isfinite(lp)), 'non-finite log-posterior among live walkers at step %d' % (done + k)
    done += k
    print('steps %6d  acceptance %.3f  logp %.2f .. %.2f  %.1f s' % (done, acc[-1], lp.min(), lp.max(), time.perf_counter() - t), flush=True)
a = post.ctx.audit(x)
l1, l2, l3 = post.log_prob(x), post.log_prob(x), post.log_prob(x[::-1].copy())[::-1]
print('audit of the live walkers against the literal sequence:', a)
print('re-evaluation bitwise equal:', bool(np.array_equal(l1, l2)), '| reversed order bitwise equal:', bool(np.array_equal(l1, l3)),
      '| agrees with the sampler\'s own values: max abs diff %.3e' % float(np.max(np.abs(l1 - lp))))
post.close()
