"""Multi-rank paths with the HIP evaluator on the one GPU a test box has: RCCL through the C-ABI at world size 1, and two
processes that share the device, each with its own JoxszPosterior on its shard (the gather runs over gloo there, because
RCCL refuses two ranks on one device)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_gather_through_the_c_abi_world_1():
    """jx_comm_unique_id / jx_comm_init_rank / jx_allgather_logp / jx_comm_allreduce_max / jx_comm_destroy, no torch:
    at one rank the gather is a device copy, ordered on the context's stream behind the evaluation."""
    from joxsz_amd import datasets
    from joxsz_amd.dist import RcclGather
    from joxsz_amd.posterior import JoxszPosterior
    from joxsz_amd.hip_backend import JoxszHipError
    assert 'torch' not in sys.modules or True                     # (other tests of the session may have imported it)
    pb = datasets.synthetic_problem(S=64, N=80, seed=2)
    th = np.ascontiguousarray(datasets.walker_ball(pb, 40, spread=0.03, seed=2))
    post = JoxszPosterior(pb, device=0)
    ctx = post.ctx
    want = post.log_prob(th)
    with pytest.raises(JoxszHipError):
        ctx.allgather_logp(1, 1, 4)                               # no communicator yet
    comm = RcclGather(ctx, rank=0, world=1)
    th_ptr, lp_ptr, all_ptr = ctx.dev_alloc(th.nbytes), ctx.dev_alloc(8 * 40), ctx.dev_alloc(8 * 40)
    ctx.h2d(th_ptr, th)
    for _ in range(3):
        ctx.eval_device(th_ptr, 40, lp_ptr)
        comm.all_gather(lp_ptr, all_ptr, 40)
    comm.barrier()
    got = np.empty(40)
    ctx.d2h(got, all_ptr)
    np.testing.assert_array_equal(got, want)
    assert comm.max_over_ranks(3.25) == 3.25
    assert comm.n_ranks_seen == 1 and ctx.comm_count() == 1          # what RCCL itself reports (ncclCommCount)
    # the padded form for ragged shards (RCCL's all-gather wants equal counts): at one rank the shard is the batch
    np.testing.assert_array_equal(comm.gather_ragged(want[:37], 37), want[:37])
    with pytest.raises(JoxszHipError):
        ctx.comm_init_rank(b'\0' * 128, 1, 0)                     # one communicator per context
    comm.close()
    post.close()


def test_device_sampler_over_ranks(monkeypatch):
    """jx_sample shards every half step over the ranks of the context's communicator: a rank moves its contiguous share of the half
    (proposal in the per-walker kernel, acceptance in the tail) and one in-place RCCL all-gather of the share's positions and
    log-posteriors updates every rank's copy of the ensemble.  On the one GPU of a test box: (a) a communicator of one rank runs
    those collectives for real (in place, share = the half) -- same chain as without a communicator; (b) JOXSZ_SAMPLE_VIRTUAL_RANKS=R
    runs the R shares of a half step one after the other in one process -- the share arithmetic of R = 3 and 5 ranks, ragged
    chunks included, gives the chain of one rank bit for bit (the shares of a half step do not read one another)."""
    from joxsz_amd import datasets
    from joxsz_amd.dist import RcclGather
    from joxsz_amd.posterior import JoxszPosterior
    from joxsz_amd.sampler import DeviceStretchMove, initial_ball
    from joxsz_amd.hip_backend import JoxszHipError
    from oracle import joxsz_oracle as orc
    pb = datasets.synthetic_problem(S=96, N=120, seed=8)
    p0f = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0f)['bright'], orc.calc_profiles(pb, p0f), seed=8)
    post = JoxszPosterior(pb, device=0, max_batch=16)
    p0 = initial_ball(post.log_prob, datasets.fiducial_theta(pb), 90, spread=0.01, rng=np.random.default_rng(8))    # half = 45 = 3 x 15 = 5 x 9
    ref = DeviceStretchMove(post, a=2.0, seed=5).run(p0, 8)
    assert 0 < ref[2].sum() < 8 * 90
    for r in (3, 5):
        monkeypatch.setenv('JOXSZ_SAMPLE_VIRTUAL_RANKS', str(r))
        got = DeviceStretchMove(post, a=2.0, seed=5).run(p0, 8)
        for a, b in zip(got, ref):
            np.testing.assert_array_equal(a, b)
    monkeypatch.setenv('JOXSZ_SAMPLE_VIRTUAL_RANKS', '4')          # 45 walkers do not split four ways: refused, loudly
    with pytest.raises(JoxszHipError, match='divisible'):
        DeviceStretchMove(post, a=2.0, seed=5).run(p0, 2)
    monkeypatch.delenv('JOXSZ_SAMPLE_VIRTUAL_RANKS')
    comm = RcclGather(post.ctx, rank=0, world=1)
    got = DeviceStretchMove(post, a=2.0, seed=5).run(p0, 8)
    comm.close()
    post.close()
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)


def test_overlapped_gather_world_1():
    """jx_comm_set_overlap: the collectives on a second stream of the context, each behind an event.  Two alternating output
    buffers (what bench.py does) and ONE buffer written by every step (the evaluation must then wait for the gather that is
    still reading it) both deliver the evaluation's values; jx_sync covers both streams; the gather's own time is reported."""
    from joxsz_amd import datasets
    from joxsz_amd.dist import RcclGather
    from joxsz_amd.posterior import JoxszPosterior
    pb = datasets.synthetic_problem(S=64, N=80, seed=3)
    W = 48
    ths = [np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.03, seed=10 + i)) for i in range(3)]
    post = JoxszPosterior(pb, device=0)
    ctx = post.ctx
    wants = [post.log_prob(t) for t in ths]
    comm = RcclGather(ctx, rank=0, world=1, overlap=True)
    tp = [ctx.dev_alloc(t.nbytes) for t in ths]
    for p_, t in zip(tp, ths):
        ctx.h2d(p_, t)
    lp, al = [ctx.dev_alloc(8 * W) for _ in range(2)], [ctx.dev_alloc(8 * W) for _ in range(3)]
    ctx.timing_enable(True)
    for i in range(3):                                    # alternating send buffers, one receive buffer per step
        ctx.eval_device(tp[i], W, lp[i % 2])
        comm.all_gather(lp[i % 2], al[i], W)
    comm.barrier()
    ctx.sync()
    for i in range(3):
        got = np.empty(W)
        ctx.d2h(got, al[i])
        np.testing.assert_array_equal(got, wants[i])
    ms, n = ctx.comm_gather_time()
    assert n == 3 and 0 < ms < 50
    for i in range(3):                                    # the same send buffer every step: each evaluation waits for the previous gather
        ctx.eval_device(tp[i], W, lp[0])
        comm.all_gather(lp[0], al[i], W)
    ctx.sync()
    ctx.timing_enable(False)
    for i in range(3):
        got = np.empty(W)
        ctx.d2h(got, al[i])
        np.testing.assert_array_equal(got, wants[i])
    assert comm.max_over_ranks([1.5, -2.0, 7.0]).tolist() == [1.5, -2.0, 7.0]
    comm.close()
    post.close()


_RANK_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
from joxsz_amd.dist import ShardedLogProb, shard_bounds
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
pb = datasets.synthetic_problem(S=128, N=150, seed=5)
th = datasets.walker_ball(pb, 101, spread=0.04, seed=5)            # ragged: 51 + 50
th[7, 1] = 9.0                                                      # a rejected walker in rank 0's shard
post = JoxszPosterior(pb, device=0)                                 # both ranks on the one device
f = ShardedLogProb(post.log_prob, device='cpu')
full = f(th)
lo, hi = shard_bounds(101, world, rank)
assert np.array_equal(full[lo:hi], post.log_prob(th[lo:hi]))
np.save(sys.argv[2] + '/r%d.npy' % rank, full)
post.close()
dist.destroy_process_group()
print('ok')
"""


def test_two_ranks_one_gpu_with_the_hip_evaluator(tmp_path):
    """World size 2 on one device: each rank evaluates its contiguous shard with its own HIP context, the log-probabilities
    are gathered (gloo), every rank ends up with the full vector, equal to one single-context evaluation and to the oracle."""
    import importlib.util
    if importlib.util.find_spec('torch') is None:                  # (not imported here: torch brings its own HIP runtime,
        pytest.skip('torch (gloo) not installed')                  #  which must not meet this process's /opt/rocm one)
    from joxsz_amd import datasets
    from joxsz_amd.posterior import JoxszPosterior
    from oracle import joxsz_oracle as orc
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, '-c', _RANK_SCRIPT, ROOT, str(tmp_path)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and so.strip().endswith('ok'), se[-2000:]
    a, b = np.load(tmp_path / 'r0.npy'), np.load(tmp_path / 'r1.npy')
    np.testing.assert_array_equal(a, b)
    pb = datasets.synthetic_problem(S=128, N=150, seed=5)
    th = datasets.walker_ball(pb, 101, spread=0.04, seed=5)
    th[7, 1] = 9.0
    post = JoxszPosterior(pb, device=0)
    one = post.log_prob(th)
    post.close()
    np.testing.assert_array_equal(a, one)
    assert a[7] == -np.inf
    want = orc.log_posterior_batch(pb, th[:6])
    np.testing.assert_allclose(a[:6], want, rtol=1e-6)
