"""The N>1 path on CPU: world_size 2 and 3 over gloo.  The local evaluator is the
oracle here (no GPU in this container); on a GPU rank it is JoxszPosterior.log_prob."""
import os
import socket

import numpy as np
import pytest

from joxsz_amd.dist import shard_bounds


def test_shard_bounds_cover_everything():
    for W in (0, 1, 7, 30, 1024, 4097):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(W, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == W
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from joxsz_amd import datasets
    from joxsz_amd.dist import ShardedLogProb
    from oracle import joxsz_oracle as orc
    pb = datasets.synthetic_problem(S=31, N=40, seed=2, step=6., fwhm=8.5)
    th = datasets.walker_ball(pb, W, spread=0.03, seed=2)
    th[0, 1] = 9.0                                             # a rejected walker travels as -inf
    calls = []

    def evaluate(t):
        calls.append(len(t))
        return orc.log_posterior_batch(pb, t)

    sharded = ShardedLogProb(evaluate)
    th = sharded.broadcast_theta(th if rank == 0 else np.zeros_like(th))
    full = sharded(th)
    q.put((rank, full, calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,W', [(2, 10), (3, 7)])
def test_sharded_logprob_gloo(world, W):
    import torch.multiprocessing as mp
    from joxsz_amd import datasets
    from oracle import joxsz_oracle as orc
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pb = datasets.synthetic_problem(S=31, N=40, seed=2, step=6., fwhm=8.5)
    th = datasets.walker_ball(pb, W, spread=0.03, seed=2)
    th[0, 1] = 9.0
    want = orc.log_posterior_batch(pb, th)
    assert want[0] == -np.inf and np.isfinite(want[1:]).any()
    for rank, full, calls in res:
        np.testing.assert_array_equal(full, want)            # same code on the same inputs: bitwise
        lo, hi = shard_bounds(W, world, rank)
        assert calls == [hi - lo]                             # each rank evaluated only its shard
