"""Walker sharding across GPUs: one process per GPU, one context per process.

Walkers are independent (the reference maps them over a process pool,
joxsz_main.py:203-206), so rank g evaluates walkers [lo_g, hi_g) and the only
exchange step is an all-gather of the log-probabilities -- RCCL over xGMI with
the ``nccl`` backend on GPUs, ``gloo`` on CPUs (tests).  The message is a few
KiB per rank: latency-bound, no data-path collective besides it.

``torch.distributed`` is used as plumbing only (rendezvous + the collective);
it is imported lazily so that single-GPU use needs no torch at all.
"""
import numpy as np


def shard_bounds(nwalkers, world, rank):
    """Contiguous, balanced partition: the first ``nwalkers % world`` ranks get one more."""
    base, rem = divmod(int(nwalkers), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedLogProb:
    """Callable ``theta[W, ndim] -> logp[W]`` valid on every rank.

    ``evaluate`` is the local batched evaluator (``JoxszPosterior.log_prob`` on a
    GPU rank).  Every rank must call with the same ``theta`` (emcee's proposal is
    deterministic given the shared seed, or rank 0 broadcasts it with
    ``broadcast_theta``)."""

    def __init__(self, evaluate, group=None, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised (launch with torch.distributed.run)')
        self._torch, self._dist = torch, dist
        self.evaluate = evaluate
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if device is None:
            device = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
        self.device = device

    def broadcast_theta(self, theta, src=0):
        t = self._torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64)).to(self.device)
        self._dist.broadcast(t, src=src, group=self.group)
        return t.cpu().numpy()

    def __call__(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if theta.ndim == 1:
            theta = theta[None, :]
        W = theta.shape[0]
        lo, hi = shard_bounds(W, self.world, self.rank)
        local = np.asarray(self.evaluate(theta[lo:hi]), dtype=np.float64) if hi > lo else np.zeros(0)
        width = -(-W // self.world)                       # uniform slot per rank (ragged shards padded)
        buf = np.full(width, -np.inf)
        buf[:hi - lo] = local
        torch, dist = self._torch, self._dist
        mine = torch.from_numpy(buf).to(self.device)
        out = torch.empty(width * self.world, dtype=torch.float64, device=self.device)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        out = out.cpu().numpy()
        full = np.empty(W)
        for r in range(self.world):
            a, b = shard_bounds(W, self.world, r)
            full[a:b] = out[r * width:r * width + (b - a)]
        return full
