"""Walker sharding across GPUs: one process per GPU, one context per process.

Walkers are independent (the reference maps them over a process pool,
joxsz_main.py:203-206), so rank g evaluates walkers [lo_g, hi_g) and the only
exchange step is an all-gather of the log-probabilities -- RCCL over xGMI with
the ``nccl`` backend on GPUs, ``gloo`` on CPUs (tests).  The message is a few
KiB per rank: latency-bound, no data-path collective besides it.

Two carriers of that one exchange:

* ``RcclGather`` -- the GPU path: RCCL through the library's own C-ABI (``jx_comm_*``, ``jx_allgather_logp``), device
  buffers, enqueued on the context's stream behind the evaluation.  No torch anywhere; the 128-byte RCCL id travels from
  rank 0 to the others through a file (one node) -- ``exchange_unique_id``.
* ``ShardedLogProb`` -- ``torch.distributed`` with the ``gloo`` backend, for the CPU tests of the sharding logic and for
  ranks that share one device (RCCL refuses two ranks on one GPU).  torch is imported lazily, only there.
"""
import os
import time

import numpy as np

_PROCESS_T0 = time.time()


def launch_env():
    """(rank, world, local_rank) from the environment ``torch.distributed.run`` / any MPI-style launcher sets."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    return rank, world, local_rank


def _rdzv_path():
    d = os.environ.get('JOXSZ_RDZV_DIR', '/tmp')
    # (the restart count: an id file left by an attempt of the same run that died between publishing and joining is not this attempt's)
    tag = os.environ.get('JOXSZ_RDZV_TAG') or '%s_%s_%d_r%s' % (os.environ.get('TORCHELASTIC_RUN_ID', 'run'), os.environ.get('MASTER_PORT', '0'),
                                                                os.getppid(), os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'))
    return os.path.join(d, 'joxsz_rccl_%s.id' % tag)


def _launch_start_time():
    """A time that precedes the start of every rank of this launch and follows every earlier launch that could have used
    the same rendezvous path: ``JOXSZ_RDZV_T0`` when the launcher sets it (bench.py's own does), else the start time of the
    parent process -- the launcher all ranks share (torch.distributed.run's agent, mpirun, a shell), whose pid is part of the
    default path.  NOT this process's own start: ranks come up seconds apart (a cold python import on a fresh box takes
    a minute), and a slow rank must still accept the file rank 0 wrote before it was up."""
    t0 = os.environ.get('JOXSZ_RDZV_T0')
    if t0:
        return float(t0)
    try:
        with open('/proc/%d/stat' % os.getppid()) as f:
            ticks = float(f.read().rsplit(')', 1)[1].split()[19])          # field 22: start time in clock ticks since boot
        with open('/proc/stat') as f:
            btime = next(float(l.split()[1]) for l in f if l.startswith('btime'))
        return btime + ticks / os.sysconf('SC_CLK_TCK')
    except Exception:
        return _PROCESS_T0 - 600.0                                         # no /proc: ten minutes of grace


def exchange_unique_id(make_id, rank, world, timeout=600.0, not_before=None):
    """Rank 0 calls ``make_id()`` (``jx_comm_unique_id``) and publishes the 128 bytes; the others read them.
    One node: a file under ``JOXSZ_RDZV_DIR`` (default /tmp) named after ``JOXSZ_RDZV_TAG`` (bench.py's own launcher sets a
    fresh one per launch) or the launcher's run id, MASTER_PORT and pid, written atomically (rename).  A file left behind
    by an earlier launch with the same name is never taken for this launch's: rank 0 removes exactly that path before it
    writes, and a reader only accepts a file whose modification time is not older than ``not_before`` -- the launch's
    start time (``_launch_start_time``: ``JOXSZ_RDZV_T0`` from the launcher, or the start of the parent process)."""
    if world == 1:
        return make_id()
    path = _rdzv_path()
    if not_before is None:
        not_before = _launch_start_time()
    if rank == 0:
        try:
            os.unlink(path)                                   # this launch's path only, never a glob
        except FileNotFoundError:
            pass
        uid = make_id()
        tmp = path + '.tmp%d' % os.getpid()
        with open(tmp, 'wb') as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while True:
        try:
            st = os.stat(path)
            if st.st_mtime >= not_before - 1.0:               # (1 s: file systems that round modification times down)
                with open(path, 'rb') as f:
                    uid = f.read()
                if len(uid) == 128:
                    return uid
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise RuntimeError('rank %d: no RCCL id from rank 0 at %s after %.0f s' % (rank, path, timeout))
        time.sleep(0.02)


class RcclGather:
    """All-gather of per-rank log-probabilities over RCCL, through the C-ABI of the HIP library (no torch).

    ``ctx`` is this rank's ``HipContext`` (one per process, one process per GPU).  ``all_gather(send_ptr, recv_ptr, n)``
    takes device pointers (``ctx.dev_alloc``) and is asynchronous, ordered behind ``ctx.eval_device`` -- on the context's
    stream (strict, default) or with ``overlap=True`` on a second stream, so that the next evaluation (into another buffer)
    runs beside it; ``barrier()`` and ``max_over_ranks(x)`` serve the timing protocol of bench.py;
    ``gather_ragged`` is the padded form for shards of unequal length."""

    def __init__(self, ctx, rank=None, world=None, overlap=False):
        env_rank, env_world, _ = launch_env()
        self.ctx = ctx
        self.overlap = bool(overlap)
        self.rank = env_rank if rank is None else rank
        self.world = env_world if world is None else world
        uid = exchange_unique_id(ctx.comm_unique_id, self.rank, self.world)
        ctx.comm_init_rank(uid, self.world, self.rank)
        self._scratch = ctx.dev_alloc(8)
        self._scratch_n = 1
        self._pad = None
        self.n_ranks_seen = ctx.comm_count()                  # what RCCL itself says (ncclCommCount)
        if self.overlap:
            # the gather of step n on a second stream beside step n+1's kernels (the caller alternates two output buffers);
            # strict mode (default) keeps every collective in order on the compute stream
            ctx.comm_set_overlap(True)
        if self.rank == 0 and self.world > 1:
            try:                                              # every rank has read the id once the communicator exists
                os.unlink(_rdzv_path())
            except OSError:
                pass

    def all_gather(self, send_ptr, recv_ptr, count):
        self.ctx.allgather_logp(send_ptr, recv_ptr, count)

    def gather_ragged(self, local_logp, nwalkers):
        """All ranks' log-probabilities of a batch of ``nwalkers`` sharded by ``shard_bounds`` (shards differ by at most one
        walker): ``local_logp`` is this rank's shard (host array).  RCCL's all-gather wants equal counts, so every rank sends
        a slot of ceil(nwalkers / world) values, the unused tail of a short shard filled with -inf; returns the full vector."""
        width = -(-int(nwalkers) // self.world)
        lo, hi = shard_bounds(nwalkers, self.world, self.rank)
        assert len(local_logp) == hi - lo
        if self._pad is None or self._pad[0] < width:
            self._pad = (width, self.ctx.dev_alloc(8 * width), self.ctx.dev_alloc(8 * width * self.world))
        buf = np.full(width, -np.inf)
        buf[:hi - lo] = local_logp
        self.ctx.h2d(self._pad[1], buf)
        self.all_gather(self._pad[1], self._pad[2], width)
        self.ctx.sync()
        out = np.empty(width * self.world)
        self.ctx.d2h(out, self._pad[2])
        full = np.empty(int(nwalkers))
        for r in range(self.world):
            a, b = shard_bounds(nwalkers, self.world, r)
            full[a:b] = out[r * width:r * width + (b - a)]
        return full

    def max_over_ranks(self, value):
        """Maximum over the ranks of a number, or element-wise of a sequence of numbers (one collective either way)."""
        scalar = np.ndim(value) == 0
        buf = np.atleast_1d(np.asarray(value, dtype=np.float64)).copy()
        if buf.size > self._scratch_n:
            if self._scratch_n:
                self.ctx.dev_free(self._scratch)              # (the vector grew: the old scratch goes back)
            self._scratch_n = int(buf.size)
            self._scratch = self.ctx.dev_alloc(8 * self._scratch_n)
        self.ctx.h2d(self._scratch, buf)
        self.ctx.comm_allreduce_max(self._scratch, int(buf.size))
        self.ctx.sync()
        self.ctx.d2h(buf, self._scratch)
        return float(buf[0]) if scalar else buf

    def barrier(self):
        self.max_over_ranks(0.0)

    def close(self):
        self.ctx.comm_destroy()


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
