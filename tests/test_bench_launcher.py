"""bench.py as its own launcher (`python bench.py --gpus N` without torch.distributed.run) and the file rendezvous of the
RCCL id, on the CPU: stub ranks stand in for the evaluator (the GPU leg is rehearsed on a GPU box with
JOXSZ_BENCH_FORCE_DIST=1, profiles/r03_bench_force_dist.json)."""
import json
import os
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _stub(tmp_path, body):
    p = tmp_path / 'stub_rank.py'
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_spawn_relays_rank0_line_and_exit_codes(tmp_path):
    """Two fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and one rendezvous tag per launch; rank 0's
    last stdout line comes back; a failing or hanging rank makes the launcher exit non-zero and takes the others down."""
    import bench
    stub = _stub(tmp_path, '''
        import os, sys, json, time
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        assert float(os.environ["JOXSZ_RDZV_T0"]) <= time.time() + 1
        if "--fail" in sys.argv and r == 1: sys.exit(3)
        if "--hang" in sys.argv and r == 1: time.sleep(60)
        if r == 0:
            print("noise before the line")
            print(json.dumps({"world": w, "tag": os.environ["JOXSZ_RDZV_TAG"], "args": sys.argv[1:]}))
    ''')
    rc, line = bench.spawn_ranks(2, ['--steps', '3'], script=stub)
    assert rc == 0
    a = json.loads(line)
    assert a['world'] == 2 and a['args'] == ['--steps', '3']
    rc, line = bench.spawn_ranks(2, ['--steps', '3'], script=stub)
    assert rc == 0 and json.loads(line)['tag'] != a['tag']            # a fresh tag per launch
    assert bench.spawn_ranks(2, ['--fail'], script=stub)[0] == 3
    t = time.time()
    assert bench.spawn_ranks(3, ['--hang'], script=stub, timeout_s=1.5)[0] == 124
    assert time.time() - t < 30


def test_bench_becomes_the_launcher_without_one(tmp_path, monkeypatch):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the parent spawns, the children see WORLD_SIZE.
    (The children are bench.py itself and fail here for want of a GPU: the launcher must report that as a failure and must
    not have touched a device itself.)"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['JOXSZ_LIB'] = str(tmp_path / 'no_such_library.so')            # the ranks die at load time, quickly and for certain
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'not found' in r.stderr or 'JoxszHipError' in r.stderr


def test_rendezvous_ignores_a_stale_id_file(tmp_path, monkeypatch):
    """A 128-byte file left at the rendezvous path by an earlier launch is older than this launch: readers wait for rank 0's
    fresh one; rank 0 removes exactly its own path before writing."""
    import multiprocessing as mp
    from joxsz_amd import dist
    monkeypatch.setenv('JOXSZ_RDZV_DIR', str(tmp_path))
    monkeypatch.setenv('JOXSZ_RDZV_TAG', 'same_tag_as_last_time')
    path = dist._rdzv_path()
    with open(path, 'wb') as f:
        f.write(b'S' * 128)
    old = time.time() - 3600
    os.utime(path, (old, old))
    other = os.path.join(str(tmp_path), 'joxsz_rccl_someone_elses.id')
    open(other, 'wb').write(b'O' * 128)
    monkeypatch.setenv('JOXSZ_RDZV_T0', repr(time.time()))

    def reader(q):
        q.put(dist.exchange_unique_id(lambda: b'X' * 128, 1, 2, timeout=20.0))

    q = mp.get_context('fork').Queue()
    p = mp.get_context('fork').Process(target=reader, args=(q,))
    p.start()
    time.sleep(0.5)
    assert q.empty()                                                   # the stale file was not accepted
    assert dist.exchange_unique_id(lambda: b'N' * 128, 0, 2) == b'N' * 128
    got = q.get(timeout=20)
    p.join(20)
    assert got == b'N' * 128
    assert os.path.exists(other)                                       # nobody else's file was touched
    with pytest.raises(RuntimeError):
        os.unlink(path)
        dist.exchange_unique_id(lambda: b'', 1, 2, timeout=0.3)       # no rank 0: a timeout, not a hang


def test_rendezvous_accepts_the_id_a_faster_rank_0_wrote_before_this_rank_was_up(tmp_path, monkeypatch):
    """Without a launcher-provided start time the reference is the PARENT's start (the launcher every rank shares), not the
    reading rank's own: a rank that comes up seconds after rank 0 has already published the id must take it."""
    from joxsz_amd import dist
    monkeypatch.setenv('JOXSZ_RDZV_DIR', str(tmp_path))
    monkeypatch.setenv('JOXSZ_RDZV_TAG', 'slow_rank')
    monkeypatch.delenv('JOXSZ_RDZV_T0', raising=False)
    t_launch = dist._launch_start_time()
    assert t_launch <= dist._PROCESS_T0 + 1.0                             # the parent (pytest's launcher) predates this process
    path = dist._rdzv_path()
    with open(path, 'wb') as f:
        f.write(b'E' * 128)
    early = max(t_launch + 0.5, dist._PROCESS_T0 - 30.0)                  # written after the launch began, before "this rank" was up
    if early < dist._PROCESS_T0:
        os.utime(path, (early, early))
    monkeypatch.setattr(dist, '_PROCESS_T0', time.time() + 5.0)           # this rank: up five seconds from now
    assert dist.exchange_unique_id(lambda: b'', 1, 2, timeout=2.0) == b'E' * 128


_SIDE_STUB = '''
    import os, sys, json, time, glob
    import numpy as np
    sys.path.insert(0, %r)
    import bench
    r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    d = sys.argv[sys.argv.index("--dir") + 1]

    class FileComm:                                   # an all-reduce(max) through files: blocks until every rank has arrived
        def max_over_ranks(self, vec):
            np.save(os.path.join(d, "v%%d.npy" %% r), np.asarray(vec, float))
            t0 = time.time()
            while len(glob.glob(os.path.join(d, "v*.npy"))) < w:
                if time.time() - t0 > 120: raise RuntimeError("collective timed out")
                time.sleep(0.02)
            time.sleep(0.2)
            return np.max([np.load(os.path.join(d, "v%%d.npy" %% k)) for k in range(w)], axis=0)

    headline = {"value": 1.0 + r}                     # fixed before the side measurements
    dts, err = [0.5 + 0.01 * r, 0.25], None
    try:                                              # rank-local, no collective inside
        if "--raise" in sys.argv and r == 3: raise MemoryError("allocation failed building the 8192-walker context")
        if "--die" in sys.argv and r == 3: os._exit(5)
    except Exception as exc:
        dts, err = [np.inf, 0.25], str(exc)
    out = bench.agree_on_side_times(FileComm(), dts)  # outside the handler: every living rank gets here
    if r == 0:
        print(json.dumps({"headline": headline, "side": [None if not np.isfinite(v) else v for v in out]}))
''' % ROOT


def test_a_rank_that_throws_in_the_side_measurements_does_not_hang_the_collective(tmp_path):
    """Eight stand-in ranks run bench.py's side-measurement protocol (rank-local measurement inside try/except, then ONE
    collective outside it).  A rank that raises still reaches the collective: the launch ends with rank 0's line, the failed
    row marked, the headline intact.  A rank that dies outright takes the launch down through the launcher at once -- non-zero
    exit long before the collective's own timeout."""
    import bench
    stub = _stub(tmp_path, _SIDE_STUB)
    d1 = tmp_path / 'a'; d1.mkdir()
    t = time.time()
    rc, line = bench.spawn_ranks(8, ['--raise', '--dir', str(d1)], script=stub, timeout_s=100)
    assert rc == 0 and time.time() - t < 60
    a = json.loads(line)
    assert a['headline'] == {'value': 1.0} and a['side'][0] is None and abs(a['side'][1] - 0.25) < 1e-12
    d2 = tmp_path / 'b'; d2.mkdir()
    t = time.time()
    rc, line = bench.spawn_ranks(8, ['--die', '--dir', str(d2)], script=stub, timeout_s=100)
    assert rc == 5 and time.time() - t < 40            # the others were blocked in the collective; the launcher ended them
    d3 = tmp_path / 'c'; d3.mkdir()
    rc, line = bench.spawn_ranks(8, ['--dir', str(d3)], script=stub, timeout_s=100)
    assert rc == 0 and abs(json.loads(line)['side'][0] - 0.57) < 1e-12     # the maximum over the ranks
