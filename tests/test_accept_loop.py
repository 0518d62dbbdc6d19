"""The batched accept loop and its all-gather (compute.accept_loop, distributed.py) on CPU:
single process vs 2 gloo ranks vs a literal sequential loop over candidates."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NS = 9


def fake_candidate(c):
    """Deterministic pseudo-candidate: (counted, passed, accepted, basin, month, row)."""
    rng = np.random.default_rng(1000003 * 7 + int(c))
    counted = rng.random() < 0.7
    passed = counted and rng.random() < 0.4
    accepted = passed and rng.random() < 0.3
    return counted, passed, accepted, int(rng.integers(0, 7)), int(rng.integers(1, 13)), rng.random(9 * NS)


def fake_round(cand0, count):
    rows = [fake_candidate(cand0 + i) for i in range(count)]
    acc = [i for i, r in enumerate(rows) if r[2]]
    return dict(counted=np.array([r[0] for r in rows]), basin_idx=np.array([r[3] for r in rows]),
                month=np.array([r[4] for r in rows]), acc_cand=np.array([cand0 + i for i in acc], dtype=np.int64),
                acc_rows=np.array([rows[i][5] for i in acc]).reshape(len(acc), 9 * NS),
                acc_month=np.array([rows[i][4] for i in acc]), acc_basin=np.array([rows[i][3] for i in acc]))


def sequential(n_tracks):
    """The reference's loop shape: walk candidates in order until n_tracks are accepted."""
    n_seeds = np.zeros((7, 12)); rows = []; cands = []
    c = 0
    while len(rows) < n_tracks:
        counted, passed, accepted, b, m, row = fake_candidate(c)
        if counted:
            n_seeds[b, m - 1] += 1
        if accepted:
            rows.append(row); cands.append(c)
        c += 1
    return np.array(rows), np.array(cands), n_seeds


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, per_rank, n_tracks, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from tropical_cyclone_risk_amd import compute, distributed as D
    D.init_from_env(backend='gloo')
    res = compute.accept_loop(fake_round, n_tracks, per_rank, NS)
    q.put((rank, res['rows'], res['cand'], res['n_seeds'], res['rounds']))
    D.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.parametrize('per_rank', [64, 257])
def test_single_process_matches_sequential(per_rank):
    sys.path.insert(0, ROOT)
    from tropical_cyclone_risk_amd import compute
    res = compute.accept_loop(fake_round, 25, per_rank, NS)
    rows, cands, n_seeds = sequential(25)
    assert np.array_equal(res['cand'], cands)
    assert np.array_equal(res['rows'], rows)
    assert np.array_equal(res['n_seeds'], n_seeds)


@pytest.mark.parametrize('world,per_rank', [(2, 64), (2, 150), (4, 37), (8, 16)])
def test_gloo_ranks_match_sequential(world, per_rank):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, 25, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rows, cands, n_seeds = sequential(25)
    for rank, r_rows, r_cand, r_seeds, rounds in got:
        assert np.array_equal(r_cand, cands), rank
        assert np.array_equal(r_rows, rows), rank
        assert np.array_equal(r_seeds, n_seeds), rank


def _overflow_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from tropical_cyclone_risk_amd import compute, distributed as D
    D.init_from_env(backend='gloo')

    def round_fn(cand0, count):
        out = fake_round(cand0, count)
        out['bad'] = 3 if rank == 1 else 0          # only rank 1 has storms that overflowed their step record
        return out
    try:
        compute.accept_loop(round_fn, 25, 64, NS)
        q.put((rank, 'no error'))
    except RuntimeError as e:
        q.put((rank, str(e)))
    D.barrier()                                    # both ranks get here: nobody is left waiting in a collective
    import torch.distributed as dist
    dist.destroy_process_group()


def test_step_record_overflow_raises_on_every_rank():
    """A storm that overflows its step record on ONE rank must abort the round on EVERY rank (the count is part of
    the round's one all-gather), not leave the others hanging in the row all-gather."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all('gpu_max_rk_steps' in got[r] and got[r].startswith('3 storms') for r in (0, 1)), got


def _grow_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from tropical_cyclone_risk_amd import compute, distributed as D
    D.init_from_env(backend='gloo')

    class Round:
        """Rank 1's storms overflow until the record has been doubled twice; rank 0 never sees an overflow itself."""
        cap, calls, grown = 16, 0, 0

        def __call__(self, cand0, count):
            self.calls += 1
            out = fake_round(cand0, count)
            out['bad'] = 2 if (rank == 1 and self.cap < 64) else 0
            return out

        def grow(self):
            self.cap *= 2
            self.grown += 1
            return True
    rf = Round()
    res = compute.accept_loop(rf, 25, 64, NS)
    q.put((rank, res['cand'], res['n_seeds'], rf.grown, rf.calls, res['rounds']))
    D.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


def test_step_record_overflow_grows_on_every_rank():
    """A round function that can grow its step record (compute.GpuRound.grow) is asked to — on EVERY rank, because the
    overflow count is part of the round's one all-gather — and the round is integrated again; the result is the
    sequential loop's."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows, cands, n_seeds = sequential(25)
    for rank, cand, seeds, grown, calls, rounds in got:
        assert np.array_equal(cand, cands) and np.array_equal(seeds, n_seeds), rank
        assert grown == 2 and calls == rounds + 2, (rank, grown, calls, rounds)       # both ranks grew, both re-ran the first round twice


def _refuse_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from tropical_cyclone_risk_amd import compute, distributed as D
    D.init_from_env(backend='gloo')

    class Round:
        """Rank 0's storms overflow; rank 0 could grow its record, rank 1 has no memory for it (GpuRound.can_grow is a
        local fact: free HBM)."""
        grown = 0

        def __call__(self, cand0, count):
            out = fake_round(cand0, count)
            out['bad'] = 2 if rank == 0 else 0
            return out

        def can_grow(self):
            return rank == 0

        def grow(self):
            self.grown += 1
            return True
    rf = Round()
    try:
        compute.accept_loop(rf, 25, 64, NS)
        q.put((rank, 'no error', rf.grown))
    except RuntimeError as e:
        q.put((rank, str(e), rf.grown))
    D.barrier()                                    # both ranks get here: nobody is left waiting in a collective
    import torch.distributed as dist
    dist.destroy_process_group()


def test_grow_refused_by_one_rank_raises_on_every_rank():
    """whether the step record can grow depends on a rank's free memory.  If ONE rank cannot, NO rank grows and
    every rank raises — a rank that grew and re-ran the round would wait for ever in its all-gather."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_refuse_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, msg, grown in got:
        assert 'gpu_max_rk_steps' in msg and grown == 0, (rank, msg, grown)


def test_allgather_rows_ragged_gloo():
    """Ragged all-gather incl. an empty contribution and the rank-order guarantee."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, shapes, ok in got:
        assert ok, (rank, shapes)


def _ragged_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch
    from tropical_cyclone_risk_amd import distributed as D
    D.init_from_env(backend='gloo')
    ok, shapes = True, []
    for counts in ([3, 5], [0, 4], [2, 0], [0, 0]):
        n = counts[rank]
        rows = torch.full((max(n, 1), 6), float(rank)) + torch.arange(max(n, 1)).reshape(-1, 1) * 0.01
        out, cs = D.allgather_rows(rows[:n] if n else rows[:0], torch.tensor([n]))
        shapes.append(tuple(out.shape))
        ok &= cs == counts and out.shape[0] == sum(counts)
        if sum(counts):
            exp = torch.cat([torch.full((counts[r], 6), float(r)) + torch.arange(counts[r]).reshape(-1, 1) * 0.01
                             for r in range(world)])
            ok &= bool(torch.equal(out, exp))
        work, fin = D.allgather_rows(rows[:n] if n else rows[:0], None, counts=counts, async_op=True)
        out2, _ = fin()
        ok &= bool(torch.equal(out2, out))
    q.put((rank, shapes, bool(ok)))
    D.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


def test_deferred_row_gather_gloo():
    """distributed.DeferredRowGather (bench.py's N > 1 path): counts read `lag` batches late, rotating
    buffers, capacity clipping, rank order — every batch's rows arrive exactly once and intact."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_deferred_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, msg in got:
        assert ok, (rank, msg)


def _deferred_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch
    from tropical_cyclone_risk_amd import distributed as D
    D.init_from_env(backend='gloo')
    cap, width, lag, n_batches = 5, 4, 3, 11
    counts_of = lambda k, r: (3 * k + 2 * r) % 8            # 0..7: some batches exceed cap = 5, some are empty
    seen = []

    def on_rows(parts, counts):
        seen.append(([p.clone() for p in parts], list(counts)))
    g = D.DeferredRowGather(cap, width, 'cpu', lag=lag, on_rows=on_rows)
    for k in range(n_batches):
        buf = g.buffer()
        n = counts_of(k, rank)
        buf.fill_(-1.0)
        for i in range(min(n, cap)):
            buf[i] = 1000.0 * k + 10.0 * rank + i
        g.submit(torch.tensor([n], dtype=torch.int64))
    g.drain()
    ok, msg = True, ''
    exp_total = sum(min(counts_of(k, r), cap) for k in range(n_batches) for r in range(world))
    exp_clip = sum(max(0, counts_of(k, r) - cap) for k in range(n_batches) for r in range(world))
    if g.rows_gathered != exp_total or g.rows_clipped != exp_clip or len(seen) != n_batches:
        ok, msg = False, 'totals %s %s %s' % (g.rows_gathered, g.rows_clipped, len(seen))
    for k, (parts, counts) in enumerate(seen):              # batches finish in submission order
        for r in range(world):
            n = min(counts_of(k, r), cap)
            exp = torch.tensor([[1000.0 * k + 10.0 * r + i] * width for i in range(n)], dtype=torch.float64).reshape(n, width)
            if counts[r] != n or not torch.equal(parts[r], exp):
                ok, msg = False, 'batch %d rank %d: %s vs %s' % (k, r, parts[r], exp)
    q.put((rank, ok, msg))
    D.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# years sharded over the ranks (compute.run_downscaling with at least as many years as ranks)
def _year_tuple(i, T, ns):
    rng = np.random.default_rng(77 + i)
    from tropical_cyclone_risk_amd.basins import BASIN_IDS
    def plane():
        a = rng.standard_normal((T, ns))
        a[:, ns - 1 - i % 3:] = np.nan
        return a
    return (plane(), plane(), plane(), plane(), plane(), rng.standard_normal((T, ns, 4)), rng.integers(1, 13, T).astype(float),
            np.array([BASIN_IDS[k] for k in rng.integers(0, 7, T)], dtype='U2'), rng.integers(0, 90, (7, 12)).astype(float))


def _years_worker(rank, world, port, n_years, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import types
    import torch
    from tropical_cyclone_risk_amd import compute, distributed as D
    D.init_from_env(backend='gloo')
    T, ns = 5, 361
    nl = types.SimpleNamespace(tracks_per_year=T, total_track_time_days=15, output_interval_s=3600)
    years = list(range(2000, 2000 + n_years))
    mine = list(range(rank, n_years, world))
    out = [None] * n_years
    from tropical_cyclone_risk_amd.basins import BASIN_IDS
    for i in mine:
        # a year as the accept loop leaves it on the device: records + (candidate index, month, basin index), n_seeds
        t9 = _year_tuple(i, T, ns)
        rows = np.concatenate([t9[0], t9[1], t9[2], t9[3], t9[4], t9[5].reshape(T, ns * 4), np.arange(T, dtype=np.float64)[:, None],
                               t9[6][:, None], np.array([BASIN_IDS.index(x) for x in t9[7]], dtype=np.float64)[:, None]], axis=1)
        out[i] = dict(rows_dev=torch.from_numpy(rows), n_seeds_dev=torch.from_numpy(t9[8].reshape(-1)))
    # (rank 0 is the one that writes the file; here every rank rebuilds the tuples so that each can check them)
    res = compute._allgather_years(out, mine, years, nl, torch.device('cpu'), to_host=True)
    ok = all(all(np.array_equal(a, b, equal_nan=(a.dtype.kind == 'f')) for a, b in zip(res[i], _year_tuple(i, T, ns))) for i in range(n_years))
    ok = ok and compute._allgather_years(out, mine, years, nl, torch.device('cpu'), to_host=False) is None
    # the single-rank form of the accept loop (distributed.Local) runs without touching the process group
    loc = compute.accept_loop(fake_round, 12, 64, NS, ops=D.Local)
    q.put((rank, ok, loc['cand']))
    D.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.parametrize('n_years', [2, 5])
def test_year_sharded_allgather_of_final_tracks(n_years):
    """Years sharded over two ranks (gloo): the all-gather of the final tracks hands every rank every year's 9-tuple,
    bit for bit, also when the ranks hold different numbers of years; and a rank working a year on its own
    (distributed.Local) gets the sequential loop's tracks."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_years_worker, args=(r, 2, port, n_years, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
    _, cands, _ = sequential(12)
    for rank, ok, cand in got:
        assert ok, rank
        assert np.array_equal(cand, cands)
