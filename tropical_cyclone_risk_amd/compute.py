"""Per-year driver: the batched, multi-GPU form of the reference's ``run_tracks``
and ``run_downscaling`` (`util/compute.py:64-270`).

``run_tracks(year, n_tracks, b)`` returns the same 9-tuple the reference returns
(`compute.py:210`).  Its sequential "draw a seed, integrate, keep it if it became
a TC, until n_tracks are kept" loop (`compute.py:134-209`) is restated as rounds:

    round r covers global candidates [r*W*C, (r+1)*W*C); rank k owns the block
    [(r*W + k)*C, (r*W + k + 1)*C); every candidate has its own Philox stream, so
    the set of accepted candidates does not depend on W, C or the GPU count.

A round seeds its candidates on the device, integrates the ones whose seed
passed, and reports which were accepted.  The loop stops after the round in which
the n_tracks-th acceptance happens; tracks are the first n_tracks accepted
candidates *in candidate order*, and ``n_seeds`` counts every counted candidate
up to and including the one that completed the quota — exactly what the
sequential loop would have counted.

Deliberate deviation (documented in DESIGN.md): the reference writes a candidate's
lon/lat/v/m/env-wind rows *before* the vmax test (`compute.py:193-202`), so rows of
candidates rejected by that test leave stale tails under the next accepted
track.  Here every returned row is NaN beyond its own track end.
"""
import time

import numpy as np

from . import distributed as D
from . import namelist as default_namelist
from .basins import BASIN_IDS, TC_Basin

ROW_VARS = 9          # lon, lat, v, m, vmax, u250, v250, u850, v850 per output sample


def accept_loop(round_fn, n_tracks, per_rank, n_steps, max_rounds=10000):
    """Order-preserving accept loop over rounds of candidates (CPU logic, backend-agnostic).

    round_fn(cand0, count) -> dict with, for the candidates [cand0, cand0+count) of THIS rank:
        counted   bool[count]   counts toward n_seeds (compute.py:165-167)
        basin_idx int[count], month int[count] (1..12)
        acc_cand  int64[a]      global candidate index of each accepted track, ascending
        acc_rows  float64[a, 9*n_steps]  survivor records (lon, lat, v, m, vmax, envw[ns][4])
        acc_month int[a], acc_basin int[a]
    Returns dict(rows [n_tracks, 9*n_steps], month, basin_idx, cand, n_seeds [7, 12], rounds).
    Every rank returns the same result.
    """
    import torch
    W, rk = D.world(), D.rank()
    got_rows, got_cand, got_month, got_basin = [], [], [], []
    n_seeds = np.zeros((len(BASIN_IDS), 12))
    total = 0
    pending_counts = []          # per-round (cand index, counted, basin, month) of this rank, for the cutoff
    for r in range(max_rounds):
        cand0 = D.round_block(r, per_rank, rk, W)
        out = round_fn(cand0, per_rank)
        a = len(out['acc_cand'])
        dev = out.get('device', 'cpu')
        # ---- the data-path collective: all-gather of this round's survivor records
        meta = np.stack([np.asarray(out['acc_cand'], dtype=np.float64), np.asarray(out['acc_month'], dtype=np.float64),
                         np.asarray(out['acc_basin'], dtype=np.float64)], axis=1) if a else np.zeros((0, 3))
        rows = np.concatenate([np.asarray(out['acc_rows'], dtype=np.float64).reshape(a, 9 * n_steps), meta], axis=1)
        t_rows = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
        cnt = torch.tensor([a], dtype=torch.int64, device=dev)
        gathered, counts = D.allgather_rows(t_rows, cnt)
        g = gathered.cpu().numpy()
        # rank blocks are contiguous and ascending, so rank order == candidate order
        assert (np.diff(g[:, -3]) > 0).all() if len(g) > 1 else True
        got_rows.append(g[:, :-3]); got_cand.append(g[:, -3].astype(np.int64))
        got_month.append(g[:, -2].astype(np.int64)); got_basin.append(g[:, -1].astype(np.int64))
        pending_counts.append((cand0, np.asarray(out['counted'], bool), np.asarray(out['basin_idx']),
                               np.asarray(out['month'])))
        total += len(g)
        if total >= n_tracks:
            break
    else:
        raise RuntimeError('accept_loop: quota not reached after %d rounds' % max_rounds)
    rows = np.concatenate(got_rows)[:n_tracks]
    cand = np.concatenate(got_cand)[:n_tracks]
    month = np.concatenate(got_month)[:n_tracks]
    basin = np.concatenate(got_basin)[:n_tracks]
    cutoff = cand[-1]            # the candidate that completed the quota
    # ---- n_seeds: counted candidates with index <= cutoff (compute.py:167), summed over ranks
    for cand0, counted, bidx, mo in pending_counts:
        idx = cand0 + np.arange(len(counted))
        keep = counted & (idx <= cutoff)
        np.add.at(n_seeds, (bidx[keep], mo[keep] - 1), 1)
    t_seeds = torch.from_numpy(n_seeds)
    if D.world() > 1:
        t_seeds = t_seeds.to(dev)
        D.allreduce_sum_(t_seeds)
        n_seeds = t_seeds.cpu().numpy()
    return dict(rows=rows, month=month, basin_idx=basin, cand=cand, n_seeds=n_seeds, rounds=r + 1)


class GpuRound:
    """round_fn backed by the device pipeline: seed → select → integrate → pack."""

    def __init__(self, engine, year, per_rank, experiment_seed=None):
        import torch
        from .pipeline import DevicePipeline
        self.torch = torch
        self.eng = engine
        self.year = int(year)
        self.seed = experiment_seed
        self.pipe = DevicePipeline(engine, per_rank, per_rank, tc_rows_only=True)
        self.packed = None

    def __call__(self, cand0, count):
        torch, p = self.torch, self.pipe
        ns = self.eng.n_steps
        p.seed_round(self.year, cand0, count, self.seed)
        p.select_passed(count)
        n_pass = min(int(p.n_passed.item()), count)
        res = dict(device=p.dev)
        flags = p.cand['seed_flags'][:count].cpu().numpy()
        res['counted'] = (flags & 1) != 0
        res['basin_idx'] = p.cand['basin_idx'][:count].cpu().numpy()
        res['month'] = p.cand['slot'][:count].cpu().numpy() + 1
        if n_pass == 0:
            res.update(acc_cand=np.zeros(0, np.int64), acc_rows=np.zeros((0, ROW_VARS * ns)),
                       acc_month=np.zeros(0, np.int64), acc_basin=np.zeros(0, np.int64))
            return res
        p.integrate(n_pass)
        bad = int((p.tracks['status'][:n_pass] == -3).sum().item())
        if bad:
            raise RuntimeError('%d storms needed more than max_rk_steps accepted RK steps; raise '
                               'tcr_params.max_rk_steps' % bad)
        p.select_accepted()
        n_acc = int(p.n_accepted.item())
        if self.packed is None or self.packed.shape[0] < max(n_acc, 1):
            self.packed = torch.empty(max(n_acc, 1024), ROW_VARS * ns, dtype=torch.float64, device=p.dev)
        p.pack_accepted(self.packed, n_acc)
        dense = p.acc_idx[:n_acc].long()                     # position in the dense batch
        cand_local = p.cand_idx[:n_pass].long()[dense]       # position in this rank's candidate block
        res['acc_cand'] = (cand_local + int(cand0)).cpu().numpy()
        res['acc_rows'] = self.packed[:n_acc]
        res['acc_rows'] = res['acc_rows'].cpu().numpy()
        res['acc_month'] = (p.storms['slot'][:n_pass][dense] + 1).cpu().numpy()
        res['acc_basin'] = p.storms['basin_idx'][:n_pass][dense].cpu().numpy()
        return res


def rows_to_tuple(res, n_steps):
    """Survivor records -> the reference's 9-tuple layout (compute.py:124-133, 210)."""
    rows = res['rows']
    n = rows.shape[0]
    ns = n_steps
    tc_lon, tc_lat, tc_v, tc_m, tc_vmax = (rows[:, k * ns:(k + 1) * ns].copy() for k in range(5))
    tc_env_wnds = rows[:, 5 * ns:].reshape(n, ns, 4).copy()
    tc_month = res['month'].astype(np.float64)
    tc_basin = np.array([BASIN_IDS[i] for i in res['basin_idx']], dtype='U2')
    return (tc_lon, tc_lat, tc_v, tc_m, tc_vmax, tc_env_wnds, tc_month, tc_basin, res['n_seeds'])


def run_tracks(year, n_tracks, b, engine=None, env=None, nl=None, per_rank=None, info=None):
    """Generate n_tracks TC tracks in basin b for one year (reference: compute.py:64-210).

    Returns (tc_lon, tc_lat, tc_v, tc_m, tc_vmax, tc_env_wnds, tc_month, tc_basin, n_seeds).
    ``engine`` is a staged TCEngine; if omitted one is built from ``env`` (a field set shaped
    like ``synthetic.SyntheticEnv``) on this rank's GPU.  ``info`` (a dict, optional) receives what the
    reference's loop keeps implicit: ``cand`` (global candidate index of every returned track) and ``rounds``.
    """
    nl = nl or default_namelist
    basin_id = b.basin_id if isinstance(b, TC_Basin) else b
    own = engine is None
    if own:
        from .engine import TCEngine
        import os
        engine = TCEngine(basin_id, device=D.local_device(os.environ.get('LOCAL_RANK', '0')), nl=nl).stage_env(env)
    per_rank = int(per_rank or max(4096, min(nl.gpu_candidate_round, 64 * n_tracks)))
    rf = GpuRound(engine, year, per_rank)
    res = accept_loop(rf, n_tracks, per_rank, engine.n_steps)
    if info is not None:
        info.update(cand=res['cand'], rounds=res['rounds'], per_rank=per_rank)
    if own:
        engine.close()
    return rows_to_tuple(res, engine.n_steps)


def run_downscaling(basin_id, env=None, nl=None, out_dir=None):
    """Run every year of the namelist for one basin and write the track file
    (reference: compute.py:216-270).  Returns the output file name (rank 0) or None."""
    from . import io as tio
    nl = nl or default_namelist
    import os
    from .engine import TCEngine
    b = TC_Basin(basin_id)
    if env is None:
        env = tio.load_env(nl)
    s = time.time()
    eng = TCEngine(basin_id, device=D.local_device(os.environ.get('LOCAL_RANK', '0')), nl=nl).stage_env(env)
    out = []
    years = list(range(nl.start_year, nl.end_year + 1))
    for yr in years:
        if hasattr(env, 'for_year'):
            eng.stage_env(env.for_year(yr))
        out.append(run_tracks(yr, nl.tracks_per_year, b, engine=eng, nl=nl))
    eng.close()
    fn = None
    if D.rank() == 0:
        fn = tio.write_tracks(out, years, b, nl, out_dir)
        print('Saved %s' % fn)
        print(time.time() - s)
    D.barrier()
    return fn
