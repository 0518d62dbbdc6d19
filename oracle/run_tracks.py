"""ORACLE (test infrastructure) — the reference's per-year driver `run_tracks`
(`util/compute.py:134-210`) as the literal sequential loop it is:

    while nt < n_tracks:
        while not seed_passed:   draw a seed (oracle/seeding.py: compute.py:136-169)
        v0, m0, h_bl             (compute.py:172-175)
        res = gen_track(...)     (oracle/tc_oracle.c through c_oracle: coupled_fast.py:229-267)
        accept test 1, env winds at the samples, vmax, accept test 2   (compute.py:178-209)

one candidate at a time, in candidate order, for the same Philox key the device uses (the reference
itself draws from a wall-clock-seeded MT19937, `track/bam_track.py:37-42`, so its stream cannot be
reproduced; "candidate c" = pass c of the inner `while not seed_passed` body, exactly as in the seeding
fixtures).  It returns the reference's 9-tuple plus the candidate index of every kept track, so that
the product's batched, multi-GPU `compute.run_tracks` can be compared end to end: which candidates
end up in the output, in which order, and where the `n_seeds` count stops.

The pieces are pinned separately against outputs of the reference itself (tests/test_oracle_golden.py,
tests/test_seeding.py); this file adds only the loop.

`stale_tails=True` reproduces the reference's row bookkeeping to the letter: rows are written *before* the
vmax test (compute.py:193-202), so a candidate rejected by that test leaves its samples under the next
kept track wherever that one is shorter.  The product pads every row with NaN beyond its own end
(deliberate deviations); the default here does the same.

Only tests/ may import this.
"""
import numpy as np

from . import c_oracle, seeding
from .scipy_port import Params


def run_tracks(env, basin, year, n_tracks, seed, prm=None, stale_tails=False, max_candidates=10_000_000):
    prm = prm or Params()
    ns = prm.n_steps
    se = seeding.SeedEnv(env, basin)
    ens = c_oracle.Ensemble(env, basin, prm)
    ids = np.array(seeding.BASIN_IDS)
    n_seeds = np.zeros((len(ids), 12))
    tc_lon = np.full((n_tracks, ns), np.nan); tc_lat = np.full((n_tracks, ns), np.nan)
    tc_v = np.full((n_tracks, ns), np.nan); tc_m = np.full((n_tracks, ns), np.nan)
    tc_vmax = np.full((n_tracks, ns), np.nan); tc_env_wnds = np.full((n_tracks, ns, 4), np.nan)
    tc_month = np.full(n_tracks, np.nan); tc_basin = np.full(n_tracks, '', dtype='U2')
    kept = np.zeros(n_tracks, np.int64)
    nt, cand, n_integrated, n_is_tc = 0, 0, 0, 0
    while nt < n_tracks:
        seed_passed = False
        while not seed_passed:                                      # compute.py:136-169
            if cand >= max_candidates:
                raise RuntimeError('quota not reached after %d candidates' % cand)
            c = seeding.seed_candidate(se, seed, year, cand, prm.N_series)
            cand += 1
            if c['flags'] & 1:
                n_seeds[c['basin_idx'], c['month'] - 1] += 1
            seed_passed = bool(c['flags'] & 2)
        one = dict(lon=[c['lon']], lat=[c['lat']], v0=[c['v0']], m0=[c['m0']], h_bl=[c['h_bl']],
                   month=[c['month']], phases=c['phases'][None])
        o = ens.run(one)                                            # compute.py:176-204
        n_integrated += 1
        if not o['is_tc'][0]:
            continue
        n_is_tc += 1
        n_time = int(o['n_valid'][0])
        if not stale_tails:
            for a in (tc_lon, tc_lat, tc_v, tc_m, tc_vmax, tc_env_wnds):
                a[nt] = np.nan
        tc_lon[nt, :n_time] = o['traj'][0, 0, :n_time]; tc_lat[nt, :n_time] = o['traj'][0, 1, :n_time]
        tc_v[nt, :n_time] = o['traj'][0, 2, :n_time]; tc_m[nt, :n_time] = o['traj'][0, 3, :n_time]
        tc_env_wnds[nt, :n_time] = o['envw'][0, :n_time]
        if o['accepted'][0]:                                        # compute.py:205-209
            tc_vmax[nt, :n_time] = o['vmax'][0, :n_time]
            tc_month[nt] = c['month']
            tc_basin[nt] = ids[c['basin_idx']]
            kept[nt] = cand - 1
            nt += 1
    return dict(tuple9=(tc_lon, tc_lat, tc_v, tc_m, tc_vmax, tc_env_wnds, tc_month, tc_basin, n_seeds),
                cand=kept, n_candidates=cand, n_integrated=n_integrated, n_is_tc=n_is_tc)
