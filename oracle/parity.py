"""ORACLE-side checker (test infrastructure): pointwise parity of a batch of tracks, with a
decision-forced replay for the storms whose `land == 1` decisions differ.

The reference's over-land test ``f_land.ev(lon, lat) == 1`` (intensity/coupled_fast.py:35-38) runs
on a bilinear sum whose weights add up to 1 only within rounding, so in the interior of land it is
True or False with the last bits of lon/lat ("flicker").  Where the interpolated PI is non-zero the
RHS then jumps between PI and 0, and two implementations whose trajectories differ by 1e-13 stop
agreeing at the first evaluation where the decision lands differently — by an amount bounded only
by the integrator's own rtol.  Every implementation therefore records the decision of every RHS
evaluation (the "decision probe": bit 0 `land == 1`, bit 1 PI != 0, bit 2 land within 1e-12 of 1), and
this checker asserts

  * storms whose decision sequences agree: identical discrete results (status, n_valid, nfev, accept
    flags, step counters when given) and the stated fp64 tolerance on every sample;
  * storms whose sequences differ at evaluation k:
      - the differing evaluation itself is rounding-sensitive (bit 2 on either side AT evaluation k) —
        a differing decision anywhere else is a bug (coastline indexing, wrong cell) and fails;
      - everything emitted before the step attempt that contains evaluation k agrees (prefix check);
      - **decision-forced replay**: the C oracle is run again taking the other side's decision at every
        rounding-sensitive evaluation (tc_oracle.c, orc_run_ensemble_forced; pinned to the reference's own
        code by tests/golden/forced_*.npz).  Both sides then walk the same branch sequence, so the WHOLE
        track is held to the same discrete equalities and the same pointwise tiers as the storms above;
        the replay must leave no differing decision and report no hard mismatch.
    No storm is skipped and none is only partly checked.

Only tests/ and __graft_entry__.smoke() import this.
"""
import numpy as np

# Tiers on max |got - want| over a storm's hourly lon / lat / v / m / env winds / vmax, about 10x what was measured on
# 20 000 storms per basin (profiles/r03_parity_study.json: p95 1.0-2.2e-12, p99 3-11e-11).  The intensity equation
# amplifies a perturbation while a storm intensifies, so the far tail is a handful of storms that the oracle itself
# amplifies the same way.  The bound on EVERY sample is therefore stated PER STORM wherever a replayer is at hand (every
# caller in tests/ and smoke()): a storm passes outright under TOL_ALL_FLOOR; a storm above it must be one of the oracle's
# own amplifiers — its difference at most TWIN_FACTOR x what the oracle moves ON THAT VERY STORM when one input changes by
# one ulp (`replay.twin_storms`: v0 up, v0 down, lon up; the maximum over the twins whose decisions and counters stay the
# same) —, there may be at most n // 1000 + 2 such storms, and none may exceed TOL_TAIL_CAP.  A chaotic storm thus widens
# nobody's bound but its own, and each one is printed.  TWIN_FACTOR: the twin injects one ulp once, at t = 0,
# into one variable; two libm builds differ by an ulp or two in several operations of every one of a storm's 100-300
# evaluations, each amplified from where it enters (measured: 16x on the one storm of 2 000 that exceeded the floor,
# tests/test_static_store.py; at most 9.4x over 4 x 20 000 storms, profiles/r05_parity_study.json — round 6 sets the factor to 30,
# about twice the largest ratio ever measured).  The cap stays at 1e-4: round 6 tried 5e-5 (2.5x round 5's largest difference,
# 1.94e-5) and the 20 000-storm study failed it on ONE NA storm: GPU - oracle 6.9e-5 there with round 6's arithmetic, 4.5 x what the
# oracle moves on that very storm under one-ulp changes (the one-ulp twin ensemble itself reaches 1.9e-4 on another storm of the same
# 20 000, profiles/r06_parity_study.json).  An absolute bound on a chaotic storm is a statement about which last bits happen to
# differ — that is what the per-storm factor is for; the cap is the backstop, a tenth of the integrator's own rtol = 1e-3.
# TOL_ALL is the fixed bound of a comparison without a replayer.
TOL_ALL_FLOOR = 1e-7  # every sample of every storm: passes without looking at the twin
TOL_ALL = 1e-6        # every sample of every storm when there is no oracle twin to measure against
TWIN_FACTOR = 30.0    # a storm above the floor: at most this x the oracle's own one-ulp response on the same storm
TOL_TAIL_CAP = 1e-4   # ... and never above this
TOL_99 = 1e-9         # 99 % of the storms: at most n // 100 + 1 storms above it
TOL_95 = 2e-11        # 95 % of the storms: at most n // 20 + 2 storms above it
# vmax (wind/tc_wind.py:6-21) contains the translation speed, a centred difference of hourly positions
# (util/sphere.py:58-83): a derivative of the track, so its tiers are 5x the track's
TIER_SCALE = {'vmax': 5.0}
NOT_EVAL = 0xff


def ragged_to_padded(dec, off, cap, fill=NOT_EVAL, dtype=np.uint8):
    """Fixture layout (storm i owns dec[off[i]:off[i+1]]) -> [n, cap] padded with `fill`."""
    n = len(off) - 1
    out = np.full((n, cap), fill, dtype=dtype)
    for i in range(n):
        m = min(cap, int(off[i + 1] - off[i]))
        out[i, :m] = dec[off[i]:off[i] + m]
    return out


def first_divergence(dec_a, dec_b):
    """Per storm: index of the first RHS evaluation whose `land == 1` decision (bit 0) differs where it
    matters (bit 1, PI != 0, on either side); -1 if the sequences agree over their common length."""
    a, b = np.asarray(dec_a), np.asarray(dec_b)
    w = min(a.shape[1], b.shape[1])
    a, b = a[:, :w], b[:, :w]
    both = (a != NOT_EVAL) & (b != NOT_EVAL)
    diff = both & (((a ^ b) & 1) != 0) & (((a | b) & 2) != 0)
    return np.where(diff.any(axis=1), diff.argmax(axis=1), -1)


def _storm_maxdiff(a, b):
    n = a.shape[0]
    assert np.array_equal(np.isnan(a), np.isnan(b)), 'NaN padding differs'
    d = np.abs(np.nan_to_num(a) - np.nan_to_num(b)).reshape(n, -1)
    return d.max(axis=1) if d.size else np.zeros(n)


def _subset(d, idx, keys):
    return {k: np.asarray(d[k])[idx] for k in keys if k in d}


def check_tracks(tag, got, want, dec_got, dec_want, t0_want, t_s, counters=('status', 'n_valid', 'nfev'),
                 flags=('is_tc', 'accepted'), names=('traj', 'envw', 'vmax'), verbose=True, tol_all=None,
                 replay=None, replay_as='want', tol_99=TOL_99, tol_95=TOL_95):
    """Assert parity of `got` against `want` (dicts of arrays: traj [n,4,ns], envw [n,ns,4], vmax [n,ns],
    status, n_valid, nfev, is_tc, accepted ...) — every storm pointwise over its whole track.

    dec_*: [n, cap] decision probes (bit0 `land == 1`, bit1 PI != 0, bit2 land within 1e-12 of 1; 0xff
    = not evaluated); t0_want [n, cap]: start time of the step attempt of each evaluation of `want`.

    replay(idx, dec_force) -> result dict of the C oracle for storms `idx` with `dec_force` [len(idx), cap]
    taken at the rounding-sensitive evaluations (c_oracle.replayer).  replay_as says which side the oracle plays:
      'want' (GPU vs oracle): forced with got's decisions, `got[idx]` is compared with the replay;
      'got'  (oracle vs a reference fixture): forced with want's decisions, the replay is compared with `want[idx]`.
    Without `replay` the storms with a differing decision are only prefix-checked and the summary says so
    (`unreplayed`); every caller in tests/ and smoke() passes one.
    Returns a summary dict (counts of storms per class, exposure among accepted storms, worst differences).
    tol_all: the bound on every sample.  None (default): the per-storm rule of the module header (floor, else the storm's
    own oracle twin x TWIN_FACTOR, few such storms, none above TOL_TAIL_CAP), or TOL_ALL when `replay` offers no twin; a
    number: that bound (the large-ensemble study passes inf and states its own bounds on the tail,
    tests/test_gpu_parity.py::test_parity_study_at_scale)."""
    n = len(want['n_valid'])
    twin_cache = {}
    amplified = []                 # (storm, output, difference, the oracle's own one-ulp response on that storm)

    def within(name, d, storm=None):
        """Is difference `d` of output `name` on storm `storm` (index into this batch) acceptable?"""
        sc = TIER_SCALE.get(name, 1.0)
        if tol_all is not None:
            return d <= tol_all
        if d <= sc * TOL_ALL_FLOOR:
            return True                                   # under the floor: no need to run the twin
        if storm is None or not hasattr(replay, 'twin_storms'):
            return d <= sc * TOL_ALL
        storm = int(storm)
        if storm not in twin_cache:
            # (the twins walk the decision sequence both compared runs walked on this storm)
            walked = np.ascontiguousarray((np.asarray(dec_got) if replay_as == 'want' else np.asarray(dec_want))[[storm]])
            twin_cache[storm] = {k: float(v[0]) for k, v in replay.twin_storms([storm], walked).items()}
        tw = twin_cache[storm].get(name, float('nan'))
        ok = bool(d <= TWIN_FACTOR * tw) and d <= sc * TOL_TAIL_CAP          # (a NaN twin — every twin changed a decision — fails)
        amplified.append((storm, name, float(d), tw))
        if verbose:
            print('%s: storm %d %s differs by %.3g > %.0e; the oracle itself moves by %.3g on this storm when one input '
                  'changes by one ulp (x%.1f) -> %s' % (tag, storm, name, d, sc * TOL_ALL_FLOOR, tw, d / tw if tw > 0 else float('inf'),
                                                        'ok' if ok else 'FAIL'))
        return ok
    dec_got, dec_want = np.asarray(dec_got), np.asarray(dec_want)
    k = first_divergence(dec_got, dec_want)
    len_g = (dec_got != NOT_EVAL).sum(axis=1)
    len_w = (dec_want != NOT_EVAL).sum(axis=1)
    agree = k < 0
    cap = min(dec_got.shape[1], dec_want.shape[1])
    # no differing decision => the same evaluations were made (beyond the probe's capacity the
    # discrete results below still pin it)
    same_len = (len_g == len_w) | (np.minimum(len_g, len_w) >= cap)
    assert same_len[agree].all(), (tag, 'evaluation count differs without a differing land decision',
                                   np.nonzero(agree & ~same_len)[0][:8])
    div = np.nonzero(~agree)[0]
    # ---- a differing decision may only occur AT an evaluation that is rounding-sensitive (bit 2 on either side, at
    #      evaluation k itself): anything else is a wrong land decision, not flicker
    if len(div):
        at_k = dec_got[div, k[div]] | dec_want[div, k[div]]
        bad = div[(at_k & 4) == 0]
        assert bad.size == 0, (tag, 'land decision differs at an evaluation that is not within 1e-12 of land == 1',
                               [(int(i), int(k[i])) for i in bad[:8]])
    # ---- decision-identical storms: discrete results
    for key in tuple(counters) + tuple(flags):
        bad = agree & (np.asarray(got[key]) != np.asarray(want[key]))
        assert not bad.any(), (tag, key, np.nonzero(bad)[0][:8])
    # ---- storms with a differing decision: prefix parity against `want` up to the attempt that contains it
    pref_worst, pref_samples, pref_max = 0.0, 0, []
    for i in div:
        t0 = float(np.asarray(t0_want)[i, k[i]])
        gated = (got['status'][i] == -1) or (want['status'][i] == -1)
        if t0 <= 0.0:
            n_pref = 0 if gated else 1          # sample 0 is the seed itself
        else:
            n_pref = int(np.searchsorted(t_s, t0 - 1e-6, side='right'))
        assert got['n_valid'][i] >= n_pref and want['n_valid'][i] >= n_pref, (tag, 'track shorter than its common prefix', i)
        if n_pref == 0:
            continue
        for name in names:
            a, b = np.asarray(got[name])[i], np.asarray(want[name])[i]
            if name == 'traj':
                a, b = a[:, :n_pref], b[:, :n_pref]
            elif name == 'vmax':                # vmax at sample j needs sample j + 1 (centred difference)
                a, b = a[:max(n_pref - 1, 0)], b[:max(n_pref - 1, 0)]
            else:
                a, b = a[:n_pref], b[:n_pref]
            if a.size == 0:
                continue
            assert np.array_equal(np.isnan(a), np.isnan(b)), (tag, name, i)
            d = np.abs(np.nan_to_num(a) - np.nan_to_num(b)).max()
            pref_worst = max(pref_worst, float(d))
            pref_max.append(float(d))
            assert within(name, d, i), (tag, name, 'prefix of storm %d (first differing decision at evaluation %d, '
                                        't = %.0f s, %d samples)' % (i, k[i], t0, n_pref), d)
        pref_samples += n_pref
    # ---- ... and the decision-forced replay: the whole track of every such storm, pointwise
    keys = tuple(counters) + tuple(flags) + tuple(names)
    overridden = hard = 0
    per_storm = {name: np.full(n, np.nan) for name in names}
    for name in names:
        per_storm[name][agree] = _storm_maxdiff(np.asarray(got[name])[agree], np.asarray(want[name])[agree])
    replayed = 0
    if replay is not None and len(div):
        src_dec = dec_got if replay_as == 'want' else dec_want
        alt = replay(div, np.ascontiguousarray(src_dec[div]))
        hard = int(np.asarray(alt['hard_mismatch']).sum())
        overridden = int(np.asarray(alt['overridden']).sum())
        assert hard == 0, (tag, 'forced replay: decisions differ at evaluations that are not rounding-sensitive',
                           div[np.asarray(alt['hard_mismatch']) > 0][:8])
        left = first_divergence(np.asarray(alt['dec']), src_dec[div])
        assert (left < 0).all(), (tag, 'forced replay still differs in a land decision', div[left >= 0][:8], left[left >= 0][:8])
        a, b = (_subset(got, div, keys), alt) if replay_as == 'want' else (alt, _subset(want, div, keys))
        for key in tuple(counters) + tuple(flags):
            bad = np.asarray(a[key]) != np.asarray(b[key])
            assert not bad.any(), (tag, key, 'after forced replay', div[bad][:8])
        for name in names:
            per_storm[name][div] = _storm_maxdiff(np.asarray(a[name]), np.asarray(b[name]))
        replayed = len(div)
    worst = {}
    for name in names:
        d = per_storm[name][~np.isnan(per_storm[name])]
        worst[name] = float(d.max()) if d.size else 0.0
        if verbose:
            print('%s %-5s pointwise over whole tracks: max %.3g  p99 %.3g  p95 %.3g   (n=%d, %d of them replayed)'
                  % (tag, name, worst[name], np.percentile(d, 99) if d.size else 0, np.percentile(d, 95) if d.size else 0,
                     d.size, replayed))
        # every sample of every storm: the floor, or — per storm — the oracle's own amplification of that storm
        ps = np.nan_to_num(per_storm[name], nan=0.0)
        sc = TIER_SCALE.get(name, 1.0)
        over = np.nonzero(ps > (sc * TOL_ALL_FLOOR if tol_all is None else tol_all))[0]
        if tol_all is None:
            assert len(over) <= n // 1000 + 2, (tag, name, 'storms above the floor', len(over), n)
        for i in over[np.argsort(-ps[over])]:
            assert within(name, float(ps[i]), i), (tag, name, 'storm %d' % i, float(ps[i]), twin_cache.get(int(i)))
        # the tiers as counts, at every ensemble size: at most 1 % (+1) of the storms above tol_99, 5 % (+2) above tol_95
        # (the additive slack only matters for the small curated golden sets, which over-sample intense storms)
        assert (d > sc * tol_99).sum() <= d.size // 100 + 1, (tag, name, 'p99 tier', int((d > sc * tol_99).sum()), d.size)
        assert (d > sc * tol_95).sum() <= d.size // 20 + 2, (tag, name, 'p95 tier', int((d > sc * tol_95).sum()), d.size)
    exposed = ((dec_want != NOT_EVAL) & ((dec_want & 6) == 6)).any(axis=1)
    acc = np.asarray(want['accepted'], bool)
    out = dict(n=n, identical=int(agree.sum()), diverged=int(len(div)), replayed=int(replayed),
               unreplayed=int(len(div) - replayed), pointwise=int(agree.sum() + replayed),
               overridden=overridden, hard_mismatch=hard, exposed=int(exposed.sum()),
               exposed_identical=int((exposed & agree).sum()), accepted=int(acc.sum()),
               accepted_exposed=int((acc & exposed).sum()), accepted_diverged=int((acc & ~agree).sum()),
               prefix_samples=int(pref_samples), prefix_worst=pref_worst, worst=worst, per_storm=per_storm, amplified=amplified)
    if verbose:
        print('%s: %d storms — %d decision-identical + %d decision-forced replays = %d pointwise over the whole track '
              '(%d forced decisions, %d unforced mismatches), %d only prefix-checked; prefixes: %d samples, worst %.3g; '
              '%d flicker-exposed of which %d decision-identical; accepted %d, of which exposed %d (%.0f %%), diverged %d'
              % (tag, n, out['identical'], replayed, out['pointwise'], overridden, hard, out['unreplayed'], pref_samples,
                 pref_worst, out['exposed'], out['exposed_identical'], out['accepted'], out['accepted_exposed'],
                 100.0 * out['accepted_exposed'] / max(1, out['accepted']), out['accepted_diverged']))
    return out
