"""ORACLE-side checker (test infrastructure): pointwise / prefix parity of a batch of tracks.

The reference's over-land test ``f_land.ev(lon, lat) == 1`` (intensity/coupled_fast.py:35-38) runs
on a bilinear sum whose weights add up to 1 only within rounding, so in the interior of land it is
True or False with the last bits of lon/lat ("flicker").  Where the interpolated PI is non-zero the
RHS then jumps between PI and 0, and two implementations whose trajectories differ by 1e-13 stop
agreeing at the first evaluation where the decision lands differently — by an amount bounded only
by the integrator's own rtol.  Instead of waving such storms through, every implementation records
the decision of every RHS evaluation (the "decision probe"), and this checker asserts

  * storms whose decision sequences agree: identical discrete results (status, n_valid, nfev, accept
    flags, step counters when given) and the stated fp64 tolerance on every sample;
  * storms whose sequences differ at evaluation k: everything emitted before the step attempt that
    contains evaluation k — i.e. every hourly sample up to that attempt's start time — agrees to the
    same tolerance tiers, and both tracks are at least that long.  Nothing is skipped.

Only tests/ and __graft_entry__.smoke() import this.
"""
import numpy as np

TOL_ALL = 1e-6        # every sample of every decision-identical storm (ensembles of up to a few thousand storms; the
                      # tail grows with the ensemble exactly as the oracle's own response to a one-ulp input change does:
                      # profiles/r02_parity_study.json)
TOL_99 = 1e-8         # 99 % of those storms (max over the storm's samples)
TOL_95 = 1e-9         # 95 % of those storms
# samples before the first differing decision: the same tiers, over the storms that have such a prefix
# (all <= TOL_ALL; 99 % <= TOL_99 and 95 % <= TOL_95 when there are enough of them, else all <= TOL_99)
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


def check_tracks(tag, got, want, dec_got, dec_want, t0_want, t_s, counters=('status', 'n_valid', 'nfev'),
                 flags=('is_tc', 'accepted'), names=('traj', 'envw', 'vmax'), verbose=True, tol_all=TOL_ALL):
    """Assert pointwise / prefix parity of `got` against `want` (dicts of arrays: traj [n,4,ns],
    envw [n,ns,4], vmax [n,ns], status, n_valid, nfev, is_tc, accepted ...).

    dec_*: [n, cap] decision probes (bit0 `land == 1`, bit1 PI != 0, bit2 land within 1e-12 of 1; 0xff
    = not evaluated); t0_want [n, cap]: start time of the step attempt of each evaluation of `want`.
    Returns a summary dict (counts of storms per class, exposure among accepted storms).
    tol_all: the bound on every sample (TOL_ALL; the large-ensemble study passes its own, see
    tests/test_gpu_parity.py::test_parity_study_at_scale)."""
    n = len(want['n_valid'])
    ns = len(t_s)
    k = first_divergence(dec_got, dec_want)
    len_g = (np.asarray(dec_got) != NOT_EVAL).sum(axis=1)
    len_w = (np.asarray(dec_want) != NOT_EVAL).sum(axis=1)
    agree = k < 0
    cap = min(np.asarray(dec_got).shape[1], np.asarray(dec_want).shape[1])
    # no differing decision => the same evaluations were made (beyond the probe's capacity the
    # discrete results below still pin it)
    same_len = (len_g == len_w) | (np.minimum(len_g, len_w) >= cap)
    assert same_len[agree].all(), (tag, 'evaluation count differs without a differing land decision',
                                   np.nonzero(agree & ~same_len)[0][:8])
    # ---- decision-identical storms: full pointwise parity
    for key in tuple(counters) + tuple(flags):
        bad = agree & (np.asarray(got[key]) != np.asarray(want[key]))
        assert not bad.any(), (tag, key, np.nonzero(bad)[0][:8])
    worst = {}
    for name in names:
        d = _storm_maxdiff(np.asarray(got[name])[agree], np.asarray(want[name])[agree])
        worst[name] = float(d.max()) if d.size else 0.0
        if verbose:
            print('%s %-5s identical decisions: max %.3g  p99 %.3g  p95 %.3g   (n=%d)'
                  % (tag, name, worst[name], np.percentile(d, 99) if d.size else 0, np.percentile(d, 95) if d.size else 0, d.size))
        assert worst[name] <= tol_all, (tag, name, worst[name])
        if d.size >= 100:
            assert np.percentile(d, 99) <= TOL_99, (tag, name)
        if d.size >= 20:
            assert np.percentile(d, 95) <= TOL_95, (tag, name)
    # ---- storms with a differing decision: prefix parity up to the attempt that contains it
    div = np.nonzero(~agree)[0]
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
            assert d <= tol_all, (tag, name, 'prefix of storm %d (first differing decision at evaluation %d, '
                                  't = %.0f s, %d samples)' % (i, k[i], t0, n_pref), d)
        pref_samples += n_pref
    if pref_max:
        pm = np.array(pref_max)
        if pm.size >= 300:
            assert np.percentile(pm, 99) <= TOL_99, (tag, 'prefix p99', np.percentile(pm, 99))
        if pm.size >= 60:
            assert np.percentile(pm, 95) <= TOL_95, (tag, 'prefix p95', np.percentile(pm, 95))
        else:
            assert pm.max() <= TOL_99, (tag, 'prefix max', pm.max())
    exposed = ((np.asarray(dec_want) != NOT_EVAL) & ((np.asarray(dec_want) & 6) == 6)).any(axis=1)
    acc = np.asarray(want['accepted'], bool)
    out = dict(n=n, identical=int(agree.sum()), diverged=int(len(div)), exposed=int(exposed.sum()),
               exposed_identical=int((exposed & agree).sum()), accepted=int(acc.sum()),
               accepted_exposed=int((acc & exposed).sum()), accepted_diverged=int((acc & ~agree).sum()),
               prefix_samples=int(pref_samples), prefix_worst=pref_worst, worst=worst)
    # a differing decision can only come from an evaluation the probe marks as rounding-sensitive
    assert (exposed | agree).all(), (tag, 'decision differs at a point that is not within 1e-12 of land == 1',
                                     np.nonzero(~exposed & ~agree)[0][:8])
    if verbose:
        print('%s: %d storms — %d decision-identical (pointwise), %d diverged (prefix-checked, %d samples, worst %.3g); '
              '%d flicker-exposed of which %d pointwise; accepted %d, of which exposed %d (%.0f %%), diverged %d'
              % (tag, n, out['identical'], out['diverged'], pref_samples, pref_worst, out['exposed'],
                 out['exposed_identical'], out['accepted'], out['accepted_exposed'],
                 100.0 * out['accepted_exposed'] / max(1, out['accepted']), out['accepted_diverged']))
    return out
