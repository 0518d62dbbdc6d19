"""(test utility, not collected by pytest: lives under tests/ because it calls the oracle, which only tests may do)
Parity distribution of ONE build of the library against the golden fixtures and the C oracle (GPU box):
    python tests/parity_probe.py [LIB.so] [--n 10000]
Prints, per set, the per-storm maximum |GPU - reference| over the hourly lon / lat / v / m of the decision-identical storms:
counts above the tiers of oracle/parity.py (1e-9: allowed n // 100 + 1; 2e-11: n // 20 + 2; 1e-7: the floor), percentiles,
and the number of decision-identical storms whose counters (status / n_valid / nfev / n_accept / n_reject) differ.
The oracle runs are cached in /tmp (they do not depend on the library)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tropical_cyclone_risk_amd import _lib, synthetic            # noqa: E402
args = sys.argv[1:]
n_ens = 10000
if '--n' in args:
    i = args.index('--n'); n_ens = int(args[i + 1]); del args[i:i + 2]
if args:
    _lib.LIB_PATH = os.path.abspath(args[0])
from tropical_cyclone_risk_amd.engine import TCEngine            # noqa: E402
from oracle import c_oracle, parity                              # noqa: E402
G = os.path.join(ROOT, 'tests', 'golden')
meta = np.load(os.path.join(G, 'tracks_NA.npz'))
env = synthetic.make_env(shape=str(meta['meta_env_shape']), seed=int(meta['meta_env_seed']), zero_cov_patch=bool(meta['meta_env_zero_cov_patch']))
CAP = 1024


def report(tag, got, want, dec_w):
    k = parity.first_divergence(np.asarray(got['dec']), np.asarray(dec_w))
    agree = k < 0
    a, b = np.asarray(got['traj'])[agree], np.asarray(want['traj'])[agree]
    both = ~np.isnan(a) & ~np.isnan(b)
    d = np.where(both, np.abs(np.where(both, a, 0.0) - np.where(both, b, 0.0)), 0.0).reshape(a.shape[0], -1).max(axis=1)
    bad = 0
    for key in ('status', 'n_valid', 'nfev', 'n_accept', 'n_reject'):
        if key in want and key in got:
            bad += int((agree & (np.asarray(got[key]) != np.asarray(want[key]))).sum())
    n = d.size
    print('%-12s n %5d (identical decisions %5d of %5d)  >1e-7 %3d  >1e-9 %4d (allowed %4d)  >2e-11 %4d (allowed %4d)  p50 %.2e p95 %.2e p99 %.2e p99.9 %.2e max %.2e  counter flips %d'
          % (tag, n, int(agree.sum()), len(agree), int((d > 1e-7).sum()), int((d > 1e-9).sum()), n // 100 + 1, int((d > 2e-11).sum()), n // 20 + 2,
             np.percentile(d, 50), np.percentile(d, 95), np.percentile(d, 99), np.percentile(d, 99.9), d.max(), bad), flush=True)


for basin in ('NA', 'AU', 'GL'):
    g = np.load(os.path.join(G, 'tracks_%s.npz' % basin))
    storms = dict(lon=g['lon0'], lat=g['lat0'], v0=g['v0'], m0=g['m0'], h_bl=g['h_bl'], month=g['month'], phases=g['phases'])
    eng = TCEngine(basin, device=0).stage_env(env)
    out = eng.integrate(storms, probe_cap=CAP)
    report('golden-' + basin, out, g, parity.ragged_to_padded(g['dec'], g['dec_off'], CAP))
    if basin == 'NA':
        st = synthetic.draw_storm_inputs(n_ens, 'NA', seed=77)
        fn = '/tmp/parity_probe_oracle_NA_%d.npz' % n_ens
        if os.path.exists(fn):
            ref = dict(np.load(fn))
        else:
            ref = c_oracle.run_ensemble(env, 'NA', st, probe=True, post=False)
            np.savez(fn, **{k: v for k, v in ref.items() if k in ('traj', 'dec', 'status', 'n_valid', 'nfev', 'n_accept', 'n_reject')})
        got = eng.integrate(st, probe_cap=CAP)
        report('ens-NA-%d' % n_ens, got, ref, ref['dec'])
    eng.close()
