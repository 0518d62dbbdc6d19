"""TEST INFRASTRUCTURE — CPU restatement of the reference's monthly wind statistics.

Parity status: **unpinned against the reference itself** — `calc_wnd_stat` (track/env_wind.py:180-228)
is written on xarray objects and xarray is not installable here, so this file restates what those
xarray calls compute with the NumPy reductions they dispatch to:
    DataArray.mean(dim)       -> x.mean(axis=0)                        (:218)
    DataArray.var(dim)        -> ((x - x.mean(0))**2).mean(axis=0)      ddof = 0   (:221)
    xr.cov(a, b, dim)         -> ((a - a.mean(0)) * (b - b.mean(0))).sum(axis=0) / (n - 1)   ddof = 1   (:223)
    groupby("time.day").mean  -> per calendar day x[day].mean(axis=0)   (:199-203)
It is pinned by hand-computed cases in tests/test_host_units.py.  Only tests/ may import it.
"""
import numpy as np

TRIL = [(i, j) for i in range(4) for j in range(i + 1)]


def wind_stats(planes, day_start=None):
    """planes: 4 arrays [n_samples, ...]; returns [14, ...] in the order of `wnd_stats` (:226-229)."""
    x = [np.asarray(p, dtype=np.float64) for p in planes]
    if day_start is not None:
        x = [np.stack([p[day_start[d]:day_start[d + 1]].mean(axis=0) for d in range(len(day_start) - 1)]) for p in x]
    n = x[0].shape[0]
    mean = [p.mean(axis=0) for p in x]
    out = list(mean)
    for (i, j) in TRIL:
        if i == j:
            out.append(((x[i] - mean[i]) ** 2).mean(axis=0))
        else:
            out.append(((x[i] - mean[i]) * (x[j] - mean[j])).sum(axis=0) / (n - 1))
    return np.stack(out)
