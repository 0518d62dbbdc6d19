"""TEST INFRASTRUCTURE — CPU restatement of the reference's monthly wind statistics.

`calc_wnd_stat` (track/env_wind.py:180-228) is written on xarray objects, and xarray is not installable here,
so this file writes down — with NumPy only — what those xarray calls execute when the data are NumPy-backed
(no dask chunking of the time axis within a month, no bottleneck: neither is in the reference's
environment.yml), *in the dtype of the file*: ERA5 u / v are float32 and xarray keeps float32 through these
reductions.  The call chain, per statistic (xarray 2023-2025; `skipna` defaults to True for floats):

  DataArray.mean(dim)   (:218)  duck_array_ops.mean -> nanops.nanmean -> np.nanmean(a, axis):
                                NaN -> 0, tot = np.sum(axis=0) in a's dtype (a reduction over the leading,
                                non-contiguous axis: plain accumulation in time order, no pairwise blocks),
                                cnt = non-NaN count, tot / cnt evaluated in float64 and cast back to a's dtype
  DataArray.var(dim)    (:221)  nanops.nanvar -> np.nanvar(a, axis, ddof=0): the mean as above (keepdims),
                                a - mean, NaN -> 0, square, sum in a's dtype, / cnt via float64, cast back
  xr.cov(a, b, dim)     (:223)  computation._cov_corr(ddof=1): valid = a.notnull() & b.notnull();
                                a, b = a.where(valid), b.where(valid); (a - a.mean(dim)) * (b - b.mean(dim)) in
                                a's dtype; .sum(dim, skipna=True, min_count=1) in a's dtype;
                                / (valid.sum(dim) - 1): float32 / int64 -> **float64**
  wnd_stats[i] = stats[i]  (:226-229)  every statistic is widened to float64 on assignment
  groupby("time.day").mean (:199-203) per calendar day np.nanmean(axis=0) — reached only when the time step
                                exceeds one day (the reference tests `dt_step < 0`), i.e. never for sub-daily data

Parity status: **pinned to NumPy's own reductions** (the functions below call np.nanmean / np.nanvar
literally, so "what NumPy does" is not restated but executed); the mapping xarray -> NumPy above is from
reading xarray's source and is **unpinned against a running xarray** (not installable here).  Hand-computed
cases are in tests/test_host_units.py.  Only tests/ may import this file.
"""
import numpy as np

TRIL = [(i, j) for i in range(4) for j in range(i + 1)]


def _cov(a, b):
    """xr.cov(a, b, dim='time') for NumPy-backed [time, ...] arrays of one dtype (ddof = 1)."""
    valid = ~np.isnan(a) & ~np.isnan(b)
    aw, bw = np.where(valid, a, np.nan).astype(a.dtype), np.where(valid, b, np.nan).astype(b.dtype)
    da = aw - np.nanmean(aw, axis=0)
    db = bw - np.nanmean(bw, axis=0)
    prod = da * db
    s = np.sum(np.where(np.isnan(prod), 0, prod).astype(prod.dtype), axis=0)       # nansum, in the data's dtype
    n_valid = valid.sum(axis=0)
    s = np.where(n_valid >= 1, s, np.nan)                                          # min_count = 1
    return s / (n_valid - 1)                                                       # float32 / int64 -> float64


def wind_stats(planes, day_start=None):
    """planes: 4 arrays [n_samples, ...] of one float dtype (float32 as in ERA5 files, or float64);
    returns float64 [14, ...] in the order of `wnd_stats` (:226-229)."""
    dt = np.result_type(*[np.asarray(p).dtype for p in planes])
    dt = dt if dt in (np.float32, np.float64) else np.float64
    x = [np.asarray(p, dtype=dt) for p in planes]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if day_start is not None:
            x = [np.stack([np.nanmean(p[day_start[d]:day_start[d + 1]], axis=0) for d in range(len(day_start) - 1)]) for p in x]
        out = [np.nanmean(p, axis=0).astype(np.float64) for p in x]
        for (i, j) in TRIL:
            if i == j:
                out.append(np.nanvar(x[i], axis=0).astype(np.float64))
            else:
                out.append(np.asarray(_cov(x[i], x[j]), dtype=np.float64))
    return np.stack(out)
