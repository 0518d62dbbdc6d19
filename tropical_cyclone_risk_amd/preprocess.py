"""Preprocessing next to the hot path: monthly wind mean / covariance on the GPU.

Host mirror of `track/env_wind.calc_wnd_stat(ua, va, dt)` (env_wind.py:180-228) over plain arrays:
the month mask (:184-190), the pressure-level pick by units (:192-196), the optional per-day
averaging (:198-206) and the stacking order of the 14 statistics (:226-229) are host logic; the
reduction itself is `k_wind_stats` (csrc/tcr_prep.hip) behind `tcr_wind_stats_*`.

Two things of the reference are reproduced on purpose and named:
  * variances use ddof = 0 (`.var`) but covariances ddof = 1 (`xr.cov`);
  * `dt_step = 1 day - step` is tested with `< 0`, so the per-day averaging only happens for records
    *coarser* than daily — sub-daily records are reduced sample by sample (group_days='reference').
    group_days=True gives what the comment in the reference says (daily means first).
The reduction runs in the dtype of the input, as xarray's does: float32 planes (ERA5 files) are reduced in
float32 (`tcr_wind_stats_f32_*`), anything else in float64; NaN samples are skipped (`skipna`).
"""
import ctypes as C
import datetime

import numpy as np

from . import _lib

MEAN_NAMES = ['ua250_Mean', 'va250_Mean', 'ua850_Mean', 'va850_Mean']


def month_mask(times, year, month):
    """env_wind.py:184-190: samples in [first of the month, first of the next month)."""
    t0 = datetime.datetime(year, month, 1)
    t1 = datetime.datetime(year + 1, 1, 1) if month == 12 else datetime.datetime(year, month + 1, 1)
    return np.array([(t >= t0) and (t < t1) for t in times], dtype=bool)


def pick_levels(levels, units):
    """env_wind.py:192-196: 250 / 850 in hPa or 25000 / 85000 in Pa; returns (i_upper, i_lower)."""
    levels = np.asarray(levels)
    up, lo = (250, 850) if units in ('millibars', 'hPa') else (25000, 85000)
    iu, il = np.where(levels == up)[0], np.where(levels == lo)[0]
    if len(iu) != 1 or len(il) != 1:
        raise KeyError('levels %s do not contain %s and %s' % (levels, up, lo))
    return int(iu[0]), int(il[0])


def day_groups(times):
    """Sample offsets of the calendar days of a time-sorted month (groupby("time.day"))."""
    days = np.array([t.day for t in times])
    start = [0] + [i for i in range(1, len(days)) if days[i] != days[i - 1]] + [len(days)]
    return np.asarray(start, dtype=np.int32)


def calc_wnd_stat(engine, ua, va, levels, level_units, times, year, month, group_days='reference'):
    """ua, va: [time, level, lat, lon]; times: datetimes (sorted).  Returns wnd_stats [14, lat, lon]."""
    ua, va = np.asarray(ua), np.asarray(va)
    keep = month_mask(times, year, month)
    t_sel = [t for t, k in zip(times, keep) if k]
    iu, il = pick_levels(levels, level_units)
    is_f32 = lambda a: a.dtype.kind == 'f' and a.dtype.itemsize == 4        # any byte order (NetCDF-3 is big-endian)
    dt = np.float32 if (is_f32(ua) and is_f32(va)) else np.float64
    planes = [np.ascontiguousarray(a[keep][:, lev], dtype=dt)
              for a, lev in ((ua, iu), (va, iu), (ua, il), (va, il))]
    step = (times[1] - times[0]).total_seconds()
    if group_days == 'reference':
        group_days = (86400.0 - step) < 0          # env_wind.py:198-199, as written
    ds = day_groups(t_sel) if group_days else None
    return engine.wind_stats(planes, ds)


def wind_stats_host(engine, planes, day_start=None):
    """planes: 4 arrays [n_samples, ...] (same trailing shape) -> float64 [14, ...].  All float32: reduced in
    float32 like xarray does on ERA5 files; otherwise in float64."""
    f32 = all(np.asarray(p).dtype.kind == 'f' and np.asarray(p).dtype.itemsize == 4 for p in planes)
    planes = [np.ascontiguousarray(p, dtype=np.float32 if f32 else np.float64) for p in planes]
    n = planes[0].shape[0]
    shape = planes[0].shape[1:]
    npts = int(np.prod(shape))
    for p in planes:
        if p.shape != planes[0].shape:
            raise ValueError('the four wind components must have the same shape')
    out = np.empty((14,) + shape)
    ptrs = (C.c_void_p * 4)(*[p.ctypes.data for p in planes])
    ds = None if day_start is None else np.ascontiguousarray(day_start, dtype=np.int32)
    nd = 0 if ds is None else len(ds) - 1
    fn = engine.L.tcr_wind_stats_f32_host if f32 else engine.L.tcr_wind_stats_host
    engine._ck(fn(engine.h, n, npts, ptrs, None if ds is None else ds.ctypes.data, nd, out.ctypes.data))
    return out


# --------------------------------------------------------------------------------------
# thermodynamic preprocessing
def load_entropy_table(fn=None, nl=None):
    """The reference's `thermo/entropy_table.npz` (thermo.py:272-277): p [Pa], s [J/kg/K], T[p][s].
    The table was made by a Nelder-Mead inversion (thermo.py:451-469), so results match the reference's
    only with *its* table; point `fn` (or namelist.src_directory) at it."""
    import os
    from . import namelist as default_namelist
    nl = nl or default_namelist
    fn = fn or os.path.join(nl.src_directory, 'thermo', 'entropy_table.npz')
    with np.load(fn) as t:
        return np.array(t['p'], dtype=np.float64), np.array(t['s'], dtype=np.float64), np.array(t['T'], dtype=np.float64)


def stage_entropy_table(engine, p, s, T):
    p, s, T = (np.ascontiguousarray(x, dtype=np.float64) for x in (p, s, T))
    if T.shape != (len(p), len(s)):
        raise ValueError('T must be [len(p), len(s)]')
    dp = lambda a: a.ctypes.data_as(_lib.DP)
    engine._ck(engine.L.tcr_entropy_table_upload(engine.h, len(p), len(s), dp(p), dp(s), dp(T)))


def potential_intensity(engine, sst, p_surf, p_env, T_env, r_env, nl=None):
    """thermo.CAPE_PI_vectorized(sst, p_surf, p_env, T_env, r_env): p_env [L] Pa lowest level first,
    T_env / r_env [L, lat, lon]; returns PI [lat, lon]."""
    from . import namelist as default_namelist
    nl = nl or default_namelist
    sst, p_surf = np.ascontiguousarray(sst, dtype=np.float64), np.ascontiguousarray(p_surf, dtype=np.float64)
    T_env, r_env = np.ascontiguousarray(T_env, dtype=np.float64), np.ascontiguousarray(r_env, dtype=np.float64)
    p_env = np.ascontiguousarray(p_env, dtype=np.float64)
    L = len(p_env)
    if T_env.shape != (L,) + sst.shape or r_env.shape != T_env.shape or p_surf.shape != sst.shape:
        raise ValueError('shapes: sst/p_surf [..], T_env/r_env [L, ..]')
    out = np.empty(sst.shape)
    dp = lambda a: a.ctypes.data_as(_lib.DP)
    engine._ck(engine.L.tcr_potential_intensity_host(engine.h, sst.size, L, dp(p_env), dp(sst), dp(p_surf), dp(T_env),
                                                     dp(r_env), float(nl.Ck / nl.Cd), dp(out)))
    return out


def chi_rh(engine, sst, p_surf, T_mid, q_mid, p_mid):
    """(thermo.sat_deficit, thermo.conv_q_to_rh) at the mid level; chi is not clipped here."""
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (sst, p_surf, T_mid, q_mid)]
    chi, rh = np.empty(a[0].shape), np.empty(a[0].shape)
    dp = lambda x: x.ctypes.data_as(_lib.DP)
    engine._ck(engine.L.tcr_chi_rh_host(engine.h, a[0].size, dp(a[0]), dp(a[1]), dp(a[2]), dp(a[3]), float(p_mid), dp(chi), dp(rh)))
    return chi, rh


def compute_thermo(engine, sst_K, psl, levels, level_units, ta, hus, nl=None):
    """One time sample of thermo/calc_thermo.compute_thermo (:36-74) over plain arrays already on the
    atmospheric grid: orders the levels lowest first (:50-54), converts hPa to Pa (:56-59), picks the
    level nearest namelist.p_midlevel (:59, 65-68), and returns (vmax, chi clipped to [0, 10], rh_mid).
    As in the reference, specific humidity is handed to the PI routine as if it were mixing ratio (:63)."""
    from . import namelist as default_namelist
    nl = nl or default_namelist
    lev = np.array(levels, dtype=np.float64)
    ta, hus = np.asarray(ta, dtype=np.float64), np.asarray(hus, dtype=np.float64)
    if lev[0] - lev[1] < 0:
        lev, ta, hus = lev[::-1], ta[::-1], hus[::-1]
    if level_units in ('millibars', 'hPa'):
        lev = lev * 100
    k_mid = int(np.argmin(np.abs(lev - nl.p_midlevel)))
    vmax = potential_intensity(engine, sst_K, psl, lev, ta, hus, nl)
    chi, rh = chi_rh(engine, sst_K, psl, ta[k_mid], hus[k_mid], float(lev[k_mid]))
    return vmax, np.minimum(np.maximum(chi, 0), 10), rh


# --------------------------------------------------------------------------------------
# file drivers: the reference's gen_wind_mean_cov / gen_thermo over NetCDF files
def _datetimes(ta):
    """Datetimes of a fields.TimeAxis (input.convert_to_datetime)."""
    out = []
    for sec in ta.t:
        if ta.calendar in ('noleap', '365_day'):
            days, rem = divmod(sec, 86400.0)
            y, doy = divmod(int(days), 365)
            mo = int(np.searchsorted(np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]), doy, side='right'))
            d = doy - int(([0] + list(np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])))[mo])
            out.append(datetime.datetime(y + 1, mo + 1, d + 1) + datetime.timedelta(seconds=rem))
        else:
            out.append(datetime.datetime(1, 1, 1) + datetime.timedelta(seconds=float(sec)))
    return out


def _bounding_times(nl):
    """input.get_bounding_times (:135-139)."""
    import calendar
    s = datetime.datetime(nl.start_year, nl.start_month, 1)
    e = datetime.datetime(nl.end_year, nl.end_month, calendar.monthrange(nl.end_year, nl.end_month)[1])
    return s, e


def _write_nc3(fn, coords, variables, time_days, year0):
    from scipy.io import netcdf_file
    with netcdf_file(fn, 'w', version=2) as f:
        f.createDimension('time', len(time_days)); f.createDimension('lat', len(coords['lat'])); f.createDimension('lon', len(coords['lon']))
        v = f.createVariable('time', 'd', ('time',)); v[:] = time_days
        v.units = 'days since %04d-01-01 00:00:00' % year0; v.calendar = 'standard'
        for k in ('lat', 'lon'):
            v = f.createVariable(k, 'd', (k,)); v[:] = np.asarray(coords[k], dtype=np.float64)
        for name, arr in variables.items():
            v = f.createVariable(name, 'd', ('time', 'lat', 'lon')); v[:] = np.asarray(arr, dtype=np.float64)


def gen_wind_mean_cov(engine, fns_ua, fns_va, out_fn, nl=None, group_days='reference'):
    """track/env_wind.gen_wind_mean_cov + wnd_stat_wrapper (:83-176) for pairs of daily u / v files:
    every month covered by a file pair between the namelist's bounding times gets its 14 statistics
    (GPU), and the result is written with the reference's variable names (`<c>_Mean`, `<c>_Var`,
    `<a>_<b>_cov`) on (time, lat, lon).  As in the reference the first record of a file is stamped with
    the start of its period and the following ones with the 15th of their month (:139-152)."""
    from . import fields, namelist as default_namelist
    nl = nl or default_namelist
    vk = nl.var_keys[nl.dataset_type]
    dt_start, dt_end = _bounding_times(nl)
    stamps, stats, grid = [], [], None
    for fu, fv in zip(fns_ua, fns_va):
        du, dv = fields._Dataset(fu), fields._Dataset(fv)
        times = _datetimes(fields.TimeAxis(du['time'], du.attrs['time']))
        t0 = max(dt_start, times[0])
        t_months = [t0]
        while t_months[-1] <= min(dt_end, times[-1]):
            y, m = t_months[-1].year, t_months[-1].month
            t_months.append(datetime.datetime(y + 1, 1, 15) if m == 12 else datetime.datetime(y, m + 1, 15))
        t_months = t_months[:-1]
        units = str(du.attrs[vk['lvl']].get('units', 'Pa'))
        for tm in t_months:
            stats.append(calc_wnd_stat(engine, du[vk['u']], dv[vk['v']], du[vk['lvl']], units, times, tm.year, tm.month, group_days))
            stamps.append(tm)
        grid = dict(lat=du[vk['lat']], lon=du[vk['lon']])
    if not stats:
        raise ValueError('no month of the namelist period is covered by the given files')
    stats = np.stack(stats)                                    # [time, 14, lat, lon]
    names = MEAN_NAMES + [n for n in _cov_names()]
    year0 = stamps[0].year
    days = [(t - datetime.datetime(year0, 1, 1)).total_seconds() / 86400.0 for t in stamps]
    _write_nc3(out_fn, grid, {n: stats[:, i] for i, n in enumerate(names)}, days, year0)
    return out_fn


def _cov_names():
    from .fields import cov_name
    return [cov_name(i, j) for i in range(4) for j in range(i + 1)]


class MultiFile:
    """`xr.open_mfdataset(fns, concat_dim="time", combine="nested")` (util/input.py:14-21) over the dataset facade:
    one file or a list of files of one variable; records are concatenated along time in the order of the list (the
    reference passes its sorted glob), coordinates and attributes come from the first file, and every file's time axis
    is decoded with its own units / calendar."""

    def __init__(self, fns):
        from . import fields
        fns = [fns] if isinstance(fns, str) else list(fns)
        if not fns:
            raise ValueError('no input file')
        self.parts = [fields._Dataset(f) for f in fns]
        self.attrs = self.parts[0].attrs
        self.n_time = [len(np.atleast_1d(p['time'])) for p in self.parts]
        self.times = [t for p in self.parts for t in _datetimes(fields.TimeAxis(p['time'], p.attrs['time']))]
        if any(b < a for a, b in zip(self.times, self.times[1:])):
            raise ValueError('files are not in time order: %s' % fns)

    def __getitem__(self, k):
        first = np.asarray(self.parts[0][k])
        # Coordinates come from the first file (combine="nested" concatenates data variables along `time` only).  Every
        # 1-D variable other than `time` is a coordinate — a level axis whose length happens to equal the first file's
        # record count (12 levels, 12 monthly records) must not be concatenated.
        is_coord = first.ndim == 0 or (first.ndim == 1 and k != 'time') or first.shape[0] != self.n_time[0]
        if len(self.parts) == 1 or is_coord:
            return self.parts[0][k]
        return np.concatenate([np.asarray(p[k]) for p in self.parts], axis=0)

    def record(self, k, i):
        """Record i (along the concatenated time axis) of variable k without concatenating the files."""
        for p, n in zip(self.parts, self.n_time):
            if i < n:
                return np.asarray(p[k])[i]
            i -= n
        raise IndexError(i)


def gen_thermo(engine, fn_sst, fn_mslp, fn_temp, fn_sp_hum, out_fn, nl=None, table=None):
    """thermo/calc_thermo.gen_thermo + compute_thermo (:24-117) for monthly sst / mslp / temperature /
    specific-humidity files: sst is regridded bilinearly (after nan_to_num) onto the atmospheric grid
    (:37-41, Celsius -> Kelvin by the units attribute), PI / chi / rh_mid come from the GPU kernels, and
    records are stamped on the 15th of their month (:101-106).  Axes must ascend in latitude (the
    reference's RectBivariateSpline regridding needs that too).  Every fn_* is one file or a list of files in time
    order, as the reference's `open_mfdataset` over its sorted glob (util/input.py:14-58)."""
    from . import fields, namelist as default_namelist
    nl = nl or default_namelist
    vk = nl.var_keys[nl.dataset_type]
    if table is not None:
        stage_entropy_table(engine, *table)
    ds_sst, ds_psl, ds_ta, ds_q = (MultiFile(f) for f in (fn_sst, fn_mslp, fn_temp, fn_sp_hum))
    dt_start, dt_end = _bounding_times(nl)
    times = ds_psl.times
    for d in (ds_sst, ds_ta, ds_q):
        if d.times != times:
            raise ValueError('the monthly variables do not share one time axis')
    keep = [i for i, t in enumerate(times) if dt_start <= t <= dt_end]
    lon_a, lat_a = np.asarray(ds_ta[vk['lon']], dtype=np.float64), np.asarray(ds_ta[vk['lat']], dtype=np.float64)
    lev = ds_ta[vk['lvl']]
    lev_units = str(ds_ta.attrs[vk['lvl']].get('units', 'Pa'))
    celsius = 'C' in str(ds_sst.attrs[vk['sst']].get('units', 'K'))
    vmax, chi, rh = [], [], []
    for i in keep:
        sst = fields.interp_2d_grid(ds_sst[vk['lon']], ds_sst[vk['lat']], np.nan_to_num(np.asarray(ds_sst.record(vk['sst'], i), dtype=np.float64)),
                                    lon_a, lat_a)
        if celsius:
            sst = sst + 273.15
        v, c, r = compute_thermo(engine, sst, ds_psl.record(vk['mslp'], i), lev, lev_units, ds_ta.record(vk['temp'], i),
                                 ds_q.record(vk['sp_hum'], i), nl)
        vmax.append(v); chi.append(c); rh.append(r)
    stamps = [datetime.datetime(times[i].year, times[i].month, 15) for i in keep]
    year0 = stamps[0].year
    days = [(t - datetime.datetime(year0, 1, 1)).total_seconds() / 86400.0 for t in stamps]
    _write_nc3(out_fn, dict(lat=ds_psl[vk['lat']], lon=ds_psl[vk['lon']]),
               dict(vmax=np.stack(vmax), chi=np.stack(chi), rh_mid=np.stack(rh)), days, year0)
    return out_fn


def glob_prefix(nl, var_key):
    """input._glob_prefix (:22-27): files under base_directory whose name carries the experiment prefix
    and `_<var>_` (or `<var>_`)."""
    import glob
    fns = glob.glob('%s/**/*%s*.nc' % (nl.base_directory, nl.exp_prefix), recursive=True)
    out = sorted(x for x in fns if '_%s_' % var_key in x)
    return out or sorted(x for x in fns if '%s_' % var_key in x)


def run_preprocessing(engine, nl=None, table=None):
    """What the reference's run.py does before the downscaling (run.py:14-15): write env_wnd_*.nc and
    thermo_*.nc next to the data unless they exist."""
    import os
    from . import fields, namelist as default_namelist
    nl = nl or default_namelist
    fl = fields.default_files(nl)
    vk = nl.var_keys[nl.dataset_type]
    if not os.path.exists(fl['env_wnd']):
        fu, fv = glob_prefix(nl, vk['u']), glob_prefix(nl, vk['v'])
        n = min(len(fu), len(fv))
        gen_wind_mean_cov(engine, fu[:n], fv[:n], fl['env_wnd'], nl)
        print('Saved %s' % fl['env_wnd'])
    if not os.path.exists(fl['thermo']):
        one = {}
        for k in ('sst', 'mslp', 'temp', 'sp_hum'):
            one[k] = glob_prefix(nl, vk[k])                  # all files of the variable, in name order (input._load_var)
            if not one[k]:
                raise FileNotFoundError('no %s file under %s' % (vk[k], nl.base_directory))
        gen_thermo(engine, one['sst'], one['mslp'], one['temp'], one['sp_hum'], fl['thermo'], nl,
                   table=table or load_entropy_table(nl=nl))
        print('Saved %s' % fl['thermo'])
    return fl
