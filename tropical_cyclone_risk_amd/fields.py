"""Input-field loader: the reference's preprocessed files -> one year's environment.

What `util/compute.run_tracks` does before its seeding loop (`compute.py:64-118`), for the files the
reference's preprocessing leaves behind:

  ``thermo_<prefix>_<dates>.nc``   vmax, rh_mid, chi [time, lat, lon]     (thermo/calc_thermo.py:16-21)
  ``env_wnd_<prefix>_<dates>.nc``  ua250_Mean ... va850_va850_cov [time, lat, lon]
                                                                    (track/env_wind.py:20-47, 83-118)
  ``intensity/data/{mld,strat}_climatology.nc``  mixed_layer / strat [lat, lon, month]
                                                                    (intensity/ocean.py:11-64)
  ``intensity/data/{land,bathymetry}.nc``        land / bathymetry [lat, lon]   (intensity/geo.py:9-34)
  ``land/<B>.nc``                                basin [lat, lon]               (compute.py:87-97)

and returns the same 12-month object `synthetic.make_env` returns, so `engine.stage_env`,
`compute.run_tracks` and `run.py` work on real fields unchanged.

Files are read through xarray when it is installed; otherwise NetCDF-3 (classic / 64-bit offset) through
``scipy.io.netcdf_file`` and NetCDF-4 through the built-in minimal HDF5 reader `hdf5lite` (neither the
build nor the bench container has xarray, netCDF4 or h5py).  `write_reference_files` writes the same
schema as NetCDF-3, which is what the round-trip tests read and which lets the *reference* be run on
this project's synthetic fields; the HDF5 path is tested on the reference's own `intensity/data/land.nc`.

Time handling follows the reference: the thermo record is cut to ``[Dec 31 of year-1, Dec 31 of
year]`` and interpolated linearly to the 15th of every month (NaN outside the record: vpot -> 0,
chi -> 5, `compute.py:108-113`); wind statistics are interpolated to the same dates over the whole
file (`bam_track.py:86-91`).  Calendars: standard/gregorian and noleap (`util/input.py:111-133`).
"""
import datetime
import os
import re

import numpy as np
from scipy.interpolate import RectBivariateSpline

from . import namelist as _default_namelist
from .basins import BASIN_IDS, TC_Basin

from .synthetic import N_WIND, TRIL, SyntheticEnv, chi_transform

MEAN_NAMES = ['ua250_Mean', 'va250_Mean', 'ua850_Mean', 'va850_Mean']       # env_wind.py:20-24


def cov_name(i, j):
    """env_wind.py:29-41: variance `<a>_Var`, covariance `<a>_<b>_cov` (i >= j, names without _Mean)."""
    base = [x[:-5] for x in MEAN_NAMES]
    return base[i] + '_Var' if i == j else base[i] + '_' + base[j] + '_cov'


# --------------------------------------------------------------------------------------
# minimal dataset facade
class _Dataset:
    """Variables as float64/int NumPy arrays (fill values -> NaN, scale/offset applied) + attributes."""

    def __init__(self, fn):
        self.fn = fn
        self.vars, self.attrs, self.dims = {}, {}, {}
        try:
            import xarray as xr
        except Exception:
            xr = None
        if xr is not None:
            ds = xr.open_dataset(fn, decode_times=False)
            for k in list(ds.variables):
                self.vars[k] = np.asarray(ds[k].values)
                self.attrs[k] = dict(ds[k].attrs)
                self.dims[k] = tuple(ds[k].dims)
            ds.close()
            return
        with open(fn, 'rb') as f:
            magic = f.read(4)
        if magic == b'\x89HDF':                                 # NetCDF-4: the built-in minimal HDF5 reader
            from . import hdf5lite
            for k, (a, at) in hdf5lite.read_variables(fn).items():
                self.vars[k], self.attrs[k], self.dims[k] = a, at, ()
            return
        if magic[:3] != b'CDF':
            raise RuntimeError('%s is neither NetCDF-3 nor HDF5 (magic %r)' % (fn, magic))
        from scipy.io import netcdf_file
        with netcdf_file(fn, 'r', mmap=False, maskandscale=True) as f:
            for k, v in f.variables.items():
                a = v[:]
                if np.ma.isMaskedArray(a):
                    a = a.astype(np.float64).filled(np.nan) if a.dtype.kind == 'f' or a.mask.any() else a.data
                self.vars[k] = np.array(a)
                self.attrs[k] = {n: (x.decode() if isinstance(x, bytes) else x)
                                 for n, x in v._attributes.items()}
                self.dims[k] = tuple(v.dimensions)

    def __getitem__(self, k):
        return self.vars[k]

    def __contains__(self, k):
        return k in self.vars

    def first(self, *names):
        for n in names:
            if n in self.vars:
                return n
        raise KeyError('%s: none of %s among %s' % (self.fn, names, sorted(self.vars)))


# --------------------------------------------------------------------------------------
# CF time axis -> seconds on a calendar-specific linear scale
_CUM = np.cumsum([0, 31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30])


def _parse_units(units):
    m = re.match(r'\s*(\w+)\s+since\s+(\d+)-(\d+)-(\d+)(?:[ T](\d+):(\d+)(?::(\d+(?:\.\d*)?))?)?', units)
    if not m:
        raise ValueError('cannot parse time units %r' % units)
    scale = {'seconds': 1.0, 'second': 1.0, 'minutes': 60.0, 'minute': 60.0, 'hours': 3600.0, 'hour': 3600.0,
             'days': 86400.0, 'day': 86400.0}[m.group(1).lower()]
    y, mo, d = int(m.group(2)), int(m.group(3)), int(m.group(4))
    hh, mi, ss = int(m.group(5) or 0), int(m.group(6) or 0), float(m.group(7) or 0)
    return scale, (y, mo, d, hh, mi, ss)


def _linear_seconds(calendar, y, mo, d, hh=0, mi=0, ss=0.0):
    """Seconds since 0001-01-01 on the file's calendar."""
    if calendar in ('noleap', '365_day'):
        days = 365 * (y - 1) + int(_CUM[mo - 1]) + (d - 1)
    else:
        days = (datetime.date(y, mo, d) - datetime.date(1, 1, 1)).days
    return days * 86400.0 + hh * 3600.0 + mi * 60.0 + ss


class TimeAxis:
    def __init__(self, values, attrs):
        self.calendar = str(attrs.get('calendar', 'standard')).lower()
        if self.calendar not in ('standard', 'gregorian', 'proleptic_gregorian', 'noleap', '365_day'):
            raise ValueError('unsupported calendar %r (input.py:111-133 knows datetime64 and noleap)' % self.calendar)
        scale, ref = _parse_units(str(attrs['units']))
        self.t = _linear_seconds(self.calendar, *ref) + np.asarray(values, dtype=np.float64) * scale

    def at(self, y, mo, d):
        return _linear_seconds(self.calendar, y, mo, d)


def interp_time(t_axis, values, t_new):
    """`DataArray.interp(time=...)` = scipy interp1d(kind='linear', bounds_error=False): bracket by
    searchsorted, slope * (x - x_lo) + y_lo, NaN outside the record.  values is [time, ...]."""
    t_axis = np.asarray(t_axis, dtype=np.float64)
    if t_new < t_axis[0] or t_new > t_axis[-1] or len(t_axis) < 2:
        if len(t_axis) == 1 and t_new == t_axis[0]:
            return np.array(values[0], dtype=np.float64)
        return np.full(values.shape[1:], np.nan)
    hi = int(np.clip(np.searchsorted(t_axis, t_new), 1, len(t_axis) - 1))
    lo = hi - 1
    slope = (values[hi] - values[lo]) / (t_axis[hi] - t_axis[lo])
    return slope * (t_new - t_axis[lo]) + values[lo]


def _interp2_fx(lon, lat, X):
    """util/mat.py:142-154 `interp2_fx`: RectBivariateSpline(lon, lat, X.T, kx=1, ky=1), with a descending
    latitude axis (and the field with it) reversed first, since the spline needs increasing axes."""
    lat, X = np.asarray(lat), np.asarray(X)
    if lat[1] - lat[0] < 0:
        lat, X = np.flip(lat, 0), np.flip(X, 0)
    return RectBivariateSpline(lon, lat, X.T, kx=1, ky=1)


def _ascending_lat(lat, *fields):
    """North-to-south files: flip to the ascending latitude the staging (and FITPACK) requires.  The reference
    does this for every field it reads through `interp2_fx` (mld / strat, land/<B>.nc, rh_mid) and for the thermo
    record (compute.py:80-84); flipping is exact, so it is applied to every [.., lat, lon] input here."""
    lat = np.asarray(lat, dtype=np.float64)
    if lat[0] - lat[1] > 0:
        return (lat[::-1].copy(),) + tuple(np.asarray(f)[..., ::-1, :].copy() for f in fields)
    return (lat,) + tuple(fields)


def interp_2d_grid(lon, lat, X, lon_grid, lat_grid):
    """util/mat.py:159-164."""
    LON, LAT = np.meshgrid(lon_grid, lat_grid)
    return _interp2_fx(lon, lat, X).ev(LON, LAT)


# --------------------------------------------------------------------------------------
def default_files(nl):
    """Where the reference looks (calc_thermo.py:16-21, env_wind.py:12-18, geo.py, ocean.py, compute.py:91)."""
    tag = '%s_%d%02d_%d%02d' % (nl.exp_prefix, nl.start_year, nl.start_month, nl.end_year, nl.end_month)
    src = nl.src_directory
    return dict(thermo='%s/thermo_%s.nc' % (nl.output_directory, tag),
                env_wnd='%s/env_wnd_%s.nc' % (nl.output_directory, tag),
                mld='%s/intensity/data/mld_climatology.nc' % src,
                strat='%s/intensity/data/strat_climatology.nc' % src,
                land='%s/intensity/data/land.nc' % src,
                bathy='%s/intensity/data/bathymetry.nc' % src,
                basin_dir='%s/land' % src)


def _climatology(fn, var, lon_t, lat_t):
    """ocean.py:11-64 + compute.py:116-117: [lat, lon, month] with a wrap column, regridded bilinearly
    (after nan_to_num) onto the thermo grid for each month."""
    ds = _Dataset(fn)
    X = np.asarray(ds[var], dtype=np.float64)
    lon, lat = np.asarray(ds['lon'], dtype=np.float64), np.asarray(ds['lat'], dtype=np.float64)
    gl = TC_Basin('GL')
    out = []
    for i in range(12):
        lon_b, lat_b, Xb = gl.transform_global_field(lon[0:-1], lat, X[:, 0:-1, i])
        out.append(interp_2d_grid(lon_b, lat_b, np.nan_to_num(Xb), lon_t, lat_t))
    return np.stack(out)


def load_year_env(year, nl=None, files=None, cache=None):
    """One year's 12 monthly field sets from the reference's files (see the module docstring).  `cache` (a dict the caller
    keeps between years, e.g. FileEnvironment's): the parsed files and everything that does not depend on the year — land,
    bathymetry, basin masks, ocean climatologies — are read once, and the static arrays of consecutive years are the SAME
    objects, which is how `TCEngine.stage_env` knows not to stage them again."""
    nl = nl or _default_namelist
    fl = default_files(nl)
    fl.update(files or {})
    cache = {} if cache is None else cache

    def opened(key):
        k = ('file', fl[key])
        if k not in cache:
            cache[k] = _Dataset(fl[key])
        return cache[k]

    # ---- thermo record of the year (compute.py:66-85)
    ds = opened('thermo')
    ta = TimeAxis(ds['time'], ds.attrs['time'])
    keep = (ta.t >= ta.at(year - 1, 12, 31)) & (ta.t <= ta.at(year, 12, 31))
    tt = ta.t[keep]
    if not keep.any():
        raise ValueError('%s holds no record between %d-12-31 and %d-12-31' % (fl['thermo'], year - 1, year))
    lon = np.asarray(ds['lon'], dtype=np.float64)
    lat = np.asarray(ds['lat'], dtype=np.float64)
    vpot_all = np.asarray(ds['vmax'], dtype=np.float64)[keep] * nl.PI_reduc * np.sqrt(nl.Ck / nl.Cd)
    rh_all = np.asarray(ds['rh_mid'], dtype=np.float64)[keep]
    chi_all = np.asarray(ds['chi'], dtype=np.float64)[keep]
    if lat[0] - lat[1] > 0:
        lat = lat[::-1]
        vpot_all, rh_all, chi_all = vpot_all[:, ::-1], rh_all[:, ::-1], chi_all[:, ::-1]

    # ---- wind statistics (bam_track.py:76-91)
    dw = opened('env_wnd')
    tw = TimeAxis(dw['time'], dw.attrs['time'])
    wlon = np.asarray(dw['lon'], dtype=np.float64)
    wlat = np.asarray(dw['lat'], dtype=np.float64)
    wflip = wlat[0] - wlat[1] > 0          # RectBivariateSpline needs ascending axes; the reference's files are ascending
    if wflip:
        wlat = wlat[::-1]

    vpot, rh_mid, chi, mean, cov = [], [], [], [], []
    for mo in range(1, 13):
        t_th = ta.at(year, mo, 15)
        vpot.append(np.nan_to_num(interp_time(tt, vpot_all, t_th), nan=0.0))
        rh_mid.append(interp_time(tt, rh_all, t_th))
        chi.append(chi_transform(interp_time(tt, chi_all, t_th)))
        t_w = tw.at(year, mo, 15)
        get = lambda name: np.nan_to_num(interp_time(tw.t, np.asarray(dw[name], dtype=np.float64), t_w)[::-1 if wflip else 1])
        mean.append(np.stack([get(n) for n in MEAN_NAMES]))
        cov.append(np.stack([get(cov_name(i, j)) for (i, j) in TRIL]))

    skey = ('static', fl['mld'], fl['strat'], fl['land'], fl['bathy'], fl['basin_dir'], lon.tobytes(), lat.tobytes())
    if skey not in cache:
        cache[skey] = _load_static(fl, lon, lat)
    mld, strat, hlon, hlat, land, bathy, blon, blat, same_static, masks, mgrid = cache[skey]
    env = SyntheticEnv(lon=lon, lat=lat, wlon=wlon, wlat=wlat, wnd_mean=np.stack(mean), wnd_cov=np.stack(cov),
                       vpot=np.stack(vpot), chi=np.stack(chi), mld=mld, strat=strat, rh_mid=np.stack(rh_mid),
                       hlon=hlon, hlat=hlat, land=land, bathy=bathy, basin_masks=masks, seed=0, shape='files')
    if mgrid is not None and not (np.array_equal(mgrid[0], hlon) and np.array_equal(mgrid[1], hlat)):
        env.mlon, env.mlat = mgrid
    if not same_static:                  # land.nc and bathymetry.nc each keep their own grid (intensity/geo.py:9-34)
        env.blon, env.blat = blon, blat
    return env


def _load_static(fl, lon, lat):
    """What does not depend on the year: ocean climatologies on the thermo grid, land, bathymetry, basin masks."""
    mld = _climatology(fl['mld'], 'mixed_layer', lon, lat)
    strat = _climatology(fl['strat'], 'strat', lon, lat)

    # ---- static hi-res fields and basin masks
    dl = _Dataset(fl['land'])
    hlon, hlat = np.asarray(dl['lon'], dtype=np.float64), np.asarray(dl['lat'], dtype=np.float64)
    land = np.asarray(dl['land'], dtype=np.float64)
    hlat, land = _ascending_lat(hlat, land)
    db = _Dataset(fl['bathy'])
    bathy = np.asarray(db['bathymetry'], dtype=np.float64)
    blon, blat = np.asarray(db['lon'], dtype=np.float64), np.asarray(db['lat'], dtype=np.float64)
    blat, bathy = _ascending_lat(blat, bathy)
    same_static = bathy.shape == land.shape and np.array_equal(blon, hlon) and np.array_equal(blat, hlat)
    masks = {}
    mgrid = None
    for b in list(BASIN_IDS) + ['GL']:
        fn = '%s/%s.nc' % (fl['basin_dir'], b)
        if not os.path.exists(fn):
            continue
        dm = _Dataset(fn)
        mlat_b, mask_b = _ascending_lat(np.asarray(dm['lat'], dtype=np.float64), np.asarray(dm['basin'], dtype=np.float64))
        g = (np.asarray(dm['lon'], dtype=np.float64), mlat_b)
        if mgrid is None:
            mgrid = g
        elif not (np.array_equal(g[0], mgrid[0]) and np.array_equal(g[1], mgrid[1])):
            raise NotImplementedError('basin masks on different grids')
        masks[b] = mask_b
    return mld, strat, hlon, hlat, land, bathy, blon, blat, same_static, masks, mgrid


class FileEnvironment:
    """All years of an experiment: `for_year(y)` loads a year (the parsed files and the static planes are kept between
    years; the last few years' environments too); attribute access falls through to the first year so that it can be staged
    like a single-year environment.  Safe to call from the worker threads of `compute.run_downscaling`."""

    def __init__(self, nl=None, files=None):
        import threading
        self.nl = nl or _default_namelist
        self.files = files
        self._cache, self._years, self._lock = {}, {}, threading.Lock()

    def for_year(self, year):
        with self._lock:
            env = self._years.get(year)
            if env is None:
                env = load_year_env(year, self.nl, self.files, cache=self._cache)
                self._years[year] = env
                while len(self._years) > 4:
                    del self._years[next(iter(self._years))]
            return env

    def __getattr__(self, k):
        if k.startswith('_') or k in ('nl', 'files'):
            raise AttributeError(k)
        return getattr(self.for_year(self.nl.start_year), k)


# --------------------------------------------------------------------------------------
def write_reference_files(env, out_dir, year, nl=None, calendar='standard', last_year=None):
    """Write a 12-month environment in the reference's file schema (NetCDF-3 classic): the inverse of
    `load_year_env`, and the way to run the *reference* on this project's synthetic fields.
    Monthly records are stamped on the 15th and repeated for every year of [year, last_year]; December
    of year-1 and January of last_year+1 are copies of the neighbouring months so that every mid-month
    date lies inside the record.  Returns the `files` dict for `load_year_env`."""
    from scipy.io import netcdf_file
    nl = nl or _default_namelist
    os.makedirs(out_dir + '/land', exist_ok=True)
    last_year = year if last_year is None else last_year
    stamps = [(year - 1, 12, 15)] + [(y, m, 15) for y in range(year, last_year + 1) for m in range(1, 13)] + [(last_year + 1, 1, 15)]
    pick = [11] + list(range(12)) * (last_year - year + 1) + [0]
    t0 = _linear_seconds(calendar, year - 1, 1, 1)
    days = np.array([(_linear_seconds(calendar, *s) - t0) / 86400.0 for s in stamps])

    def new(fn, dims):
        f = netcdf_file(fn, 'w', version=2)
        for k, n in dims.items():
            f.createDimension(k, n)
        return f

    def put(f, name, arr, dims, **attrs):
        v = f.createVariable(name, 'd', dims)
        v[:] = np.asarray(arr, dtype=np.float64)
        for k, a in attrs.items():
            setattr(v, k, a)

    def time_var(f):
        put(f, 'time', days, ('time',), units='days since %04d-01-01 00:00:00' % (year - 1), calendar=calendar)

    files = {}
    fac = nl.PI_reduc * np.sqrt(nl.Ck / nl.Cd)
    fn = files['thermo'] = '%s/thermo_synth.nc' % out_dir
    with new(fn, dict(time=len(days), lat=len(env.lat), lon=len(env.lon))) as f:
        time_var(f); put(f, 'lat', env.lat, ('lat',)); put(f, 'lon', env.lon, ('lon',))
        put(f, 'vmax', env.vpot[pick] / fac, ('time', 'lat', 'lon'))
        put(f, 'rh_mid', env.rh_mid[pick], ('time', 'lat', 'lon'))
        # inverse of compute.py:113-115 inside its clip range
        raw = np.exp(np.log(np.maximum(env.chi[pick] - nl.chi_fac, 1e-300)) - nl.log_chi_fac) - 1e-3
        put(f, 'chi', raw, ('time', 'lat', 'lon'))
    fn = files['env_wnd'] = '%s/env_wnd_synth.nc' % out_dir
    with new(fn, dict(time=len(days), lat=len(env.wlat), lon=len(env.wlon))) as f:
        time_var(f); put(f, 'lat', env.wlat, ('lat',)); put(f, 'lon', env.wlon, ('lon',))
        for i, n in enumerate(MEAN_NAMES):
            put(f, n, env.wnd_mean[pick, i], ('time', 'lat', 'lon'))
        for k, (i, j) in enumerate(TRIL):
            put(f, cov_name(i, j), env.wnd_cov[pick, k], ('time', 'lat', 'lon'))
    for key, var, arr in (('mld', 'mixed_layer', env.mld), ('strat', 'strat', env.strat)):
        fn = files[key] = '%s/%s_climatology.nc' % (out_dir, key)
        lonw = np.append(env.lon, env.lon[0] + 360.0)                     # the wrap column ocean.py drops
        X = np.concatenate([arr, arr[:, :, :1]], axis=2).transpose(1, 2, 0)   # [lat, lon+1, month]
        with new(fn, dict(lat=len(env.lat), lon=len(lonw), month=12)) as f:
            put(f, 'lat', env.lat, ('lat',)); put(f, 'lon', lonw, ('lon',))
            put(f, 'month', np.arange(1, 13), ('month',))
            put(f, var, X, ('lat', 'lon', 'month'))
    for key, var, arr in (('land', 'land', env.land), ('bathy', 'bathymetry', env.bathy)):
        fn = files[key] = '%s/%s.nc' % (out_dir, 'land' if key == 'land' else 'bathymetry')
        own = key == 'bathy' and getattr(env, 'blon', None) is not None
        glon, glat = (env.blon, env.blat) if own else (env.hlon, env.hlat)
        with new(fn, dict(lat=len(glat), lon=len(glon))) as f:
            put(f, 'lat', glat, ('lat',)); put(f, 'lon', glon, ('lon',))
            put(f, var, arr, ('lat', 'lon'))
    files['basin_dir'] = out_dir + '/land'
    mlon, mlat = getattr(env, 'mlon', None), getattr(env, 'mlat', None)
    mlon, mlat = (env.hlon if mlon is None else mlon), (env.hlat if mlat is None else mlat)
    for b, m in env.basin_masks.items():
        with new('%s/land/%s.nc' % (out_dir, b), dict(lat=len(mlat), lon=len(mlon))) as f:
            put(f, 'lat', mlat, ('lat',)); put(f, 'lon', mlon, ('lon',))
            put(f, 'basin', m, ('lat', 'lon'))
    return files
