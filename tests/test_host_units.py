"""Host logic against reference known-answer vectors (tests/golden/units.npz). CPU only."""
import datetime
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def units():
    return np.load(os.path.join(GOLDEN, 'units.npz'))


def test_transform_global_field_matches_reference(units):
    from tropical_cyclone_risk_amd import basins, namelist
    namelist.basin_bounds['XW'] = ['100W', '5N', '10W', '55N']
    namelist.basin_bounds['XE'] = ['20E', '40S', '120E', '10S']
    try:
        for bid in ('NA', 'SI', 'GL', 'XW', 'XE'):
            lo, la, X = basins.TC_Basin(bid).transform_global_field(units['tgf_in_lon'], units['tgf_in_lat'], units['tgf_in_X'])
            assert np.array_equal(lo, units['tgf_%s_lon' % bid]), bid
            assert np.array_equal(la, units['tgf_%s_lat' % bid]), bid
            assert np.array_equal(X, units['tgf_%s_X' % bid]), bid
        lo, la, X = basins.TC_Basin('XE').transform_global_field(units['tgf_in_lon_pm'], units['tgf_in_lat'], units['tgf_in_X'])
        assert np.array_equal(lo, units['tgf_pm_XE_lon']) and np.array_equal(X, units['tgf_pm_XE_X'])
    finally:
        del namelist.basin_bounds['XW'], namelist.basin_bounds['XE']


def test_basin_errors_and_bounds():
    from tropical_cyclone_risk_amd import basins
    with pytest.raises(ValueError):
        basins.TC_Basin('ZZ')
    b = basins.TC_Basin('SI')
    x0, y0, x1, y1 = b.get_bounds()
    assert (x0, y0, x1) == (20.0, -45.0, 100.0) and y1 == 0 and np.signbit(y1)    # '0S' -> -0.0
    assert b.in_basin(50, -20, 1) and not b.in_basin(20.5, -20, 1) and not b.in_basin(50, -0.5, 1)
    assert basins.BASIN_IDS == ('AU', 'EP', 'NA', 'NI', 'SI', 'SP', 'WP')


def test_minit_and_params(units):
    from tropical_cyclone_risk_amd import namelist
    assert np.allclose(np.maximum(0, namelist.f_mInit(units['minit_rh'])), units['minit_m'], rtol=0, atol=1e-15)


def test_oracle_steering_matches_reference(units):
    from oracle import scipy_port as P
    st = P.Storm.__new__(P.Storm)
    st.prm = P.Params()
    got = np.array([st.steering(v) for v in units['steer_v']])
    assert np.array_equal(got, units['steer_coefs'])


def test_chi_transform():
    from tropical_cyclone_risk_amd import synthetic
    x = np.array([0.0, 0.3, 1.0, 4.0, np.nan])
    y = synthetic.chi_transform(x)
    want = np.clip(np.exp(np.log(np.array([0.0, 0.3, 1.0, 4.0, 5.0]) + 1e-3) + 0.5) + 1.3, 1e-5, 5)
    assert np.allclose(y, want) and y[-1] == 5


def test_namelist_overlay(tmp_path):
    from tropical_cyclone_risk_amd import namelist
    f = tmp_path / 'nl.py'
    f.write_text('tracks_per_year = 123\nexp_name = "x"\n')
    old = namelist.tracks_per_year, namelist.exp_name
    try:
        namelist.load(str(f))
        assert namelist.tracks_per_year == 123 and namelist.exp_name == 'x'
    finally:
        namelist.tracks_per_year, namelist.exp_name = old


def test_track_file_schema_roundtrip(tmp_path):
    """Output file: the reference's variables / dimensions (compute.py:250-264, README "Model
    Output"), duplicate-name suffixing (compute.py:52-58), round trip through the NetCDF writer."""
    import types
    from tropical_cyclone_risk_amd import io as tio, namelist
    from tropical_cyclone_risk_amd.basins import TC_Basin
    nl = types.SimpleNamespace(**{k: getattr(namelist, k) for k in dir(namelist) if not k.startswith('__')})
    nl.output_directory = str(tmp_path); nl.exp_name = 'unit'; nl.start_year, nl.end_year = 2001, 2002
    rng = np.random.default_rng(0)
    ns = 361

    def fake_year(n):
        lon = rng.random((n, ns)); lon[:, 200:] = np.nan
        return (lon, lon + 1, lon + 2, lon + 3, lon + 4, rng.random((n, ns, 4)), rng.integers(1, 13, n).astype(float),
                np.array(['NA', 'EP'] * (n // 2), dtype='U2'), rng.integers(0, 9, (7, 12)).astype(float))
    out = [fake_year(4), fake_year(6)]
    fn = tio.write_tracks(out, [2001, 2002], TC_Basin('NA'), nl)
    assert os.path.basename(fn) == 'tracks_NA_era5_200101_200212.nc'
    fn2 = tio.write_tracks(out, [2001, 2002], TC_Basin('NA'), nl)
    assert fn2.endswith('_e0.nc')
    d = tio.read_tracks(fn)
    for k in ('lon_trks', 'lat_trks', 'u250_trks', 'v250_trks', 'u850_trks', 'v850_trks', 'v_trks', 'm_trks',
              'vmax_trks', 'tc_month', 'tc_basins', 'tc_years', 'seeds_per_month', 'time', 'year', 'basin', 'month'):
        assert k in d, k
    assert d['lon_trks'].shape == (10, ns) and d['seeds_per_month'].shape == (2, 7, 12)
    assert np.array_equal(d['lon_trks'], np.concatenate([out[0][0], out[1][0]]), equal_nan=True)
    assert np.array_equal(d['u850_trks'], np.concatenate([out[0][5][:, :, 2], out[1][5][:, :, 2]]))
    assert list(d['tc_basins'][:2]) == ['NA', 'EP'] and list(d['basin']) == ['AU', 'EP', 'NA', 'NI', 'SI', 'SP', 'WP']
    assert list(d['tc_years']) == [2001] * 4 + [2002] * 6 and d['time'][1] == 3600.0


# ---- input-field loader: the reference's file schema, NetCDF-3 round trip
def _small_env():
    from tropical_cyclone_risk_amd import synthetic
    env = synthetic.make_env('gfdl', seed=7)          # two different grids (wind 2x2.5, thermo 1x1.25)
    # thin the 0.25-degree static grid: the loader does not care and the files stay small
    env.hlon, env.hlat = env.hlon[::4], env.hlat[::4]
    env.land, env.bathy = env.land[::4, ::4], env.bathy[::4, ::4]
    env.basin_masks = {k: v[::4, ::4] for k, v in env.basin_masks.items()}
    return env


@pytest.mark.parametrize('calendar', ['standard', 'noleap'])
def test_field_loader_round_trip(tmp_path, calendar):
    from tropical_cyclone_risk_amd import fields, namelist
    env = _small_env()
    files = fields.write_reference_files(env, str(tmp_path), 2003, namelist, calendar=calendar)
    got = fields.load_year_env(2003, namelist, files)
    for k in ('lon', 'lat', 'wlon', 'wlat', 'hlon', 'hlat', 'land', 'bathy', 'rh_mid', 'wnd_mean', 'wnd_cov', 'mld', 'strat'):
        np.testing.assert_allclose(getattr(got, k), getattr(env, k), rtol=1e-13, atol=1e-13, err_msg=k)
    # vmax is stored without the PI_reduc * sqrt(Ck/Cd) factor, chi before compute.py:113-115's transform
    np.testing.assert_allclose(got.vpot, env.vpot, rtol=1e-13, atol=1e-13)
    inside = (env.chi > 1e-5 + namelist.chi_fac) & (env.chi < 5)
    np.testing.assert_allclose(got.chi[inside], env.chi[inside], rtol=1e-12)
    assert sorted(got.basin_masks) == sorted(env.basin_masks)
    for b in env.basin_masks:
        assert np.array_equal(got.basin_masks[b], env.basin_masks[b])


def test_field_loader_keeps_independent_static_grids(tmp_path):
    """land.nc and bathymetry.nc on different grids (two independent interpolators, intensity/geo.py:9-34) and
    north-to-south latitudes (flipped like mat.interp2_fx does, util/mat.py:147-154)."""
    from scipy.io import netcdf_file
    from tropical_cyclone_risk_amd import fields, namelist
    env = _small_env()
    env.blon, env.blat, env.bathy = env.hlon[::2].copy(), env.hlat[::2].copy(), np.ascontiguousarray(env.bathy[::2, ::2])
    files = fields.write_reference_files(env, str(tmp_path), 2003, namelist)
    got = fields.load_year_env(2003, namelist, files)
    assert np.array_equal(got.blon, env.blon) and np.array_equal(got.blat, env.blat)
    assert np.array_equal(got.bathy, env.bathy) and np.array_equal(got.land, env.land) and got.bathy.shape != got.land.shape
    # a descending-latitude basin mask and land file come back ascending
    fn = files['basin_dir'] + '/NA.nc'
    with netcdf_file(fn, 'r', mmap=False) as f:
        lat, lon, m = f.variables['lat'][:].copy(), f.variables['lon'][:].copy(), f.variables['basin'][:].copy()
    with netcdf_file(fn, 'w', version=2) as f:
        f.createDimension('lat', len(lat)); f.createDimension('lon', len(lon))
        v = f.createVariable('lat', 'd', ('lat',)); v[:] = lat[::-1]
        v = f.createVariable('lon', 'd', ('lon',)); v[:] = lon
        v = f.createVariable('basin', 'd', ('lat', 'lon')); v[:] = m[::-1]
    for b in got.basin_masks:
        if b != 'NA':
            os.remove(files['basin_dir'] + '/%s.nc' % b)
    again = fields.load_year_env(2003, namelist, files)
    assert np.array_equal(again.basin_masks['NA'], got.basin_masks['NA'])
    f2 = fields._interp2_fx(lon, lat[::-1], m[::-1])
    f1 = fields._interp2_fx(lon, lat, m)
    assert f2.ev(lon[5] + 0.1, lat[7] + 0.05) == f1.ev(lon[5] + 0.1, lat[7] + 0.05)


def test_field_loader_time_semantics(tmp_path):
    """Monthly records stamped on the 1st: the 15th of each month is a blend of two records and
    December falls off the end of the year's slice -> vpot 0, chi 5 -> transform (compute.py:70-71, 108-113)."""
    from scipy.io import netcdf_file
    from tropical_cyclone_risk_amd import fields, namelist
    env = _small_env()
    files = fields.write_reference_files(env, str(tmp_path), 2003, namelist)
    fn = str(tmp_path / 'thermo_first.nc')
    lat, lon = env.lat[::-1], env.lon                       # descending latitude, as ERA5 delivers it
    days = np.array([(datetime.date(2003, m, 1) - datetime.date(2002, 1, 1)).days for m in range(1, 13)], dtype=float)
    vm = np.arange(12, dtype=float)[:, None, None] * 10 + np.zeros((12, len(lat), len(lon)))
    with netcdf_file(fn, 'w', version=2) as f:
        f.createDimension('time', 12); f.createDimension('lat', len(lat)); f.createDimension('lon', len(lon))
        v = f.createVariable('time', 'd', ('time',)); v[:] = days; v.units = 'days since 2002-01-01'
        for name, arr in (('lat', lat), ('lon', lon)):
            v = f.createVariable(name, 'd', (name,)); v[:] = arr
        for name in ('vmax', 'rh_mid', 'chi'):
            v = f.createVariable(name, 'd', ('time', 'lat', 'lon')); v[:] = vm if name == 'vmax' else vm * 0 + 0.7
    files['thermo'] = fn
    got = fields.load_year_env(2003, namelist, files)
    assert np.array_equal(got.lat, env.lat)                 # flipped to ascending
    fac = namelist.PI_reduc * np.sqrt(namelist.Ck / namelist.Cd)
    assert abs(got.vpot[0, 3, 3] - fac * 10 * 14 / 31) < 1e-12          # Jan 15 between Jan 1 (0) and Feb 1 (10)
    assert abs(got.vpot[1, 3, 3] - fac * (10 + 10 * 14 / 28)) < 1e-12
    assert np.all(got.vpot[11] == 0.0)                                   # Dec 15 is past the last record
    assert np.allclose(got.chi[11], min(max(np.exp(np.log(5 + 1e-3) + namelist.log_chi_fac) + namelist.chi_fac, 1e-5), 5))
    assert np.isnan(got.rh_mid[11]).all()


def test_loader_time_interp_known_answers(tmp_path):
    """`DataArray.interp(time=the 15th)` (compute.py:108-113) is xarray.core.missing -> scipy.interpolate.interp1d(
    kind='linear', bounds_error=False, fill_value=nan) along time.  SciPy is installed here, so fields.interp_time is
    pinned against that very call — at record edges, exactly on records, and outside — plus hand-computed
    calendar cases for standard and noleap axes and the inclusive year slice (compute.py:68-71)."""
    from scipy.interpolate import interp1d
    from scipy.io import netcdf_file
    from tropical_cyclone_risk_amd import fields, namelist
    rng = np.random.default_rng(3)
    t = np.sort(rng.uniform(0, 400, 9)) * 86400.0
    y = rng.normal(size=(9, 4, 5))
    f = interp1d(t, y, kind='linear', axis=0, bounds_error=False, fill_value=np.nan)
    for tq in list(t) + [t[0] - 1.0, t[-1] + 1.0, 0.5 * (t[3] + t[4]), t[0] + 1e-3, t[-1] - 1e-3, np.nextafter(t[2], np.inf)]:
        assert np.array_equal(fields.interp_time(t, y, tq), f(tq), equal_nan=True), tq
    assert np.isnan(fields.interp_time(t, y, t[0] - 1.0)).all() and not np.isnan(fields.interp_time(t, y, t[-1])).any()
    # calendars: Feb 15 of a leap year between records on Feb 1 and Mar 1
    for cal, w in (('standard', 14.0 / 29.0), ('noleap', 14.0 / 28.0)):
        ax = fields.TimeAxis(np.array([31.0, 59.0 if cal == 'noleap' else 60.0]), dict(units='days since 2004-01-01', calendar=cal))
        assert ax.at(2004, 2, 1) == ax.t[0] and ax.at(2004, 3, 1) == ax.t[1]
        v = fields.interp_time(ax.t, np.array([[10.0], [30.0]]), ax.at(2004, 2, 15))
        assert abs(v[0] - (10.0 + 20.0 * w)) < 1e-12, cal
    assert fields._linear_seconds('noleap', 2004, 3, 1) - fields._linear_seconds('noleap', 2004, 2, 28) == 86400.0
    assert fields._linear_seconds('standard', 2004, 3, 1) - fields._linear_seconds('standard', 2004, 2, 28) == 2 * 86400.0
    assert fields._linear_seconds('noleap', 2005, 1, 1) - fields._linear_seconds('noleap', 2004, 1, 1) == 365 * 86400.0
    # the year slice [Dec 31 of year-1, Dec 31 of year] is inclusive at both ends (ds.sel(time=slice(a, b))): records stamped
    # exactly on those two days take part, records one day outside do not — visible in January / December of the result
    env = _small_env()
    files = fields.write_reference_files(env, str(tmp_path), 2003, namelist)
    lat, lon = env.lat, env.lon
    stamps = [datetime.date(2002, 12, 30), datetime.date(2002, 12, 31), datetime.date(2003, 7, 1), datetime.date(2003, 12, 31), datetime.date(2004, 1, 1)]
    days = np.array([(d - datetime.date(2002, 1, 1)).days for d in stamps], dtype=float)
    vals = np.array([1000.0, 10.0, 20.0, 40.0, 1000.0])
    vm = vals[:, None, None] + np.zeros((5, len(lat), len(lon)))
    fn = str(tmp_path / 'thermo_edges.nc')
    with netcdf_file(fn, 'w', version=2) as fo:
        fo.createDimension('time', 5); fo.createDimension('lat', len(lat)); fo.createDimension('lon', len(lon))
        v = fo.createVariable('time', 'd', ('time',)); v[:] = days; v.units = 'days since 2002-01-01'
        for name, arr in (('lat', lat), ('lon', lon)):
            v = fo.createVariable(name, 'd', (name,)); v[:] = arr
        for name in ('vmax', 'rh_mid', 'chi'):
            v = fo.createVariable(name, 'd', ('time', 'lat', 'lon')); v[:] = vm
    files['thermo'] = fn
    got = fields.load_year_env(2003, namelist, files)
    fac = namelist.PI_reduc * np.sqrt(namelist.Ck / namelist.Cd)
    d0, d1, d2 = (datetime.date(2002, 12, 31), datetime.date(2003, 7, 1), datetime.date(2003, 12, 31))
    jan = 10.0 + 10.0 * (datetime.date(2003, 1, 15) - d0).days / (d1 - d0).days          # the Dec-30 record (1000) is outside the slice
    dec = 20.0 + 20.0 * (datetime.date(2003, 12, 15) - d1).days / (d2 - d1).days         # ... and so is Jan 1 of the next year
    assert abs(got.vpot[0, 2, 2] - fac * jan) < 1e-12 and abs(got.vpot[11, 2, 2] - fac * dec) < 1e-12
    assert abs(got.rh_mid[6, 2, 2] - (20.0 + 20.0 * 14 / (d2 - d1).days)) < 1e-12      # Jul 15


def test_hdf5lite_reads_the_reference_land_mask():
    """the minimal HDF5 reader on the reference's own NetCDF-4 data file
    (intensity/data/land.nc, kept as a fixture: chunked, deflate + shuffle, v2 object headers), checked
    against geography rather than against another reader (none is installed)."""
    from tropical_cyclone_risk_amd import fields, hdf5lite
    fn = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_land.nc')
    f = hdf5lite.File(fn)
    assert f.keys() == ['land', 'lat', 'lon']
    land, lat, lon = f['land'], f['lat'], f['lon']
    assert land.shape == (1440, 2880) and land.dtype == np.int8 and set(np.unique(land)) == {0, 1}
    assert lat.dtype == np.float32 and np.all(np.diff(lat) > 0) and lat[0] == -89.875 and lat[-1] == 90.0
    assert np.all(np.diff(lon) > 0) and lon[0] == 0.0 and lon[-1] == 359.875
    assert f.attrs('land')['units'] == 'boolean' and f.attrs('lat')['units'] == 'degrees'
    at = lambda la, lo: int(land[np.argmin(np.abs(lat - la)), np.argmin(np.abs(lon - lo % 360))])
    assert [at(23, 10), at(-5, -60), at(-25, 135), at(-85, 0), at(38, -98)] == [1, 1, 1, 1, 1]      # Sahara, Amazon, Australia, Antarctica, Kansas
    assert [at(0, 180), at(30, -50), at(-20, 80), at(-50, -140)] == [0, 0, 0, 0]                    # open ocean
    assert 0.30 < land.mean() < 0.37
    # and through the loader's dataset facade (what fields.load_year_env uses when xarray is absent)
    ds = fields._Dataset(fn)
    assert ds['land'].shape == (1440, 2880) and float(ds['lat'][0]) == -89.875


REF_DATA = '/root/reference/intensity/data'


@pytest.mark.skipif(not os.path.exists(REF_DATA + '/mld_climatology.nc'),
                    reason='needs the reference data files (build container only; they never travel to the GPU box)')
def test_ocean_climatologies_through_hdf5lite_match_the_reference_code():
    """intensity/ocean.py:11-64 + util/compute.py:117-118 on the reference's own mld / strat climatology files
    (NetCDF-4: HDF5 with shuffle + deflate), read through hdf5lite and regridded by fields._climatology, against
    tests/golden/clim_ref.npz — what the reference's `mld_climatology`, `strat_climatology` and `mat.interp_2d_grid` lines
    returned for months 1 and 7 (tests/golden/make_golden_clim.py)."""
    from tropical_cyclone_risk_amd import fields, hdf5lite
    g = np.load(os.path.join(GOLDEN, 'clim_ref.npz'))
    for name, var, fn in (('mld', 'mixed_layer', 'mld_climatology.nc'), ('strat', 'strat', 'strat_climatology.nc')):
        f = hdf5lite.File(os.path.join(REF_DATA, fn))
        X = np.asarray(f[var])
        lon, lat = np.asarray(f['lon']), np.asarray(f['lat'])
        assert X.shape == (180, 361, 12) and X.dtype == np.float32 and lon[0] == 0.0 and lon[-1] == 360.0
        # the decoding, by the file's own redundancy: its last longitude column (360 E) repeats the first (0 E)
        assert np.array_equal(X[:, 360, :], X[:, 0, :], equal_nan=True)
        assert abs(np.isnan(X).mean() - 0.2924) < 1e-3 and (name != 'mld' or np.nanmin(X) >= 0)      # land is NaN; depths are not negative
        got = fields._climatology(os.path.join(REF_DATA, fn), var, g['lon'], g['lat'])
        assert got.shape == (12, 73, 144) and not np.isnan(got).any()                    # NaN -> 0 before regridding (compute.py:117)
        for mo in (1, 7):
            assert np.array_equal(got[mo - 1], g['%s_%d' % (name, mo)]), (name, mo)      # bit for bit
        # the wrap column: the source axis the reference regrids from stops at 359 E (ocean.py:26 drops lon[-1]), so the
        # target column 357.5 E interpolates between 357 and 358, and 0 E is the file's first column itself
        assert np.array_equal(g[name + '_src_lon'], np.arange(360.0))
        j = 36                                                                            # the equator row of the 2.5-degree target
        src_eq = np.nan_to_num(X[:, :, 0].astype(np.float64))
        assert got[0][j, 0] == pytest.approx(0.5 * (src_eq[89, 0] + src_eq[90, 0]), rel=1e-12)
        assert (got[0] == 0).mean() > 0.2                                                 # land (and ice) came through as 0, not NaN


def test_hdf5lite_rejects_garbage(tmp_path):
    from tropical_cyclone_risk_amd import fields, hdf5lite
    fn = tmp_path / 'x.nc'
    fn.write_bytes(b'not a netcdf file at all' * 10)
    with pytest.raises(RuntimeError, match='neither NetCDF-3 nor HDF5'):
        fields._Dataset(str(fn))
    with pytest.raises(hdf5lite.Unsupported):
        hdf5lite.File(str(fn))


# ---- monthly wind statistics: oracle and host logic
def test_wind_stats_oracle_hand_case():
    from oracle import wind_stats as ws
    # two grid points, three days; component c at point p on day d = table below
    x0 = np.array([[1.0, 10.0], [2.0, 10.0], [6.0, 13.0]])
    x1 = np.array([[0.0, 1.0], [4.0, 1.0], [2.0, 4.0]])
    z = np.zeros_like(x0)
    out = ws.wind_stats([x0, x1, z, z + 5.0])
    assert out.shape == (14, 2)
    assert np.allclose(out[0], [3.0, 11.0]) and np.allclose(out[1], [2.0, 2.0]) and np.allclose(out[3], [5.0, 5.0])
    assert np.allclose(out[4], [(4 + 1 + 9) / 3.0, (1 + 1 + 4) / 3.0])                 # var(x0), ddof 0
    assert np.allclose(out[5], [((-2) * (-2) + (-1) * 2 + 3 * 0) / 2.0, ((-1) * (-1) + (-1) * (-1) + 2 * 2) / 2.0])   # cov, ddof 1
    assert np.allclose(out[6], [(4 + 4 + 0) / 3.0, (1 + 1 + 4) / 3.0])                 # var(x1)
    assert np.all(out[7:] == 0.0)
    # per-day grouping: 6 samples, days of 1, 2 and 3 samples
    s = np.array([[3.0], [1.0], [3.0], [2.0], [4.0], [12.0]])
    g = ws.wind_stats([s, s, s, s], day_start=np.array([0, 1, 3, 6]))
    assert np.allclose(g[0], [(3 + 2 + 6) / 3.0]) and np.allclose(g[4], [((3 - 11 / 3) ** 2 + (2 - 11 / 3) ** 2 + (6 - 11 / 3) ** 2) / 3])


def test_wind_stats_float32_path_is_what_numpy_does():
    """The documented float32 path of mean / var / xr.cov (oracle/wind_stats.py: np.nanmean / np.nanvar / the cov
    steps) against explicit scalar loops: float32 sums in time order, divisions through fp64 rounded back to
    float32, per-pair NaN masks and means for the covariances, cov = float32 sum / (n - 1) in fp64 — which is also
    what k_wind_stats<float> does, operation for operation."""
    from oracle import wind_stats as ws
    f32 = np.float32
    # accumulation order: 16777216 + 1 + 1 stays 16777216 in float32 when added in time order
    big = np.array([[16777216.0], [1.0], [1.0]], dtype=f32)
    one = np.ones((3, 1), dtype=f32)
    out = ws.wind_stats([big, one, one, one])
    assert out[0, 0] == float(f32(16777216.0 / 3.0)) and out.dtype == np.float64
    rng = np.random.default_rng(5)
    T, P = 29, 6
    x = [(rng.normal(10 * c, 4, size=(T, P)) * (1 + 100 * (c == 2))).astype(f32) for c in range(4)]
    x[0][3, 1] = np.nan; x[2][3, 1] = np.nan; x[1][7, 1] = np.nan; x[3][:, 4] = np.nan     # scattered and a whole column
    got = ws.wind_stats(x)
    want = np.zeros((14, P))
    for p in range(P):
        def nanmean(v, ok):
            s, n = f32(0), 0
            for t in range(T):
                if ok[t]:
                    s = f32(s + v[t]); n += 1
            return (f32(np.float64(s) / n) if n else f32(np.nan)), n
        col = [a[:, p] for a in x]
        fin = [~np.isnan(c) for c in col]
        k = 4
        for i in range(4):
            m, n = nanmean(col[i], fin[i])
            want[i, p] = m
        for i in range(4):
            for j in range(i + 1):
                if i == j:
                    m, n = nanmean(col[i], fin[i])
                    s = f32(0)
                    for t in range(T):
                        if fin[i][t]:
                            d = f32(col[i][t] - m); s = f32(s + f32(d * d))
                    want[k, p] = f32(np.float64(s) / n) if n else np.nan
                else:
                    both = fin[i] & fin[j]
                    mi, n = nanmean(col[i], both); mj, _ = nanmean(col[j], both)
                    s = f32(0)
                    for t in range(T):
                        if both[t]:
                            s = f32(s + f32(f32(col[i][t] - mi) * f32(col[j][t] - mj)))
                    want[k, p] = np.float64(s) / (n - 1) if n >= 1 else np.nan
                k += 1
    with np.errstate(all='ignore'):
        assert np.array_equal(got, want, equal_nan=True)
    assert np.isnan(got[3, 4]) and np.isnan(got[13, 4]) and np.isnan(got[10, 4]) and not np.isnan(got[0, 4])
    # float64 planes take the same path in float64 (round-1 formula when there is no NaN)
    y = [a.astype(np.float64) for a in x]
    y[3][:, 4] = 1.0
    for a in y:
        a[np.isnan(a)] = 0.5
    o64 = ws.wind_stats(y)
    assert np.array_equal(o64[5], ((y[1] - y[1].mean(0)) * (y[0] - y[0].mean(0))).sum(0) / (T - 1))
    assert np.array_equal(o64[4], ((y[0] - y[0].mean(0)) ** 2).mean(0))


def test_multi_file_monthly_inputs(tmp_path):
    """preprocess.MultiFile = xr.open_mfdataset(fns, concat_dim="time", combine="nested") (util/input.py:14-21):
    records of several files concatenated in list order, each time axis decoded with its own units."""
    from scipy.io import netcdf_file
    from tropical_cyclone_risk_amd import preprocess as pp
    rng = np.random.default_rng(1)
    recs = rng.normal(size=(5, 3, 4))
    spec = [(slice(0, 2), 'days since 2001-01-01', [14.0, 45.0]), (slice(2, 5), 'hours since 2001-01-01', [73 * 24.0, 104 * 24.0, 134 * 24.0])]
    fns = []
    for j, (sl, units, tv) in enumerate(spec):
        fn = str(tmp_path / ('v_%d.nc' % j)); fns.append(fn)
        with netcdf_file(fn, 'w', version=2) as f:
            f.createDimension('time', len(tv)); f.createDimension('lat', 3); f.createDimension('lon', 4)
            v = f.createVariable('time', 'd', ('time',)); v[:] = tv; v.units = units; v.calendar = 'standard'
            v = f.createVariable('lat', 'd', ('lat',)); v[:] = [0, 1, 2]
            v = f.createVariable('lon', 'd', ('lon',)); v[:] = [0, 1, 2, 3]
            v = f.createVariable('x', 'd', ('time', 'lat', 'lon')); v[:] = recs[sl]
            # a level axis as long as the FIRST file's record count: a coordinate, not a record
            f.createDimension('lev', 2)
            v = f.createVariable('lev', 'd', ('lev',)); v[:] = [85000.0, 25000.0]
    m = pp.MultiFile(fns)
    assert np.array_equal(m['lev'], [85000.0, 25000.0])
    assert [(t.month, t.day) for t in m.times] == [(1, 15), (2, 15), (3, 15), (4, 15), (5, 15)]
    assert np.array_equal(m['x'], recs) and np.array_equal(m['lat'], [0, 1, 2])
    for i in range(5):
        assert np.array_equal(m.record('x', i), recs[i])
    assert np.array_equal(pp.MultiFile(fns[0])['x'], recs[:2])
    with pytest.raises(ValueError, match='time order'):
        pp.MultiFile(fns[::-1])


def test_wind_stats_host_logic():
    from tropical_cyclone_risk_amd import preprocess as pp
    times = [datetime.datetime(2001, 1, 30) + datetime.timedelta(hours=12 * k) for k in range(8)]
    m = pp.month_mask(times, 2001, 1)
    assert list(m) == [True] * 4 + [False] * 4                                      # Jan 30 00 .. Jan 31 12
    assert list(pp.month_mask(times, 2001, 2)) == [False] * 4 + [True] * 4
    assert list(pp.day_groups([t for t, k in zip(times, m) if k])) == [0, 2, 4]
    assert pp.pick_levels([1000, 850, 500, 250], 'hPa') == (3, 1)
    assert pp.pick_levels([25000, 85000], 'Pa') == (0, 1)
    with pytest.raises(KeyError):
        pp.pick_levels([1000, 500], 'hPa')
    dec = [datetime.datetime(2001, 12, 31, 12), datetime.datetime(2002, 1, 1)]
    assert list(pp.month_mask(dec, 2001, 12)) == [True, False]


def test_track_file_writer_streams_the_same_file(tmp_path):
    """io.TrackFileWriter (years handed over as they complete, in any order, rows written at their final place by a
    background thread) leaves byte for byte the file io.write_tracks writes from the assembled years
    (util/compute.py:233-264's Dataset in NetCDF-3), and the transform cache of TC_Basin returns what the uncached
    selection returned."""
    import types
    from tropical_cyclone_risk_amd import io as tio, namelist
    from tropical_cyclone_risk_amd.basins import BASIN_IDS, TC_Basin
    if tio._try_xarray() is not None:
        pytest.skip('xarray present: the file is NetCDF-4, written at close')
    nl = types.SimpleNamespace(**{k: getattr(namelist, k) for k in dir(namelist) if not k.startswith('__')})
    nl.start_year, nl.end_year, nl.tracks_per_year = 2000, 2004, 9
    rng = np.random.default_rng(7)
    years, ns, n = list(range(2000, 2005)), 361, 9

    def year_tuple():
        def plane():
            a = rng.standard_normal((n, ns))
            for r in range(n):
                a[r, rng.integers(2, ns):] = np.nan
            return a
        env = rng.standard_normal((n, ns, 4))
        return (plane(), plane(), plane(), plane(), plane(), env, rng.integers(1, 13, n).astype(float),
                np.array([BASIN_IDS[i] for i in rng.integers(0, 7, n)], dtype='U2'), rng.integers(0, 50, (7, 12)).astype(float))
    out = [year_tuple() for _ in years]
    b = TC_Basin('GL', nl)
    nl.output_directory, nl.exp_name = str(tmp_path), 'whole'
    fn1 = tio.write_tracks(out, years, b, nl)
    nl.exp_name = 'streamed'
    w = tio.TrackFileWriter(years, b, nl)
    for i in (3, 0, 4, 2, 1):
        w.put(i, out[i])
    fn2 = w.close()
    assert open(fn1, 'rb').read() == open(fn2, 'rb').read()
    got = tio.read_tracks(fn2)
    assert np.array_equal(got['v_trks'][n:2 * n], out[1][2], equal_nan=True) and list(got['tc_basins'][:n]) == list(out[0][7])
    # a year with the wrong number of tracks is an error at close, not a corrupt file
    w = tio.TrackFileWriter(years, b, nl)
    bad = tuple(x[:5] if i < 8 else x for i, x in enumerate(out[0]))
    w.put(0, bad)
    for i in range(1, 5):
        w.put(i, out[i])
    with pytest.raises(ValueError):
        w.close()
    # the streaming layout leans on scipy internals; they are checked once on a scratch file, and when the check
    # fails the writer falls back to write_tracks at close — the same file
    assert tio._streaming_supported() and w.streaming
    try:
        tio._STREAMING_OK = False
        nl.exp_name = 'fallback'
        w = tio.TrackFileWriter(years, b, nl)
        assert not w.streaming
        for i in range(5):
            w.put(i, out[i])
        assert open(w.close(), 'rb').read() == open(fn1, 'rb').read()
    finally:
        tio._STREAMING_OK = None


def test_rows_to_tuple_never_aliases_the_round_buffer():
    """with ONE survivor row the env-wind slice of the packed row is already C-contiguous, so an
    `ascontiguousarray` would return a view of the round's reused pinned buffer and year k's tc_env_wnds would be
    overwritten by year k + 1's round.  Every array of the tuple must own its memory, for 1 row and for several."""
    from tropical_cyclone_risk_amd import compute
    ns = 7
    for n in (1, 3):
        buf = np.arange(n * 9 * ns, dtype=np.float64).reshape(n, 9 * ns)          # stands for GpuRound's host_rows view
        res = dict(rows=buf, month=np.ones(n, np.int32), basin_idx=np.zeros(n, np.int64), n_seeds=np.zeros((7, 12)))
        tup = compute.rows_to_tuple(res, ns)
        want_env = buf[:, 5 * ns:].copy().reshape(n, ns, 4)
        for a in tup[:6]:
            assert not np.shares_memory(a, buf)
        buf[:] = -1.0                                                             # the next year's round reuses the buffer
        assert np.array_equal(tup[5], want_env)
        assert np.array_equal(tup[0], np.arange(n * 9 * ns, dtype=np.float64).reshape(n, 9 * ns)[:, :ns])


def test_design_cites_current_round_or_names_the_round():
    """DESIGN.md states the current state — a sentence that cites a profile of rounds 1-4 must say which round it is
    from, and every profile file it names exists."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'DESIGN.md')).read()
    for ln, line in enumerate(text.splitlines(), 1):
        if re.search(r'r0[1-4]_', line):
            assert re.search(r'[Rr]ound', line), 'DESIGN.md:%d cites an old profile without naming its round' % ln
    for name in set(re.findall(r'profiles/(r0\d_[A-Za-z0-9_.]+?\.(?:json|txt|csv))', text)):
        assert os.path.exists(os.path.join(root, 'profiles', name)), name
    assert os.path.exists(os.path.join(root, 'DESIGN_LOG.md'))
