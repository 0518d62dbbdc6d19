"""PI / chi / RH preprocessing: oracle vs the reference's golden vectors (CPU),
HIP kernels vs oracle and golden (GPU)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def cases():
    return np.load(os.path.join(GOLDEN, 'thermo_cases.npz'))


@pytest.fixture(scope='module')
def table():
    return np.load(os.path.join(GOLDEN, 'entropy_table.npz'))


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_oracle_matches_reference(cases, table, tag):
    from oracle import thermo_oracle as to
    tb = to.Table(table['p'], table['s'], table['T'])
    p, sst, psl, T, r = (cases[tag + '_' + k] for k in ('p', 'sst', 'psl', 'T', 'r'))
    k_mid = int(cases[tag + '_k_mid'])
    pi, chi, rh = to.column_fields(tb, float(cases['Ck_over_Cd']), p, sst, psl, T, r, k_mid)
    # same formulas, same libm: the scalar restatement reproduces the vectorised reference to rounding
    np.testing.assert_allclose(pi, cases[tag + '_PI'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(chi, cases[tag + '_chi'], rtol=1e-12, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(rh, cases[tag + '_rh_mid'], rtol=1e-13, atol=0, equal_nan=True)
    for idx in [(0, 0), (1, 1), (2, 2), (5, 5), (7, 3)]:
        aux = to.potential_intensity(tb, 1.0, float(sst[idx]), float(psl[idx]), p, T[(slice(None),) + idx], r[(slice(None),) + idx])[1]
        np.testing.assert_allclose(aux['p_lcl'], cases[tag + '_pLCL'][idx], rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(aux['s_ns'], cases[tag + '_s_ns'][idx], rtol=1e-13, equal_nan=True)
        np.testing.assert_allclose(aux['ss'], cases[tag + '_ss'][idx], rtol=1e-13, equal_nan=True)


def _gpu_fields(eng, table, cases, tag):
    from tropical_cyclone_risk_amd import preprocess as pp
    pp.stage_entropy_table(eng, table['p'], table['s'], table['T'])
    p, sst, psl, T, r = (cases[tag + '_' + k] for k in ('p', 'sst', 'psl', 'T', 'r'))
    k_mid = int(cases[tag + '_k_mid'])
    pi = pp.potential_intensity(eng, sst, psl, p, T, r)
    chi, rh = pp.chi_rh(eng, sst, psl, T[k_mid], r[k_mid], float(p[k_mid]))
    return pi, chi, rh


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_kernels_match_reference_golden(cases, table, built_lib, tag):
    """k_potential_intensity / k_chi_rh against the reference's own outputs.  Tolerance: 1e-9 relative
    (device libm vs glibc in exp/log/pow, own Lambert W); zeros and NaN-handling cases must agree exactly."""
    from tropical_cyclone_risk_amd.engine import TCEngine
    eng = TCEngine('GL', device=0)
    pi, chi, rh = _gpu_fields(eng, table, cases, tag)
    eng.close()
    ref = cases[tag + '_PI']
    assert np.array_equal(pi == 0, ref == 0)
    err = np.abs(pi - ref) / np.maximum(ref, 1.0)
    print('PI: max rel err %.2e, p99 %.2e' % (err.max(), np.percentile(err, 99)))
    assert err.max() < 1e-9
    np.testing.assert_allclose(chi, cases[tag + '_chi'], rtol=1e-9, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(rh, cases[tag + '_rh_mid'], rtol=1e-12, atol=0, equal_nan=True)


@pytest.mark.gpu
def test_compute_thermo_host_mirror(cases, table, built_lib):
    """calc_thermo.compute_thermo's array handling: level order, hPa, mid-level pick, chi clip."""
    from tropical_cyclone_risk_amd import preprocess as pp
    from tropical_cyclone_risk_amd.engine import TCEngine
    eng = TCEngine('GL', device=0)
    pi, chi, rh = _gpu_fields(eng, table, cases, 'b')
    p, sst, psl, T, r = (cases['b_' + k] for k in ('p', 'sst', 'psl', 'T', 'r'))
    v2, c2, r2 = pp.compute_thermo(eng, sst, psl, (p / 100)[::-1], 'hPa', T[::-1], r[::-1])     # top-down, hPa
    # (p / 100) * 100 is not p bit for bit, so agreement is to rounding, not exact
    np.testing.assert_allclose(v2, pi, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(r2, rh, rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(c2, np.minimum(np.maximum(chi, 0), 10), rtol=1e-9, atol=1e-12, equal_nan=True)
    assert np.nanmin(c2) >= 0 and np.nanmax(c2) <= 10
    with pytest.raises(Exception, match='lowest'):
        dp = lambda a: np.ascontiguousarray(a).ctypes.data_as(pp._lib.DP)
        out = np.empty(sst.shape)
        eng._ck(eng.L.tcr_potential_intensity_host(eng.h, sst.size, len(p), dp(p[::-1].copy()), dp(sst), dp(psl), dp(T), dp(r), 1.0, dp(out)))
    eng2 = TCEngine('GL', device=0)
    with pytest.raises(Exception, match='entropy table'):
        pp.potential_intensity(eng2, sst, psl, p, T, r)
    eng.close(); eng2.close()


def _nc3(fn, dims, variables):
    """variables: name -> (dims tuple, array, attrs)"""
    from scipy.io import netcdf_file
    with netcdf_file(fn, 'w', version=2) as f:
        for k, n in dims.items():
            f.createDimension(k, n)
        for name, (d, arr, attrs) in variables.items():
            a = np.asarray(arr)
            v = f.createVariable(name, 'f' if a.dtype == np.float32 else 'd', d)
            v[:] = a
            for k, x in attrs.items():
                setattr(v, k, x)


@pytest.mark.gpu
def test_file_drivers_wind_and_thermo(cases, table, built_lib, tmp_path):
    """gen_wind_mean_cov / gen_thermo over NetCDF files in the ERA5 layout of namelist.var_keys: daily u, v
    on (time, level, latitude, longitude) -> env_wnd file; monthly sst / sp / t / q -> thermo file; both are
    then read back through the field loader's dataset facade."""
    import datetime, types
    from oracle import wind_stats as ws
    from tropical_cyclone_risk_amd import fields, namelist, preprocess as pp
    from tropical_cyclone_risk_amd.engine import TCEngine
    nl = types.SimpleNamespace(**{k: getattr(namelist, k) for k in dir(namelist) if not k.startswith('__')})
    nl.dataset_type = 'ERA5'; nl.start_year, nl.start_month, nl.end_year, nl.end_month = 2001, 1, 2001, 3
    eng = TCEngine('GL', device=0, nl=nl)
    rng = np.random.default_rng(5)
    # ---- daily winds, Jan 1 .. Mar 31 2001, three levels in hPa
    nt, lat, lon = 90, np.linspace(-30, 30, 13), np.arange(0, 60, 5.0)
    days = np.arange(nt, dtype=float)
    lev = np.array([850.0, 500.0, 250.0])
    u = rng.normal(size=(nt, 3, len(lat), len(lon))).astype(np.float32)
    v = rng.normal(size=(nt, 3, len(lat), len(lon))).astype(np.float32)
    common = {'time': (('time',), days, dict(units='days since 2001-01-01 00:00:00', calendar='standard')),
              'level': (('level',), lev, dict(units='hPa')), 'latitude': (('latitude',), lat, {}), 'longitude': (('longitude',), lon, {})}
    dims = dict(time=nt, level=3, latitude=len(lat), longitude=len(lon))
    for name, arr in (('u', u), ('v', v)):
        _nc3(str(tmp_path / ('era5_%s_daily.nc' % name)), dims, dict(common, **{name: (('time', 'level', 'latitude', 'longitude'), arr, {})}))
    out = pp.gen_wind_mean_cov(eng, [str(tmp_path / 'era5_u_daily.nc')], [str(tmp_path / 'era5_v_daily.nc')], str(tmp_path / 'env_wnd.nc'), nl)
    ds = fields._Dataset(out)
    t = [datetime.datetime(2001, 1, 1) + datetime.timedelta(days=float(x)) for x in ds['time']]
    assert [(x.month, x.day) for x in t] == [(1, 1), (2, 15), (3, 15)]          # env_wind.py:139-152 stamps
    for k, (m0, m1) in enumerate([(0, 31), (31, 59), (59, 90)]):
        ref = ws.wind_stats([u[m0:m1, 2], v[m0:m1, 2], u[m0:m1, 0], v[m0:m1, 0]])
        assert np.array_equal(ds['ua250_Mean'][k], ref[0]) and np.array_equal(ds['va850_Mean'][k], ref[3])
        assert np.array_equal(ds['ua250_Var'][k], ref[4]) and np.array_equal(ds['va250_ua250_cov'][k], ref[5])
        assert np.array_equal(ds['va850_Var'][k], ref[13]) and np.array_equal(ds['va850_ua850_cov'][k], ref[12])
    # ---- monthly thermo: the golden soundings of case b as two monthly records; sst in Celsius on its own grid
    p, sst, psl, T, r = (cases['b_' + k] for k in ('p', 'sst', 'psl', 'T', 'r'))
    nla, nlo = sst.shape
    lat, lon = np.linspace(-40, 40, nla), np.linspace(100, 100 + 2.0 * (nlo - 1), nlo)
    tm = np.array([14.0, 45.0])
    tv = ('time', (('time',), tm, dict(units='days since 2001-01-01', calendar='standard')))
    ax = dict([tv, ('latitude', (('latitude',), lat, {})), ('longitude', (('longitude',), lon, {}))])
    d2 = dict(time=2, latitude=nla, longitude=nlo)
    _nc3(str(tmp_path / 'sst.nc'), d2, dict(ax, sst=(('time', 'latitude', 'longitude'), np.stack([sst, sst]) - 273.15, dict(units='degC'))))
    _nc3(str(tmp_path / 'sp.nc'), d2, dict(ax, sp=(('time', 'latitude', 'longitude'), np.stack([psl, psl]), dict(units='Pa'))))
    d3 = dict(d2, level=len(p))
    axl = dict(ax, level=(('level',), p[::-1] / 100.0, dict(units='hPa')))              # top-down in hPa, as ERA5 delivers
    # temperature comes as two files of one record each (open_mfdataset over a sorted glob, util/input.py:14-58)
    d31 = dict(d3, time=1)
    for j, day in enumerate(tm):
        ax1 = dict(axl, time=(('time',), np.array([day]), dict(units='days since 2001-01-01', calendar='standard')))
        _nc3(str(tmp_path / ('t_%d.nc' % j)), d31, dict(ax1, t=(('time', 'level', 'latitude', 'longitude'), T[::-1][None], {})))
    _nc3(str(tmp_path / 'q.nc'), d3, dict(axl, q=(('time', 'level', 'latitude', 'longitude'), np.stack([r[::-1], r[::-1]]), {})))
    out = pp.gen_thermo(eng, str(tmp_path / 'sst.nc'), [str(tmp_path / 'sp.nc')], [str(tmp_path / 't_0.nc'), str(tmp_path / 't_1.nc')], str(tmp_path / 'q.nc'),
                        str(tmp_path / 'thermo.nc'), nl, table=(table['p'], table['s'], table['T']))
    ds = fields._Dataset(out)
    assert ds['vmax'].shape == (2, nla, nlo)
    tt = [datetime.datetime(2001, 1, 1) + datetime.timedelta(days=float(x)) for x in ds['time']]
    assert [(x.month, x.day) for x in tt] == [(1, 15), (2, 15)]
    ok = cases['b_PI'] > 0
    # sst went through Celsius and back and the levels through hPa: agreement to rounding
    np.testing.assert_allclose(ds['vmax'][0][ok], cases['b_PI'][ok], rtol=1e-7)
    np.testing.assert_allclose(ds['vmax'][1], ds['vmax'][0], rtol=0, atol=0)
    assert np.nanmin(ds['chi']) >= 0 and np.nanmax(ds['chi']) <= 10
    eng.close()


@pytest.mark.gpu
def test_device_pointer_entry_points(cases, table, built_lib):
    """tcr_potential_intensity_dev / tcr_wind_stats_dev (device buffers, caller's stream) give what the host
    entry points give."""
    import ctypes as C
    import torch
    from tropical_cyclone_risk_amd import preprocess as pp
    from tropical_cyclone_risk_amd.engine import TCEngine
    eng = TCEngine('GL', device=0)
    pp.stage_entropy_table(eng, table['p'], table['s'], table['T'])
    p, sst, psl, T, r = (cases['a_' + k] for k in ('p', 'sst', 'psl', 'T', 'r'))
    ref = pp.potential_intensity(eng, sst, psl, p, T, r)
    dev = torch.device('cuda', 0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    dp, dsst, dpsl, dT, dr = t(p), t(sst.ravel()), t(psl.ravel()), t(T.reshape(len(p), -1)), t(r.reshape(len(p), -1))
    out = torch.empty(sst.size, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    eng._ck(eng.L.tcr_potential_intensity_dev(eng.h, sst.size, len(p), dp.data_ptr(), dsst.data_ptr(), dpsl.data_ptr(),
                                              dT.data_ptr(), dr.data_ptr(), float(cases['Ck_over_Cd']), out.data_ptr(), C.c_void_p(st)))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(sst.shape), ref)
    rng = np.random.default_rng(2)
    planes = [rng.normal(size=(20, 500)) for _ in range(4)]
    want = eng.wind_stats(planes)
    dpl = [t(x) for x in planes]
    ptrs = (C.c_void_p * 4)(*[x.data_ptr() for x in dpl])
    o2 = torch.empty(14, 500, dtype=torch.float64, device=dev)
    eng._ck(eng.L.tcr_wind_stats_dev(eng.h, 20, 500, ptrs, None, 0, o2.data_ptr(), C.c_void_p(st)))
    torch.cuda.synchronize()
    assert np.array_equal(o2.cpu().numpy(), want)
    eng.close()
