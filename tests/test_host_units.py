"""Host logic against reference known-answer vectors (tests/golden/units.npz). CPU only."""
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
