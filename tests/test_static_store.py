"""Land / bathymetry on the grid and in the type the reference ships them, and their exact narrow storage in HBM.

The reference reads `intensity/data/land.nc` — int8 0 / 1 on lon 0 .. 359.875 (2880), lat -89.875 .. 90 (1440) — and a
`bathymetry.nc` it does not ship, crops both to the basin and hands them to RectBivariateSpline(kx=1, ky=1), which computes in
float64 (intensity/geo.py:9-34; `_get_over_land` tests the bilinear sum `== 1`, coupled_fast.py:35-38).  The library keeps the
two planes narrow when the VALUES allow it (tcr_static_store): one uint16 per grid point (land bit + whole metres), or a
uint8 + a float plane; otherwise the fp64 planes of rounds 1-4.  The bar: every output — rows, counters, flags and the
per-evaluation `land == 1` decision probe — is BIT-IDENTICAL to the fp64 storage of the same planes, and the run is held to
the C oracle on the same 0.125-degree planes like every other parity test.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PROBE_CAP = 1024
ROWS = ('lon', 'lat', 'v', 'm', 'vmax', 'envw')
COUNTERS = ('status', 'n_valid', 'nfev', 'flags', 'n_accept', 'n_reject')


@pytest.fixture(scope='module')
def envs(built_lib):
    from tropical_cyclone_risk_amd import synthetic
    cache = {}

    def get(kind, shape='era5', res=0.125):
        key = (kind, shape, res)
        if key not in cache:
            cache[key] = synthetic.make_env(shape, static_res=res, bathy_kind=kind)
        return cache[key]
    return get


def _engine(basin, env, store='auto'):
    import types
    from tropical_cyclone_risk_amd import namelist
    from tropical_cyclone_risk_amd.engine import TCEngine
    nl = types.SimpleNamespace(**{k: getattr(namelist, k) for k in dir(namelist) if not k.startswith('__')})
    nl.gpu_static_store = store
    return TCEngine(basin, device=0, nl=nl).stage_env(env)


def _same(a, b, probe=True):
    for k in ROWS:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    for k in COUNTERS:
        assert np.array_equal(a[k], b[k]), k
    if probe:
        assert np.array_equal(a['dec'], b['dec'])


@pytest.mark.parametrize('basin,kind,mode,n', [('GL', 'i16', 'pack16', 3000), ('NA', 'i16', 'pack16', 1500),
                                              ('GL', 'f32', 'u8_f32', 2000), ('AU', 'f64', 'f64', 600)])
def test_narrow_storage_is_bit_identical_and_matches_the_oracle(envs, basin, kind, mode, n):
    """0.125-degree int8 land + {whole-metre, float32, float64} bathymetry: the mode the upload picks, its footprint, bitwise
    equality with the fp64 planes (decision probe included), and the C oracle on the same planes."""
    from oracle import c_oracle, parity
    from tropical_cyclone_risk_amd import synthetic
    env = envs(kind)
    assert env.land.dtype == np.int8 and env.land.shape == (1440, 2880)
    storms = synthetic.draw_storm_inputs(n, basin, seed=500 + n)
    eng = _engine(basin, env)
    got_mode, nbytes = eng.static_info()
    assert got_mode == mode
    ref_eng = _engine(basin, env, store='f64')
    assert ref_eng.static_info()[0] == 'f64'
    if basin == 'GL':
        # the reference's real static shape: 66 MB as fp64 planes, <= 13 MB narrow
        assert ref_eng.static_info()[1] == 1440 * 2880 * 16
        assert nbytes == {'pack16': 1440 * 2880 * 2, 'u8_f32': 1440 * 2880 * 5}[mode]
        assert mode != 'pack16' or nbytes <= 13e6
    got = eng.integrate(storms, probe_cap=PROBE_CAP)
    want = ref_eng.integrate(storms, probe_cap=PROBE_CAP)
    _same(got, want)
    plain = eng.integrate(storms)                    # the production instantiation
    _same(plain, got, probe=False)
    # the fp32 variant reads the same narrow planes ((float) of an exact value == (float) of its double)
    a32, b32 = eng.integrate(storms, dtype='f32'), ref_eng.integrate(storms, dtype='f32')
    _same(a32, b32, probe=False)
    # gen_track(m=None) goes through the same lookup (k_init_m)
    st = dict(storms); st.pop('m0')
    assert np.array_equal(eng.init_m(st), ref_eng.init_m(st))
    ref_eng.close()
    # ... and against the oracle on the same planes
    ref = c_oracle.run_ensemble(env, basin, storms, probe=True)
    s = parity.check_tracks('static-%s-%s' % (basin, mode), got, ref, got['dec'], ref['dec'], ref['dec_t0'], eng.t_s,
                            counters=('status', 'n_valid', 'nfev', 'n_accept', 'n_reject'), replay=c_oracle.replayer(env, basin, storms))
    assert s['pointwise'] == s['n'] and s['unreplayed'] == 0 and s['hard_mismatch'] == 0
    assert s['exposed'] > 0                      # storms whose `land == 1` decisions are rounding-sensitive are in the sample
    eng.close()


def test_rhs_points_on_the_real_static_shape(envs):
    """The RHS itself at random points over land, coast and shelf: narrow storage against the C oracle's lookups on the same
    planes (alpha = the ocean feedback, which reads the bathymetry; dv/dt switches on `land == 1`)."""
    from oracle import c_oracle
    env = envs('i16')
    eng = _engine('NA', env)
    assert eng.static_info()[0] == 'pack16'
    rng = np.random.default_rng(9)
    n = 4000
    lon, lat = rng.uniform(262, 358, n), rng.uniform(2, 58, n)
    v, m, t = rng.uniform(5, 60, n), rng.uniform(0.1, 0.95, n), rng.uniform(0, 15 * 86400.0, n)
    ph = rng.uniform(0, 1, (4, 15))
    Fs = eng.fourier_table(ph[None])[0]
    cme = c_oracle.CMonthEnv(env, 'NA', 8)
    d_ref, w_ref, a_ref = c_oracle.rhs_points(cme, c_oracle.fourier_table(ph), 1400.0, t, lon, lat, v, m)
    d, w, a = eng.probe_rhs(8, 1400.0, Fs, t, lon, lat, v, m)
    assert (a_ref < 1).sum() > 100 and (d_ref[:, 2] < 0).sum() > 100
    scale = np.abs(d_ref).max(axis=0)
    assert (np.abs(d - d_ref) / scale).max() < 1e-12
    assert np.abs(a - a_ref).max() < 1e-13
    eng.close()


def test_two_grids_and_the_general_knot_search(envs):
    """(a) land on the 0.125-degree grid, bathymetry on its own 0.25-degree grid (two independent interpolators, geo.py:9-34):
    uint8 + float planes on two grids.  (b) a context whose grids are not all affine (a Gaussian-like latitude axis): the
    narrow planes are widened to fp64 once — the narrow kernels exist for affine grids only — and results are those of the
    fp64 storage bit for bit in both cases."""
    import dataclasses
    from tropical_cyclone_risk_amd import synthetic
    e8, e4 = envs('f32'), envs('f32', res=0.25)
    env = dataclasses.replace(e8, bathy=e4.bathy, blon=e4.hlon, blat=e4.hlat)
    storms = synthetic.draw_storm_inputs(800, 'NA', seed=41)
    a, b = _engine('NA', env), _engine('NA', env, store='f64')
    assert a.static_info()[0] == 'u8_f32' and b.static_info()[0] == 'f64_split'
    _same(a.integrate(storms, probe_cap=PROBE_CAP), b.integrate(storms, probe_cap=PROBE_CAP))
    a.close(); b.close()
    g = envs('i16', shape='gaussian')
    a, b = _engine('NA', g), _engine('NA', g, store='f64')
    assert a.static_info()[0] == 'pack16'
    ra = a.integrate(storms, probe_cap=PROBE_CAP)
    assert a.static_info()[0] == 'f64'                 # widened at the first launch
    _same(ra, b.integrate(storms, probe_cap=PROBE_CAP))
    a.close(); b.close()


def test_restaging_may_change_the_mode(envs):
    """A context re-staged with other planes picks the mode again (and results follow the planes staged last)."""
    from tropical_cyclone_risk_amd import synthetic
    storms = synthetic.draw_storm_inputs(300, 'NA', seed=3)
    eng = _engine('NA', envs('i16'))
    first = eng.integrate(storms)
    e64 = envs('f64')
    eng.stage_static(e64.hlon, e64.hlat, e64.land, e64.bathy)
    assert eng.static_info()[0] == 'f64'
    other = _engine('NA', e64)
    _same(eng.integrate(storms), other.integrate(storms), probe=False)
    e16 = envs('i16')
    eng.stage_static(e16.hlon, e16.hlat, e16.land, e16.bathy)
    assert eng.static_info()[0] == 'pack16'
    _same(eng.integrate(storms), first, probe=False)
    eng.close(); other.close()
