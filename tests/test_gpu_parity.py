"""GPU parity tests proper: the HIP path (through the C ABI) against
(a) the committed golden vectors produced by the reference's own code and
(b) the C oracle on larger seeded ensembles.

Stated fp64 tolerance (oracle/parity.py).  The reference integrates adaptively, so ulp-level libm /
summation-order differences are amplified along a trajectory, and its over-land test
``f_land.ev(lon, lat) == 1`` (intensity/coupled_fast.py:35-38) is decided by rounding in the
interior of land ("flicker").  Every implementation therefore records the decision of every RHS
evaluation (tcr_integrate_probe_host; the fixtures hold the reference's own decisions), and the bar is,
for EVERY storm and over its WHOLE track (none skipped, none prefix-only):
  * storms whose decision sequences agree (all clean storms and ~92 % of the exposed ones): discrete results
    (status, n_valid, nfev, accepted / rejected step counts, accept flags) identical and the pointwise tiers below;
  * storms with a differing decision at evaluation k: the differing evaluation itself must be rounding-sensitive
    (probe bit 2 at evaluation k), the samples before the step attempt that contains it agree, and the C oracle is
    run again with the other side's decisions forced at its rounding-sensitive evaluations (decision-forced replay,
    pinned to the reference by tests/golden/forced_*.npz) — after which the whole track is held to the same discrete
    equalities and the same tiers; the replay must leave no differing decision and report 0 hard mismatches;
  * tiers on the per-storm maximum over the hourly lon / lat / v / m / env winds (vmax: 5x), as counts: at most
    n/20 + 2 storms above 2e-11, at most n/100 + 1 above 1e-9 (the curated golden sets and the run_tracks composition
    state 1e-10 / 2e-10 for the 95 % tier, in the test, with the reason), and on EVERY sample
    max(1e-7, 10 x the oracle's own response to a one-ulp change of v0 on the same storms) — the intensity equation
    amplifies a last-bit difference while a storm intensifies, and the oracle's one-ulp twin is the yardstick for how far
    (test_parity_study_at_scale runs that comparison on 4 000 storms per basin by default, TCR_PARITY_STUDY=<n> for more).
The exposed fraction among accepted storms is printed and checked.
"""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

pytestmark = pytest.mark.gpu

PROBE_CAP = 1024


def _storms(g):
    return dict(lon=g['lon0'], lat=g['lat0'], v0=g['v0'], m0=g['m0'], h_bl=g['h_bl'],
                month=g['month'], phases=g['phases'])


def _check(tag, got, want, t_s, replay, counters=('status', 'n_valid', 'nfev'), **tiers):
    """got: engine.integrate(..., probe_cap=PROBE_CAP); want: c_oracle.run_ensemble(..., probe=True) or a
    golden fixture (decisions of the reference itself, ragged); replay: c_oracle.replayer(env, basin, storms) — storms
    whose `land == 1` decisions differ are re-run on the C oracle with the GPU's decisions forced at the
    rounding-sensitive evaluations and then held, over their whole track, to the same bar as all the others."""
    from oracle import parity
    if 'dec_off' in want:
        dec_w = parity.ragged_to_padded(want['dec'], want['dec_off'], PROBE_CAP)
        t0_w = parity.ragged_to_padded(want['dec_t0'], want['dec_off'], PROBE_CAP, fill=np.nan, dtype=np.float64)
    else:
        dec_w, t0_w = want['dec'], want['dec_t0']
    s = parity.check_tracks(tag, got, want, got['dec'], dec_w, t0_w, t_s, counters=counters, replay=replay, **tiers)
    # every storm was checked pointwise over its whole track — none skipped, none only prefix-checked
    assert s['pointwise'] == s['n'] and s['unreplayed'] == 0 and s['hard_mismatch'] == 0
    return s


@pytest.fixture(scope='module')
def engines(golden_env, built_lib):
    from tropical_cyclone_risk_amd.engine import TCEngine
    cache = {}

    def get(basin):
        if basin not in cache:
            cache[basin] = TCEngine(basin, device=0).stage_env(golden_env)
        return cache[basin]
    yield get
    for e in cache.values():
        e.close()


@pytest.mark.parametrize('basin', ['NA', 'AU', 'GL'])
def test_tracks_vs_reference_golden(engines, golden_env, basin):
    from oracle import c_oracle
    g = np.load(os.path.join(GOLDEN, 'tracks_%s.npz' % basin))
    eng = engines(basin)
    out = eng.integrate(_storms(g), probe_cap=PROBE_CAP)
    # the golden sets are curated, not random: 27 % of the NA tracks are accepted storms (5.7 % in a seeded ensemble) — the
    # long-lived intensifying ones that amplify a last-bit difference — so the 95 % tier is 1e-10 here instead of 2e-11
    s = _check('golden-' + basin, out, g, eng.t_s, c_oracle.replayer(golden_env, basin, _storms(g)), tol_95=1e-10)
    # the golden sets hold 10 (NA), 5 (AU), 3 (GL) flicker-exposed tracks; each is pointwise- or prefix-checked
    assert s['exposed'] == {'NA': 10, 'AU': 5, 'GL': 3}[basin]
    assert s['exposed'] == int((c_oracle.run_ensemble(golden_env, basin, _storms(g), post=False)['flicker'] > 0).sum())


@pytest.mark.parametrize('case', ['uncoupled', 'physics'])
def test_tracks_vs_reference_golden_namelist_variations(golden_env, built_lib, case):
    """The kernel's uncoupled-steering branch (tcr_device.h rhs_track: namelist.coupled_track = False -> steering_coefs,
    coupled_fast.py:190-191) and the physics scalars that reach the kernels through tcr_params (u_beta, v_beta, Ck,
    v_2d_thresh; PI_reduc through the staged PI; atm_bl_depth through h_bl), against tracks the reference itself produced
    with those namelist values (tests/golden/make_golden_namelist.py) — same bar as the default-namelist golden sets."""
    from oracle import c_oracle
    from tests.test_oracle_golden import namelist_case
    from tropical_cyclone_risk_amd.engine import TCEngine
    g, prm, env = namelist_case(golden_env, case)
    if case == 'uncoupled':
        nl = _namelist_with(coupled_track=False, steering_coefs=[float(x) for x in g['nl_steering_coefs']])
    else:
        nl = _namelist_with(u_beta=float(g['nl_u_beta']), v_beta=float(g['nl_v_beta']), Ck=float(g['nl_Ck']),
                            PI_reduc=float(g['nl_PI_reduc']), seed_v_2d_threshold_ms=float(g['nl_seed_v_2d_threshold_ms']))
    eng = TCEngine('NA', device=0, nl=nl).stage_env(env)
    out = eng.integrate(_storms(g), probe_cap=PROBE_CAP)
    t_s = eng.t_s
    eng.close()
    s = _check('golden-' + case, out, g, t_s, c_oracle.replayer(env, 'NA', _storms(g), prm=prm), tol_95=1e-10)
    assert s['n'] == len(g['status']) >= 24          # (status, n_valid, nfev, is_tc, accepted are part of the check)


@pytest.mark.parametrize('name,slot', [('NA', 8), ('SI', 1)])
def test_rhs_vs_reference_golden(engines, name, slot):
    g = np.load(os.path.join(GOLDEN, 'rhs_%s.npz' % name))
    eng = engines(name)
    dydt, envw, alpha = eng.probe_rhs(slot, float(g['h_bl']), g['Fs'], g['t'], g['lon'], g['lat'], g['v'], g['m'])
    scale = np.abs(g['dydt']).max(axis=0)
    assert (np.abs(dydt - g['dydt']) / scale).max() < 1e-13
    assert np.abs(envw - g['envw']).max() < 1e-12
    assert np.abs(alpha - g['alpha']).max() < 1e-13
    Fs = eng.fourier_table(g['phases'][None])[0]
    assert np.abs(Fs - g["Fs"]).max() < 5e-14      # periodic-table evaluation on the matrix cores, see k_fourier_mfma


@pytest.mark.parametrize('basin,n,seed', [('NA', 10000, 77), ('GL', 2000, 78), ('SI', 1000, 79)])
def test_ensemble_vs_c_oracle(engines, golden_env, basin, n, seed):
    """Seeded ensembles: GPU vs the C restatement, including step counters.  NA / 10 000 storms is BASELINE config 2's size
    ("NA basin, 1 year, 10k storms on 1 x MI355X, synthetic ERA5-shaped env fields, fp64")."""
    from oracle import c_oracle
    from tropical_cyclone_risk_amd import synthetic
    storms = synthetic.draw_storm_inputs(n, basin, seed=seed)
    eng = engines(basin)
    got = eng.integrate(storms, probe_cap=PROBE_CAP)
    # one storm in 10 000 needs more accepted RK steps than the default step record holds (65 > gpu_max_rk_steps = 64; SciPy
    # is unbounded): the library reports it (status -3) and the product's accept loop doubles the record and integrates the
    # round again (compute.accept_loop, tests/test_configs.py::test_step_record_grows_instead_of_aborting) — so does this test
    while (got['status'] == -3).any():
        assert basin == 'NA' and (got['status'] == -3).sum() <= 2 and eng.grow_step_record()
        got = eng.integrate(storms, probe_cap=PROBE_CAP)
    ref = c_oracle.run_ensemble(golden_env, basin, storms, probe=True)
    s = _check('oracle-' + basin, got, ref, eng.t_s, c_oracle.replayer(golden_env, basin, storms),
               counters=('status', 'n_valid', 'nfev', 'n_accept', 'n_reject'))
    assert s['accepted'] > 20 and s['accepted_exposed'] > 0      # accepted storms make landfall: the exposed ones are in the sample
    # the probe instantiation of the integrator is the production arithmetic
    plain = eng.integrate(storms)
    for k in ('lon', 'lat', 'v', 'm', 'vmax', 'envw'):
        assert np.array_equal(plain[k], got[k], equal_nan=True), k
    for k in ('status', 'n_valid', 'nfev', 'flags', 'n_accept', 'n_reject'):
        assert np.array_equal(plain[k], got[k]), k


def test_init_m_vs_reference(engines, golden_env):
    """gen_track(clon, clat, v, m=None) (coupled_fast.py:258-261): k_init_m against the reference's own `_init_m`
    (coupled_fast.py:153-173) at 240 points for dvdt = 0 and dvdt != 0, and twelve whole reference tracks started
    without m — `TCEngine.integrate` fills a missing / NaN m0 through tcr_init_m_host first."""
    from oracle import c_oracle, parity
    g = np.load(os.path.join(GOLDEN, 'init_m_NA.npz'))
    eng = engines('NA')
    n = len(g['lon'])
    st = dict(lon=g['lon'], lat=g['lat'], v0=g['v'], h_bl=np.full(n, float(g['h_bl'])), month=np.full(n, int(g['month'])), phases=g['phases'])
    for key, dvdt in (('m_dvdt0', 0.0), ('m_dvdt2em5', float(g['dvdt1']))):
        got = eng.init_m(st, dvdt)
        assert np.abs(got - g[key]).max() <= 1e-13, (key, np.abs(got - g[key]).max())
        assert np.array_equal(got == 1, g[key] == 1) and np.array_equal(got == 0, g[key] == 0)      # the clips land on the same points
    # a given m0 is kept; NaN entries are initialised
    m_in = np.where(np.arange(n) % 2 == 0, 0.37, np.nan)
    got = eng.init_m(dict(st, m0=m_in))
    assert (got[::2] == 0.37).all() and np.abs(got[1::2] - g['m_dvdt0'][1::2]).max() <= 1e-13
    # whole tracks without m
    storms = dict(lon=g['t_lon0'], lat=g['t_lat0'], v0=g['t_v0'], h_bl=g['t_h_bl'], month=g['t_month'], phases=g['t_phases'])
    want = dict(traj=g['t_traj'], status=g['t_status'], n_valid=g['t_n_valid'], nfev=g['t_nfev'], accepted=np.zeros(12, bool))
    out = eng.integrate(storms, probe_cap=PROBE_CAP)
    alive = want['n_valid'] > 0
    assert np.abs(out['m'][alive, 0] - want['traj'][alive, 3, 0]).max() <= 1e-13
    dec_w = parity.ragged_to_padded(g['t_dec'], g['t_dec_off'], PROBE_CAP)
    t0_w = parity.ragged_to_padded(g['t_dec_t0'], g['t_dec_off'], PROBE_CAP, fill=np.nan, dtype=np.float64)
    st2 = dict(storms, m0=eng.init_m(storms))
    s = parity.check_tracks('init-m', out, want, out['dec'], dec_w, t0_w, eng.t_s, flags=(), names=('traj',),
                            replay=c_oracle.replayer(golden_env, 'NA', st2), tol_95=1e-10)
    assert s['pointwise'] == 12


def test_empty_and_single(engines):
    """Edge cases: n = 0 and n = 1 batches, a v0 <= 4 seed (1-sample track), a gated seed."""
    from tropical_cyclone_risk_amd import synthetic
    eng = engines('NA')
    s = synthetic.draw_storm_inputs(1, 'NA', seed=3)
    empty = {k: v[:0] for k, v in s.items()}
    out = eng.integrate(empty)
    assert out['lon'].shape == (0, eng.n_steps)
    s['lon'][:] = 310.0; s['lat'][:] = 18.0; s['v0'][:] = 3.5; s['month'][:] = 9
    out = eng.integrate(s)
    assert out['status'][0] == 1 and out['n_valid'][0] == 1
    assert out['lon'][0, 0] == 310.0 and np.isnan(out['lon'][0, 1:]).all()
    assert np.isnan(out['vmax'][0]).all() and not out['accepted'][0]


@pytest.mark.parametrize('shape,basin', [('gfdl', 'WP'), ('gaussian', 'EP'), ('gaussian', 'GL')])
def test_two_grid_and_nonuniform_fields(built_lib, shape, basin):
    """GFDL-shaped fields: wind grid (2 x 2.5 deg) differs from the thermo grid (1 x 1.25 deg)
    (bam_track.py:82-83); 'gaussian' adds non-uniform latitudes, which takes the general
    knot-search path of tcr_device.h (`locate_t<false>`) instead of the affine fast path."""
    from oracle import c_oracle
    from tropical_cyclone_risk_amd import synthetic
    from tropical_cyclone_risk_amd.engine import TCEngine
    env = synthetic.make_env(shape, seed=7)
    storms = synthetic.draw_storm_inputs(1500, basin, seed=91)
    eng = TCEngine(basin, device=0).stage_env(env)
    got = eng.integrate(storms, probe_cap=PROBE_CAP)
    t_s = eng.t_s
    eng.close()
    ref = c_oracle.run_ensemble(env, basin, storms, probe=True)
    _check('%s-%s' % (shape, basin), got, ref, t_s, c_oracle.replayer(env, basin, storms),
           counters=('status', 'n_valid', 'nfev', 'n_accept', 'n_reject'))


def test_independent_land_and_bathymetry_grids(built_lib):
    """intensity/geo.py:9-34 builds f_bath and f_land as two independent interpolators; here the bathymetry sits on a
    0.5-degree grid and the land mask on the 0.25-degree one (tcr_static_upload2, the SPLIT instantiations).
    fp64 against the C oracle with the decision probe; the fp32 variant runs the same arrangement."""
    import copy
    from oracle import c_oracle
    from tropical_cyclone_risk_amd import synthetic
    from tropical_cyclone_risk_amd.engine import TCEngine
    env = copy.copy(synthetic.make_env('era5', seed=11))
    env.blon, env.blat, env.bathy = env.hlon[::2].copy(), env.hlat[::2].copy(), np.ascontiguousarray(env.bathy[::2, ::2])
    storms = synthetic.draw_storm_inputs(1500, 'WP', seed=17)
    eng = TCEngine('WP', device=0).stage_env(env)
    got = eng.integrate(storms, probe_cap=PROBE_CAP)
    f32 = eng.integrate(storms, dtype='f32')
    t_s = eng.t_s
    # the probe kernel (single evaluations) on the same arrangement
    cme = c_oracle.CMonthEnv(env, 'WP', 8)
    rng = np.random.default_rng(2)
    n = 400
    t = rng.uniform(0, 15 * 86400.0, n); lon = rng.uniform(101, 179, n); lat = rng.uniform(1, 59, n)
    v = rng.uniform(1, 70, n); m = rng.uniform(0, 1.1, n)
    Fs = eng.fourier_table(storms['phases'][:1])[0]
    dydt, envw, alpha = eng.probe_rhs(8, 1800.0, Fs, t, lon, lat, v, m)
    eng.close()
    o_dydt, o_envw, o_alpha = c_oracle.rhs_points(cme, Fs, 1800.0, t, lon, lat, v, m)
    assert (np.abs(dydt - o_dydt) / np.abs(o_dydt).max(axis=0)).max() < 1e-12 and np.abs(alpha - o_alpha).max() < 1e-12
    ref = c_oracle.run_ensemble(env, 'WP', storms, probe=True)
    _check('split-static-WP', got, ref, t_s, c_oracle.replayer(env, 'WP', storms),
           counters=('status', 'n_valid', 'nfev', 'n_accept', 'n_reject'))
    # and it matters: the same storms on the shared 0.25-degree bathymetry give different tracks
    env1 = synthetic.make_env('era5', seed=11)
    ref1 = c_oracle.run_ensemble(env1, 'WP', storms)
    assert (ref1['n_valid'] != ref['n_valid']).any()
    assert (f32['status'] == got['status']).mean() > 0.99 and (np.abs(f32['n_valid'] - got['n_valid']) <= 6).mean() > 0.97


def _namelist_with(**over):
    import types
    from tropical_cyclone_risk_amd import namelist
    nl = types.SimpleNamespace(**{k: getattr(namelist, k) for k in dir(namelist) if not k.startswith('__')})
    for k, v in over.items():
        setattr(nl, k, v)
    return nl


@pytest.mark.parametrize('dt_out,days,T_days', [(5400, 10, 20), (7000, 15, 20), (3600, 15, 17.3), (21600, 6, 20), (1800, 9, 20)])
def test_other_output_grids(golden_env, built_lib, dt_out, days, T_days):
    """Non-default namelist time settings: n_steps != 361, output intervals that do not divide
    the track length (np.linspace step != dt_out, bam_track.py:54-55), and a Fourier period that
    is not a whole number of output intervals (direct k_fourier_direct path instead of the
    periodic table); 1800 s x 9 d = 433 samples is beyond the matrix-core kernel's 384 (k_fourier_periodic)."""
    from oracle import c_oracle, scipy_port
    from tropical_cyclone_risk_amd import synthetic
    from tropical_cyclone_risk_amd.engine import TCEngine
    nl = _namelist_with(output_interval_s=dt_out, total_track_time_days=days, T_days=T_days)
    prm = scipy_port.Params(dt_out=float(dt_out), total_time=days * 86400.0, T_Fs=T_days * 86400.0)
    storms = synthetic.draw_storm_inputs(600, 'NA', seed=5 + dt_out)
    eng = TCEngine('NA', device=0, nl=nl).stage_env(golden_env)
    assert eng.n_steps == prm.n_steps
    got = eng.integrate(storms, probe_cap=PROBE_CAP)
    Fs = eng.fourier_table(storms['phases'][:3])
    t_s = eng.t_s
    eng.close()
    ref = c_oracle.run_ensemble(golden_env, 'NA', storms, prm=prm, probe=True)
    for i in range(3):
        assert np.abs(Fs[i] - c_oracle.fourier_table(storms['phases'][i], prm)).max() < 5e-14
    _check('dt%d-%dd' % (dt_out, days), got, ref, t_s, c_oracle.replayer(golden_env, 'NA', storms, prm=prm))


def test_full_size_ensemble_properties(golden_env, built_lib):
    """BASELINE's full size (100 000 storms, GL, device-seeded) through size-independent properties:
      * launch-shape invariance: the same batch integrated with a different number of persistent
        waves (different storm-to-lane assignment and queue order; no chain of passes, no table segments) and with the
        forcing table written in one piece is bitwise identical;
      * batch-composition invariance: a storm integrated inside the 100k batch equals the same
        storm integrated in a small batch, bitwise;
      * structure: NaN padding exactly beyond n_valid, accepted => is_tc, sample 0 == the seed;
      * a random 1 500-storm subsample agrees with the C oracle within the stated tolerance."""
    import torch
    from oracle import c_oracle
    from tropical_cyclone_risk_amd.engine import TCEngine
    from tropical_cyclone_risk_amd.pipeline import DevicePipeline
    B = 100_000
    eng = TCEngine('GL', device=0).stage_env(golden_env)
    pipe = DevicePipeline(eng, 560_000, B)
    pipe.seed_round(2005, 0)
    pipe.select_passed(B)
    assert int(pipe.n_passed.item()) >= B
    pipe.integrate(B)
    a = pipe.host_tracks()
    eng.tune(waves=700)
    try:
        pipe.integrate(B)
        b = pipe.host_tracks()
    finally:
        eng.tune(waves=-1)
    for k in ('lon', 'lat', 'v', 'm', 'vmax', 'envw'):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    for k in ('n_valid', 'status', 'flags', 'nfev', 'n_accept', 'n_reject'):
        assert np.array_equal(a[k], b[k]), k
    # the forcing table in one piece (every storm, before the chain) instead of two segments (the second only for the
    # storms the first pass parks): same chain of passes otherwise, bitwise the same result
    eng.tune(table_segments=0)
    try:
        pipe.integrate(B)
        c = pipe.host_tracks()
    finally:
        eng.tune(table_segments=-1)
    for k in ('lon', 'lat', 'v', 'm', 'vmax', 'envw'):
        assert np.array_equal(a[k], c[k], equal_nan=True), k
    for k in ('n_valid', 'status', 'flags', 'nfev', 'n_accept', 'n_reject'):
        assert np.array_equal(a[k], c[k]), k
    # structure
    ns = eng.n_steps
    idx = np.arange(ns)[None, :]
    valid = idx < a['n_valid'][:, None]
    for k in ('lon', 'lat', 'v', 'm'):
        assert np.array_equal(np.isnan(a[k]), ~valid), k
    assert np.array_equal(np.isnan(a['envw'][:, :, 0]), ~valid)
    assert not (a['accepted'] & ~a['is_tc']).any()
    seeds = {k: pipe.storms[k][:B].cpu().numpy() for k in ('lon0', 'lat0', 'v0', 'm0', 'h_bl', 'slot')}
    alive = a['n_valid'] > 0
    assert np.array_equal(a['lon'][alive, 0], seeds['lon0'][alive]) and np.array_equal(a['v'][alive, 0], seeds['v0'][alive])
    # subsample: small-batch bitwise equality + oracle tolerance
    rng = np.random.default_rng(3)
    sub = np.sort(rng.choice(B, 1500, replace=False))
    storms = dict(lon=seeds['lon0'][sub], lat=seeds['lat0'][sub], v0=seeds['v0'][sub], m0=seeds['m0'][sub],
                  h_bl=seeds['h_bl'][sub], month=seeds['slot'][sub] + 1,
                  phases=pipe.storms['phases'][:B].cpu().numpy()[sub].reshape(len(sub), 4, -1))
    small = eng.integrate(storms, probe_cap=PROBE_CAP)
    for k in ('lon', 'lat', 'v', 'm', 'vmax', 'envw'):
        assert np.array_equal(small[k], a[k][sub], equal_nan=True), k
    t_s = eng.t_s
    eng.close()
    ref = c_oracle.run_ensemble(golden_env, 'GL', storms, probe=True)
    print('full size: %d storm-steps, %.1f %% accepted' % (np.clip(a['n_valid'] - 1, 0, None).sum(), 100 * a['accepted'].mean()))
    _check('full-size-sample', small, ref, t_s, c_oracle.replayer(golden_env, 'GL', storms))


def test_pad_state_reuse_is_bit_identical(golden_env, built_lib):
    """tcr_tracks.pad_state (tcrisk_hip.h): re-using the plane buffers batch after batch, with only the
    stale part of each row re-padded, must give the planes a full NaN padding gives, bit for bit."""
    import torch
    from tropical_cyclone_risk_amd.engine import TCEngine
    from tropical_cyclone_risk_amd.pipeline import DevicePipeline
    B = 4000
    eng = TCEngine('GL', device=0).stage_env(golden_env)
    reused = DevicePipeline(eng, 40_000, B)
    assert int(reused.tracks['pad_state'].min().item()) == -1
    keys = ('lon', 'lat', 'v', 'm', 'vmax', 'envw')
    for year in (2001, 2002, 2003):
        reused.seed_round(year, 0); reused.select_passed(B); reused.integrate(B)
        got = reused.host_tracks()
        assert torch.equal(reused.tracks['pad_state'][:B].cpu(), torch.as_tensor(got['n_valid']))
        fresh = DevicePipeline(eng, 40_000, B)
        for k in keys:                       # poison: a fresh buffer holds garbage, not NaN
            fresh.tracks[k].fill_(123.0)
        fresh.seed_round(year, 0); fresh.select_passed(B); fresh.integrate(B)
        ref = fresh.host_tracks()
        for k in keys:
            assert np.array_equal(got[k], ref[k], equal_nan=True), (year, k)
        assert np.array_equal(got['flags'], ref['flags'])
        del fresh
    eng.close()


def test_tc_rows_only_matches_all_rows(golden_env, built_lib):
    """tcr_tracks.tc_rows_only: accept test 1 decided from the v series alone (k_screen), env winds / vmax / rows
    only for the storms that pass it — what the reference does (compute.py:185-204).  Flags and counters must equal
    the all-rows mode for every storm, rows of is_tc storms must be bit-identical, rows of the others untouched;
    also with the plane buffers re-used batch after batch (pad_state)."""
    import torch
    from tropical_cyclone_risk_amd.engine import TCEngine
    from tropical_cyclone_risk_amd.pipeline import DevicePipeline
    B = 30_000
    eng = TCEngine('GL', device=0).stage_env(golden_env)
    keys = ('lon', 'lat', 'v', 'm', 'vmax', 'envw')
    tc = DevicePipeline(eng, 200_000, B, tc_rows_only=True)
    for k in keys:
        tc.tracks[k].fill_(123.0)
    prev_tc = None
    for year in (2001, 2002):
        full = DevicePipeline(eng, 200_000, B)
        for p in (full, tc):
            p.seed_round(year, 0); p.select_passed(B)
            assert int(p.n_passed.item()) >= B
            p.integrate(B)
        a, b = full.host_tracks(), tc.host_tracks()
        for k in ('n_valid', 'status', 'flags', 'nfev', 'n_accept', 'n_reject'):
            assert np.array_equal(a[k], b[k]), (year, k)
        is_tc = a['is_tc']
        assert 0.02 < is_tc.mean() < 0.5 and a['accepted'].sum() > 100
        for k in keys:
            assert np.array_equal(a[k][is_tc], b[k][is_tc], equal_nan=True), (year, k)
        # rows of storms that are not TCs were not touched: still the previous batch's contents
        before = prev_tc if prev_tc is not None else {k: np.full_like(b[k], 123.0) for k in keys}
        for k in keys:
            assert np.array_equal(b[k][~is_tc], before[k][~is_tc], equal_nan=True), (year, k)
        ps = tc.tracks['pad_state'][:B].cpu().numpy()
        assert np.array_equal(ps[is_tc], a['n_valid'][is_tc])
        prev_tc = {k: b[k].copy() for k in keys}
        stats = torch.zeros(10, dtype=torch.int64, device=tc.dev)
        tc.add_stats(stats); torch.cuda.synchronize()
        st = stats.cpu().numpy()
        assert st[4] == is_tc.sum() and st[5] == a['n_valid'][is_tc].sum() and st[3] == a['accepted'].sum()
        assert st[0] == np.clip(a['n_valid'] - 1, 0, None).sum() and st[1] == a['nfev'].sum() and st[2] == a['n_valid'].sum()
        del full
    # the bounded grids of k_dense / k_emit over the TC list: with 64 workgroup rows every workgroup takes ~30 list
    # entries in turn; rows and flags must not depend on the grid
    eng.tune(emit_grid_cap=64)
    try:
        tc2 = DevicePipeline(eng, 200_000, B, tc_rows_only=True)
        tc2.seed_round(2002, 0); tc2.select_passed(B); tc2.integrate(B)
        c = tc2.host_tracks()
    finally:
        eng.tune(emit_grid_cap=-1)
    assert np.array_equal(c['flags'], b['flags']) and c['is_tc'].sum() > 64 * 3
    for k in keys:
        assert np.array_equal(c[k][is_tc], b[k][is_tc], equal_nan=True), k
    eng.close()


def test_fp32_variant_within_stated_tolerance(golden_env, built_lib):
    """BASELINE config 5: the fp32 variant (tcr_integrate_f32_*: fp32 fields / state / RHS / rows, fp64 time and step
    controller) against the fp64 path on identical storms.  Stated tolerance (profiles/r02_fp32_study.json has the
    full distributions at 100 000 storms; the bounds below leave a 3-10x margin over them):
      * status identical for >= 99.9 % of the storms, track length for >= 97 %, within 6 h for >= 99.5 %;
      * where the lengths agree: |dv| median <= 1e-4 m/s, p99 <= 0.1 m/s; |dlon|, |dlat| median <= 1e-4 deg,
        p99 <= 5e-3 deg; |dm| p99 <= 1e-3 (per-storm maxima over the whole track);
      * the first 24 h (before the adaptive integrator has amplified anything): |dv| p99.9 <= 0.03 m/s;
      * accept decisions: is_tc flips <= 0.05 % of the storms, accepted flips <= 1 % of the accepted tracks;
      * the TC-rows-only mode of the fp32 path gives the same flags as its all-rows mode and bit-identical rows."""
    import torch
    from tropical_cyclone_risk_amd.engine import TCEngine
    from tropical_cyclone_risk_amd.pipeline import DevicePipeline
    B = 40_000
    eng = TCEngine('GL', device=0).stage_env(golden_env)
    res = {}
    for tag, kw in (('f64', dict()), ('f32', dict(dtype='f32')), ('f32tc', dict(dtype='f32', tc_rows_only=True))):
        p = DevicePipeline(eng, 260_000, B, **kw)
        p.seed_round(2007, 0); p.select_passed(B)
        assert int(p.n_passed.item()) >= B
        p.integrate(B); torch.cuda.synchronize()
        res[tag] = p.host_tracks()
        assert res[tag]['lon'].dtype == (np.float64 if tag == 'f64' else np.float32)
        del p
    eng.close()
    a, b, c = res['f64'], res['f32'], res['f32tc']
    assert (a['status'] == b['status']).mean() >= 0.999
    dn = np.abs(b['n_valid'].astype(np.int64) - a['n_valid'])
    assert (dn == 0).mean() >= 0.97 and (dn <= 6).mean() >= 0.995
    same = dn == 0
    idx = np.arange(a['lon'].shape[1])[None, :]
    for k, med, p99 in (('v', 1e-4, 0.1), ('lon', 1e-4, 5e-3), ('lat', 1e-4, 5e-3), ('m', 1e-5, 1e-3)):
        d = np.abs(np.nan_to_num(a[k][same]) - np.nan_to_num(b[k][same]).astype(np.float64)).max(axis=1)
        print('fp32 %-3s per-storm max |d|: median %.3g  p99 %.3g  max %.3g' % (k, np.median(d), np.percentile(d, 99), d.max()))
        assert np.median(d) <= med and np.percentile(d, 99) <= p99, k
    m24 = (idx <= 24) & (idx < np.minimum(a['n_valid'], b['n_valid'])[:, None])
    d24 = np.abs(a['v'] - b['v'].astype(np.float64))[m24]
    assert np.percentile(d24, 99.9) <= 0.03
    assert (a['is_tc'] != b['is_tc']).mean() <= 5e-4
    flips = (a['accepted'] != b['accepted']).sum()
    print('fp32: %d accepted (fp64 %d), %d flips' % (b['accepted'].sum(), a['accepted'].sum(), flips))
    assert flips <= 0.01 * a['accepted'].sum()
    # fp32 TC-rows-only vs fp32 all rows
    for k in ('n_valid', 'status', 'flags', 'nfev'):
        assert np.array_equal(b[k], c[k]), k
    tc = b['is_tc']
    for k in ('lon', 'lat', 'v', 'm', 'vmax', 'envw'):
        assert np.array_equal(b[k][tc], c[k][tc], equal_nan=True), k


def test_wind_stats_kernel_vs_oracle(built_lib):
    """k_wind_stats against the NumPy restatement — fp64 inputs, sums in day order on both
    sides, so bit for bit; plus the host mirror of calc_wnd_stat (month mask, levels, grouping rule)."""
    import datetime
    from oracle import wind_stats as ws
    from tropical_cyclone_risk_amd import preprocess as pp
    from tropical_cyclone_risk_amd.engine import TCEngine
    rng = np.random.default_rng(11)
    eng = TCEngine('GL', device=0)
    planes = [rng.normal(5 * c, 3 + c, size=(31, 91, 180)) for c in range(4)]
    got = eng.wind_stats(planes)
    assert np.array_equal(got, ws.wind_stats(planes))
    # sub-daily record with ragged days
    ds = np.array([0, 3, 4, 8, 12, 13, 17, 21, 25, 31], dtype=np.int32)
    assert np.array_equal(eng.wind_stats(planes, ds), ws.wind_stats(planes, ds))
    # float32 planes — the dtype ERA5 u / v files hold: reduced in float32 exactly as NumPy does (np.nanmean /
    # np.nanvar / the xr.cov steps, oracle/wind_stats.py), NaN samples skipped per statistic and per pair
    p32 = [(p * (1 + 50 * (c == 2))).astype(np.float32) for c, p in enumerate(planes)]
    assert np.array_equal(eng.wind_stats(p32), ws.wind_stats(p32))
    assert not np.array_equal(eng.wind_stats(p32), eng.wind_stats([p.astype(np.float64) for p in p32]))   # it is not the fp64 reduction
    p32[0][3, 10, 20] = np.nan; p32[2][3, 10, 20] = np.nan; p32[1][7, 10, 20] = np.nan; p32[3][:, 50, 60] = np.nan
    with np.errstate(all='ignore'):
        a, b = eng.wind_stats(p32), ws.wind_stats(p32)
    assert np.array_equal(a, b, equal_nan=True) and np.isnan(a[3, 50, 60]) and not np.isnan(a[:, 10, 20]).any()
    p64 = [p.copy() for p in planes]
    p64[1][5, 2, 3] = np.nan
    assert np.array_equal(eng.wind_stats(p64), ws.wind_stats(p64), equal_nan=True)
    # covariance matrix is symmetric positive semi-definite up to the ddof mix: check the diagonal dominates
    assert (got[4] > 0).all() and (got[6] > 0).all()
    # calc_wnd_stat: 6-hourly record over two months, 3 levels in Pa
    times = [datetime.datetime(2001, 1, 25) + datetime.timedelta(hours=6 * k) for k in range(80)]
    ua = rng.normal(size=(80, 3, 20, 30)).astype(np.float32)
    va = rng.normal(size=(80, 3, 20, 30)).astype(np.float32)
    lev = [85000, 50000, 25000]
    out = pp.calc_wnd_stat(eng, ua, va, lev, 'Pa', times, 2001, 2)
    keep = pp.month_mask(times, 2001, 2)
    sel = [ua[keep][:, 2], va[keep][:, 2], ua[keep][:, 0], va[keep][:, 0]]         # float32, like the files
    assert sel[0].dtype == np.float32
    assert np.array_equal(out, ws.wind_stats(sel))                                  # the reference's `< 0` test: no grouping
    out_d = pp.calc_wnd_stat(eng, ua, va, lev, 'Pa', times, 2001, 2, group_days=True)
    assert np.array_equal(out_d, ws.wind_stats(sel, pp.day_groups([t for t, k in zip(times, keep) if k])))
    with pytest.raises(Exception, match='at least two days'):
        eng.wind_stats([p[:1] for p in planes])
    eng.close()


def test_parity_study_at_scale(golden_env, built_lib):
    """The parity tiers of this file on a large random ensemble per basin — 4 000 storms by default (about a minute),
    TCR_PARITY_STUDY=<storms per basin> for the 20 000 of profiles/r03_parity_study.json; writes
    gpurun_out/parity_study.json: every storm
    checked pointwise over its whole track against the C oracle — decision-identical storms directly, the others after
    the decision-forced replay; none skipped.  The intensity equation amplifies perturbations
    (an e-folding of hours while a storm intensifies), so over 20 000 fifteen-day tracks a last-bit difference of the
    device's libm grows to 1e-5 in a handful of storms — the same storms, and the same amounts, by which the ORACLE
    moves when one of its inputs (v0) is changed by one ulp.  Asserted per basin, on ALL storms
    (per-storm maximum over lon / lat / v / m):
      * the p95 / p99 tiers of oracle/parity.py (2e-11 / 1e-9), as counts;
      * p99.9: no more than 10x the oracle's own p99.9 response to the one-ulp change (or 1e-6, whichever is larger);
        storms above 1e-6: at most 3x the twin's count + 3; none above 1e-4."""
    import json
    from oracle import c_oracle, parity
    from tropical_cyclone_risk_amd import synthetic
    from tropical_cyclone_risk_amd.engine import TCEngine
    n = int(os.environ.get('TCR_PARITY_STUDY') or 4000)
    out = {'storms_per_basin': n,
           'd_gpu': 'per storm max |GPU - C oracle| over lon, lat, v, m of the hourly samples, ALL storms (the ones with a '
                    'differing land == 1 decision against the decision-forced replay; d_gpu_replayed_storms = those alone)',
           'd_ulp': 'the same between the C oracle and the C oracle with v0 -> nextafter(v0) (same decision sequence and counters)'}

    def md(a, b):
        return np.abs(np.nan_to_num(a) - np.nan_to_num(b)).reshape(len(a), -1).max(axis=1)

    for basin in ('NA', 'AU', 'GL', 'WP'):
        storms = synthetic.draw_storm_inputs(n, basin, seed=9000 + len(basin) + ord(basin[0]))
        eng = TCEngine(basin, device=0).stage_env(golden_env)
        got = eng.integrate(storms, probe_cap=PROBE_CAP)
        t_s = eng.t_s
        eng.close()
        ref = c_oracle.run_ensemble(golden_env, basin, storms, probe=True)
        pert = dict(storms)
        pert['v0'] = np.nextafter(storms['v0'], np.inf)
        ref2 = c_oracle.run_ensemble(golden_env, basin, pert, probe=True)
        from oracle import parity as P
        s = P.check_tracks('study-' + basin, got, ref, got['dec'], ref['dec'], ref['dec_t0'], t_s, tol_all=np.inf,
                           replay=c_oracle.replayer(golden_env, basin, storms))
        assert s['pointwise'] == s['n'] and s['unreplayed'] == 0 and s['hard_mismatch'] == 0
        agree = P.first_divergence(got['dec'], ref['dec']) < 0
        twin = (P.first_divergence(ref2['dec'], ref['dec']) < 0) & (ref2['nfev'] == ref['nfev']) & (ref2['n_valid'] == ref['n_valid'])
        d, u = s['per_storm']['traj'], md(ref2['traj'], ref['traj'])[twin]
        q = lambda x: dict(zip(('p50', 'p95', 'p99', 'p99.9', 'max'), (float(v) for v in np.percentile(x, [50, 95, 99, 99.9, 100]))))
        qd, qu, qr = q(d), q(u), q(d[~agree]) if (~agree).any() else None
        qo = {name: q(s['per_storm'][name]) for name in ('envw', 'vmax')}
        # the far tail is a handful of storms whose intensification amplifies a last-bit difference by 1e9 and more; the
        # oracle's own one-ulp twin has the same tail (its maximum over 20 000 storms ranges from 6e-8 to 2e-4 between
        # basins and runs), so: p99.9 within 10x of the twin's, storms above 1e-6 at most 3x the twin's count + 3, none above 1e-4
        # (the integrator's own tolerance is rtol = 1e-3)
        assert qd['p99.9'] <= max(1e-6, 10 * qu['p99.9']), (basin, qd, qu)
        assert int((d > 1e-6).sum()) <= 3 * int((u > 1e-6).sum()) + 3 and qd['max'] <= P.TOL_TAIL_CAP, (basin, qd, qu, int((d > 1e-6).sum()), int((u > 1e-6).sum()))
        # ... and storm by storm (round 5): every storm above 1e-7 is one the oracle itself amplifies — its difference at most
        # parity.TWIN_FACTOR x what the oracle moves on THAT storm under a one-ulp change of one input (three twins walking the
        # same decision sequence, c_oracle.replayer.twin_storms): the rule the other tests apply to every sample, here at scale
        rp = c_oracle.replayer(golden_env, basin, storms)
        big = np.nonzero(d > P.TOL_ALL_FLOOR)[0]
        ratios = []
        if len(big):
            tw = rp.twin_storms(big, np.ascontiguousarray(got['dec'][big]))['traj']
            ratios = [(int(i), float(d[i]), float(t), float(d[i] / t) if t > 0 else float('inf')) for i, t in zip(big, tw)]
            print(basin, 'storms above 1e-7 (index, d, own twin, ratio):', ratios)
            assert all(r[3] <= P.TWIN_FACTOR for r in ratios), (basin, ratios)
        out[basin] = dict(summary={k: v for k, v in s.items() if not isinstance(v, (dict, list))}, worst=s['worst'], d_gpu=qd, d_ulp=qu,
                          max_ratio_to_own_twin=max([r[3] for r in ratios], default=0.0), twin_factor=P.TWIN_FACTOR, tail_cap=P.TOL_TAIL_CAP,
                          storms_over_1e7_vs_their_own_twin=[dict(storm=r[0], d_gpu=r[1], own_twin=r[2], ratio=r[3]) for r in ratios],
                          d_gpu_replayed_storms=qr, d_gpu_envw=qo['envw'], d_gpu_vmax=qo['vmax'],
                          storms_over_1e9=int((d > 1e-9).sum()), oracle_twins_over_1e9=int((u > 1e-9).sum()),
                          storms_over_1e6=int((d > 1e-6).sum()), oracle_twins_over_1e6=int((u > 1e-6).sum()),
                          replayed_over_1e9=int((d[~agree] > 1e-9).sum()),
                          accepted=int(ref['accepted'].sum()), is_tc=int(ref['is_tc'].sum()))
        print(basin, 'd_gpu', qd, 'd_ulp', qu, 'max ratio to own twin %.2f' % out[basin]['max_ratio_to_own_twin'])
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_study.json', 'w') as f:
        json.dump(out, f, indent=1, default=lambda o: o.item() if hasattr(o, 'item') else str(o))


def test_integrator_arithmetic_helpers_against_numpy(built_lib):
    """Round 6: the integrator's division / square root without range scaling and its three libm-class substitutions
    (csrc/tcr_device.h, "Arithmetic policy"), each on its own through tcr_probe_math_host, against NumPy in extended precision:
      * qdiv_nz, qsqrt, qsqrt_pos: the IEEE result bit for bit on the operands the path has (what makes them parity-neutral);
      * inv_fifth_root (err ** -0.2, rk.py:160) and strat_pow (t_strat ** -0.4, coupled_fast.py:91): within 2 ulp;
      * cos_lat (np.cos(np.deg2rad(lat)), bam_track.py:139): within 1.5 ulp equatorward of 60 degrees, 5 ulp up to 80 degrees
        (where the track stops, bam_track.py:134);
    and the special values the callers rely on."""
    import ctypes as C
    from tropical_cyclone_risk_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    assert L.tcr_ctx_create(0, C.byref(h)) == 0

    def run(fn, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = np.empty_like(a)
        bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
        rc = L.tcr_probe_math_host(h, fn, a.size, a.ctypes.data_as(_lib.DP), bb.ctypes.data_as(_lib.DP) if bb is not None else None,
                                   out.ctypes.data_as(_lib.DP))
        assert rc == 0, L.tcr_last_error(h).decode()
        return out

    def ulps(got, want_ld):
        want = want_ld.astype(np.float64)
        return np.abs((got.astype(np.longdouble) - want_ld) / np.spacing(np.abs(want)).astype(np.longdouble)).astype(np.float64)
    rng = np.random.default_rng(6)
    n = 400_000
    # ---- division: dividends of either sign over 40 decades, divisors the path has (grid steps, cos(lat), h_bl, error scales, sigma)
    a = rng.normal(size=n) * 10.0 ** rng.uniform(-20, 20, n)
    b = np.concatenate([rng.uniform(0.17, 1.0, n // 4), rng.uniform(500, 3000, n // 4), 10.0 ** rng.uniform(-6, 2, n // 4),
                        rng.uniform(0.5, 30, n - 3 * (n // 4))])
    q = run(0, a, b)
    assert np.array_equal(q, a / b), 'qdiv_nz differs from IEEE division on %d of %d operands' % ((q != a / b).sum(), n)
    assert np.isnan(run(0, [np.nan, 1.0], [2.0, np.nan])).all()
    # ---- square roots: sums of squares of speeds, variances
    x = 10.0 ** rng.uniform(-12, 8, n)
    for fn in (1, 2):
        assert np.array_equal(run(fn, x), np.sqrt(x)), fn
    s = run(1, [0.0, np.inf, -1.0, np.nan, 4.0])
    assert s[0] == 0.0 and s[1] == np.inf and np.isnan(s[2]) and np.isnan(s[3]) and s[4] == 2.0
    # ---- a ** (-1/5): error norms from 1e-14 to 1e6, stratifications from 1e-4 to 1e3 K / 100 m
    # (what the reference computes is pow with the DOUBLE exponents -0.2 and -0.4, which are not -1/5 and -2/5: |ln a| x 1.1e-17 resp.
    # 2.2e-17 relative, up to 3 ulp over these ranges — the helper corrects for it, tcr_device.h; ue / u4e: against the exact roots)
    e = 10.0 ** rng.uniform(-14, 6, n)
    u = ulps(run(3, e), np.power(e.astype(np.longdouble), np.longdouble(-0.2)))
    ue = ulps(run(3, e), np.power(e.astype(np.longdouble), -np.longdouble(1) / np.longdouble(5)))
    assert u.max() <= 2.0, (u.max(), ue.max())
    g = 10.0 ** rng.uniform(-4, 3, n)
    u4 = ulps(run(4, g), np.power(g.astype(np.longdouble), np.longdouble(-0.4)))
    u4e = ulps(run(4, g), np.power(g.astype(np.longdouble), -np.longdouble(2) / np.longdouble(5)))
    assert u4.max() <= 2.0, (u4.max(), u4e.max())
    sp = run(4, [-1.0, np.nan, 1.0, 32.0])
    assert np.isnan(sp[0]) and np.isnan(sp[1]) and sp[2] == 1.0 and abs(sp[3] - 0.25) <= 2 * np.spacing(0.25)
    # step-size control at the ends of the range: a tiny / huge error norm ends at the clamps like pow does (rk.py:157-165)
    pw = 0.9 * run(3, [1e-300, 0.0, 1e300, np.nan])
    assert np.fmin(10.0, pw[0]) == 10.0 and np.fmin(10.0, pw[1]) == 10.0 and np.fmax(0.2, pw[2]) == 0.2 and np.fmax(0.2, pw[3]) == 0.2
    # ---- cos(lat)
    lat = rng.uniform(-80, 80, n)
    xr = lat * (np.pi / 180.0)
    u = ulps(run(5, xr), np.cos(xr.astype(np.longdouble)))
    assert u[np.abs(lat) <= 60].max() <= 1.5 and u.max() <= 5.0, (u[np.abs(lat) <= 60].max(), u.max())
    c = run(5, [0.0, np.nan, np.pi / 2])
    assert c[0] == 1.0 and np.isnan(c[1]) and abs(c[2]) < 1e-15 and c[2] != 0.0
    print('arithmetic helpers: division / sqrt bit-identical on %d operands each; a ** (-1/5): %.2f ulp from the exact root, %.2f from '
          'pow(a, -0.2) over 20 decades; t_strat ** -0.4: %.2f / %.2f; cos(lat) max %.2f ulp (%.2f within 60 degrees)'
          % (n, ue.max(), ulps(run(3, e), np.power(e.astype(np.longdouble), np.longdouble(-0.2))).max(), u4e.max(), u4.max(), u.max(), u[np.abs(lat) <= 60].max()))
    L.tcr_ctx_destroy(h)
