#!/usr/bin/env python3
"""Generate the golden fixtures by running the reference's own code.

Run ONLY in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

Outputs small .npz files next to this script.  Each holds *data only*: the
inputs handed to the reference (storm seeds, Fourier phases, query points; the
fields are regenerated from ``synthetic.make_env(shape, seed, zero_cov_patch)``)
and the outputs the reference produced.  SciPy / NumPy versions are recorded
because the integrator and the spline evaluator live there.
"""
import os
import sys
import warnings

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from tests.golden import ref_harness as H            # noqa: E402
from tropical_cyclone_risk_amd import synthetic      # noqa: E402

ENV_KW = dict(shape='era5', seed=20250614, zero_cov_patch=True)
N_STEPS = 361
META = dict(scipy=scipy.__version__, numpy=np.__version__,
            env_shape=ENV_KW['shape'], env_seed=ENV_KW['seed'],
            env_zero_cov_patch=ENV_KW['zero_cov_patch'])


def classify(ref, f, inp, res):
    """Label a finished reference track with the branch it exercised."""
    tags = []
    if res['status'] < 0:
        return ['gated']
    n = res['n']
    lon, lat, v = res['y'][0], res['y'][1], res['y'][2]
    if n == 1:
        tags.append('v0_le_4')
    if res['status'] == 0:
        tags.append('full')
    else:
        x0, y0, x1, y1 = f.basin.get_bounds()
        # the event is tested at the step end, which is ≥ the last emitted sample
        if v[-1] < 4.6 and n > 1:
            tags.append('dissipated')
        if min(lon[-1] - x0, x1 - lon[-1], lat[-1] - y0, y1 - lat[-1]) < 1.6:
            tags.append('basin_exit')
        if abs(lat[-1]) < 2.6:
            tags.append('low_lat')
    over_land = np.array([f._get_over_land(a, b) for a, b in zip(lon[::6], lat[::6])])
    if over_land.any():
        tags.append('land')
    shallow = np.array([f._get_current_bathymetry(a, b) > -80 for a, b in zip(lon[::6], lat[::6])])
    if (shallow & ~over_land).any():
        tags.append('shelf')
    if inp['month'] == 9 and ((lon > 300) & (lon < 312) & (lat > 22) & (lat < 30)).any():
        tags.append('chol_fail')
    if v.max() > 33:
        tags.append('hurricane')
    return tags or ['other']


def run_set(ref, env, basin, n_scan, seed, per_class, extra=(), mutate=None):
    """Scan random seeds, keep a class-balanced subset, return fixture arrays.
    mutate (make_golden_namelist.py): callable applied to the drawn storm inputs before they are handed to the reference."""
    S = synthetic.draw_storm_inputs(n_scan, basin, seed)
    if mutate is not None:
        mutate(S)
    fast = {}
    keep, counts = [], {}
    cands = [dict(lon=S['lon'][i], lat=S['lat'][i], month=int(S['month'][i]), v0=S['v0'][i],
                  m0=S['m0'][i], h_bl=S['h_bl'][i], phases=S['phases'][i]) for i in range(n_scan)]
    cands = list(extra) + cands
    for inp in cands:
        mo = inp['month'] - 1
        if mo not in fast:
            fast[mo] = H.build_coupled_fast(ref, env, basin, mo)
        f = fast[mo]
        res = H.gen_track(ref, f, inp['lon'], inp['lat'], inp['v0'], inp['m0'], inp['h_bl'], inp['phases'])
        tags = classify(ref, f, inp, res)
        want = inp.get('force', False) or any(counts.get(t, 0) < per_class for t in tags)
        if not want:
            continue
        for t in tags:
            counts[t] = counts.get(t, 0) + 1
        post = H.post_track(ref, f, res)
        keep.append((inp, res, post, tags))
    n = len(keep)
    out = dict(
        lon0=np.array([k[0]['lon'] for k in keep]), lat0=np.array([k[0]['lat'] for k in keep]),
        v0=np.array([k[0]['v0'] for k in keep]), m0=np.array([k[0]['m0'] for k in keep]),
        h_bl=np.array([k[0]['h_bl'] for k in keep]),
        month=np.array([k[0]['month'] for k in keep], dtype=np.int32),
        phases=np.array([k[0]['phases'] for k in keep]),
        status=np.array([k[1]['status'] for k in keep], dtype=np.int32),
        n_valid=np.array([k[1]['n'] for k in keep], dtype=np.int32),
        nfev=np.array([k[1]['nfev'] for k in keep], dtype=np.int32),
        is_tc=np.array([k[2]['is_tc'] for k in keep]), accepted=np.array([k[2]['accepted'] for k in keep]),
        tags=np.array([','.join(k[3]) for k in keep]),
        basin=np.array(basin),
    )
    traj = np.full((n, 4, N_STEPS), np.nan)
    envw = np.full((n, N_STEPS, 4), np.nan)
    vmax = np.full((n, N_STEPS), np.nan)
    for i, (inp, res, post, tags) in enumerate(keep):
        m = res['n']
        traj[i, :, :m] = res['y']
        envw[i, :m] = post['envw']
        vmax[i, :m] = post['vmax'][:m] if m else []
    out.update(traj=traj, envw=envw, vmax=vmax)
    # decision probe of the reference itself (ref_harness.gen_track), ragged: storm i owns [dec_off[i], dec_off[i+1])
    out['dec_off'] = np.concatenate([[0], np.cumsum([len(k[1]['dec']) for k in keep])]).astype(np.int64)
    out['dec'] = np.concatenate([k[1]['dec'] for k in keep]).astype(np.uint8)
    out['dec_t0'] = np.concatenate([k[1]['dec_t0'] for k in keep]).astype(np.float64)
    print('%s: kept %d of %d scanned; classes %s' % (basin, n, len(cands), counts))
    return out


def rhs_level(ref, env, basin, month, n, seed):
    """Random (t, y) → dydt, _env_winds, _calc_alpha, bilinear lookups."""
    rng = np.random.default_rng(seed)
    f = H.build_coupled_fast(ref, env, basin, month - 1)
    phases = rng.uniform(0, 1, (4, 15))
    with H.InjectedRandom(phases):
        f.Fs = f.gen_synthetic_f()
    from scipy.interpolate import interp1d
    f.Fs_i = interp1d(f.t_s, f.Fs, axis=1)
    f.h_bl = 1400.0
    x0, y0, x1, y1 = f.basin.get_bounds()
    t = rng.uniform(0, f.total_time, n)
    t[:8] = f.t_s[[0, 1, 2, 100, 359, 360, 17, 240]]          # exact knots incl. both ends
    lon = rng.uniform(x0 - 1.5, x1 + 1.5, n)                   # includes clamped points
    lat = rng.uniform(max(y0 - 1.5, -89.9), min(y1 + 1.5, 89.9), n)
    # snap a share of the points onto grid lines / grid nodes (interval-edge cases)
    lon[8:72] = np.round(lon[8:72])
    lat[40:104] = np.round(lat[40:104] * 4) / 4
    v = rng.uniform(0.5, 80, n)
    m = rng.uniform(0.0, 1.2, n)
    dydt = np.zeros((n, 4)); envw = np.zeros((n, 4)); alpha = np.zeros(n)
    look = np.zeros((n, 6))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for i in range(n):
            y = np.array([lon[i], lat[i], v[i], m[i]])
            dydt[i] = f.dydt(t[i], y)
            envw[i] = f._env_winds(lon[i], lat[i], t[i])
            sc = f._calc_steering_coefs(v[i])
            vb, _ = f._step_bam_track(lon[i], lat[i], t[i], sc)
            alpha[i] = f._calc_alpha(lon[i], lat[i], vb, v[i])
            look[i] = [f.f_vpot.ev(lon[i], lat[i]).item(), f.f_chi.ev(lon[i], lat[i]).item(),
                       f.f_mld.ev(lon[i], lat[i]).item(), f.f_strat.ev(lon[i], lat[i]).item(),
                       f.f_land.ev(lon[i], lat[i]).item(), f.f_bath.ev(lon[i], lat[i]).item()]
    return dict(basin=np.array(basin), month=np.int32(month), h_bl=1400.0, phases=phases,
                Fs=f.Fs, t=t, lon=lon, lat=lat, v=v, m=m, dydt=dydt, envw=envw,
                alpha=alpha, lookups=look)


def unit_level(ref):
    """Small known-answer vectors for host utilities."""
    rng = np.random.default_rng(7)
    out = {}
    # transform_global_field incl. the negative-longitude rotation (basins.py:57-75)
    lon = np.arange(0, 360, 2.5); lat = np.linspace(-90, 90, 73)
    X = rng.normal(size=(lat.size, lon.size))
    nl = ref.namelist
    nl.basin_bounds['XW'] = ['100W', '5N', '10W', '55N']       # west-positive box
    nl.basin_bounds['XE'] = ['20E', '40S', '120E', '10S']
    for bid in ('NA', 'SI', 'GL', 'XW', 'XE'):
        b = ref.basins.TC_Basin(bid)
        lo, la, Xb = b.transform_global_field(lon, lat, X)
        out['tgf_%s_lon' % bid] = lo; out['tgf_%s_lat' % bid] = la; out['tgf_%s_X' % bid] = Xb
    lon_pm = np.arange(-180, 180, 2.5)
    lo, la, Xb = ref.basins.TC_Basin('XE').transform_global_field(lon_pm, lat, X)
    out['tgf_pm_XE_lon'] = lo; out['tgf_pm_XE_X'] = Xb
    out['tgf_in_lon'] = lon; out['tgf_in_lat'] = lat; out['tgf_in_X'] = X; out['tgf_in_lon_pm'] = lon_pm
    del nl.basin_bounds['XW'], nl.basin_bounds['XE']
    # steering coefficients (coupled_fast.py:183-192)
    vv = np.array([0.0, 5.0, 12.3, 20.0, 33.0, 50.0, 70.0, 90.0, np.nan])
    f = ref.coupled_fast.Coupled_FAST.__new__(ref.coupled_fast.Coupled_FAST)
    out['steer_v'] = vv
    out['steer_coefs'] = np.array([f._calc_steering_coefs(x) for x in vv])
    # f_mInit and low-latitude filter (namelist.py:94, compute.py:164)
    rh = np.linspace(0, 1, 21)
    out['minit_rh'] = rh; out['minit_m'] = np.maximum(0, nl.f_mInit(rh))
    return out


def seeding_level(ref, env, basin, seed, year, cand0, n):
    """Transcription of the seed loop of util/compute.py:134-175 over the reference's own
    interpolators (mat.interp2_fx on the basin masks and rh_mid, Coupled_FAST.f_vpot), with
    np.random replaced by the per-candidate Philox stream of oracle/seeding.py:
        np.random.uniform(a, b, 1)[0] -> a + (b - a) * u      (NumPy's own formula)
        np.random.randint(1, 13)      -> int(u * 12) + 1
        np.random.randn(1)[0]         -> Box-Muller of two uniforms
    One candidate = one pass of the `while not seed_passed` body."""
    from oracle import seeding as OS
    nl = ref.namelist
    b = ref.basins.TC_Basin(basin)
    basin_ids = np.array(sorted([k for k in nl.basin_bounds if k != 'GL']))
    f_basins = {bid: ref.mat.interp2_fx(env.hlon, env.hlat, env.basin_masks[bid]) for bid in basin_ids}
    f_b = ref.mat.interp2_fx(env.hlon, env.hlat, env.basin_masks[basin])
    b_bounds = b.get_bounds()
    cpl_fast = [H.build_coupled_fast(ref, env, basin, mo) for mo in range(12)]
    m_init_fx = [ref.mat.interp2_fx(env.lon, env.lat, env.rh_mid[mo]) for mo in range(12)]
    out = {k: [] for k in ('lon', 'lat', 'month', 'basin_idx', 'flags', 'v0', 'm0', 'h_bl', 'redraw')}
    for cand in range(cand0, cand0 + n):
        lat_min = 3 if np.sign(b_bounds[1]) >= 0 else -45
        lat_max = 45 if np.sign(b_bounds[3]) >= 0 else -3
        y_min = np.sin(np.pi / 180 * lat_min)
        y_max = np.sin(np.pi / 180 * lat_max)
        u0, u1 = OS.uniform2(seed, year, cand, 0, 0)
        gen_lon = b_bounds[0] + (b_bounds[2] - b_bounds[0]) * float(u0)
        gen_lat = np.arcsin(y_min + (y_max - y_min) * float(u1)) * 180 / np.pi
        redraw = 0
        while f_b.ev(gen_lon, gen_lat) < 1e-2:
            redraw += 1
            u0, u1 = OS.uniform2(seed, year, cand, 0, redraw)
            gen_lon = b_bounds[0] + (b_bounds[2] - b_bounds[0]) * float(u0)
            gen_lat = b_bounds[1] + (b_bounds[3] - b_bounds[1]) * float(u1)
        um, ul = OS.uniform2(seed, year, cand, 1, 0)
        month_seed = min(int(float(um) * 12.0) + 1, 12)
        fast = cpl_fast[month_seed - 1]
        basin_val = np.zeros(len(basin_ids))
        for (b_idx, basin_id) in enumerate(basin_ids):
            basin_val[b_idx] = f_basins[basin_id].ev(gen_lon, gen_lat).item()
        basin_idx = np.argmax(basin_val)
        pi_gen = float(fast.f_vpot.ev(gen_lon, gen_lat).item())
        lat_vort_power = nl.lat_vort_power[basin_ids[basin_idx]]
        prob_lowlat = np.power(np.minimum(np.maximum((np.abs(gen_lat) - nl.lat_vort_fac) / 12.0, 0), 1), lat_vort_power)
        rand_lowlat = float(ul)
        flags = 0
        if (np.nanmax(basin_val) > 1e-3) and (rand_lowlat < prob_lowlat):
            flags |= 1                         # n_seeds[basin_idx, month_seed-1] += 1
            if (pi_gen > 35):
                flags |= 2                     # seed_passed = True
        n0, n1 = OS.uniform2(seed, year, cand, 1, 1)
        v_init = nl.seed_v_init_ms + np.sqrt(-2.0 * np.log(1.0 - float(n0))) * np.cos(2. * np.pi * float(n1))
        rh_init = float(m_init_fx[month_seed - 1].ev(gen_lon, gen_lat).item())
        m_init = np.maximum(0, nl.f_mInit(rh_init))
        for k, v in zip(out, (gen_lon, gen_lat, month_seed, basin_idx, flags, v_init, m_init,
                              nl.atm_bl_depth[basin_ids[basin_idx]], redraw)):
            out[k].append(v)
    res = {k: np.array(v) for k, v in out.items()}
    res.update(seed=np.uint64(seed), year=np.int32(year), cand0=np.int64(cand0), basin=np.array(basin))
    return res


def main():
    ref = H.import_reference()
    env = synthetic.make_env(**ENV_KW)
    warnings.simplefilter('ignore')

    # hand-built extras: v0 <= 4 (1-sample result) and a Cholesky-failure start
    ph = np.random.default_rng(99).uniform(0, 1, (4, 4, 15))
    extra_na = [
        dict(lon=310.0, lat=18.0, month=9, v0=3.5, m0=0.3, h_bl=1400.0, phases=ph[0], force=True),
        dict(lon=306.0, lat=26.0, month=9, v0=12.0, m0=0.5, h_bl=1400.0, phases=ph[1], force=True),
        dict(lon=301.5, lat=20.5, month=9, v0=14.0, m0=0.6, h_bl=1400.0, phases=ph[2], force=True),
        dict(lon=330.0, lat=3.2, month=9, v0=9.0, m0=0.5, h_bl=1400.0, phases=ph[3], force=True),
    ]
    sets = {
        'tracks_NA': run_set(ref, env, 'NA', 700, 101, per_class=7, extra=extra_na),
        'tracks_AU': run_set(ref, env, 'AU', 160, 202, per_class=3),
        'tracks_GL': run_set(ref, env, 'GL', 160, 303, per_class=3),
    }
    for name, d in sets.items():
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **d, **{'meta_' + k: v for k, v in META.items()})
    d = rhs_level(ref, env, 'NA', 9, 1500, 11)
    np.savez_compressed(os.path.join(HERE, 'rhs_NA.npz'), **d, **{'meta_' + k: v for k, v in META.items()})
    d = rhs_level(ref, env, 'SI', 2, 600, 12)
    np.savez_compressed(os.path.join(HERE, 'rhs_SI.npz'), **d, **{'meta_' + k: v for k, v in META.items()})
    np.savez_compressed(os.path.join(HERE, 'units.npz'), **unit_level(ref))
    for bsn, c0 in (('NA', 0), ('GL', 1 << 33), ('SI', 12345)):
        d = seeding_level(ref, env, bsn, 20250614, 2001, c0, 1500)
        print('seeds %s: counted %.3f passed %.3f mean redraws %.2f' % (bsn, (d['flags'] & 1).mean() if False else np.mean((d['flags'] & 1) != 0),
                                                                     np.mean((d['flags'] & 2) != 0), d['redraw'].mean()))
        np.savez_compressed(os.path.join(HERE, 'seeds_%s.npz' % bsn), **d, **{'meta_' + k: v for k, v in META.items()})
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith('.npz'):
            print('%-20s %8d bytes' % (fn, os.path.getsize(os.path.join(HERE, fn))))


if __name__ == '__main__':
    main()
