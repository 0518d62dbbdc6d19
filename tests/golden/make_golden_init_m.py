#!/usr/bin/env python3
"""Fixture for gen_track(clon, clat, v, m=None): the reference's `Coupled_FAST._init_m` (intensity/coupled_fast.py:153-173)
at random points, and a few whole tracks started without m (coupled_fast.py:258-261).  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_init_m.py
"""
import os
import sys
import warnings

import numpy as np
from scipy.interpolate import interp1d

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from tests.golden import ref_harness as H                         # noqa: E402
from tests.golden.make_golden import ENV_KW, META, N_STEPS       # noqa: E402
from tropical_cyclone_risk_amd import synthetic                  # noqa: E402


def main():
    ref = H.import_reference()
    env = synthetic.make_env(**ENV_KW)
    warnings.simplefilter('ignore')
    rng = np.random.default_rng(31)
    basin, month, h_bl = 'NA', 9, 1400.0
    f = H.build_coupled_fast(ref, env, basin, month - 1)
    f.h_bl = h_bl
    n = 240
    x0, y0, x1, y1 = f.basin.get_bounds()
    lon = rng.uniform(x0 + 0.5, x1 - 0.5, n); lat = rng.uniform(y0 + 0.5, y1 - 0.5, n); v = rng.uniform(3, 40, n)
    lon[:40] = np.round(lon[:40] * 4) / 4                         # the +-0.25 degree probes land on grid lines
    lat[20:60] = np.round(lat[20:60] * 4) / 4
    phases = rng.uniform(0, 1, (n, 4, 15))
    m0 = np.zeros(n); m1 = np.zeros(n)
    for i in range(n):
        with H.InjectedRandom(phases[i]):
            f.Fs = f.gen_synthetic_f()
        f.Fs_i = interp1d(f.t_s, f.Fs, axis=1)
        y = np.asarray([lon[i], lat[i], v[i]])
        m0[i] = f._init_m(y, 0)
        m1[i] = f._init_m(y, 2e-5)
    out = dict(basin=np.array(basin), month=np.int32(month), h_bl=h_bl, lon=lon, lat=lat, v=v, phases=phases,
               m_dvdt0=m0, m_dvdt2em5=m1, dvdt1=2e-5)
    print('_init_m: %d points, m in [%.3f, %.3f], %d clipped to 1, %d NaN' % (n, np.nanmin(m0), np.nanmax(m0), (m0 == 1).sum(), np.isnan(m0).sum()))
    # whole tracks started with m=None
    S = synthetic.draw_storm_inputs(12, basin, 555)
    fast = {}
    rows = []
    for i in range(12):
        mo = int(S['month'][i]) - 1
        if mo not in fast:
            fast[mo] = H.build_coupled_fast(ref, env, basin, mo)
        res = H.gen_track(ref, fast[mo], S['lon'][i], S['lat'][i], S['v0'][i], None, S['h_bl'][i], S['phases'][i])
        rows.append(res)
    traj = np.full((12, 4, N_STEPS), np.nan)
    for i, r in enumerate(rows):
        traj[i, :, :r['n']] = r['y']
    out.update(t_lon0=S['lon'], t_lat0=S['lat'], t_v0=S['v0'], t_h_bl=S['h_bl'], t_month=S['month'].astype(np.int32),
               t_phases=S['phases'], t_traj=traj, t_status=np.array([r['status'] for r in rows], np.int32),
               t_n_valid=np.array([r['n'] for r in rows], np.int32), t_nfev=np.array([r['nfev'] for r in rows], np.int32),
               t_dec_off=np.concatenate([[0], np.cumsum([len(r['dec']) for r in rows])]).astype(np.int64),
               t_dec=np.concatenate([r['dec'] for r in rows]).astype(np.uint8),
               t_dec_t0=np.concatenate([r['dec_t0'] for r in rows]).astype(np.float64))
    print('tracks with m=None: status', out['t_status'], 'n', out['t_n_valid'], 'm0', np.round(traj[:, 3, 0], 4))
    out.update({'meta_' + k: np.array(v) for k, v in META.items()})
    np.savez_compressed(os.path.join(HERE, 'init_m_NA.npz'), **out)


if __name__ == '__main__':
    main()
