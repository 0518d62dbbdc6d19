"""ORACLE (test infrastructure, not product code) — NumPy/SciPy restatement of the
reference hot path, one storm at a time, with the *same library calls* the
reference makes: ``scipy.integrate.solve_ivp`` (RK45, rtol 1e-3, atol 1e-6,
max_step 86400 s, hourly ``t_eval``, one terminal event),
``RectBivariateSpline(kx=1, ky=1).ev``, ``interp1d`` and ``numpy.linalg.cholesky``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  It is the timed "NumPy/SciPy CPU baseline"
(``cpu_baseline.kind = "port"``).

Parity pin: checked against ``tests/golden/*.npz`` which were produced by running
the reference's own code (`tests/golden/make_golden.py`) under SciPy 1.15.3 /
NumPy 2.2.6 — see ``tests/test_oracle_golden.py``.

Reference lines restated (all under /root/reference):
  field samplers        intensity/coupled_fast.py:217-225, track/bam_track.py:72-91,
                        intensity/geo.py:9-34, util/basins.py:57-75
  Fourier forcing       track/bam_track.py:23-31, 111-113; coupled_fast.py:234-235
  env winds             track/bam_track.py:93-128
  beta-advection        track/bam_track.py:131-144; coupled_fast.py:183-192
  intensity RHS         coupled_fast.py:35-58, 65-94, 115-131, 141-150, 175-180, 196-207
  gate/event/solve_ivp  coupled_fast.py:229-267
  accept + post-step    util/compute.py:178-209; wind/tc_wind.py:6-21; util/sphere.py:15-83
"""
import warnings
from dataclasses import dataclass

import numpy as np
from scipy.integrate import solve_ivp
from scipy.interpolate import RectBivariateSpline, interp1d

EARTH_R = 6.3781e6
KT_PER_MS = 1.94384
TRIL = [(i, j) for i in range(4) for j in range(i + 1)]

BASIN_BOUNDS = {'EP': (180., 0., 290., 60.), 'NA': (260., 0., 360., 60.),
                'NI': (30., 0., 100., 50.), 'SI': (20., -45., 100., 0.),
                'AU': (100., -45., 180., 0.), 'SP': (180., -45., 250., 0.),
                'WP': (100., 0., 180., 60.), 'GL': (0., -90., 360., 90.)}


@dataclass
class Params:
    """Scalars of namelist.py:56-94 + coupled_fast.py:23-27 the path reads."""
    Ck: float = 1.2e-3
    epsilon: float = 0.33
    kappa: float = 0.1
    u_beta: float = -1.0
    v_beta: float = 2.5
    T_Fs: float = 20 * 86400.0
    N_series: int = 15
    y_alpha: tuple = (0.17, 0.83)
    m_alpha: tuple = (0.0025, -0.0025)
    alpha_max: tuple = (0.41, 0.78)
    alpha_min: tuple = (0.22, 0.59)
    coupled_track: bool = True            # namelist.py:72
    steering_coefs: tuple = (0.2, 0.8)    # namelist.py:71 (used when not coupled_track)
    dt_out: float = 3600.0
    total_time: float = 15 * 86400.0
    rtol: float = 1e-3
    atol: float = 1e-6
    max_step: float = 86400.0
    v_thresh: float = 15.0
    v_2d_thresh: float = 6.5
    vmax_thresh: float = 18.0

    @property
    def beta(self):
        return 1 - self.epsilon - self.kappa

    @property
    def n_steps(self):
        return int(self.total_time / self.dt_out) + 1

    @property
    def t_s(self):
        return np.linspace(0, self.total_time, self.n_steps)


def crop_to_box(bounds, lon, lat, X):
    """Longitude rotation + crop of util/basins.py:57-75 for a bounds tuple."""
    x0, y0, x1, y1 = bounds
    lon = np.asarray(lon); lat = np.asarray(lat)
    if lon[0] >= -1e-5 and (x0 < 0 or x1 < 0):
        hi = lon >= 180 - 1e-5
        X = np.concatenate((X[:, hi], X[:, ~hi]), axis=1)
        lon = np.hstack((lon[hi] - 360, lon[~hi]))
    elif (lon < 0).any() and x0 >= 0:
        neg = lon < -1e-5
        X = np.concatenate((X[:, ~neg], X[:, neg]), axis=1)
        lon = np.hstack((lon[~neg], lon[neg] + 360))
    mx = (lon <= x1 + 1e-5) & (lon >= x0 - 1e-5)
    my = (lat >= y0 - 1e-5) & (lat <= y1 + 1e-5)
    return lon[mx], lat[my], X[my][:, mx]


def _spline(bounds, lon, lat, X, nan_to_num=False):
    lo, la, Xb = crop_to_box(bounds, lon, lat, X)
    if nan_to_num:
        Xb = np.nan_to_num(Xb)
    return RectBivariateSpline(lo, la, Xb.T, kx=1, ky=1)


class MonthEnv:
    """The 20 bilinear samplers of one (basin, month) field set."""

    def __init__(self, env, basin, month0, bounds=None):
        self.bounds = b = BASIN_BOUNDS[basin] if bounds is None else bounds
        self.mean = [_spline(b, env.wlon, env.wlat, env.wnd_mean[month0, i], True) for i in range(4)]
        self.cov = {ij: _spline(b, env.wlon, env.wlat, env.wnd_cov[month0, k], True)
                    for k, ij in enumerate(TRIL)}
        self.vpot = _spline(b, env.lon, env.lat, env.vpot[month0])
        self.chi = _spline(b, env.lon, env.lat, env.chi[month0])
        self.mld = _spline(b, env.lon, env.lat, env.mld[month0])
        self.strat = _spline(b, env.lon, env.lat, env.strat[month0])
        self.land = _spline(b, env.hlon, env.hlat, env.land)
        self.bathy = _spline(b, getattr(env, 'blon', None) if getattr(env, 'blon', None) is not None else env.hlon,
                             getattr(env, 'blat', None) if getattr(env, 'blat', None) is not None else env.hlat, env.bathy)    # its own grid (geo.py:9-20)

    def inside(self, lon, lat, dx):
        x0, y0, x1, y1 = self.bounds
        return (x0 + dx) < lon < (x1 - dx) and (y0 + dx) < lat < (y1 - dx)


def _at(spl, lon, lat):
    return spl.ev(lon, lat).flatten()[0]


def fourier_table(phases, prm):
    """Fs[4, n_steps]: sqrt(2/Σn⁻³)·Σ n^-1.5 sin(2π(n t/T + x_n)) (bam_track.py:23-31)."""
    t = prm.t_s
    N = prm.N_series
    n = np.linspace(1, N, N)
    amp = np.sqrt(2 / np.sum(np.power(n, -3)))
    wgt = np.tile(np.power(n, -1.5), (np.size(t), 1)).T
    fs = np.zeros((4, t.size))
    for i in range(4):
        x = np.tile(np.asarray(phases[i], dtype=float).reshape(N, 1), (1, t.size))
        fs[i] = amp * np.sum(np.multiply(wgt, np.sin(2. * np.pi * (np.outer(n, t) / prm.T_Fs + x))), axis=0)
    return fs


class Storm:
    """RHS closure for one storm (frozen month environment + its forcing table)."""

    def __init__(self, me, prm, h_bl, Fs):
        self.me, self.prm, self.h_bl = me, prm, h_bl
        self.Fs = Fs
        self.Fs_i = interp1d(prm.t_s, Fs, axis=1)

    # -- track/bam_track.py:116-128
    def env_winds(self, lon, lat, t):
        if np.isnan(lon) or np.isnan(t):
            return np.zeros(4)
        me = self.me
        mu = np.zeros(4)
        C = np.zeros((4, 4))
        for i in range(4):
            mu[i] = me.mean[i].ev(lon, lat)
            for j in range(i + 1):
                C[i, j] = me.cov[(i, j)].ev(lon, lat)
        for i in range(4):
            for j in range(i, 4):
                C[i, j] = C[j, i]
        try:
            A = np.linalg.cholesky(C)
        except np.linalg.LinAlgError:
            return np.zeros(4)
        return mu + np.matmul(A, self.Fs_i(t))

    # -- coupled_fast.py:183-192
    def steering(self, v):
        p = self.prm
        if not p.coupled_track:
            return np.array(p.steering_coefs)
        a = (v * KT_PER_MS) * np.array(p.m_alpha) + np.array(p.y_alpha)
        a = np.maximum(np.minimum(a, p.alpha_max), p.alpha_min)
        if np.any(np.isnan(a)):
            a = np.array(p.y_alpha)
        return a

    # -- bam_track.py:131-144
    def translation(self, lon, lat, t, coefs):
        if np.abs(lat) >= 80:
            return np.zeros(2), np.zeros(4)
        w = self.env_winds(lon, lat, t)
        cl = np.cos(np.deg2rad(lat))
        vb = np.zeros(2)
        vb[0] = np.dot(w[[0, 2]], coefs) + self.prm.u_beta * cl
        vb[1] = np.dot(w[[1, 3]], coefs) + np.sign(lat) * self.prm.v_beta * cl
        return vb, w

    # -- coupled_fast.py:35-58
    def vpot_here(self, lon, lat):
        if _at(self.me.land, lon, lat) == 1:
            return 0
        return _at(self.me.vpot, lon, lat)

    # -- coupled_fast.py:65-94
    def ocean_alpha(self, lon, lat, vb, v):
        me = self.me
        h_m = _at(me.mld, lon, lat)
        gam = _at(me.strat, lon, lat)
        vp = self.vpot_here(lon, lat)
        uT = np.linalg.norm(vb)
        bathy = _at(me.bathy, lon, lat)
        if bathy >= 0 or -h_m <= bathy or gam == 0:
            return 1
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', category=RuntimeWarning)
            z = 0.01 * (gam ** -0.4) * h_m * uT * vp / v
        return 1 - 0.87 * np.exp(-np.clip(z, 0, 100))

    def shear(self, w):
        return np.linalg.norm(np.array([w[0] - w[2], w[1] - w[3]]))

    # -- coupled_fast.py:196-207
    def rhs(self, t, y):
        p = self.prm
        lon, lat, v, m = y
        vb, w = self.translation(lon, lat, t, self.steering(v))
        dlon = vb[0] / EARTH_R * 180. / np.pi / (np.cos(lat * np.pi / 180.))
        dlat = vb[1] / EARTH_R * 180. / np.pi
        vp = self.vpot_here(lon, lat)
        a = self.ocean_alpha(lon, lat, vb, v)
        gamma = p.epsilon + a * p.kappa
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', category=RuntimeWarning)
            dv = 0.5 * p.Ck / self.h_bl * (a * p.beta * (vp ** 2) * (m ** 3) - (1 - gamma * (m ** 3)) * (v ** 2))
            if np.isnan(dv):
                dv = 0
            venti = self.shear(w) * _at(self.me.chi, lon, lat)
            dm = 0.5 * p.Ck / self.h_bl * ((1 - m) * v - venti * m)
        return np.array([dlon, dlat, dv, dm])


def integrate_storm(me, prm, lon, lat, v0, m0, h_bl, phases):
    """coupled_fast.py:229-267.  status: -1 gated, 0 ran to 15 d, 1 terminal event."""
    st = Storm(me, prm, h_bl, fourier_table(phases, prm))
    S = st.shear(st.env_winds(lon, lat, 0))
    vp = st.vpot_here(lon, lat)
    chi = _at(me.chi, lon, lat)
    if vp > 0 and S * chi / vp >= 1:
        return dict(status=-1, n=0, t=np.zeros(0), y=np.zeros((4, 0)), nfev=0, storm=st)

    def stop(t, y):
        if not me.inside(y[0], y[1], 1):
            return 0
        if np.abs(y[1]) <= 2:
            return 0
        return np.maximum(0, y[2] - 4)
    stop.terminal = True

    res = solve_ivp(st.rhs, (0, prm.total_time), np.asarray([lon, lat, v0, m0]),
                    t_eval=np.linspace(0, prm.total_time, prm.n_steps), events=stop,
                    max_step=prm.max_step, rtol=prm.rtol, atol=prm.atol)
    return dict(status=int(res.status), n=int(res.t.size), t=res.t, y=res.y,
                nfev=int(res.nfev), storm=st)


# ---------------------------------------------------------------- post-step
def haversine_km(lon1, lat1, lon2, lat2):
    lon1, lat1, lon2, lat2 = map(np.deg2rad, (lon1, lat1, lon2, lat2))
    a = (np.square(np.sin((lat2 - lat1) / 2)) +
         np.cos(lat1) * np.cos(lat2) * np.square(np.sin((lon2 - lon1) / 2)))
    return (EARTH_R / 1000.) * 2 * np.arcsin(np.sqrt(a))


def translation_speed(lon, lat, dt):
    """util/sphere.py:58-83 for a 1-D track."""
    if len(lon) <= 1:
        return np.full(1, np.nan), np.full(1, np.nan)
    el = np.hstack((2 * lon[0] - lon[1], lon, 2 * lon[-1] - lon[-2]))
    ep = np.hstack((2 * lat[0] - lat[1], lat, 2 * lat[-1] - lat[-2]))
    dx = 0.5 * np.sign(el[2:] - el[:-2]) * haversine_km(el[2:], ep[1:-1], el[:-2], ep[1:-1])
    dy = 0.5 * np.sign(ep[2:] - ep[:-2]) * haversine_km(el[1:-1], ep[2:], el[1:-1], ep[:-2])
    return dx * 1000. / dt, dy * 1000. / dt


def max_wind(lon, lat, dt, v, envw):
    """wind/tc_wind.py:6-21."""
    ut, vt = translation_speed(lon, lat, dt)
    G = np.minimum(1., 0.8 + 0.35 * (1. + np.tanh((lat - 35.) / 10.)))
    Ui = G * ut + 0.1 * (envw[:, 0] - envw[:, 2]) * v / 15.
    Vi = G * vt + 0.1 * (envw[:, 1] - envw[:, 3]) * v / 15.
    mag = np.sqrt(np.power(Ui, 2) + np.power(Vi, 2))
    fac = np.minimum(1, (v * 0.50) / mag)
    th = np.arctan2(-Ui, Vi)
    ug = v * -np.sin(th) + Ui * fac
    vg = v * np.cos(th) + Vi * fac
    return np.sqrt(np.power(ug, 2) + np.power(vg, 2))


def post_storm(prm, res, tc_only=False):
    """util/compute.py:178-209 for one candidate.  tc_only: env winds and vmax only `if is_tc:`, as the reference
    computes them (compute.py:190-204); the default computes them for every candidate (what the parity tests compare)."""
    n = res['n']
    out = dict(is_tc=False, accepted=False, envw=np.zeros((n, 4)), vmax=np.full(n, np.nan))
    if res['status'] < 0:
        return out
    st = res['storm']
    lon, lat, v = res['y'][0], res['y'][1], res['y'][2]
    v2d = np.interp(2 * 86400, res['t'], v)
    out['is_tc'] = bool(np.any(v >= prm.v_thresh) and v2d >= prm.v_2d_thresh)
    if tc_only and not out['is_tc']:
        return out
    t_s = prm.t_s
    for i in range(n):
        out['envw'][i] = st.env_winds(lon[i], lat[i], t_s[i])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out['vmax'] = np.asarray(max_wind(lon, lat, prm.dt_out, v, out['envw'])).reshape(-1)
        out['accepted'] = bool(out['is_tc'] and np.nanmax(out['vmax']) >= prm.vmax_thresh)
    return out


def run_ensemble(env, basin, storms, prm=None, post=True, index=None, cache=None):
    """Integrate (and post-process) a set of storms; returns padded arrays + counters.

    storms: dict of arrays as produced by ``synthetic.draw_storm_inputs``.  cache: a dict that keeps the month
    environments (the twelve `cpl_fast` objects of run_tracks) between calls.
    """
    prm = prm or Params()
    idx = range(len(storms['lon'])) if index is None else index
    idx = list(idx)
    ns = prm.n_steps
    out = dict(traj=np.full((len(idx), 4, ns), np.nan), envw=np.full((len(idx), ns, 4), np.nan),
               vmax=np.full((len(idx), ns), np.nan), n_valid=np.zeros(len(idx), np.int32),
               status=np.zeros(len(idx), np.int32), nfev=np.zeros(len(idx), np.int32),
               is_tc=np.zeros(len(idx), bool), accepted=np.zeros(len(idx), bool))
    cache = {} if cache is None else cache
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for k, i in enumerate(idx):
            mo = int(storms['month'][i]) - 1
            if mo not in cache:
                cache[mo] = MonthEnv(env, basin, mo)
            r = integrate_storm(cache[mo], prm, storms['lon'][i], storms['lat'][i], storms['v0'][i],
                                storms['m0'][i], storms['h_bl'][i], storms['phases'][i])
            n = r['n']
            out['status'][k] = r['status']; out['n_valid'][k] = n; out['nfev'][k] = r['nfev']
            out['traj'][k, :, :n] = r['y']
            if post:                 # True: every candidate; 'tc': only candidates that pass accept test 1 (the reference)
                p = post_storm(prm, r, tc_only=(post == 'tc'))
                out['envw'][k, :n] = p['envw']; out['vmax'][k, :n] = p['vmax'][:n] if n else []
                out['is_tc'][k] = p['is_tc']; out['accepted'][k] = p['accepted']
    return out
