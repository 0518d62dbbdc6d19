"""ORACLE (test infrastructure) — ctypes front-end of ``oracle/liborc.so``
(the plain-C restatement in ``tc_oracle.c``).  Build with ``make -C oracle``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this.  Field cropping reuses ``scipy_port.crop_to_box`` (the
oracle's own restatement of util/basins.py:57-75); nothing here imports the
product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .scipy_port import BASIN_BOUNDS, Params, crop_to_box

HERE = os.path.dirname(os.path.abspath(__file__))
_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int)


class _Grid(C.Structure):
    _fields_ = [('nlon', C.c_int), ('nlat', C.c_int), ('lon', _DP), ('lat', _DP)]


class _Env(C.Structure):
    _fields_ = [('wg', _Grid), ('tg', _Grid), ('hg', _Grid), ('bg', _Grid),
                ('mean', _DP * 4), ('cov', _DP * 10),
                ('vpot', _DP), ('chi', _DP), ('mld', _DP), ('strat', _DP),
                ('land', _DP), ('bathy', _DP), ('box', C.c_double * 4)]


class _Params(C.Structure):
    _fields_ = [('Ck', C.c_double), ('epsilon', C.c_double), ('kappa', C.c_double),
                ('u_beta', C.c_double), ('v_beta', C.c_double), ('T_Fs', C.c_double),
                ('y_alpha', C.c_double * 2), ('m_alpha', C.c_double * 2),
                ('alpha_max', C.c_double * 2), ('alpha_min', C.c_double * 2),
                ('dt_out', C.c_double), ('total_time', C.c_double), ('rtol', C.c_double),
                ('atol', C.c_double), ('max_step', C.c_double),
                ('v_thresh', C.c_double), ('v_2d_thresh', C.c_double),
                ('vmax_thresh', C.c_double), ('earth_R', C.c_double),
                ('n_series', C.c_int), ('n_steps', C.c_int),
                ('coupled_track', C.c_int), ('steering_coefs', C.c_double * 2)]


def build(force=False):
    so = os.path.join(HERE, 'liborc.so')
    src = os.path.join(HERE, 'tc_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', HERE, '-s', 'liborc.so'])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_bilinear.restype = C.c_double
        _lib.orc_bilinear.argtypes = [C.POINTER(_Grid), _DP, C.c_double, C.c_double]
        _lib.orc_integrate.restype = C.c_int
    return _lib


def _dp(a):
    return a.ctypes.data_as(_DP)


def _ip(a):
    return a.ctypes.data_as(_IP)


def c_params(prm=None):
    prm = prm or Params()
    p = _Params()
    for k in ('Ck', 'epsilon', 'kappa', 'u_beta', 'v_beta', 'T_Fs', 'dt_out', 'total_time',
              'rtol', 'atol', 'max_step', 'v_thresh', 'v_2d_thresh', 'vmax_thresh'):
        setattr(p, k, float(getattr(prm, k)))
    for k in ('y_alpha', 'm_alpha', 'alpha_max', 'alpha_min', 'steering_coefs'):
        setattr(p, k, (C.c_double * 2)(*getattr(prm, k)))
    p.coupled_track = 1 if prm.coupled_track else 0
    p.earth_R = 6.3781e6
    p.n_series = prm.N_series
    p.n_steps = prm.n_steps
    return p


_STATIC_CROPS = {}


class CMonthEnv:
    """Cropped [lat][lon] planes of one (basin, month) + the C struct pointing at them."""

    def __init__(self, env, basin, month0, bounds=None):
        b = BASIN_BOUNDS[basin] if bounds is None else bounds
        keep = []

        def crop(lon, lat, X, nan0=False):
            lo, la, Xb = crop_to_box(b, lon, lat, X)
            Xb = np.ascontiguousarray(np.nan_to_num(Xb) if nan0 else Xb, dtype=np.float64)
            keep.append(Xb)
            return np.ascontiguousarray(lo, dtype=np.float64), np.ascontiguousarray(la, dtype=np.float64), Xb

        e = _Env()
        wl, wa, _ = crop(env.wlon, env.wlat, env.wnd_mean[month0, 0])
        for i in range(4):
            e.mean[i] = _dp(crop(env.wlon, env.wlat, env.wnd_mean[month0, i], True)[2])
        for k in range(10):
            e.cov[k] = _dp(crop(env.wlon, env.wlat, env.wnd_cov[month0, k], True)[2])
        tl, ta, _ = crop(env.lon, env.lat, env.vpot[month0])
        for name in ('vpot', 'chi', 'mld', 'strat'):
            setattr(e, name, _dp(crop(env.lon, env.lat, getattr(env, name)[month0])[2]))
        # the static planes are the same for the twelve months of an (environment, basin): cropped once, shared (the
        # reference's 0.125-degree land.nc is 33 MB per float64 copy)
        key = (id(env.land), id(env.bathy), tuple(b))
        st = _STATIC_CROPS.get(key)
        if st is None or st[0] is not env.land or st[1] is not env.bathy:
            hl, ha, lb_ = crop(env.hlon, env.hlat, env.land)
            # the bathymetry is its own interpolator on its own grid (geo.py:9-20); env.blon / blat when it differs
            blon, blat = getattr(env, 'blon', None), getattr(env, 'blat', None)
            bl, ba, bb = crop(env.hlon if blon is None else blon, env.hlat if blat is None else blat, env.bathy)
            if len(_STATIC_CROPS) >= 4:
                _STATIC_CROPS.clear()
            st = _STATIC_CROPS[key] = (env.land, env.bathy, hl, ha, lb_, bl, ba, bb)
        _, _, hl, ha, lb_, bl, ba, bb = st
        keep += [lb_, bb]
        e.land = _dp(lb_)
        e.bathy = _dp(bb)
        for g, (lo, la) in zip((e.wg, e.tg, e.hg, e.bg), ((wl, wa), (tl, ta), (hl, ha), (bl, ba))):
            g.nlon, g.nlat, g.lon, g.lat = lo.size, la.size, _dp(lo), _dp(la)
        keep += [wl, wa, tl, ta, hl, ha, bl, ba]
        e.box = (C.c_double * 4)(*b)
        self.c = e
        self._keep = keep
        self.grids = dict(w=(wl, wa), t=(tl, ta), h=(hl, ha), b=(bl, ba))


def fourier_table(phases, prm=None):
    p = c_params(prm)
    Fs = np.zeros((4, p.n_steps))
    ph = np.ascontiguousarray(phases, dtype=np.float64)
    lib().orc_fourier_table(C.byref(p), _dp(ph), _dp(Fs))
    return Fs


def rhs_points(cme, Fs, h_bl, t, lon, lat, v, m, prm=None):
    p = c_params(prm)
    n = len(t)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (t, lon, lat, v, m)]
    Fs = np.ascontiguousarray(Fs, dtype=np.float64)
    dydt = np.zeros((n, 4)); envw = np.zeros((n, 4)); alpha = np.zeros(n)
    lib().orc_rhs_points(C.byref(cme.c), C.byref(p), _dp(Fs), C.c_double(h_bl), C.c_int(n),
                         *[_dp(a) for a in arrs], _dp(dydt), _dp(envw), _dp(alpha))
    return dydt, envw, alpha


def init_m(cme, Fs, h_bl, lon, lat, v, dvdt=0.0, prm=None):
    """Coupled_FAST._init_m at points of one month environment with one forcing table Fs [4, n_steps]."""
    p = c_params(prm)
    Fs = np.ascontiguousarray(Fs, dtype=np.float64)
    f = lib().orc_init_m
    f.restype = C.c_double
    f.argtypes = [C.c_void_p, C.c_void_p, _DP, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
    return np.array([f(C.addressof(cme.c), C.addressof(p), _dp(Fs), float(h_bl), float(a), float(b), float(c), float(dvdt))
                     for a, b, c in zip(lon, lat, v)])


def bilinear(cme, which, plane_name, lon, lat):
    g = {'w': cme.c.wg, 't': cme.c.tg, 'h': cme.c.hg if plane_name != 'bathy' else cme.c.bg}[which]
    plane = getattr(cme.c, plane_name)
    return np.array([lib().orc_bilinear(C.byref(g), plane, float(a), float(b)) for a, b in zip(lon, lat)])


PROBE_CAP = 1024          # RHS evaluations recorded per storm by the decision probe (max observed ~400)


class Ensemble:
    """The twelve month field sets of one (environment, basin), cropped once and kept — what the list of
    `cpl_fast` objects is in run_tracks (util/compute.py:101-121).  `run(storms)` has the contract of
    scipy_port.run_ensemble plus per-storm step counters."""

    def __init__(self, env, basin, prm=None, bounds=None):
        self.env, self.basin, self.bounds = env, basin, bounds
        self.prm = prm or Params()
        self.p = c_params(self.prm)
        self.cmes = {}
        self.envs = (C.POINTER(_Env) * 12)()

    def _need(self, months):
        for mo in np.unique(months):
            if int(mo) not in self.cmes:
                self.cmes[int(mo)] = CMonthEnv(self.env, self.basin, int(mo) - 1, self.bounds)
                self.envs[int(mo) - 1] = C.pointer(self.cmes[int(mo)].c)

    def run(self, storms, post=True, probe=False, force=None):
        """force: [n, cap] uint8 decision probes of another implementation (decision-forced replay, see
        tc_oracle.c orc_storm.force): rounding-sensitive `land == 1` evaluations take that side's decision.
        Adds 'overridden' and 'hard_mismatch' [n] to the result."""
        p = self.p
        n = len(storms['lon'])
        ns = p.n_steps
        months = np.ascontiguousarray(storms['month'], dtype=np.int32)
        self._need(months)
        f64 = lambda k: np.ascontiguousarray(storms[k], dtype=np.float64)
        lon0, lat0, v0, m0, h_bl, ph = f64('lon'), f64('lat'), f64('v0'), f64('m0'), f64('h_bl'), f64('phases')
        traj = np.empty((n, 4, ns)); envw = np.empty((n, ns, 4)); vmax = np.empty((n, ns))
        n_valid = np.zeros(n, np.int32); status = np.zeros(n, np.int32)
        counters = np.zeros((n, 5), np.int32); flags = np.zeros((n, 2), np.int32)
        dec = np.full((n, PROBE_CAP), 0xff, np.uint8) if probe else None
        dec_t0 = np.full((n, PROBE_CAP), np.nan) if probe else None
        finfo = np.zeros((n, 2), np.int32)
        if force is not None:
            force = np.ascontiguousarray(force, dtype=np.uint8)
            assert force.ndim == 2 and force.shape[0] == n
        lib().orc_run_ensemble_forced(self.envs, C.byref(p), C.c_int(n), _dp(lon0), _dp(lat0), _dp(v0), _dp(m0),
                                      _dp(h_bl), _ip(months), _dp(ph), _dp(traj), _dp(envw), _dp(vmax),
                                      _ip(n_valid), _ip(status), _ip(counters), _ip(flags), C.c_int(1 if post else 0),
                                      dec.ctypes.data_as(C.c_void_p) if probe else None,
                                      dec_t0.ctypes.data_as(C.c_void_p) if probe else None, C.c_int(PROBE_CAP),
                                      force.ctypes.data_as(C.c_void_p) if force is not None else None,
                                      C.c_int(force.shape[1] if force is not None else 0), _ip(finfo))
        if not post:
            envw[:] = np.nan; vmax[:] = np.nan
        extra = dict(dec=dec, dec_t0=dec_t0) if probe else {}
        if force is not None:
            extra.update(overridden=finfo[:, 0].copy(), hard_mismatch=finfo[:, 1].copy())
        return dict(traj=traj, envw=envw, vmax=vmax, n_valid=n_valid, status=status, **extra,
                    nfev=counters[:, 0].copy(), n_accept=counters[:, 1].copy(),
                    n_reject=counters[:, 2].copy(), anomaly=counters[:, 3].copy(),
                    flicker=counters[:, 4].copy(),
                    is_tc=flags[:, 0].astype(bool), accepted=flags[:, 1].astype(bool))


def run_ensemble(env, basin, storms, prm=None, post=True, bounds=None, probe=False, force=None):
    """Same contract as scipy_port.run_ensemble, plus per-storm step counters.  probe=True adds
    'dec' [n, PROBE_CAP] uint8 (per RHS evaluation: bit0 `land == 1`, bit1 PI != 0, bit2 land within
    1e-12 of 1; 0xff = not evaluated) and 'dec_t0' (start time of the step attempt of that evaluation)."""
    return Ensemble(env, basin, prm, bounds).run(storms, post=post, probe=probe, force=force)


def replayer(env, basin, storms, prm=None, bounds=None, ensemble=None):
    """Callback for parity.check_tracks: `replay(idx, dec_force)` re-runs storms `idx` with the other
    implementation's decision sequence forced at the rounding-sensitive evaluations.

    `replay.twin_storms(idx)` is the yardstick for the far tail of a comparison (parity.check_tracks' bound on EVERY sample):
    the oracle's own response ON THOSE STORMS to one input changed by one ulp — three twins (v0 up, v0 down, lon up), each
    walking the base run's decision sequence (forced like the replay); per output and storm the maximum over the twins whose
    decisions and counters stay the same (NaN if none does)."""
    ens = ensemble or Ensemble(env, basin, prm, bounds)

    def replay(idx, dec_force):
        sub = {k: np.asarray(v)[idx] for k, v in storms.items()}
        return ens.run(sub, post=True, probe=True, force=dec_force)

    def twin_storms(idx, dec=None):
        """dec [len(idx), cap] (optional): the decision sequences the compared runs walked on those storms; base and twins
        are then run with them forced at their rounding-sensitive evaluations, like the replay itself — a flicker-exposed
        storm would otherwise flip a `land == 1` decision under the one-ulp change and show the jump between two branches
        (degrees) instead of its sensitivity along one."""
        from . import parity as P
        idx = np.atleast_1d(np.asarray(idx, dtype=np.int64))
        sub = {k: np.asarray(v)[idx] for k, v in storms.items()}
        base = ens.run(sub, post=True, probe=True, force=dec)
        out = {name: np.full(len(idx), np.nan) for name in ('traj', 'envw', 'vmax')}
        for key, sgn in (('v0', 1.0), ('v0', -1.0), ('lon', 1.0)):
            pert = dict(sub)
            pert[key] = np.nextafter(np.asarray(sub[key], dtype=np.float64), sgn * np.inf)
            twin = ens.run(pert, post=True, probe=True, force=base['dec'])
            same = ((P.first_divergence(twin['dec'], base['dec']) < 0) & (twin['nfev'] == base['nfev']) &
                    (twin['n_valid'] == base['n_valid']) & (np.asarray(twin['hard_mismatch']) == 0))
            for name in out:
                a, b = np.asarray(base[name]), np.asarray(twin[name])
                d = np.abs(np.nan_to_num(a) - np.nan_to_num(b)).reshape(len(idx), -1).max(axis=1)
                out[name] = np.fmax(out[name], np.where(same, d, np.nan))
        return out
    replay.twin_storms = twin_storms
    return replay
