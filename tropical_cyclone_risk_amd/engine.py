"""Host driver of the HIP storm integrator: one :class:`TCEngine` per (GPU, basin).

It plays the role the list of twelve ``Coupled_FAST`` objects plays in the
reference's ``run_tracks`` (`util/compute.py:101-121`): fields are cropped to the
basin exactly as ``TC_Basin.transform_global_field`` does, staged once into HBM
month slots, and whole batches of storms are then integrated per call through
the C ABI of ``include/tcrisk_hip.h``.
"""
import ctypes as C

import numpy as np

from . import _lib, constants
from . import namelist as default_namelist
from .basins import BASIN_IDS, TC_Basin, basin_ids

TRACK_F64 = ('lon', 'lat', 'v', 'm', 'vmax')
TRACK_I32 = ('n_valid', 'status', 'flags', 'nfev', 'n_accept', 'n_reject')


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(_lib.DP)


def params_from_namelist(nl, basin, n_series=None):
    """Collect the scalars the kernels read from a namelist module (namelist.py:56-94)."""
    p = _lib.Params()
    p.Ck, p.epsilon, p.kappa = nl.Ck, 0.33, 0.1                   # coupled_fast.py:23-27
    p.u_beta, p.v_beta = nl.u_beta, nl.v_beta
    p.T_Fs = nl.T_days * 24 * 60 * 60                             # bam_track.py:56
    for k in ('y_alpha', 'm_alpha', 'alpha_max', 'alpha_min', 'steering_coefs'):
        setattr(p, k, (C.c_double * 2)(*getattr(nl, k)))
    p.dt_out = float(nl.output_interval_s)
    p.total_time = float(nl.total_track_time_days * 24 * 60 * 60)
    p.rtol = getattr(nl, 'gpu_rtol', 1e-3)
    p.atol = getattr(nl, 'gpu_atol', 1e-6)
    p.max_step = getattr(nl, 'gpu_max_step_s', 86400.0)           # coupled_fast.py:266
    p.v_thresh = nl.seed_v_threshold_ms
    p.v_2d_thresh = nl.seed_v_2d_threshold_ms
    p.vmax_thresh = nl.seed_vmax_threshold_ms
    p.v_dissipate = 4.0                                           # coupled_fast.py:255
    p.earth_R = constants.earth_R
    if basin_ids(nl) != BASIN_IDS:
        raise ValueError('namelist.basin_bounds must define exactly the basins %s (+ GL): the device tables hold one entry '
                         'per reference basin in that order; got %s' % (BASIN_IDS, basin_ids(nl)))
    p.box = (C.c_double * 4)(*TC_Basin(basin, nl).get_bounds())
    N = int(n_series or getattr(nl, 'gpu_N_series', 15))          # bam_track.py:112
    n = np.linspace(1, N, N)
    p.fs_amp = float(np.sqrt(2 / np.sum(np.power(n, -3))))        # bam_track.py:28
    wgt = np.zeros(_lib.TCR_MAX_SERIES)
    wgt[:N] = np.power(n, -1.5)
    p.fs_wgt = (C.c_double * _lib.TCR_MAX_SERIES)(*wgt)
    p.n_series = N
    p.n_steps = int(p.total_time / p.dt_out) + 1                  # bam_track.py:54
    p.coupled_track = 1 if nl.coupled_track else 0
    p.max_rk_steps = int(getattr(nl, 'gpu_max_rk_steps', 64))
    p.seed_v_init = nl.seed_v_init_ms
    p.pi_gate = 35.0                                              # compute.py:168
    p.lat_vort_fac = nl.lat_vort_fac
    p.lat_vort_power = (C.c_double * 7)(*[nl.lat_vort_power[b] for b in BASIN_IDS])
    p.atm_bl_depth = (C.c_double * 7)(*[nl.atm_bl_depth[b] for b in BASIN_IDS])
    # f_mInit = a / (1 + exp(-(rh - b) * c)) + d; verified against nl.f_mInit below
    p.minit_a, p.minit_b, p.minit_c, p.minit_d = 0.20, 0.55, 10.0, 0.125
    rh = np.linspace(0, 1, 11)
    mine = p.minit_a / (1 + np.exp(-(rh - p.minit_b) * p.minit_c)) + p.minit_d
    if not np.allclose(mine, nl.f_mInit(rh), rtol=0, atol=1e-14):
        raise ValueError('namelist.f_mInit is not the logistic form the device seeding implements; '
                         'seed on the host and call integrate() instead')
    return p


class TCEngine:
    """Owns one library context: staged fields + parameters for one basin on one GPU."""

    def __init__(self, basin, device=0, nl=None):
        self.nl = nl or default_namelist
        self.basin = TC_Basin(basin, self.nl)
        self.L = _lib.lib()
        h = C.c_void_p()
        if self.L.tcr_ctx_create(int(device), C.byref(h)) != 0:
            raise _lib.TcrError(self.L.tcr_last_error(None).decode())
        self.h = h
        self.device = int(device)
        self.params = params_from_namelist(self.nl, basin)
        self._ck(self.L.tcr_params_set(self.h, C.byref(self.params)))
        self.n_steps = int(self.params.n_steps)
        self.n_series = int(self.params.n_series)
        self.t_s = np.linspace(0, self.params.total_time, self.n_steps)
        self._keep = []

    # ------------------------------------------------------------------ utils
    def _ck(self, rc):
        if rc != 0:
            raise _lib.TcrError(self.L.tcr_last_error(self.h).decode())

    def grow_step_record(self, limit=4096):
        """Double tcr_params.max_rk_steps (accepted RK45 steps recorded per storm; the workspaces — n x max_rk_steps x
        ~400 B — follow at the next integrate).  Returns False when `limit` (the ABI's 4096; the reference's solve_ivp is unbounded;
        no storm of the 40-year config 3 needs more than 128; `compute.GpuRound.grow` stops earlier when HBM would not hold it) is reached.  The value persists for the engine's later rounds and years."""
        cur = int(self.params.max_rk_steps) or 64
        if cur >= limit:
            return False
        self.params.max_rk_steps = min(limit, 2 * cur)
        self._ck(self.L.tcr_params_set(self.h, C.byref(self.params)))
        return True

    def close(self):
        if getattr(self, 'h', None):
            self.L.tcr_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _grid(self, lon, lat):
        lon, lat = _f64(lon), _f64(lat)
        g = _lib.Grid(lon.size, lat.size, _dp(lon), _dp(lat))
        g._keep = (lon, lat)
        return g

    # --------------------------------------------------------------- staging
    def stage_static(self, hlon, hlat, land, bathy, blon=None, blat=None):
        """geo.read_land / read_bathy (intensity/geo.py:9-34): crop to the basin, stage.  The two are independent
        interpolators in the reference; (blon, blat) is the bathymetry's own grid when it differs from the land mask's."""
        lo, la, land_b = self.basin.transform_global_field(hlon, hlat, land)
        blo, bla, bathy_b = self.basin.transform_global_field(hlon if blon is None else blon, hlat if blat is None else blat, bathy)
        g, gb = self._grid(lo, la), self._grid(blo, bla)
        # (the reference passes whatever dtype the files hold — its land.nc is int8 — to RectBivariateSpline, which computes
        # in float64; the library decides from the values whether a narrow storage is exact: tcr_static_store)
        land_b, bathy_b = _f64(land_b), _f64(bathy_b)
        self._ck(self.L.tcr_static_store(self.h, 1 if str(getattr(self.nl, 'gpu_static_store', 'auto')) == 'f64' else 0))
        self._ck(self.L.tcr_static_upload2(self.h, C.byref(g), _dp(land_b), C.byref(gb), _dp(bathy_b)))
        self._staged_static = None          # (stage_env's identity shortcut: whatever it remembered is no longer what is staged)

    def static_info(self):
        """(storage mode, bytes) of the staged land / bathymetry planes: 'f64' | 'f64_split' | 'pack16' | 'u8_f32'."""
        m, b = C.c_int32(0), C.c_int64(0)
        self._ck(self.L.tcr_static_info(self.h, C.byref(m), C.byref(b)))
        return _lib.STATIC_MODES[m.value], int(b.value)

    def stage_month(self, slot, wlon, wlat, wnd_mean, wnd_cov, lon, lat, vpot, chi, mld, strat, rh_mid=None):
        """One month's field set: `_load_wnd_stat` (bam_track.py:76-91) + `init_fields`
        (coupled_fast.py:217-225).  wnd_mean [4,nlat,nlon]; wnd_cov packed lower triangle
        [10,nlat,nlon]; thermo planes [nlat,nlon] on the global grid."""
        # (the selection of transform_global_field is looked up once per grid, not once per plane; raw addresses instead of
        # typed ctypes pointers: 25 ctypes.cast calls per slot were a third of a slot's host time)
        wplan, tplan = self.basin._crop_plan(wlon, wlat), self.basin._crop_plan(lon, lat)
        crop = self.basin._apply_plan
        planes = [_f64(crop(wplan, wnd_mean[k])) for k in range(4)] + [_f64(crop(wplan, wnd_cov[k])) for k in range(10)]
        th = [_f64(crop(tplan, x)) for x in (vpot, chi, mld, strat)]
        addr = lambda a: a.__array_interface__['data'][0]
        mean_p = (C.c_void_p * 4)(*[addr(x) for x in planes[:4]])
        cov_p = (C.c_void_p * 10)(*[addr(x) for x in planes[4:]])
        wg, tg = self._grid_cached(wplan[0], wplan[1]), self._grid_cached(tplan[0], tplan[1])
        # one call, one transfer per slot (tcr_slot_upload): the planes go to the device as they are and are interleaved there;
        # m_init_fx is built on the uncropped grid in the reference (compute.py:111)
        rh = _f64(rh_mid) if rh_mid is not None else None
        rg = self._grid_cached(lon, lat) if rh is not None else None
        self._ck(self.L.tcr_slot_upload(self.h, int(slot), C.byref(wg), mean_p, cov_p, C.byref(tg), addr(th[0]), addr(th[1]), addr(th[2]),
                                        addr(th[3]), C.byref(rg) if rg is not None else None, addr(rh) if rh is not None else None))

    def _grid_cached(self, lon, lat):
        """tcr_grid of a pair of axes, kept while the same axis arrays come again (a year's twelve slots share them)."""
        cache = self.__dict__.setdefault('_grids', [])
        for a, b, g in cache:
            if a is lon and b is lat:
                return g
        g = self._grid(lon, lat)
        cache.append((lon, lat, g))
        if len(cache) > 8:
            del cache[0]
        return g

    def stage_masks(self, mlon, mlat, run_mask, basin_masks):
        """land/<B>.nc indicator grids (compute.py:87-97); basin_masks: dict id -> plane."""
        run = np.ascontiguousarray(np.asarray(run_mask) > 0.5, dtype=np.uint8)
        ms = [np.ascontiguousarray(np.asarray(basin_masks[b]) > 0.5, dtype=np.uint8) for b in BASIN_IDS]
        g = self._grid(mlon, mlat)
        arr = (_lib.U8P * 7)(*[m.ctypes.data_as(_lib.U8P) for m in ms])
        self._ck(self.L.tcr_masks_upload(self.h, C.byref(g), run.ctypes.data_as(_lib.U8P), arr))
        self._staged_masks = None

    def stage_env(self, env, months=range(12)):
        """Stage a ``synthetic.SyntheticEnv``-shaped object (12 monthly field sets).  The static planes (land, bathymetry,
        basin masks) do not change from year to year: they are staged again only when the environment hands over other
        arrays than the ones already on the device.  The test is object IDENTITY (`is`), not content: an array modified in
        place must be staged explicitly (`stage_static` / `stage_masks`, which also reset this shortcut)."""
        blon, blat = getattr(env, 'blon', None), getattr(env, 'blat', None)
        static = (env.hlon, env.hlat, env.land, env.bathy, blon, blat)
        if not self._same(getattr(self, '_staged_static', None), static):
            self.stage_static(*static)
            self._staged_static = static
        for mo in months:
            self.stage_month(mo, env.wlon, env.wlat, env.wnd_mean[mo], env.wnd_cov[mo], env.lon, env.lat,
                             env.vpot[mo], env.chi[mo], env.mld[mo], env.strat[mo], env.rh_mid[mo])
        if getattr(env, 'basin_masks', None):
            mlon, mlat = getattr(env, 'mlon', None), getattr(env, 'mlat', None)
            masks = (env.hlon if mlon is None else mlon, env.hlat if mlat is None else mlat, env.basin_masks[self.basin.basin_id], env.basin_masks)
            staged = getattr(self, '_staged_masks', None)
            if not (self._same(staged and staged[:3], masks[:3]) and all(masks[3][b] is staged[3][b] for b in BASIN_IDS)):
                self.stage_masks(*masks)
                self._staged_masks = masks[:3] + (dict(masks[3]),)
        return self

    @staticmethod
    def _same(a, b):
        return a is not None and len(a) == len(b) and all(x is y for x, y in zip(a, b))

    # -------------------------------------------------------------- hot path
    def init_m(self, storms, dvdt=0.0):
        """Coupled_FAST._init_m(y, dvdt) (coupled_fast.py:153-173) for a host batch: the m0 gen_track(m=None) starts from.
        Storms whose 'm0' is a number keep it; a missing 'm0' or NaN entries are initialised."""
        n = len(storms['lon'])
        lon0, lat0, v0, h_bl = (_f64(storms[k]) for k in ('lon', 'lat', 'v0', 'h_bl'))
        m0 = _f64(storms['m0']) if storms.get('m0') is not None else None
        slot = np.ascontiguousarray(np.asarray(storms['month']) - 1, dtype=np.int32)
        ph = _f64(storms['phases']).reshape(n, 4 * self.n_series)
        out = np.empty(n)
        if n:
            si = _lib.Storms(n, lon0.ctypes.data, lat0.ctypes.data, v0.ctypes.data, m0.ctypes.data if m0 is not None else None,
                             h_bl.ctypes.data, slot.ctypes.data, ph.ctypes.data)
            self._ck(self.L.tcr_init_m_host(self.h, C.byref(si), float(dvdt), _dp(out)))
        return out

    def integrate(self, storms, probe_cap=0, dtype='f64'):
        """Integrate + post-process a batch given as host arrays (dict with lon, lat, v0,
        m0, h_bl, month (1..12), phases [n,4,N]); returns a dict of NumPy arrays.  probe_cap > 0 adds
        'dec' [n, probe_cap] uint8, the per-evaluation `land == 1` decisions (tcr_integrate_probe_host).
        dtype='f32': the fp32 variant (tcr_integrate_f32_host), rows come back as float32.
        gen_track(m=None) (coupled_fast.py:258-261): a batch without 'm0' gets it from `init_m` (dv/dt = 0 at t = 0) first.
        (A NaN *entry* of a given m0 is not "None": the reference integrates it as it is, and so does this — call `init_m`
        explicitly to fill NaN entries.)"""
        n = len(storms['lon'])
        ns = self.n_steps
        if n and storms.get('m0') is None:
            storms = dict(storms, m0=self.init_m(storms))
        lon0, lat0, v0, m0, h_bl = (_f64(storms[k]) for k in ('lon', 'lat', 'v0', 'm0', 'h_bl'))
        slot = np.ascontiguousarray(np.asarray(storms['month']) - 1, dtype=np.int32)
        ph = _f64(storms['phases']).reshape(n, 4 * self.n_series)
        assert dtype in ('f64', 'f32') and not (probe_cap and dtype == 'f32')
        rt = np.float64 if dtype == 'f64' else np.float32
        out = {k: np.empty((n, ns), rt) for k in TRACK_F64}
        out['envw'] = np.empty((n, ns, 4), rt)
        out.update({k: np.zeros(n, np.int32) for k in TRACK_I32})
        if n == 0:
            if probe_cap:
                out['dec'] = np.full((0, int(probe_cap)), 0xff, np.uint8)
            return self._finish(out)
        si = _lib.Storms(n, *[a.ctypes.data for a in (lon0, lat0, v0, m0, h_bl, slot, ph)])
        so = _lib.Tracks(*[out[k].ctypes.data for k in ('lon', 'lat', 'v', 'm', 'vmax', 'envw',
                                                         'n_valid', 'status', 'flags', 'nfev',
                                                         'n_accept', 'n_reject')])
        if probe_cap:
            out['dec'] = np.full((n, int(probe_cap)), 0xff, np.uint8)
            self._ck(self.L.tcr_integrate_probe_host(self.h, C.byref(si), C.byref(so), out['dec'].ctypes.data, int(probe_cap)))
        elif dtype == 'f32':
            self._ck(self.L.tcr_integrate_f32_host(self.h, C.byref(si), C.byref(so)))
        else:
            self._ck(self.L.tcr_integrate_host(self.h, C.byref(si), C.byref(so)))
        return self._finish(out)

    @staticmethod
    def _finish(out):
        out['is_tc'] = (out['flags'] & _lib.FLAG_IS_TC) != 0
        out['accepted'] = (out['flags'] & _lib.FLAG_ACCEPTED) != 0
        out['traj'] = np.stack([out['lon'], out['lat'], out['v'], out['m']], axis=1)
        return out

    def integrate_dev(self, storms, tracks, stream=None):
        """Device-buffer variant: ``storms`` / ``tracks`` map field name -> device pointer
        (e.g. ``tensor.data_ptr()``); asynchronous on ``stream`` (a hipStream_t as int)."""
        si = _lib.Storms(int(storms['n']), *[int(storms[k]) for k in
                                              ('lon0', 'lat0', 'v0', 'm0', 'h_bl', 'slot', 'phases')])
        so = _lib.Tracks(*[int(tracks[k]) for k in ('lon', 'lat', 'v', 'm', 'vmax', 'envw', 'n_valid',
                                                     'status', 'flags', 'nfev', 'n_accept', 'n_reject')])
        self._ck(self.L.tcr_integrate_dev(self.h, C.byref(si), C.byref(so), C.c_void_p(stream or 0)))

    def seed(self, year, cand0, n, experiment_seed=None):
        """Device-side genesis seeding of candidates [cand0, cand0+n) (compute.py:134-175)."""
        seed = int(self.nl.gpu_experiment_seed if experiment_seed is None else experiment_seed)
        out = dict(lon=np.empty(n), lat=np.empty(n), v0=np.empty(n), m0=np.empty(n), h_bl=np.empty(n),
                   slot=np.zeros(n, np.int32), phases=np.empty((n, 4, self.n_series)),
                   basin_idx=np.zeros(n, np.int32), seed_flags=np.zeros(n, np.int32))
        s = _lib.Seeds(n, *[out[k].ctypes.data for k in ('lon', 'lat', 'v0', 'm0', 'h_bl', 'slot',
                                                          'phases', 'basin_idx', 'seed_flags')])
        self._ck(self.L.tcr_seed_host(self.h, C.c_uint64(seed), int(year), int(cand0), C.byref(s)))
        out['month'] = out['slot'] + 1
        out['counted'] = (out['seed_flags'] & 1) != 0
        out['passed'] = (out['seed_flags'] & 2) != 0
        return out

    def seed_dev(self, year, cand0, seeds, stream=None, experiment_seed=None):
        seed = int(self.nl.gpu_experiment_seed if experiment_seed is None else experiment_seed)
        s = _lib.Seeds(int(seeds['n']), *[int(seeds[k]) for k in ('lon0', 'lat0', 'v0', 'm0', 'h_bl', 'slot',
                                                                   'phases', 'basin_idx', 'seed_flags')])
        self._ck(self.L.tcr_seed_dev(self.h, C.c_uint64(seed), int(year), int(cand0), C.byref(s),
                                     C.c_void_p(stream or 0)))

    # ---------------------------------------------------------------- probes
    def fourier_table(self, phases):
        """gen_f (bam_track.py:23-31): phases [n,4,N] -> Fs [n,4,n_steps]."""
        ph = _f64(phases).reshape(-1, 4 * self.n_series)
        Fs = np.empty((ph.shape[0], 4, self.n_steps))
        self._ck(self.L.tcr_fourier_table_host(self.h, ph.shape[0], _dp(ph), _dp(Fs)))
        return Fs

    def probe_rhs(self, slot, h_bl, Fs, t, lon, lat, v, m):
        """dydt / _env_winds / _calc_alpha at points (coupled_fast.py:196-207, 65-94)."""
        t, lon, lat, v, m = (_f64(x) for x in (t, lon, lat, v, m))
        Fs = _f64(Fs)
        n = t.size
        dydt = np.empty((n, 4)); envw = np.empty((n, 4)); alpha = np.empty(n)
        self._ck(self.L.tcr_probe_rhs_host(self.h, int(slot), float(h_bl), _dp(Fs), n, _dp(t), _dp(lon),
                                           _dp(lat), _dp(v), _dp(m), _dp(dydt), _dp(envw), _dp(alpha)))
        return dydt, envw, alpha

    # ----------------------------------------------------------- measurement
    def timing_enable(self, on=True):
        self._ck(self.L.tcr_timing_enable(self.h, 1 if on else 0))

    def timing_last(self):
        ms = (C.c_double * 3)()
        self._ck(self.L.tcr_timing_last(self.h, ms))
        return dict(fourier_ms=ms[0], integrate_ms=ms[1], post_ms=ms[2])

    def timing_sum(self):
        """Summed HIP-event durations of every timed integrate call since timing_enable()."""
        ms = (C.c_double * 3)()
        n = C.c_int64(0)
        self._ck(self.L.tcr_timing_sum(self.h, ms, C.byref(n)))
        return dict(fourier_ms=ms[0], integrate_ms=ms[1], post_ms=ms[2], calls=int(n.value))

    def schedule(self, storms_per_lane=1):
        """Launch shape of batches that do not fill the chip (tcr_schedule_set): 1 = lowest latency of one batch, ~4 =
        throughput with many batches in flight.  Results do not depend on it."""
        self._ck(self.L.tcr_schedule_set(self.h, int(storms_per_lane)))
        return self

    def tune(self, **kw):
        """Launch-shape knobs (tcr_tune: waves, park, park_final, table_segments, prune, emit_grid_cap, copy_threads; a negative
        value = the library's choice).  Results do not depend on them.  Returns the knobs now in effect."""
        t = _lib.Tune()
        self._ck(self.L.tcr_tune_get(self.h, C.byref(t)))
        for k, v in kw.items():
            if k not in dict(_lib.Tune._fields_) or k == 'reserved':
                raise TypeError('unknown tuning knob %r' % k)
            setattr(t, k, int(v))
        if kw:
            self._ck(self.L.tcr_tune_set(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _lib.Tune._fields_ if k != 'reserved'}

    def stage_timing(self, reset=True):
        """Host milliseconds of this engine's slot uploads since the last reset (tcr_stage_timing)."""
        ms = (C.c_double * 5)()
        self._ck(self.L.tcr_stage_timing(self.h, ms, 1 if reset else 0))
        return dict(wait_pinned_ms=ms[0], copy_ms=ms[1], enqueue_transfer_ms=ms[2], enqueue_kernel_ms=ms[3], uploads=int(ms[4]))

    def stage_trace(self, on=True):
        self._ck(self.L.tcr_stage_trace_enable(self.h, 1 if on else 0))

    def stage_trace_sum(self):
        """Per-stage stream time (ms, waiting included) summed over the directly enqueued rounds since stage_trace()."""
        ms = (C.c_double * len(_lib.STAGES))()
        n = C.c_int64(0)
        self._ck(self.L.tcr_stage_trace_sum(self.h, ms, C.byref(n)))
        return {k: ms[i] for i, k in enumerate(_lib.STAGES) if i}, int(n.value)

    def wind_stats(self, planes, day_start=None):
        """Monthly wind mean / covariance (env_wind.py:180-228) of 4 [n_samples, ...] planes -> [14, ...]."""
        from . import preprocess
        return preprocess.wind_stats_host(self, planes, day_start)

    def pass_stats(self):
        """Occupancy of the last integrate call's k_integrate passes (device must be idle):
        list of dicts {requests, parked, wave_cycles, lane_cycles, wave_ms, shader_mhz}."""
        buf = (C.c_int64 * (6 * 16))()
        n = self.L.tcr_integrate_pass_stats(self.h, buf, 16)
        if n < 0:
            self._ck(n)
        return [dict(requests=buf[6 * p], parked=buf[6 * p + 1], wave_cycles=buf[6 * p + 2],
                     lane_cycles=buf[6 * p + 3], wave_ms=buf[6 * p + 4] / 1e5,
                     shader_mhz=100.0 * buf[6 * p + 5] / max(1, buf[6 * p + 4])) for p in range(n)]

    def sync(self, stream=None):
        self._ck(self.L.tcr_sync(self.h, C.c_void_p(stream or 0)))
