"""ERA5-shaped / GFDL-shaped synthetic environment.

No reanalysis or CMIP6 data exists in the build or bench containers, so every
measurement and parity test runs on analytic, smooth, *regenerated* fields with
the shapes the reference consumes:

* thermo grid 1°x1° (lon 0..359, lat -90..90 ascending; `scripts/download_era5.py:58`)
  carrying monthly potential intensity ``vpot`` (already scaled as in
  `util/compute.py:76`), transformed saturation deficit ``chi``
  (`compute.py:113-115`), mixed-layer depth, sub-mixed-layer stratification
  (`compute.py:117-118`, NaN→0 over land) and mid-level RH;
* wind grid (same as thermo for ERA5; 2°x2.5° for the GFDL-shaped variant)
  carrying the 4 monthly-mean winds ``[ua250, va250, ua850, va850]`` and the 10
  lower-triangular covariances (`track/env_wind.py:22-42`), built as ``L Lᵀ`` of a
  smooth lower-triangular ``L`` so they are SPD everywhere;
* 0.25° land mask / bathymetry (`intensity/geo.py:9-34`) and the 7+1 basin masks
  (`scripts/generate_land_masks.py:24-110`).

Everything is a pure function of ``(shape, seed)``; nothing is stored on disk.
"""
from dataclasses import dataclass, field

import numpy as np

from . import namelist
from .basins import BASIN_IDS

N_MONTHS = 12
N_WIND = 4            # ua250, va250, ua850, va850
N_COV = 10            # packed lower triangle, row-major: (0,0),(1,0),(1,1),(2,0)...
TRIL = [(i, j) for i in range(N_WIND) for j in range(i + 1)]

# (lon_c, lat_c, a_lon, b_lat) super-ellipses making blobby "continents"
_CONTINENTS = [
    (258.0, 47.0, 28.0, 24.0),    # North America
    (268.0, 17.0, 9.0, 5.5),      # Central America
    (300.0, -14.0, 19.0, 27.0),   # South America
    (20.0, 8.0, 26.0, 33.0),      # Africa (wraps the 0/360 seam)
    (85.0, 52.0, 75.0, 24.0),     # Eurasia
    (78.0, 19.0, 8.0, 11.0),      # India
    (105.0, 18.0, 7.0, 9.0),      # Indochina
    (134.0, -25.0, 20.0, 11.0),   # Australia
    (285.0, 19.5, 4.5, 1.3),      # Caribbean island 1
    (293.5, 18.2, 1.6, 0.9),      # Caribbean island 2
    (122.0, 13.0, 2.5, 6.0),      # Philippines-like
    (138.0, 36.0, 3.0, 6.5),      # Japan-like
    (47.0, -19.0, 2.5, 6.5),      # Madagascar-like
    (0.0, -85.0, 400.0, 12.0),    # Antarctica
]


def _wrap(dlon):
    return (dlon + 180.0) % 360.0 - 180.0


def _continent_index(lon, lat):
    """g(lon,lat) > 0 over land, < 0 over ocean, smooth through coasts."""
    LON, LAT = np.meshgrid(lon, lat)
    g = np.full(LON.shape, -np.inf)
    for lc, pc, a, b in _CONTINENTS:
        r2 = (_wrap(LON - lc) / a) ** 2 + ((LAT - pc) / b) ** 2
        g = np.maximum(g, 1.0 - r2)
    return np.clip(g, -4.0, 1.0)


@dataclass
class SyntheticEnv:
    lon: np.ndarray
    lat: np.ndarray
    wlon: np.ndarray
    wlat: np.ndarray
    wnd_mean: np.ndarray      # [12, 4, nlat_w, nlon_w]
    wnd_cov: np.ndarray       # [12, 10, nlat_w, nlon_w] packed lower triangle
    vpot: np.ndarray          # [12, nlat, nlon]
    chi: np.ndarray           # [12, nlat, nlon]  (already transformed)
    mld: np.ndarray
    strat: np.ndarray
    rh_mid: np.ndarray
    hlon: np.ndarray
    hlat: np.ndarray
    land: np.ndarray          # [721, 1440] 0/1 float64 (static_res 0.125: [1440, 2880] int8, the reference's land.nc)
    bathy: np.ndarray         # same grid, metres, >0 over land
    basin_masks: dict = field(default_factory=dict)   # id -> [721,1440] float64 0/1 (always the 0.25 degree mask grid)
    seed: int = 0
    shape: str = 'era5'
    blon: np.ndarray = None   # the bathymetry's own grid when it differs from the land mask's (intensity/geo.py:9-34)
    blat: np.ndarray = None
    mlon: np.ndarray = None   # the basin masks' grid (scripts/generate_land_masks.py:24-25: always 0.25 degree) when it
    mlat: np.ndarray = None   # differs from the land mask's
    static_res: float = 0.25

    def cov_matrix(self, month):
        """Dense symmetric [4,4,nlat,nlon] view of one month's covariances."""
        out = np.empty((N_WIND, N_WIND) + self.wnd_cov.shape[2:])
        for k, (i, j) in enumerate(TRIL):
            out[i, j] = self.wnd_cov[month, k]
            out[j, i] = self.wnd_cov[month, k]
        return out


def chi_transform(chi_raw):
    """`util/compute.py:113-115`: NaN→5, then clip(exp(log(chi+1e-3)+a)+b, 1e-5, 5)."""
    chi = np.array(chi_raw, dtype=np.float64, copy=True)
    chi[np.isnan(chi)] = 5
    return np.maximum(np.minimum(
        np.exp(np.log(chi + 1e-3) + namelist.log_chi_fac) + namelist.chi_fac, 5), 1e-5)


def make_basin_masks(hlon, hlat, land):
    """Ocean-only basin indicator grids in the spirit of generate_land_masks.py."""
    LON, LAT = np.meshgrid(hlon, hlat)
    ocean = land < 0.5
    # NA/EP divide: a slanted line through the synthetic Central America
    divide = np.clip(289.0 - 1.05 * LAT, 258.0, 289.0)
    box = lambda x0, x1, y0, y1: (LON >= x0) & (LON <= x1) & (LAT >= y0) & (LAT <= y1)
    m = {
        'NA': box(255, 360, 0, 60) & (LON >= divide) & ocean,
        'EP': box(180, 290, 0, 60) & (LON < divide) & ocean,
        'WP': box(100, 180, 0, 60) & ocean,
        'NI': box(30, 100, 0, 49) & ocean,
        'SI': box(10, 100, -45, 0) & ocean,
        'AU': box(100, 170, -45, 0) & ocean,
        'SP': box(170, 260, -45, 0) & ocean,
    }
    gl = ocean.copy()
    gl[np.abs(LAT) > 50] = False
    m['GL'] = gl
    return {k: v.astype(np.float64) for k, v in m.items()}


def make_env(shape='era5', seed=20250614, wind_scale=0.7, zero_cov_patch=False, static_res=0.25, bathy_kind=None):
    """Build the 12-month synthetic environment.

    static_res: 0.25 (land / bathymetry on the 721 x 1440 grid of the basin masks, float64 — what every
           golden fixture was generated on) or 0.125: the grid and type of the file the reference actually ships,
           `intensity/data/land.nc` — int8 land on lon 0 .. 359.875 (2880), lat -89.875 .. 90 (1440) (intensity/geo.py:23-34).
           The basin masks stay on their own 0.25 degree grid (scripts/generate_land_masks.py:24-25) as `mlon` / `mlat`.
    bathy_kind: what the (absent) `bathymetry.nc` is taken to hold — 'i16' whole metres (ETOPO / GEBCO style; default at
           0.125), 'f32' float32 values, 'f64' the continuous analytic depth (default at 0.25).

    shape: 'era5' (wind grid == thermo grid 181x360) or 'gfdl' (wind 90x144 on
           2°x2.5°, thermo 180x288 on 1°x1.25°).
    zero_cov_patch: zero the covariances in a small NA box in month 9 so that
           the Cholesky-failure branch (`bam_track.py:122-126`) is exercised.
    """
    rng = np.random.default_rng(seed)
    if shape == 'era5':
        lon = np.arange(360, dtype=np.float64)
        lat = np.linspace(-90.0, 90.0, 181)
        wlon, wlat = lon.copy(), lat.copy()
    elif shape in ('gfdl', 'gaussian'):
        lon = np.arange(288, dtype=np.float64) * 1.25 + 0.625
        lat = np.linspace(-89.5, 89.5, 180)
        wlon = np.arange(144, dtype=np.float64) * 2.5 + 1.25
        wlat = np.linspace(-89.0, 89.0, 90)
        if shape == 'gaussian':
            # non-uniform latitudes (Gaussian-grid-like): exercises the general knot search
            lat = lat + 0.2 * np.sin(np.deg2rad(lat) * 3.0)
            wlat = wlat + 0.35 * np.sin(np.deg2rad(wlat) * 3.0)
    else:
        raise ValueError('unknown synthetic shape %r' % (shape,))
    mlat = np.linspace(-90.0, 90.0, 721)
    mlon = np.linspace(0.0, 360.0, 1441)[:-1]
    if static_res == 0.25:
        hlat, hlon = mlat, mlon
    elif static_res == 0.125:
        hlat = -89.875 + 0.125 * np.arange(1440)          # land.nc: float32 axes, exact multiples of 1/8
        hlon = 0.125 * np.arange(2880)
    else:
        raise ValueError('static_res must be 0.25 or 0.125')
    bathy_kind = bathy_kind or ('f64' if static_res == 0.25 else 'i16')

    # a handful of random phases/amplitudes make each seed a different climate
    ph = rng.uniform(0, 2 * np.pi, size=32)
    am = rng.uniform(0.7, 1.3, size=32)

    # ---- high-resolution land / bathymetry ---------------------------------
    g = _continent_index(hlon, hlat)
    land = (g > 0).astype(np.float64 if static_res == 0.25 else np.int8)
    depth = 4000.0 * np.clip(-g / 0.30, 0.0, 1.0) ** 1.5
    bathy = np.where(g > 0, 300.0 * g + 1.0, -depth)
    if bathy_kind == 'i16':
        bathy = np.round(bathy).astype(np.int16)
    elif bathy_kind == 'f32':
        bathy = bathy.astype(np.float32)
    elif bathy_kind != 'f64':
        raise ValueError('bathy_kind must be i16, f32 or f64')
    basin_masks = make_basin_masks(mlon, mlat, land if static_res == 0.25 else (_continent_index(mlon, mlat) > 0).astype(np.float64))

    # ---- thermo grid -------------------------------------------------------
    gt = _continent_index(lon, lat)
    ocean_t = (gt <= 0).astype(np.float64)
    LON, LAT = np.meshgrid(np.deg2rad(lon), np.deg2rad(lat))
    LATD = np.rad2deg(LAT)
    shp = (N_MONTHS,) + LON.shape
    vpot = np.empty(shp); chi = np.empty(shp); mld = np.empty(shp)
    strat = np.empty(shp); rh = np.empty(shp)
    for m in range(N_MONTHS):
        season = 2 * np.pi * (m - 3.5) / 12.0
        lat0 = 9.0 * np.sin(season)              # thermal equator migrates north in Aug
        pi = 86.0 * np.exp(-((LATD - lat0) / 25.0) ** 2)
        pi *= 1.0 + 0.08 * am[0] * np.sin(2 * LON + ph[0] + 0.3 * m) * np.cos(LAT)
        vpot[m] = pi * ocean_t                    # PI is NaN→0 over land (compute.py:108)
        chi_raw = (0.22 + 0.50 * np.sin(2.2 * LAT + 0.2 * np.sin(season)) ** 2
                   + 0.08 * am[1] * np.cos(2 * LON + ph[1]) * np.cos(LAT) ** 2)
        chi[m] = chi_transform(np.clip(chi_raw, 0.10, 1.5))
        mld[m] = (50.0 + 25.0 * am[2] * np.sin(3 * LON + ph[2] + 0.5 * m) * np.cos(2 * LAT)) * ocean_t
        strat[m] = (3.5 + 2.2 * am[3] * np.cos(2 * LON + ph[3]) * np.sin(3 * LAT + 0.4 * m)) * ocean_t
        rh[m] = 0.55 + 0.22 * am[4] * np.sin(2 * LON + ph[4] + 0.6 * m) * np.cos(3 * LAT)

    # ---- wind grid ---------------------------------------------------------
    WLON, WLAT = np.meshgrid(np.deg2rad(wlon), np.deg2rad(wlat))
    wshp = WLON.shape
    wnd_mean = np.empty((N_MONTHS, N_WIND) + wshp)
    wnd_cov = np.empty((N_MONTHS, N_COV) + wshp)
    c, s2 = np.cos(WLAT), np.sin(2 * WLAT) ** 2
    for m in range(N_MONTHS):
        season = 2 * np.pi * (m - 3.5) / 12.0
        wob = 0.3 * m
        wnd_mean[m, 0] = (-8.0 * c ** 4 + 25.0 * s2
                          + 3.0 * am[5] * np.sin(2 * WLON + ph[5] + wob) * c
                          + 2.0 * np.sin(season) * np.sin(WLAT))
        wnd_mean[m, 1] = 2.0 * am[6] * np.sin(3 * WLON + ph[6] + wob) * c
        wnd_mean[m, 2] = (-6.0 * c ** 6 + 8.0 * s2
                          + 1.5 * am[7] * np.cos(WLON + ph[7] + wob) * c)
        wnd_mean[m, 3] = 1.5 * am[8] * np.sin(2 * WLON + ph[8] + wob) * c
        # smooth lower-triangular factor; diagonal bounded away from zero
        L = np.zeros((N_WIND, N_WIND) + wshp)
        diag0 = np.array([4.2, 3.6, 2.6, 2.3]) * wind_scale
        for i in range(N_WIND):
            L[i, i] = diag0[i] * (1.0 + 0.25 * am[9 + i] * np.sin((i + 1) * WLON + ph[9 + i] + wob) * c
                                  + 0.35 * s2)
            for j in range(i):
                k = 13 + i * 3 + j
                L[i, j] = 1.1 * wind_scale * am[k] * np.cos((j + 2) * WLON + ph[k]) * np.sin(2 * WLAT + 0.2 * m)
        for k, (i, j) in enumerate(TRIL):
            wnd_cov[m, k] = sum(L[i, q] * L[j, q] for q in range(j + 1))
        if zero_cov_patch and m == 8:
            patch = ((np.rad2deg(WLON) >= 300) & (np.rad2deg(WLON) <= 312) &
                     (np.rad2deg(WLAT) >= 22) & (np.rad2deg(WLAT) <= 30))
            wnd_cov[m][:, patch] = 0.0

    return SyntheticEnv(lon=lon, lat=lat, wlon=wlon, wlat=wlat,
                        wnd_mean=wnd_mean, wnd_cov=wnd_cov, vpot=vpot, chi=chi,
                        mld=mld, strat=strat, rh_mid=rh, hlon=hlon, hlat=hlat,
                        land=land, bathy=bathy, basin_masks=basin_masks,
                        seed=seed, shape=shape, static_res=static_res,
                        mlon=None if static_res == 0.25 else mlon, mlat=None if static_res == 0.25 else mlat)


def draw_storm_inputs(n, basin, seed, env=None):
    """Simple host-side seed sampler used by parity tests and the bench
    (device-side seeding is `tcr_seed_candidates`): uniform box positions over
    the genesis latitude band, month, v0 = 5 + N(0,1), m0, h_bl, 60 phases.
    Returns a dict of float64 / int32 arrays of length n.
    """
    from .basins import TC_Basin
    rng = np.random.default_rng(seed)
    b = TC_Basin(basin)
    x0, y0, x1, y1 = b.get_bounds()
    lat_lo = 3.0 if y0 >= 0 else -45.0
    lat_hi = 45.0 if y1 > 0 else -3.0
    if y0 < 0 < y1:       # GL: both hemispheres
        sgn = rng.choice([-1.0, 1.0], size=n)
        lat = sgn * np.rad2deg(np.arcsin(rng.uniform(np.sin(np.deg2rad(3)), np.sin(np.deg2rad(45)), n)))
    else:
        lat = np.rad2deg(np.arcsin(rng.uniform(np.sin(np.deg2rad(lat_lo)), np.sin(np.deg2rad(lat_hi)), n)))
    lon = rng.uniform(x0 + 2.0, x1 - 2.0, n)
    month = rng.integers(1, 13, n).astype(np.int32)
    v0 = namelist.seed_v_init_ms + rng.standard_normal(n)
    rh = rng.uniform(0.3, 0.8, n)
    m0 = np.maximum(0.0, namelist.f_mInit(rh))
    h_bl = rng.choice([1400.0, 1500.0, 1600.0, 1800.0, 2000.0], size=n)
    phases = rng.uniform(0.0, 1.0, (n, N_WIND, namelist.gpu_N_series))
    return dict(lon=lon, lat=lat, month=month, v0=v0, m0=m0, h_bl=h_bl, phases=phases)
