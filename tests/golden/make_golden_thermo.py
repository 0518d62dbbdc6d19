"""Golden vectors for the PI / chi / RH preprocessing, made by importing the reference.

Runs the reference's own `thermo.CAPE_PI_vectorized`, `thermo.sat_deficit`, `thermo.conv_q_to_rh`,
`thermo.get_LCL`, `thermo.s_unsat`, `thermo.s_sat` (thermo/thermo.py, pure NumPy/SciPy) on synthetic
soundings and stores inputs + outputs.  Only runs where /root/reference exists:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_thermo.py

Also copies the reference's *data file* thermo/entropy_table.npz (the 200 x 200 temperature table over
pressure and entropy that `CAPE_PI_vectorized` interpolates; it was produced by a Nelder-Mead inversion,
so it cannot be regenerated bit for bit) next to the vectors: it is an input of every PI evaluation.
"""
import os
import shutil
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import REF, import_reference  # noqa: E402


def soundings(rng, nlat, nlon, nlev):
    """Tropical-to-polar columns on pressure levels 1000..50 hPa, lowest level first, plus hard cases."""
    p = np.linspace(100000.0, 5000.0, nlev)
    lat = np.linspace(-75, 75, nlat)[:, None] + np.zeros((1, nlon))
    sst = 302.0 - 28.0 * (np.abs(lat) / 75.0) ** 1.6 + rng.normal(0, 0.8, size=lat.shape)
    t_ns = sst - rng.uniform(0.3, 2.5, size=lat.shape)
    gamma = rng.uniform(0.17, 0.22, size=lat.shape)                       # d ln T / d ln p of the troposphere
    T = t_ns[None] * (p[:, None, None] / p[0]) ** gamma[None]
    T = np.maximum(T, rng.uniform(195, 215, size=lat.shape)[None])         # isothermal stratosphere
    T += rng.normal(0, 0.3, size=T.shape)
    rh = np.clip(rng.uniform(0.55, 0.9, size=lat.shape)[None] * (p[:, None, None] / p[0]) ** rng.uniform(0.5, 2.0, size=lat.shape)[None],
                 0.02, 0.98)
    tc = T - 273.0
    es = 610.94 * np.exp(np.minimum(17.625 * tc / (tc + 243.04), 10))
    r = rh * 0.622 * es / (p[:, None, None] - es)
    psl = 101000.0 + rng.normal(0, 600, size=lat.shape)
    # hard cases
    sst[0, 0] = 0.0              # nan_to_num'ed land point in Kelvin files (calc_thermo.py:40-42)
    sst[0, 1] = 273.15           # the same in Celsius files
    sst[1, 0] = 310.0            # very hot
    r[:, 1, 1] *= 0.05           # very dry column: LCL far aloft
    r[0, 2, 2] = 0.0             # zero moisture at the surface
    T[:, 3, 3] += 12.0           # warm environment: no buoyancy
    T[0, 4, 4] = np.nan          # a NaN in the sounding
    r[:, 5, 5] = rs_like(T[:, 5, 5], p) * 1.02      # supersaturated boundary layer
    return p, sst, psl, T, r


def rs_like(T, p):
    tc = T - 273.0
    es = 610.94 * np.exp(np.minimum(17.625 * tc / (tc + 243.04), 10))
    return 0.622 * es / (p - es)


def main():
    ref = import_reference()
    from thermo import thermo
    nl = ref.namelist
    assert nl.select_thermo == 1 and nl.select_interp == 2
    nl.src_directory = REF                     # where CAPE_PI_vectorized looks for the table
    rng = np.random.default_rng(20250615)
    out = {}
    for tag, (nlat, nlon, nlev) in dict(a=(16, 24, 20), b=(10, 12, 37)).items():
        p, sst, psl, T, r = soundings(rng, nlat, nlon, nlev)
        with np.errstate(all='ignore'):
            pi = thermo.CAPE_PI_vectorized(sst, psl, p, T, r)
            k_mid = int(np.argmin(np.abs(p - nl.p_midlevel)))
            chi = thermo.sat_deficit(sst, psl, T[k_mid], float(p[k_mid]), r[k_mid])
            rhm = thermo.conv_q_to_rh(T[k_mid], r[k_mid], float(p[k_mid]))
            es, rs = thermo.sat_thermo(sst, psl)
            rh_ns = r[0] / rs * (1 + rs / 0.6219718309859156) / (1 + r[0] / 0.6219718309859156)
            from util import constants as pr
            rh_ns = r[0] / rs * (1 + rs / pr.eps) / (1 + r[0] / pr.eps)
            plcl = thermo.get_LCL(p[0], T[0], r[0], rh_ns)
            s_ns = thermo.s_unsat(T[0], p[0], r[0], r[0], 1)
            ss = thermo.s_sat(sst, psl, rs, 1)
        out.update({tag + '_p': p, tag + '_sst': sst, tag + '_psl': psl, tag + '_T': T, tag + '_r': r,
                    tag + '_PI': pi, tag + '_chi': chi, tag + '_rh_mid': rhm, tag + '_k_mid': np.int64(k_mid),
                    tag + '_pLCL': np.asarray(plcl, dtype=np.float64), tag + '_s_ns': s_ns, tag + '_ss': ss})
        print(tag, 'PI range', np.nanmin(pi), np.nanmax(pi), 'zeros', int((pi == 0).sum()), 'of', pi.size)
    out['versions'] = np.array(['numpy ' + np.__version__, 'scipy ' + scipy.__version__])
    out['Ck_over_Cd'] = np.float64(nl.Ck / nl.Cd)
    out['p_midlevel'] = np.float64(nl.p_midlevel)
    np.savez_compressed(os.path.join(HERE, 'thermo_cases.npz'), **out)
    shutil.copyfile(os.path.join(REF, 'thermo', 'entropy_table.npz'), os.path.join(HERE, 'entropy_table.npz'))
    print('wrote thermo_cases.npz and entropy_table.npz')


if __name__ == '__main__':
    main()
