#!/usr/bin/env python3
"""fp32 tolerance study (BASELINE config 5: "fp32 intensity ODE with tolerance study").

Integrates the SAME device-seeded storms with the fp64 path and with the fp32 variant
(tcr_integrate_f32_dev: fp32 fields, state, stage derivatives, right-hand side and rows; fp64 time and
step-size controller) and reports, per environment:
  * agreement of the discrete results (status, track length),
  * the pointwise error distribution of lon / lat / v / m over the samples both tracks have,
  * the fraction of accept-decision flips (accept test 1 = is_tc, final acceptance),
  * the shift of the lifetime-maximum-intensity (LMI) and landfall distributions of the accepted tracks.
Runs on one MI355X:   python tools/fp32_study.py [--storms 100000] > profiles/r02_fp32_study.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pct(x, qs=(50, 90, 99, 99.9, 100)):
    x = np.asarray(x, dtype=np.float64)
    return {('p%g' % q): (float(np.percentile(x, q)) if x.size else None) for q in qs}


def landfall(env, basin, lon, lat, n_valid):
    """First sample of each track that lies over land (nearest cell of the land mask)."""
    hl, ha = np.asarray(env.hlon), np.asarray(env.hlat)
    i = np.clip(np.rint((np.nan_to_num(lon) % 360.0 - hl[0]) / (hl[1] - hl[0])).astype(int), 0, hl.size - 1)
    j = np.clip(np.rint((np.nan_to_num(lat) - ha[0]) / (ha[1] - ha[0])).astype(int), 0, ha.size - 1)
    over = (np.asarray(env.land)[j, i] > 0.5) & (np.arange(lon.shape[1])[None, :] < n_valid[:, None])
    made = over.any(axis=1)
    first = np.where(made, over.argmax(axis=1), -1)
    return made, first


def study(shape, basin, B, year):
    import torch
    from tropical_cyclone_risk_amd import synthetic
    from tropical_cyclone_risk_amd.engine import TCEngine
    from tropical_cyclone_risk_amd.pipeline import DevicePipeline
    env = synthetic.make_env(shape, static_res=0.125)      # the reference's static-field shape (round 5)
    eng = TCEngine(basin, device=0).stage_env(env)
    out = {}
    res = {}
    for dt in ('f64', 'f32'):
        pipe = DevicePipeline(eng, int(7.0 * B) + 4096, B, dtype=dt)
        pipe.seed_round(year, 0)
        pipe.select_passed(B)
        assert int(pipe.n_passed.item()) >= B, 'not enough passing seeds'
        pipe.integrate(B)
        torch.cuda.synchronize()
        h = pipe.host_tracks()
        res[dt] = {k: np.asarray(h[k]) for k in ('lon', 'lat', 'v', 'm', 'vmax', 'n_valid', 'status', 'nfev', 'is_tc', 'accepted')}
        del pipe
    eng.close()
    a, b = res['f64'], res['f32']
    n = B
    same_status = a['status'] == b['status']
    dn = b['n_valid'].astype(np.int64) - a['n_valid']
    out['storms'] = n
    out['discrete'] = dict(status_equal=float(same_status.mean()), n_valid_equal=float((dn == 0).mean()),
                           n_valid_within_1=float((np.abs(dn) <= 1).mean()), n_valid_within_6=float((np.abs(dn) <= 6).mean()),
                           abs_dn=pct(np.abs(dn)), nfev_equal=float((a['nfev'] == b['nfev']).mean()),
                           nfev_ratio=float(b['nfev'].sum() / max(1, a['nfev'].sum())))
    # pointwise errors over the samples both tracks have
    common = np.minimum(a['n_valid'], b['n_valid'])
    mask = np.arange(a['lon'].shape[1])[None, :] < common[:, None]
    err = {}
    for k in ('lon', 'lat', 'v', 'm'):
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))[mask]
        err[k] = pct(d)
        per_storm = np.where(mask, np.abs(np.nan_to_num(a[k]).astype(np.float64) - np.nan_to_num(b[k]).astype(np.float64)), 0).max(axis=1)
        err[k + '_per_storm_max'] = pct(per_storm)
    # the first day only: before the adaptive integrator has had time to amplify anything
    m24 = mask & (np.arange(a['lon'].shape[1])[None, :] <= 24)
    err['first_24h'] = {k: pct(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))[m24]) for k in ('lon', 'lat', 'v', 'm')}
    out['error_vs_fp64'] = err
    out['decisions'] = dict(is_tc_f64=float(a['is_tc'].mean()), is_tc_f32=float(b['is_tc'].mean()),
                            is_tc_flips=float((a['is_tc'] != b['is_tc']).mean()),
                            accepted_f64=int(a['accepted'].sum()), accepted_f32=int(b['accepted'].sum()),
                            accepted_flips=int((a['accepted'] != b['accepted']).sum()),
                            accepted_flips_per_accepted=float((a['accepted'] != b['accepted']).sum() / max(1, a['accepted'].sum())),
                            flips_gained=int((~a['accepted'] & b['accepted']).sum()), flips_lost=int((a['accepted'] & ~b['accepted']).sum()))
    # lifetime maximum intensity of the accepted tracks of each run (what a risk analysis reads)
    lmi = {}
    for tag, r in (('f64', a), ('f32', b)):
        acc = r['accepted']
        x = np.nanmax(np.where(np.isnan(r['vmax'][acc]), -np.inf, r['vmax'][acc].astype(np.float64)), axis=1)
        lmi[tag] = dict(mean=float(x.mean()), std=float(x.std()), **pct(x, (5, 25, 50, 75, 95, 99)),
                        frac_ge_33=float((x >= 33).mean()), frac_ge_50=float((x >= 50).mean()), frac_ge_70=float((x >= 70).mean()))
        made, first = landfall(env, basin, r['lon'][acc].astype(np.float64), r['lat'][acc].astype(np.float64), r['n_valid'][acc])
        lat_lf = np.abs(r['lat'][acc][np.arange(acc.sum()), np.clip(first, 0, None)].astype(np.float64))[made]
        v_lf = r['v'][acc][np.arange(acc.sum()), np.clip(first, 0, None)].astype(np.float64)[made]
        lmi[tag]['landfall'] = dict(fraction=float(made.mean()), hours_to_landfall=pct(first[made], (25, 50, 75, 95)),
                                    abs_lat_at_landfall=pct(lat_lf, (25, 50, 75, 95)), v_at_landfall=dict(mean=float(v_lf.mean()), **pct(v_lf, (50, 90, 99))))
    both = a['accepted'] & b['accepted']
    xa = np.nanmax(np.where(np.isnan(a['vmax'][both]), -np.inf, a['vmax'][both]), axis=1)
    xb = np.nanmax(np.where(np.isnan(b['vmax'][both]), -np.inf, b['vmax'][both].astype(np.float64)), axis=1)
    lmi['paired_abs_diff'] = pct(np.abs(xa - xb))
    # two-sample Kolmogorov-Smirnov distance of the two LMI distributions
    xs = np.sort(np.nanmax(np.where(np.isnan(a['vmax'][a['accepted']]), -np.inf, a['vmax'][a['accepted']]), axis=1))
    ys = np.sort(np.nanmax(np.where(np.isnan(b['vmax'][b['accepted']]), -np.inf, b['vmax'][b['accepted']].astype(np.float64)), axis=1))
    grid = np.concatenate([xs, ys])
    lmi['ks_distance'] = float(np.abs(np.searchsorted(xs, grid, side='right') / xs.size - np.searchsorted(ys, grid, side='right') / ys.size).max())
    out['lmi_and_landfall'] = lmi
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--storms', type=int, default=100_000)
    a = ap.parse_args()
    res = dict(what='fp32 variant (tcr_integrate_f32_dev) against the fp64 path on identical device-seeded storms',
               fp32_means='fields, state, stage derivatives, RHS, dense output and rows in fp32; time, output grid and the '
                          'step-size controller (error norm, accept test, step factor) in fp64',
               era5_GL=study('era5', 'GL', a.storms, 2005),
               gfdl_two_grid_GL=study('gfdl', 'GL', a.storms, 2005))
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
