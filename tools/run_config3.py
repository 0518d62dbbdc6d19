#!/usr/bin/env python3
"""BASELINE config 3 shape — GL, 40 years x tracks_per_year = 1000 on one MI355X — through the product's
`run_downscaling` (the reference's util/compute.py:216-270: years loop, per-year accept loop, track file).
Real ERA5 monthly fields do not exist in this container, so the 12 monthly field sets are the synthetic
ERA5-shaped ones, re-staged every year exactly as a multi-year file environment would be.

    python tools/run_config3.py [--years 40] [--tracks 1000] > profiles/r02_config3.json
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--years', type=int, default=40)
    ap.add_argument('--tracks', type=int, default=1000)
    ap.add_argument('--basin', default='GL')
    a = ap.parse_args()
    import torch
    from tropical_cyclone_risk_amd import compute, io as tio, namelist, synthetic
    nl = types.SimpleNamespace(**{k: getattr(namelist, k) for k in dir(namelist) if not k.startswith('__')})
    nl.start_year, nl.end_year, nl.tracks_per_year = 1979, 1979 + a.years - 1, a.tracks
    nl.dataset_type = 'SYNTHETIC'
    env = synthetic.make_env('era5')

    class Yearly:                       # re-stage the field set every year, as run_downscaling does for file environments
        def __getattr__(self, k):
            return getattr(env, k)

        def for_year(self, y):
            return env
    with tempfile.TemporaryDirectory() as d:
        nl.output_directory, nl.exp_name = d, 'config3'
        os.makedirs(os.path.join(d, 'config3'), exist_ok=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn = compute.run_downscaling(a.basin, env=Yearly(), nl=nl)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = tio.read_tracks(fn)
        size = os.path.getsize(fn)
    n = out['lon_trks'].shape[0]
    nv = (~np.isnan(out['lon_trks'])).sum(axis=1)
    res = dict(config='%s, %d years x %d tracks per year, synthetic ERA5-shaped monthly fields re-staged per year, one MI355X'
                      % (a.basin, a.years, a.tracks),
               wall_s=dt, s_per_year=dt / a.years, tracks=int(n), track_file_bytes=int(size),
               storm_steps_in_file=int(np.clip(nv - 1, 0, None).sum()), mean_track_hours=float(nv.mean()),
               seeds_per_year=float(out['seeds_per_month'].sum() / a.years),
               all_tracks_meet_thresholds=bool((np.nanmax(out['vmax_trks'], axis=1) >= nl.seed_vmax_threshold_ms).all()),
               basins=sorted(set(str(b) for b in out['tc_basins'])))
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
