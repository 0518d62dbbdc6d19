#!/usr/bin/env python3
"""BASELINE config 3 shape — GL, 40 years x tracks_per_year = 1000 on one MI355X — through the product's
`run_downscaling` (the reference's util/compute.py:216-270: years loop, per-year accept loop, track file).
Real ERA5 monthly fields do not exist in this container, so the 12 monthly field sets are the synthetic
ERA5-shaped ones, re-staged every year exactly as a multi-year file environment would be.

    python tools/run_config3.py [--years 40] [--tracks 1000] > profiles/r04_config3.json

Reports the wall time of the first call in a fresh process (what tests/test_configs.py::test_config3_shape sees:
library workspaces, torch buffers and graph captures included), of a second call, and — from a cProfile of a run
with ONE year in flight, where everything happens in the calling thread — where the host time goes.
"""
import argparse
import cProfile
import json
import os
import pstats
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

    def one(in_flight=None, profile=False):
        if in_flight is not None:
            nl.gpu_years_in_flight = in_flight
        with tempfile.TemporaryDirectory() as d:
            nl.output_directory, nl.exp_name = d, 'config3'
            os.makedirs(os.path.join(d, 'config3'), exist_ok=True)
            torch.cuda.synchronize()
            pr = cProfile.Profile() if profile else None
            t0 = time.perf_counter()
            if pr:
                pr.enable()
            fn = compute.run_downscaling(a.basin, env=Yearly(), nl=nl)
            if pr:
                pr.disable()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out = tio.read_tracks(fn)
            size = os.path.getsize(fn)
        return dt, out, size, pr
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        first, out, size, _ = one()
        second, _, _, _ = one()
        third, _, _, _ = one()
        serial, _, _, pr = one(in_flight=1, profile=True)
    st = pstats.Stats(pr).stats
    by = {}
    for (fn_, line, name), (cc, nc, tt, ct, callers) in st.items():
        key = '%s:%s' % (os.path.basename(fn_), name)
        by[key] = by.get(key, 0.0) + ct
    pick = lambda k: round(by.get(k, 0.0), 4)
    n = out['lon_trks'].shape[0]
    nv = (~np.isnan(out['lon_trks'])).sum(axis=1)
    res = dict(config='%s, %d years x %d tracks per year, synthetic ERA5-shaped monthly fields re-staged per year, one MI355X'
                      % (a.basin, a.years, a.tracks),
               years_in_flight=int(namelist.gpu_years_in_flight),
               wall_s_first_call=first, wall_s_second_call=second, wall_s_third_call=third, s_per_year=second / a.years,
               one_year_in_flight=dict(
                   wall_s=serial, note='cProfile cumulative seconds of the calling thread (the track-file rows are written by the background thread)',
                   stage_env=pick('engine.py:stage_env'), stage_month=pick('engine.py:stage_month'),
                   run_tracks=pick('compute.py:run_tracks'), accept_loop=pick('compute.py:accept_loop'),
                   host_sync_for_round_counts=pick("~:<method 'tolist' of 'torch._C.TensorBase' objects>"),
                   rows_to_host=pick("~:<method 'cpu' of 'torch._C.TensorBase' objects>"),
                   rows_to_tuple=pick('compute.py:rows_to_tuple'),
                   track_file_header_and_small_variables=pick('io.py:_header'), track_file_close=pick('io.py:close')),
               tracks=int(n), track_file_bytes=int(size),
               storm_steps_in_file=int(np.clip(nv - 1, 0, None).sum()), mean_track_hours=float(nv.mean()),
               seeds_per_year=float(out['seeds_per_month'].sum() / a.years),
               all_tracks_meet_thresholds=bool((np.nanmax(out['vmax_trks'], axis=1) >= nl.seed_vmax_threshold_ms).all()),
               basins=sorted(set(str(b) for b in out['tc_basins'])))
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
