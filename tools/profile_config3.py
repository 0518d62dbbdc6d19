#!/usr/bin/env python3
"""Where does BASELINE config 3 (GL, 40 years x 1000 tracks per year through run_downscaling) spend its wall time?
    python tools/profile_config3.py [--years 40] [--tracks 1000] [--top 35] > profiles/r04_config3_profile.txt
cProfile of the whole call (host side), with torch.cuda.synchronize() only where the product has it."""
import argparse
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--years', type=int, default=40)
    ap.add_argument('--tracks', type=int, default=1000)
    ap.add_argument('--basin', default='GL')
    ap.add_argument('--top', type=int, default=35)
    ap.add_argument('--in-flight', type=int, default=None, help='namelist.gpu_years_in_flight')
    a = ap.parse_args()
    import torch
    from tropical_cyclone_risk_amd import compute, namelist, synthetic
    nl = types.SimpleNamespace(**{k: getattr(namelist, k) for k in dir(namelist) if not k.startswith('__')})
    nl.start_year, nl.end_year, nl.tracks_per_year = 1979, 1979 + a.years - 1, a.tracks
    nl.dataset_type = 'SYNTHETIC'
    if a.in_flight is not None:
        nl.gpu_years_in_flight = a.in_flight
    env = synthetic.make_env('era5')

    class Yearly:
        def __getattr__(self, k):
            return getattr(env, k)

        def for_year(self, y):
            return env
    walls = []
    for rep in range(2):            # the second run is the one profiled (first: allocator / library warm-up)
        with tempfile.TemporaryDirectory() as d:
            nl.output_directory, nl.exp_name = d, 'config3'
            os.makedirs(os.path.join(d, 'config3'), exist_ok=True)
            torch.cuda.synchronize()
            pr = cProfile.Profile()
            t0 = time.perf_counter()
            pr.enable()
            compute.run_downscaling(a.basin, env=Yearly(), nl=nl)
            pr.disable()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            walls.append(dt)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(a.top)
    print('config 3: %d years x %d tracks, years in flight %s: wall %.3f s (first run, incl. allocations), %.3f s (second run, profiled: main thread only)'
          % (a.years, a.tracks, getattr(nl, 'gpu_years_in_flight', None), walls[0], walls[1]))
    print(s.getvalue())


if __name__ == '__main__':
    main()
