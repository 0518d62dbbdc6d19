#!/usr/bin/env python3
"""Host time of the month-slot uploads INSIDE run_downscaling (config 3 shape, one year in flight), split by the library's own
clock (tcr_stage_timing) — next to the same uploads in a bare loop (tools/stage_probe.py).   python tools/stage_in_run_probe.py"""
import os
import sys
import tempfile
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tropical_cyclone_risk_amd import compute, engine, namelist, synthetic       # noqa: E402

nl = types.SimpleNamespace(**{k: getattr(namelist, k) for k in dir(namelist) if not k.startswith('__')})
nl.start_year, nl.end_year, nl.tracks_per_year = 1979, 2018, 1000
nl.dataset_type = 'SYNTHETIC'
env = synthetic.make_env('era5')


class Yearly:
    def __getattr__(self, k):
        return getattr(env, k)

    def for_year(self, y):
        return env


acc = {}
_close = engine.TCEngine.close


def close(self):
    if getattr(self, 'h', None):
        for k, v in self.stage_timing().items():
            acc[k] = acc.get(k, 0.0) + v
    _close(self)


engine.TCEngine.close = close
_sm = engine.TCEngine.stage_month
py = [0.0]
per_slot = [[] for _ in range(12)]


def stage_month(self, *a, **k):
    t0 = time.perf_counter()
    r = _sm(self, *a, **k)
    dt = time.perf_counter() - t0
    py[0] += dt
    per_slot[int(a[0])].append(dt * 1e3)
    return r


engine.TCEngine.stage_month = stage_month
for in_flight in (1, 3):
    nl.gpu_years_in_flight = in_flight
    for rep in range(2):
        acc.clear(); py[0] = 0.0
        for v in per_slot:
            del v[:]
        with tempfile.TemporaryDirectory() as d:
            nl.output_directory, nl.exp_name = d, 'c3'
            os.makedirs(os.path.join(d, 'c3'), exist_ok=True)
            t0 = time.perf_counter()
            compute.run_downscaling('GL', env=Yearly(), nl=nl)
            dt = time.perf_counter() - t0
    n = max(acc.get('uploads', 1), 1)
    print('  by month slot (median / max ms over the 40 years):', ' '.join('%.2f/%.1f' % (sorted(v)[len(v) // 2], max(v)) for v in per_slot if v))
    for v in per_slot:
        del v[:]
    print('years in flight %d: wall %.3f s; %d slot uploads; stage_month (Python, summed over the worker threads) %.3f ms per slot; inside the library per slot: '
          'wait for the pinned half %.3f, copy %.3f, enqueue transfer %.3f, enqueue kernel %.3f ms'
          % (in_flight, dt, n, py[0] / n * 1e3, acc['wait_pinned_ms'] / n, acc['copy_ms'] / n, acc['enqueue_transfer_ms'] / n, acc['enqueue_kernel_ms'] / n))
