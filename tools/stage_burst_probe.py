#!/usr/bin/env python3
"""Bursts of twelve slot uploads separated by what a year's rounds do (a GPU kernel burst on another stream, a host pause):
is the slow transfer inside run_downscaling a property of the burst pattern?   python tools/stage_burst_probe.py"""
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from tropical_cyclone_risk_amd import synthetic            # noqa: E402
from tropical_cyclone_risk_amd.engine import TCEngine      # noqa: E402

env = synthetic.make_env('era5')
eng = TCEngine('GL', device=0).stage_env(env)
eng.sync()
x = torch.zeros(1 << 24, device='cuda')


def burst():
    t0 = time.perf_counter()
    for mo in range(12):
        eng.stage_month(mo, env.wlon, env.wlat, env.wnd_mean[mo], env.wnd_cov[mo], env.lon, env.lat, env.vpot[mo], env.chi[mo],
                        env.mld[mo], env.strat[mo], env.rh_mid[mo])
    t1 = time.perf_counter()
    eng.sync()
    return (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3


from tropical_cyclone_risk_amd import compute, namelist                      # noqa: E402
from tropical_cyclone_risk_amd.basins import TC_Basin                       # noqa: E402
rf = compute.GpuRound(eng, 2000, compute.default_per_rank(namelist, 1000))
side = torch.cuda.Stream()
for tag, gap in (('back to back', None), ('10 ms host pause', 'sleep'), ('GPU work on the default stream + sync', 'gpu'), ('GPU work + D2H copy of 26 MB', 'd2h'),
                 ('a year of run_tracks (default stream)', 'round'), ('a year of run_tracks (side stream)', 'round_side'),
                 ('one round, device result only', 'round_dev')):
    eng.stage_timing()
    rows = []
    for k in range(20):
        if gap == 'sleep':
            time.sleep(0.010)
        elif gap in ('gpu', 'd2h'):
            for _ in range(50):
                x.mul_(1.0001)
            if gap == 'd2h':
                x[:3_300_000].cpu()
            torch.cuda.synchronize()
        elif gap == 'round':
            compute.run_tracks(2000 + k, 1000, TC_Basin('GL'), engine=eng, round_fn=rf, ops=compute.D.Local)
        elif gap == 'round_side':
            with torch.cuda.stream(side):
                compute.run_tracks(2000 + k, 1000, TC_Basin('GL'), engine=eng, round_fn=rf, ops=compute.D.Local)
        elif gap == 'round_dev':
            compute.run_tracks(2000 + k, 1000, TC_Basin('GL'), engine=eng, round_fn=rf, ops=compute.D.Local, device_result=True)
        rows.append(burst())
    st = eng.stage_timing()
    n = st['uploads']
    print('%-40s burst of 12: %.2f ms host (+ %.2f ms final wait); per slot: wait pinned %.3f copy %.3f enqueue %.3f'
          % (tag, sum(r[0] for r in rows[2:]) / 18, sum(r[1] for r in rows[2:]) / 18, st['wait_pinned_ms'] / n, st['copy_ms'] / n,
             (st['enqueue_transfer_ms'] + st['enqueue_kernel_ms']) / n))
