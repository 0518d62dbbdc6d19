#!/usr/bin/env python3
"""Where a month slot's staging time goes (GPU box):  python tools/stage_probe.py [BASIN]
Times engine.stage_month (Python crop + ctypes + library) and the bare library call (tcr_slot_upload with prebuilt
arguments) for 1 / 2 / 4 / 8 copy threads, 240 slots each, and the final wait for the GPU."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from tropical_cyclone_risk_amd import _lib, synthetic          # noqa: E402
from tropical_cyclone_risk_amd.engine import TCEngine, _dp, _f64   # noqa: E402

basin = sys.argv[1] if len(sys.argv) > 1 else 'GL'
env = synthetic.make_env('era5')
eng = TCEngine(basin, device=0).stage_env(env)
eng.sync()
N = 240
for thr in (1, 2, 4, 8):
    eng.close()
    eng = TCEngine(basin, device=0)
    eng.tune(copy_threads=thr)
    eng.stage_env(env)
    eng.sync()
    t0 = time.perf_counter()
    for k in range(N):
        mo = k % 12
        eng.stage_month(mo, env.wlon, env.wlat, env.wnd_mean[mo], env.wnd_cov[mo], env.lon, env.lat, env.vpot[mo], env.chi[mo],
                        env.mld[mo], env.strat[mo], env.rh_mid[mo])
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    # the bare library call
    tf = eng.basin.transform_global_field
    mo = 3
    planes = [_f64(tf(env.wlon, env.wlat, env.wnd_mean[mo][k])[2]) for k in range(4)] + [_f64(tf(env.wlon, env.wlat, env.wnd_cov[mo][k])[2]) for k in range(10)]
    wlo, wla, _ = tf(env.wlon, env.wlat, env.wnd_mean[mo][0])
    tlo, tla, _ = tf(env.lon, env.lat, env.vpot[mo])
    th = [_f64(tf(env.lon, env.lat, x[mo])[2]) for x in (env.vpot, env.chi, env.mld, env.strat)]
    mean_p = (_lib.DP * 4)(*[_dp(x) for x in planes[:4]]); cov_p = (_lib.DP * 10)(*[_dp(x) for x in planes[4:]])
    wg, tg, rg = eng._grid(wlo, wla), eng._grid(tlo, tla), eng._grid(env.lon, env.lat)
    rh = _f64(env.rh_mid[mo])
    t3 = time.perf_counter()
    for k in range(N):
        eng.L.tcr_slot_upload(eng.h, k % 12, C.byref(wg), mean_p, cov_p, C.byref(tg), _dp(th[0]), _dp(th[1]), _dp(th[2]), _dp(th[3]), C.byref(rg), _dp(rh))
    t4 = time.perf_counter()
    eng.sync()
    t5 = time.perf_counter()
    mb = (sum(p.nbytes for p in planes) + sum(x.nbytes for x in th) + rh.nbytes) / 1e6
    print('%s copy_threads %d: stage_month %.3f ms/slot (+ %.3f ms final wait), bare tcr_slot_upload %.3f ms/slot (+ %.3f final wait); %.1f MB per slot'
          % (basin, thr, (t1 - t0) / N * 1e3, (t2 - t1) * 1e3, (t4 - t3) / N * 1e3, (t5 - t4) * 1e3, mb))
eng.close()
