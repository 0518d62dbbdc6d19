"""Time of the forcing-table kernels alone for a bench-shaped batch (events around tcr_integrate's Fourier stage)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tropical_cyclone_risk_amd import synthetic
from tropical_cyclone_risk_amd.engine import TCEngine
from tropical_cyclone_risk_amd.pipeline import DevicePipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
env = synthetic.make_env('era5', seed=20250614)
eng = TCEngine('GL', device=0).stage_env(env)
pipe = DevicePipeline(eng, int(5.6 * B), B)
pipe.seed_round(2005, 0); pipe.select_passed(B)
eng.timing_enable(True)
t = []
for _ in range(5):
    pipe.integrate(B); torch.cuda.synchronize()
    t.append(eng.timing_last()['fourier_ms'])
print('fourier stage ms:', ' '.join('%.3f' % x for x in t))
