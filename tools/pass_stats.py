"""Occupancy of k_integrate's passes for one bench-shaped batch (GL, device-seeded)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tropical_cyclone_risk_amd import synthetic
from tropical_cyclone_risk_amd.engine import TCEngine
from tropical_cyclone_risk_amd.pipeline import DevicePipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
env = synthetic.make_env('era5', seed=20250614, static_res=0.125)
eng = TCEngine('GL', device=0).stage_env(env)
pipe = DevicePipeline(eng, int(5.6 * B), B)
pipe.seed_round(2005, 0); pipe.select_passed(B)
eng.timing_enable(True)
for _ in range(3):
    pipe.integrate(B); torch.cuda.synchronize()
print('integrate %.3f ms' % eng.timing_last()['integrate_ms'])
ps = eng.pass_stats()
for i, p in enumerate(ps):
    uc = 1e3 * p['wave_ms'] / max(1, p['wave_cycles'])
    print('pass %2d: requests %7d parked %6d wave-cycles %6d lane-util %.3f wave-ms %8.1f us/cycle %5.1f shader %4.0f MHz kclk/cycle %5.1f' % (
        i, p['requests'], p['parked'], p['wave_cycles'], p['lane_cycles'] / max(1, 64 * p['wave_cycles']), p['wave_ms'],
        uc, p['shader_mhz'], uc * p['shader_mhz'] / 1e3))
tc = sum(p['wave_cycles'] for p in ps); tl = sum(p['lane_cycles'] for p in ps); tm = sum(p['wave_ms'] for p in ps)
print('total: wave-cycles %d lane-util %.3f wave-ms %.1f' % (tc, tl / (64.0 * tc), tm))
