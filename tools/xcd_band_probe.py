"""Would XCD-local storm queues pay?  Emulation without touching the kernels: the dense batch is permuted so that the 64 storms
wave w takes first (workgroup w runs on XCD w % 8) all come from longitude band w % 8 (equal-count bands), i.e. every XCD's
private 4 MB L2 sees one eighth of the longitudes of every field.  Later refills still come from the common queue (mixed).
Compares the integrator chain against the unpermuted batch:  python tools/xcd_band_probe.py [storms] [mode]"""
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
torch.cuda.synchronize()
keys = ('lon0', 'lat0', 'v0', 'm0', 'h_bl', 'slot', 'phases', 'basin_idx')
orig = {k: pipe.storms[k][:B].clone() for k in keys}


def run(tag, perm):
    for k in keys:
        pipe.storms[k][:B] = orig[k] if perm is None else orig[k][perm]
    eng.timing_enable(True)
    ms = []
    for _ in range(4):
        pipe.integrate(B); torch.cuda.synchronize()
        ms.append(eng.timing_last()['integrate_ms'])
    ps = eng.pass_stats()
    wm = sum(p['wave_ms'] for p in ps)
    print('%-34s chain %.3f ms (min of 4)  wave-ms %.1f  pass0 us/cycle %.1f  lane-util %.3f' % (
        tag, min(ms[1:]), wm, 1e3 * ps[0]['wave_ms'] / ps[0]['wave_cycles'],
        sum(p['lane_cycles'] for p in ps) / (64.0 * sum(p['wave_cycles'] for p in ps))), flush=True)


def banded(key, n_bands=8, chunk=64, interleave=True):
    order = torch.argsort(key, stable=True)
    per = (B + n_bands - 1) // n_bands
    bands = [order[i * per:(i + 1) * per] for i in range(n_bands)]
    if not interleave:
        return torch.cat(bands)
    out, pos = [], [0] * n_bands
    c = 0
    while sum(pos) < B:
        b = c % n_bands
        if pos[b] < len(bands[b]):
            t = bands[b][pos[b]:pos[b] + chunk]; out.append(t); pos[b] += len(t)
        c += 1
    return torch.cat(out)


lon = orig['lon0']
mode = sys.argv[2] if len(sys.argv) > 2 else 'all'
if mode != 'all':            # one ordering only (for rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum runs)
    perm = {'cand': None, 'band': banded(lon), 'sorted': banded(lon, interleave=False),
            'tiles': banded(torch.floor(orig['lat0'] / 15) * 1000 + lon)}[mode]
    run(mode, perm)
    sys.exit(0)
run('candidate order', None)
run('lon bands x8, wave w <- band w%8', banded(lon))
run('lon-sorted, not interleaved', banded(lon, interleave=False))
run('slot bands (12 -> 8 XCDs)', banded(orig['slot'].double() * 400 + lon))
cell = (torch.floor(orig['lat0'] / 15) * 1000 + lon)
run('lat/lon tiles interleaved', banded(cell))
run('random permutation', torch.randperm(B, device=lon.device))
run('candidate order', None)
