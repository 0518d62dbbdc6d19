"""ORACLE-side timed CPU baseline (bench.py's ``cpu_baseline`` leg only).

Runs ``oracle/scipy_port.py`` — the same SciPy/NumPy calls the reference makes,
one storm at a time — over a bounded sample of the bench's own storm inputs, on
1 core and on P worker processes (the reference's own parallel model is one dask
process per year, `util/compute.py:223-230`).  Executed as a subprocess so that
forking workers never happens in a process that has initialised HIP.

    python -m oracle.cpu_baseline --inputs storms.npz --basin GL --procs 8 --budget 12
"""
import os

# One BLAS / OpenMP thread per worker *before* NumPy loads: P forked workers that each spin up a
# P-thread BLAS pool oversubscribe the box by P x (round 1 measured 7.3x from 128 processes).
for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS', 'NUMEXPR_NUM_THREADS', 'VECLIB_MAXIMUM_THREADS'):
    os.environ[_v] = '1'

import argparse                     # noqa: E402
import json                         # noqa: E402
import multiprocessing as mp        # noqa: E402
import sys                          # noqa: E402
import time                         # noqa: E402

import numpy as np                  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_ENV = None
_BASIN = None
_STORMS = None


_CACHE = {}


def _work(idx, post='tc'):
    """post='tc': integration + accept tests + env-wind recompute / vmax for the candidates that pass accept test 1 —
    what the reference's loop does per candidate (compute.py:176-209) and what the GPU step does; post=False:
    integration only (gen_track)."""
    from oracle import scipy_port as P
    o = P.run_ensemble(_ENV, _BASIN, _STORMS, index=idx, post=post, cache=_CACHE)
    return int(np.clip(o['n_valid'] - 1, 0, None).sum()), int(o['nfev'].sum()), len(idx)


def _work_timed(args):
    """Work through `idx` eight storms at a time until `seconds` have passed (storm cost varies by two orders of
    magnitude with its lifetime, so a leg is sized by the clock, not by a calibration): (storm-steps, nfev, storms, seconds)."""
    idx, seconds, post = args
    t0 = time.perf_counter()
    steps = nfev = done = 0
    while done < len(idx) and time.perf_counter() - t0 < seconds:
        s, f, n = _work(idx[done:done + 8], post)
        steps += s; nfev += f; done += n
    return steps, nfev, done, time.perf_counter() - t0


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota); None if unlimited.
    The GPU boxes of this pool show 256 logical CPUs but run under `cpu.max = 1600000 100000`, i.e. 16 cores:
    forking one worker per visible core (round 1) measured the quota, not the cores."""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            return float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def usable_cores():
    """Worker processes to fork: physical cores of the affinity mask, capped by the cgroup CPU quota."""
    n = physical_cores()
    q = cgroup_cpu_quota()
    return max(1, min(n, int(q))) if q else n


def physical_cores():
    """Cores in this process's affinity mask, counting SMT siblings once."""
    cpus = sorted(os.sched_getaffinity(0))
    seen, n = set(), 0
    for c in cpus:
        try:
            sib = open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % c).read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            n += 1
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--inputs', required=True)
    ap.add_argument('--basin', default='GL')
    ap.add_argument('--shape', default='era5')
    ap.add_argument('--env-seed', type=int, default=20250614)
    ap.add_argument('--procs', type=int, default=0)
    ap.add_argument('--budget', type=float, default=12.0, help='target seconds per leg')
    ap.add_argument('--static-res', type=float, default=0.25)
    ap.add_argument('--bathy-kind', default=None)
    a = ap.parse_args()
    global _ENV, _BASIN, _STORMS
    from tropical_cyclone_risk_amd import synthetic
    _ENV = synthetic.make_env(a.shape, seed=a.env_seed, static_res=a.static_res, bathy_kind=a.bathy_kind)
    _BASIN = a.basin
    z = np.load(a.inputs)
    _STORMS = {k: z[k] for k in z.files}
    n_avail = len(_STORMS['lon'])
    procs = a.procs or usable_cores()

    # warm up (imports, SciPy's first calls; 64 storms touch all twelve month environments, each 20 spline constructions)
    _work(list(range(min(64, n_avail))))
    every = list(range(n_avail))
    steps1, nfev1, n1, dt1 = _work_timed((every, a.budget, 'tc'))
    out = dict(one_core=dict(storm_steps=steps1, seconds=dt1, storms=n1, value=steps1 / dt1, nfev=nfev1,
                             what='integration + accept tests + env winds / vmax of the candidates that pass accept test 1 (compute.py:176-209)'))
    stepsi, _, ni, dti = _work_timed((every, 0.35 * a.budget, False))
    out['one_core_integration_only'] = dict(storm_steps=stepsi, seconds=dti, storms=ni, value=stepsi / dti)
    if procs > 1:
        chunks = [list(range(i, n_avail, procs)) for i in range(procs)]
        ctx = mp.get_context('fork')
        with ctx.Pool(procs) as pool:
            pool.map(_work, [list(range(16))] * procs)      # warm the workers (imports; the month environments are inherited)
            t0 = time.perf_counter(); res = pool.map(_work_timed, [(c, a.budget, 'tc') for c in chunks]); dtP = time.perf_counter() - t0
            t0 = time.perf_counter(); res_i = pool.map(_work_timed, [(c, 0.35 * a.budget, False) for c in chunks]); dtPi = time.perf_counter() - t0
        stepsP, nP = sum(r[0] for r in res), sum(r[2] for r in res)
        out['all_cores'] = dict(storm_steps=stepsP, seconds=dtP, storms=nP, value=stepsP / dtP, procs=procs,
                                physical_cores_visible=physical_cores(), cgroup_cpu_quota=cgroup_cpu_quota(),
                                per_core_efficiency=(stepsP / dtP) / (procs * steps1 / dt1))
        stepsPi, nPi = sum(r[0] for r in res_i), sum(r[2] for r in res_i)
        out['all_cores_integration_only'] = dict(storm_steps=stepsPi, seconds=dtPi, storms=nPi, value=stepsPi / dtPi, procs=procs)
    # the plain-C restatement (oracle/tc_oracle.c) on one core, for scale: same algorithm without the
    # interpreter / SciPy call overhead that dominates the reference's own CPU path
    try:
        from oracle import c_oracle
        nc = min(n_avail, 2048)
        sub = {k: v[:nc] for k, v in _STORMS.items()}
        t0 = time.perf_counter(); o = c_oracle.run_ensemble(_ENV, _BASIN, sub); dtc = time.perf_counter() - t0
        sc = int(np.clip(o['n_valid'] - 1, 0, None).sum())
        out['c_port_one_core'] = dict(storm_steps=sc, seconds=dtc, storms=nc, value=sc / dtc)
    except Exception as e:                                            # report, never hide
        out['c_port_one_core'] = dict(error=repr(e))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
