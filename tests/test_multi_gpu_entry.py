"""The N > 1 entry points, before an eight-GPU node ever sees them.

* `bench.py --gpus N` without a launcher starts N ranks itself and NEVER reports another rank count than the one asked for:
  fewer GPUs than N is an error (CPU test: this container has none).
* TCR_FORCE_COLLECTIVES=1 puts a one-rank `nccl` (= RCCL) process group under a one-GPU run, so that every `world > 1` branch
  — the count / row all-gathers of the accept loop, bench.py's DeferredRowGather with its side stream and events, the
  device-to-device all-gather of year-sharded final tracks — executes on the GPU through RCCL and its stream ordering, and
  must reproduce the plain one-rank run bit for bit (the reference's seam: the dask fan-out and the collection of its
  workers' 9-tuples, util/compute.py:223-242).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'TCR_FORCE_COLLECTIVES', 'TCR_DIST_BACKEND')}
    env.update(kw)
    return env


def test_bench_refuses_to_run_a_smaller_job():
    """No GPU here: `--gpus 2` must fail loudly instead of printing an n_gpus = 1 line; so must a launcher / --gpus mismatch."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=_env(), cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and 'needs 2 GPUs' in r.stderr and '{' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8'], env=_env(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0'),
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr and '{' not in r.stdout


def test_forced_collectives_switch_is_off_by_default():
    from tropical_cyclone_risk_amd import distributed as D
    assert not D.forced() and not D.collective() and D.world() == 1


def _bench(extra_env, *flags):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--scaling', 'strong', '--storms', '12000', '--streams', '4', '--steps', '6',
           '--warmup', '2'] + list(flags)
    r = subprocess.run(cmd, env=_env(**extra_env), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])


@pytest.mark.gpu
def test_bench_through_rccl_on_one_rank(built_lib):
    """bench.py's N > 1 step — select / pack of the accepted tracks, DeferredRowGather (counts read late, rotating buffers,
    producer -> side-stream ordering by an event, stream-side waits) — through a one-rank RCCL group: every accepted track goes
    through the all-gather, the integer totals are those of the plain run, and the line carries roofline AND cpu_baseline."""
    plain = _bench({}, '--no-cpu-baseline')
    forced = _bench(dict(TCR_FORCE_COLLECTIVES='1'), '--cpu-budget', '1.0')
    c0, c1 = plain['config'], forced['config']
    assert forced['n_gpus'] == 1 and 'nccl = RCCL' in c1['allgather'] and 'nccl = RCCL' in c1['collectives']
    assert c0['allgather_rows'] is None and c0['collectives'] == 'none'
    assert c1['allgather_rows'] == c1['accepted_total'] > 0 and c1['allgather_rows_clipped'] == 0
    for k in ('storms_per_step', 'storm_steps_total', 'accepted_total', 'is_tc_fraction', 'emitted_samples_per_step'):
        assert c0[k] == c1[k], (k, c0[k], c1[k])
    assert forced['roofline']['frac'] > 0 and forced['roofline']['achieved'] > 0
    cpu = forced['cpu_baseline']
    assert cpu is not None and cpu['value'] and cpu['value'] > 0 and cpu['cores'] >= 1 and cpu['kind'] == 'port'


@pytest.mark.gpu
@pytest.mark.parametrize('shard_years', [True, False])
def test_run_py_through_rccl_on_one_rank(built_lib, tmp_path, shard_years):
    """`run.py GL --synthetic`, three years, through a one-rank RCCL group.  shard_years: the years are worked with
    `distributed.Local` and the final tracks all-gathered ONCE, device to device (`compute._allgather_years`); otherwise every
    round of every year goes through the accept loop's collectives (count pairs, survivor rows, n_seeds all-reduce).  Both
    write the plain run's track file bit for bit."""
    from tropical_cyclone_risk_amd import io as tio
    out = {}
    for tag, env in (('plain', _env()), ('forced', _env(TCR_FORCE_COLLECTIVES='1'))):
        nlf = tmp_path / ('nl_%s.py' % tag)
        nlf.write_text("start_year = 2001\nend_year = 2003\ntracks_per_year = 24\noutput_directory = %r\nexp_name = %r\ngpu_shard_years = %r\n"
                       % (str(tmp_path), tag, shard_years))
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'run.py'), 'GL', '--synthetic', '--namelist', str(nlf)], env=env, cwd=ROOT,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        out[tag] = tio.read_tracks(str(tmp_path / tag / 'tracks_GL_era5_200101_200312.nc'))
    for k in ('lon_trks', 'lat_trks', 'v_trks', 'm_trks', 'vmax_trks', 'u250_trks', 'v850_trks', 'tc_month', 'tc_basins', 'tc_years', 'seeds_per_month'):
        assert np.array_equal(out['plain'][k], out['forced'][k], equal_nan=(out['plain'][k].dtype.kind == 'f')), k
    assert out['plain']['lon_trks'].shape == (72, 361)


def _bench_cmd(n, *flags, timeout=1500):
    """`python bench.py --gpus N ...` exactly as the driver types it for N = 1 — no launcher: bench.py starts its N ranks itself
    (self_launch).  TCR_DIST_BACKEND=gloo lets the ranks share this box's one GPU."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '4', '--warmup', '1', '--storms', '12000'] + list(flags)
    r = subprocess.run(cmd, env=_env(TCR_DIST_BACKEND='gloo'), cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line on stdout (rank 0), got %d' % len(lines)
    return json.loads(lines[0])


@pytest.mark.gpu
def test_launcherless_bench_at_2_4_8_ranks(built_lib):
    """the command shape the driver runs (`python bench.py --gpus N --steps K --warmup W`, no launcher), at every
    rank count of the scaling table, with the N > 1 defaults (strong scaling, 16 streams): rc 0, ONE JSON line, n_gpus = N,
    every accepted track through the all-gather, and — the ensemble being fixed — the integer totals of the N = 1 line at every N."""
    one = _bench_cmd(1, '--no-cpu-baseline')
    c1 = one['config']
    assert one['n_gpus'] == 1 and c1['allgather_rows'] is None
    for n in (2, 4, 8):
        d = _bench_cmd(n, '--no-cpu-baseline')
        c = d['config']
        assert d['n_gpus'] == n and d['steps'] == 4 and d['warmup'] == 1 and d['scaling'] == 'strong' and d['value'] > 0
        assert 'sharded over %d GPU' % n in c['workload']
        assert c['allgather_rows'] == c['accepted_total'] > 0 and c['allgather_rows_clipped'] == 0
        for k in ('storm_steps_total', 'accepted_total', 'storms_per_step'):
            assert c[k] == c1[k], (n, k, c[k], c1[k])
        assert abs(c['storms_per_gpu'] * n - c['storms_per_step']) < 1e-6
        assert d['roofline']['frac'] > 0 and d['cpu_baseline'] is None


@pytest.mark.gpu
def test_launcherless_bench_two_ranks_with_cpu_baseline(built_lib):
    """... and once at N = 2 WITH the CPU baseline: rank 0 times it after the process group is gone; the line carries both objects."""
    d = _bench_cmd(2, '--cpu-budget', '1.0')
    assert d['n_gpus'] == 2 and d['roofline']['frac'] > 0 and d['roofline']['achieved'] > 0
    cpu = d['cpu_baseline']
    assert cpu is not None and cpu['value'] and cpu['value'] > 0 and cpu['cores'] >= 1 and cpu['kind'] == 'port'
