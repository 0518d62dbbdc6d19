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
    cases = [('plain', _env()), ('forced', _env(TCR_FORCE_COLLECTIVES='1'))]
    if shard_years:
        # ... and once more with the all-gather of final tracks going through the LIBRARY's RCCL communicator
        # (tcr_comm_create / tcr_allgather_dev behind the C ABI; distributed.LibComm) instead of torch.distributed's
        cases.append(('lib', _env(TCR_FORCE_COLLECTIVES='1', TCR_COLLECTIVES='lib')))
    for tag, env in cases:
        nlf = tmp_path / ('nl_%s.py' % tag)
        nlf.write_text("start_year = 2001\nend_year = 2003\ntracks_per_year = 24\noutput_directory = %r\nexp_name = %r\ngpu_shard_years = %r\n"
                       % (str(tmp_path), tag, shard_years))
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'run.py'), 'GL', '--synthetic', '--namelist', str(nlf)], env=env, cwd=ROOT,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        out[tag] = tio.read_tracks(str(tmp_path / tag / 'tracks_GL_era5_200101_200312.nc'))
    for k in ('lon_trks', 'lat_trks', 'v_trks', 'm_trks', 'vmax_trks', 'u250_trks', 'v850_trks', 'tc_month', 'tc_basins', 'tc_years', 'seeds_per_month'):
        for tag in out:
            assert np.array_equal(out['plain'][k], out[tag][k], equal_nan=(out['plain'][k].dtype.kind == 'f')), (tag, k)
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
    # (N = 1 with the N > 1 default of 16 streams: the timed steps are the ensembles behind `max(warmup, streams)` untimed ones,
    # so the same stream count makes them the same ensembles at every N)
    one = _bench_cmd(1, '--no-cpu-baseline', '--streams', '16')
    c1 = one['config']
    assert one['n_gpus'] == 1 and c1['allgather_rows'] is None and c1['streams'] == 16
    for n in (2, 4, 8):
        # (N = 2 runs WITH the CPU baseline: rank 0 times it after the process group is gone; the line carries both objects)
        d = _bench_cmd(n, *(('--cpu-budget', '1.0') if n == 2 else ('--no-cpu-baseline',)))
        c = d['config']
        assert d['n_gpus'] == n and d['steps'] == 4 and d['warmup'] == 1 and d['scaling'] == 'strong' and d['value'] > 0
        assert 'sharded over %d GPU' % n in c['workload'] and c['streams'] == 16 and c['warmup_effective'] == c1['warmup_effective']
        assert c['allgather_rows'] == c['accepted_total'] > 0 and c['allgather_rows_clipped'] == 0
        for k in ('storm_steps_total', 'accepted_total', 'storms_per_step'):
            assert c[k] == c1[k], (n, k, c[k], c1[k])
        assert abs(c['storms_per_gpu'] * n - c['storms_per_step']) < 1e-6
        assert d['roofline']['frac'] > 0 and d['roofline']['achieved'] > 0
        cpu = d['cpu_baseline']
        if n == 2:
            assert cpu is not None and cpu['value'] and cpu['value'] > 0 and cpu['cores'] >= 1 and cpu['kind'] == 'port'
        else:
            assert cpu is None


@pytest.mark.gpu
def test_library_allgather_through_rccl_on_one_rank(built_lib):
    """The C ABI's own multi-GPU exchange (tcr_comm_* / tcr_allgather_rows_dev / tcr_allgather_counts_dev / tcr_concat_rows_dev /
    tcr_allreduce_sum_i64_dev; SURVEY section 8(b) lists `allgather` among the exports): a one-rank RCCL communicator created
    from the library — RCCL loaded at run time, no torch.distributed involved — moves a ragged block of survivor records and
    packs it; `tcr_concat_rows_dev` is also checked as a multi-rank packer on a hand-made [world][cap] block (odd and even row
    strides, clipping at cap and at out_cap)."""
    import ctypes as C
    import torch
    from tropical_cyclone_risk_amd import _lib, distributed as D, synthetic
    from tropical_cyclone_risk_amd.engine import TCEngine
    eng = TCEngine('NA', device=0)
    comm = D.LibComm(eng, rank_=0, world_=1)
    L = _lib.lib()
    assert L.tcr_comm_rank(comm.h) == 0 and L.tcr_comm_world(comm.h) == 1
    dev = torch.device('cuda', 0)
    g = torch.Generator(device='cpu').manual_seed(3)
    for width in (9 * 361 + 3, 9 * 361):
        rows = torch.rand(40, width, dtype=torch.float64, generator=g).to(dev)
        cnt = torch.tensor([27], dtype=torch.int64, device=dev)
        counts = comm.allgather_counts(cnt)
        out, n_out = comm.allgather_rows(rows, counts)
        torch.cuda.synchronize()
        assert counts.tolist() == [27] and int(n_out) == 27
        assert torch.equal(out[:27], rows[:27])
    t = torch.arange(84, dtype=torch.int64, device=dev)
    comm.allreduce_sum_(t)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(84, dtype=torch.int64))
    # the packer alone, as four ranks would feed it: counts 3, 0, 7 (clipped to cap = 5), 2; then out_cap = 8 of the 10 rows
    for width in (12, 13):
        world, cap = 4, 5
        blk = torch.rand(world, cap, width, dtype=torch.float64, generator=g).to(dev)
        counts = torch.tensor([3, 0, 7, 2], dtype=torch.int64, device=dev)
        want = torch.cat([blk[0, :3], blk[2, :5], blk[3, :2]], dim=0)
        for out_cap in (20, 8):
            o = torch.full((out_cap, width), -1.0, dtype=torch.float64, device=dev)
            n = torch.zeros(1, dtype=torch.int64, device=dev)
            comm._chk(L.tcr_concat_rows_dev(eng.h, world, blk.data_ptr(), counts.data_ptr(), cap, width, o.data_ptr(), out_cap, n.data_ptr(), None))
            torch.cuda.synchronize()
            k = min(10, out_cap)
            assert int(n) == k and torch.equal(o[:k], want[:k]) and bool((o[k:] == -1.0).all())
    # out_cap smaller than the total: rows beyond it are dropped, the count says so
    rows = torch.rand(8, 16, dtype=torch.float64, generator=g).to(dev)
    counts = torch.tensor([8], dtype=torch.int64, device=dev)
    out, n_out = comm.allgather_rows(rows, counts, out_cap=5)
    torch.cuda.synchronize()
    assert int(n_out) == 5 and torch.equal(out, rows[:5])
    comm.close()
    eng.close()


@pytest.mark.gpu
def test_round_and_exchange_without_pytorch(built_lib, tmp_path):
    """The C ABI is the product, PyTorch is plumbing: tools/ctypes_only_round.py runs a round of the accept loop
    with hipMalloc'ed buffers and exchanges its accepted tracks through the library's one-rank RCCL communicator in a process
    that never imports torch — and its survivor records, n_seeds and counters equal the torch-backed pipeline's, bit for bit."""
    import torch
    from tropical_cyclone_risk_amd import synthetic, _lib
    from tropical_cyclone_risk_amd.engine import TCEngine
    from tropical_cyclone_risk_amd.pipeline import DevicePipeline
    out = tmp_path / 'rows.npz'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'ctypes_only_round.py'), '--basin', 'NA', '--cand', '65536', '--storms',
                        '8192', '--year', '2003', '--cand0', '131072', '--out', str(out)], env=_env(), cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'torch imported: False' in r.stdout and 'RCCL communicator ok' in r.stdout
    got = np.load(out)
    env = synthetic.make_env('era5', static_res=0.125)
    eng = TCEngine('NA', device=0).stage_env(env)
    pipe = DevicePipeline(eng, 65536, 8192, sort_storms=2.0, tc_rows_only=True)
    dev = torch.device('cuda', 0)
    ns = eng.n_steps
    cap = 8192 // 4
    packed = torch.empty(cap, 9 * ns + 3, dtype=torch.float64, device=dev)
    stats = torch.zeros(_lib.N_STATS, dtype=torch.int64, device=dev)
    hist = torch.zeros(84, dtype=torch.int64, device=dev)
    pipe.round(2003, 131072, stats=stats, accepted=True, packed=packed, pack_cap=cap, seed_hist=hist)
    torch.cuda.synchronize()
    n_acc = int(pipe.n_accepted.item())
    assert n_acc == int(got['n_accepted']) > 0 and int(pipe.n_passed.item()) == int(got['n_passed'])
    assert np.array_equal(stats.cpu().numpy(), got['stats'])
    assert np.array_equal(hist.cpu().numpy().reshape(7, 12), got['n_seeds'])
    assert np.array_equal(packed[:n_acc].cpu().numpy(), got['rows'], equal_nan=True)
    eng.close()
