"""Genesis seeding (util/compute.py:134-175): oracle vs the golden decisions produced by a
transcription of the reference loop over the reference's own interpolators (CPU), and the
device kernel vs the oracle (GPU)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    from oracle import seeding as S
    z = np.zeros(1, dtype=np.uint64)
    o = S.philox4x32_10(z, z, z, z, 0, 0)
    assert [int(x[0]) for x in o] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = np.full(1, 0xffffffff, dtype=np.uint64)
    o = S.philox4x32_10(f, f, f, f, 0xffffffff, 0xffffffff)
    assert [int(x[0]) for x in o] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    o = S.philox4x32_10(*[np.array([v], dtype=np.uint64) for v in (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344)],
                        0xa4093822, 0x299f31d0)
    assert [int(x[0]) for x in o] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize('basin', ['NA', 'GL', 'SI'])
def test_oracle_seeding_matches_reference_transcription(golden_env, basin):
    from oracle import seeding as S
    g = np.load(os.path.join(GOLDEN, 'seeds_%s.npz' % basin))
    se = S.SeedEnv(golden_env, basin)
    n = 400
    o = S.seed_candidates(se, int(g['seed']), int(g['year']), int(g['cand0']), n)
    for k in ('lon', 'lat', 'v0', 'm0', 'h_bl'):
        assert np.array_equal(o[k], g[k][:n]), k
    for k in ('month', 'basin_idx', 'flags', 'redraw'):
        assert np.array_equal(o[k], g[k][:n]), k


@pytest.mark.gpu
@pytest.mark.parametrize('basin', ['NA', 'GL', 'SI'])
def test_device_seeding_vs_golden(golden_env, built_lib, basin):
    """tcr_seed_host vs the golden decisions: integers identical, positions to 1e-12
    (device asin/sin/log/cos differ from libm by an ulp), phases bit-exact."""
    from oracle import seeding as S
    from tropical_cyclone_risk_amd.engine import TCEngine
    g = np.load(os.path.join(GOLDEN, 'seeds_%s.npz' % basin))
    eng = TCEngine(basin, device=0).stage_env(golden_env)
    n = len(g['lon'])
    out = eng.seed(int(g['year']), int(g['cand0']), n, experiment_seed=int(g['seed']))
    eng.close()
    assert np.abs(out['lon'] - g['lon']).max() < 1e-12
    assert np.abs(out['lat'] - g['lat']).max() < 1e-12
    for k, gk in (('month', 'month'), ('basin_idx', 'basin_idx'), ('seed_flags', 'flags')):
        assert np.array_equal(out[k], g[gk]), k
    assert np.abs(out['v0'] - g['v0']).max() < 1e-12
    assert np.abs(out['m0'] - g['m0']).max() < 1e-13
    assert np.array_equal(out['h_bl'], g['h_bl'])
    pairs = np.arange(30, dtype=np.uint64)
    for i in (0, 7, n - 1):
        a, b = S.uniform2(int(g['seed']), int(g['year']), int(g['cand0']) + i, 2, pairs)
        assert np.array_equal(out['phases'][i].reshape(-1), np.stack([a, b], 1).reshape(-1))


@pytest.mark.gpu
def test_run_tracks_end_to_end(golden_env, built_lib):
    """run_tracks through the device pipeline: quota, ordering, n_seeds bookkeeping, and
    invariance to the round size (the batched loop must not depend on how it is batched)."""
    from tropical_cyclone_risk_amd import compute
    from tropical_cyclone_risk_amd.basins import TC_Basin
    from tropical_cyclone_risk_amd.engine import TCEngine
    eng = TCEngine('NA', device=0).stage_env(golden_env)
    a = compute.run_tracks(2003, 40, TC_Basin('NA'), engine=eng, per_rank=4096)
    b = compute.run_tracks(2003, 40, TC_Basin('NA'), engine=eng, per_rank=1500)
    eng.close()
    lon, lat, v, m, vmax, envw, month, basin, n_seeds = a
    assert lon.shape == (40, 361) and envw.shape == (40, 361, 4) and n_seeds.shape == (7, 12)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True) if x.dtype.kind == 'f' else np.array_equal(x, y)
    assert (np.nanmax(vmax, axis=1) >= 18).all() and (np.nanmax(v, axis=1) >= 15).all()
    assert set(basin) <= {'NA', 'EP'} and n_seeds.sum() > 40
    n_valid = (~np.isnan(lon)).sum(axis=1)
    for i in range(40):
        assert np.isnan(lon[i, n_valid[i]:]).all() and np.isnan(vmax[i, n_valid[i]:]).all()


def _rows_vs_sequential_oracle(tag, eng, env, basin, year, cand, got, ref):
    """Rows of compute.run_tracks against the rows of the sequential oracle loop, by the probe method of
    oracle/parity.py: the kept candidates are re-seeded on both sides (device seeds within 1e-12 of the
    oracle's), re-integrated with the decision probe on both sides — the device re-run must reproduce
    run_tracks' rows bit for bit — and compared pointwise over whole tracks (storms with a differing `land == 1`
    decision against the decision-forced replay of the oracle)."""
    from oracle import c_oracle, parity, seeding as S
    seed = int(eng.nl.gpu_experiment_seed)
    se = S.SeedEnv(env, basin)
    rows = [S.seed_candidate(se, seed, year, int(c)) for c in cand]
    o_st = {k: np.array([r[k] for r in rows]) for k in ('lon', 'lat', 'v0', 'm0', 'h_bl', 'month', 'phases')}
    d_st = {k: [] for k in o_st}
    for c in cand:
        d = eng.seed(year, int(c), 1, experiment_seed=seed)
        for k in d_st:
            d_st[k].append(d[k][0])
    d_st = {k: np.array(v) for k, v in d_st.items()}
    for k in ('lon', 'lat', 'v0', 'm0'):
        assert np.abs(d_st[k] - o_st[k]).max() < 1e-12, k
    assert np.array_equal(d_st['h_bl'], o_st['h_bl']) and np.array_equal(d_st['month'], o_st['month'])
    assert np.array_equal(d_st['phases'], o_st['phases'])
    dev = eng.integrate(d_st, probe_cap=c_oracle.PROBE_CAP)
    orc = c_oracle.Ensemble(env, basin).run(o_st, probe=True)
    lon, lat, v, m, vmax, envw = got[:6]
    for a, b in ((lon, dev['lon']), (lat, dev['lat']), (v, dev['v']), (m, dev['m']), (vmax, dev['vmax']), (envw, dev['envw'])):
        assert np.array_equal(a, b, equal_nan=True)              # run_tracks' rows ARE these integrations
    # and the oracle loop's rows are the oracle's integrations of its own seeds
    r = ref['tuple9']
    assert np.array_equal(r[0], orc['traj'][:, 0], equal_nan=True) and np.array_equal(r[4], orc['vmax'], equal_nan=True)
    assert orc['accepted'].all() and dev['accepted'].all()
    # every one of these storms is an accepted track — the long-lived intensifying kind that amplifies a last-bit
    # difference (5.7 % of a seeded ensemble) — and the two sides start from seeds that differ by up to 1e-12 (the device's
    # asin / exp), not from identical inputs: the 95 % tier is 2e-10 here (measured 2-5e-11) instead of 2e-11
    s = parity.check_tracks(tag, dev, orc, dev['dec'], orc['dec'], orc['dec_t0'], eng.t_s,
                            replay=c_oracle.replayer(env, basin, o_st), tol_95=2e-10)
    assert s['pointwise'] == s['n'] and s['unreplayed'] == 0 and s['hard_mismatch'] == 0
    return s


@pytest.mark.gpu
@pytest.mark.parametrize('basin,year,n_tracks,per_rank', [('NA', 2003, 40, 1500), ('GL', 2001, 100, None)])
def test_run_tracks_vs_sequential_oracle(golden_env, built_lib, basin, year, n_tracks, per_rank):
    """the batched device run_tracks against the literal sequential loop of
    util/compute.py:134-210 (oracle/run_tracks.py) for the same Philox key — which candidates end up in the
    output, in which order, their month / basin, where the n_seeds count stops, and the rows.
    GL / 100 tracks is BASELINE config 1's size."""
    from oracle import run_tracks as RT
    from tropical_cyclone_risk_amd import compute, namelist
    from tropical_cyclone_risk_amd.basins import TC_Basin
    from tropical_cyclone_risk_amd.engine import TCEngine
    eng = TCEngine(basin, device=0).stage_env(golden_env)
    info = {}
    got = compute.run_tracks(year, n_tracks, TC_Basin(basin), engine=eng, per_rank=per_rank, info=info)
    ref = RT.run_tracks(golden_env, basin, year, n_tracks, int(namelist.gpu_experiment_seed))
    r = ref['tuple9']
    print('%s/%d: %d candidates, %d integrated, %d is_tc, %d rounds of %d' % (basin, n_tracks, ref['n_candidates'],
          ref['n_integrated'], ref['n_is_tc'], info['rounds'], info['per_rank']))
    assert np.array_equal(info['cand'], ref['cand'])             # the same candidates, in the same order
    assert np.array_equal(got[6], r[6])                          # tc_month
    assert list(got[7]) == list(r[7])                            # tc_basin
    assert np.array_equal(got[8], r[8]) and got[8].sum() > n_tracks      # n_seeds: the count stops at the same candidate
    s = _rows_vs_sequential_oracle('run_tracks-%s' % basin, eng, golden_env, basin, year, info['cand'], got, ref)
    eng.close()
    assert s['pointwise'] == n_tracks


@pytest.mark.gpu
def test_product_round_is_device_resident(golden_env, built_lib):
    """compute.GpuRound — the round function run.py's accept loop calls — keeps everything on the device: a whole
    round (seed, select, integrate, TC-rows-only post-processing, pack, meta columns, n_seeds histogram) runs
    without a single host synchronisation (torch's sync debug mode raises on one), and hands the accept loop
    device tensors; the loop itself syncs once per round, for the (count, overflow) pairs."""
    import torch
    from tropical_cyclone_risk_amd import compute
    from tropical_cyclone_risk_amd.engine import TCEngine
    eng = TCEngine('NA', device=0).stage_env(golden_env)
    rf = compute.GpuRound(eng, 2003, 4096)
    rf(0, 4096)                                   # first call sizes the library's workspaces
    cut = torch.tensor(5000.0, device=rf.pipe.dev, dtype=torch.float64)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        out = rf(4096, 4096)
        h = out['hist']()
        h2 = out['hist'](cut)
    finally:
        torch.cuda.set_sync_debug_mode('default')
    assert out['rows'].is_cuda and out['count'].is_cuda and out['bad'].is_cuda and h.is_cuda
    n_acc = int(out['count'].item())
    width = 9 * eng.n_steps
    rows = out['rows'][:n_acc].cpu().numpy()
    # (the round integrates in locality order; accept_loop sorts the accepted rows back by this candidate column)
    assert out['unordered'] and n_acc > 0 and len(np.unique(rows[:, width])) == n_acc and rows[:, width].min() >= 4096 and rows[:, width].max() < 8192
    assert int(out['bad'].sum().item()) == 0 and h.sum().item() > h2.sum().item() > 0      # (step-record overflows, seeds without room)
    # the same round through the host-visible pieces: seeds, flags
    flags = rf.pipe.tracks['flags'][:4096].cpu().numpy()
    n_pass = int(rf.pipe.n_passed.item())
    assert (flags[n_pass:] == 0).all() and ((flags[:n_pass] & 2) != 0).sum() == n_acc
    month = rows[:, width + 1]
    assert ((month >= 1) & (month <= 12)).all()
    eng.close()


@pytest.mark.gpu
def test_run_downscaling_writes_reference_schema(golden_env, built_lib, tmp_path):
    """run_downscaling (compute.py:216-270): years loop + concatenation + track file."""
    import types
    from tropical_cyclone_risk_amd import compute, io as tio, namelist
    nl = types.SimpleNamespace(**{k: getattr(namelist, k) for k in dir(namelist) if not k.startswith('__')})
    nl.output_directory = str(tmp_path); nl.exp_name = 'gpu'; nl.start_year, nl.end_year = 2010, 2011
    nl.tracks_per_year = 6; nl.gpu_candidate_round = 4096
    fn = compute.run_downscaling('NA', env=golden_env, nl=nl)
    d = tio.read_tracks(fn)
    assert d['lon_trks'].shape == (12, 361) and d['seeds_per_month'].shape == (2, 7, 12)
    assert list(d['tc_years']) == [2010] * 6 + [2011] * 6
    assert (np.nanmax(d['vmax_trks'], axis=1) >= 18).all()
    assert not np.array_equal(d['lon_trks'][:6], d['lon_trks'][6:], equal_nan=True)   # years differ


@pytest.mark.gpu
def test_run_tracks_from_reference_files(golden_env, built_lib, tmp_path):
    """SURVEY section 8 f-1, the at-GPU test of the file path: fields written in the reference's file schema and read back
    through fields.FileEnvironment (util/compute.py:64-121: time interpolation to the 15th, PI_reduc, the chi transform,
    climatologies regridded; coupled_fast.py:217-225) drive run_tracks —

      * the LOADED fields equal the in-memory ones to 1e-14 relative (the files hold vmax without the PI factor and chi before
        its transform, and the 15th is reached by interp1d's slope formula: ~1e-15, not bit for bit);
      * the GPU on the loaded fields against the ORACLE on the SAME loaded fields (round 4 compared the GPU with
        itself on two field sets whose last bits differ, and had to accept forked trajectories): the sequential loop of
        oracle/run_tracks.py keeps the same candidates, months, basins and n_seeds, and every kept track is compared pointwise
        over its whole length through parity.check_tracks with the replayer — full tiers."""
    import copy
    from oracle import run_tracks as RT
    from tropical_cyclone_risk_amd import compute, fields, namelist
    from tropical_cyclone_risk_amd.basins import TC_Basin
    from tropical_cyclone_risk_amd.engine import TCEngine
    env = copy.copy(golden_env)
    files = fields.write_reference_files(env, str(tmp_path), 2004, namelist)
    loaded = fields.FileEnvironment(namelist, files).for_year(2004)
    for name in ('wnd_mean', 'wnd_cov', 'vpot', 'chi', 'mld', 'strat', 'rh_mid', 'land', 'bathy'):
        a, b = np.asarray(getattr(loaded, name), dtype=np.float64), np.asarray(getattr(env, name), dtype=np.float64)
        assert a.shape == b.shape, name
        assert np.abs(a - b).max() <= 1e-14 * max(1.0, np.abs(b).max()), (name, np.abs(a - b).max())
    for name in ('lon', 'lat', 'wlon', 'wlat', 'hlon', 'hlat'):
        assert np.array_equal(getattr(loaded, name), getattr(env, name)), name
    for k in env.basin_masks:
        assert np.array_equal(np.asarray(loaded.basin_masks[k]) > 0.5, np.asarray(env.basin_masks[k]) > 0.5), k
    eng = TCEngine('NA', device=0).stage_env(loaded)
    info = {}
    got = compute.run_tracks(2004, 24, TC_Basin('NA'), engine=eng, per_rank=4096, info=info)
    ref = RT.run_tracks(loaded, 'NA', 2004, 24, int(namelist.gpu_experiment_seed))
    r = ref['tuple9']
    assert np.array_equal(info['cand'], ref['cand'])
    assert np.array_equal(got[6], r[6]) and list(got[7]) == list(r[7]) and np.array_equal(got[8], r[8])
    s = _rows_vs_sequential_oracle('run_tracks-from-files', eng, loaded, 'NA', 2004, info['cand'], got, ref)
    eng.close()
    assert s['pointwise'] == 24


@pytest.mark.gpu
@pytest.mark.parametrize('n_years', [1, 3])
def test_run_py_is_rank_count_invariant(built_lib, tmp_path, n_years):
    """`run.py GL --synthetic` under torchrun with two ranks (sharing this GPU, gloo as the collective
    backend) writes the same track file as a single process.  One year: its candidate blocks are sharded over the ranks
    (candidate-index sharding + ordered accept loop, an all-gather per round).  Three years: the YEARS are sharded — rank r
    works years r, r + 2 on its own — and the final tracks are all-gathered once (what the reference's one-process-per-year
    fan-out, util/compute.py:223-242, becomes)."""
    import subprocess
    import sys
    from tropical_cyclone_risk_amd import io as tio
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for tag, world in (('one', 1), ('two', 2)):
        nlf = tmp_path / ('nl_%s.py' % tag)
        nlf.write_text("start_year = 2001\nend_year = %d\ntracks_per_year = 24\noutput_directory = %r\nexp_name = %r\n"
                       % (2000 + n_years, str(tmp_path), tag))
        env = dict(os.environ, TCR_DIST_BACKEND='gloo')
        cmd = [sys.executable, os.path.join(root, 'run.py'), 'GL', '--synthetic', '--namelist', str(nlf)]
        if world > 1:
            cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
                   '--master-addr', '127.0.0.1', '--master-port', '29517'] + cmd[1:]
        r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = tio.read_tracks(str(tmp_path / tag / ('tracks_GL_era5_200101_%d12.nc' % (2000 + n_years))))
    for k in ('lon_trks', 'lat_trks', 'v_trks', 'm_trks', 'vmax_trks', 'u250_trks', 'tc_month', 'tc_basins', 'tc_years', 'seeds_per_month'):
        assert np.array_equal(out['one'][k], out['two'][k], equal_nan=(out['one'][k].dtype.kind == 'f')), k
    assert out['one']['lon_trks'].shape == (24 * n_years, 361)


@pytest.mark.gpu
def test_run_tracks_fp32_knob(golden_env, built_lib):
    """namelist.gpu_dtype = 'f32' runs the product's accept loop on the fp32 variant (BASELINE config 5): the 9-tuple
    is float64 like the reference's, the kept candidates are the fp64 run's except where an accept decision flipped
    (0.1 % of the accepted tracks in the 100k study), and the rows agree to the fp32 tolerance."""
    import types
    from tropical_cyclone_risk_amd import compute, namelist
    from tropical_cyclone_risk_amd.basins import TC_Basin
    from tropical_cyclone_risk_amd.engine import TCEngine
    nl32 = types.SimpleNamespace(**{k: getattr(namelist, k) for k in dir(namelist) if not k.startswith('__')})
    nl32.gpu_dtype = 'f32'
    out = {}
    for tag, nl in (('f64', namelist), ('f32', nl32)):
        eng = TCEngine('NA', device=0, nl=nl).stage_env(golden_env)
        info = {}
        out[tag] = (compute.run_tracks(2003, 60, TC_Basin('NA'), engine=eng, per_rank=4096, nl=nl, info=info), info['cand'])
        eng.close()
    (a, ca), (b, cb) = out['f64'], out['f32']
    assert b[0].dtype == np.float64 and b[0].shape == (60, 361)
    same = np.intersect1d(ca, cb)
    assert len(same) >= 58                                             # at most a flipped decision or two
    ia, ib = np.searchsorted(ca, same), np.searchsorted(cb, same)
    nva, nvb = (~np.isnan(a[0][ia])).sum(1), (~np.isnan(b[0][ib])).sum(1)
    ok = np.abs(nva - nvb) <= 1
    assert ok.mean() > 0.9
    m = np.minimum(nva, nvb)
    dv = [np.abs(a[2][i, :k] - b[2][j, :k]).max() for i, j, k, o in zip(ia, ib, m, ok) if o]
    assert np.median(dv) < 1e-3 and np.percentile(dv, 90) < 0.5
    assert np.array_equal(a[6][ia], b[6][ib]) and list(a[7][ia]) == list(b[7][ib])


@pytest.mark.gpu
def test_bench_multi_rank_path(built_lib, tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU) — here two ranks
    sharing this GPU with gloo as the collective backend: the step includes select / pack of the accepted tracks
    and the deferred all-gather of survivor records; the JSON line carries the whole-job aggregate.
    Both scaling modes: weak (B storms per rank and step) and strong (BASELINE config 4 as worded: one ensemble per step,
    its candidate block sharded over the ranks, accepted tracks gathered once per ensemble) — in strong mode the set of
    storms of an ensemble does not depend on the rank count, so the integer totals of a run must be identical at
    world 1 and world 2.  (The strong-mode runs replay their steps from captured hipGraphs, the weak-mode runs enqueue directly.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TCR_DIST_BACKEND='gloo')
    out = {}
    for mode, storms in (('weak', 6000), ('strong', 12000)):
        for world in (1, 2):
            cmd = [sys.executable] + (['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr',
                                       '127.0.0.1', '--master-port', '29531'] if world > 1 else []) + \
                  [os.path.join(root, 'bench.py'), '--gpus', str(world), '--steps', '4', '--warmup', '1', '--storms', str(storms),
                   '--streams', '2', '--no-cpu-baseline', '--scaling', mode] + (['--graph', 'on'] if mode == 'strong' else [])
            r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-2000:]
            out[mode, world] = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    one, two = out['weak', 1], out['weak', 2]
    assert two['n_gpus'] == 2 and two['scaling'] == 'weak' and two['value'] > 0 and two['steps'] == 4
    cfg = two['config']
    # the same per-rank work: two ranks integrate twice the storms of one
    assert cfg['storms_per_step'] == 2 * one['config']['storms_per_step'] == 12000
    assert abs(cfg['storm_steps_per_storm'] - one['config']['storm_steps_per_storm']) / one['config']['storm_steps_per_storm'] < 0.05
    assert cfg['allgather_rows'] > 0 and cfg['allgather_rows_clipped'] == 0
    # every accepted track of both ranks went through the all-gather
    assert cfg['allgather_rows'] == cfg['accepted_total'] == round(cfg['accepted_fraction'] * 6000 * 4 * 2)
    assert two['roofline']['frac'] > 0 and two['cpu_baseline'] is None and one['config']['allgather_rows'] is None
    # ---- strong scaling: the ensemble is fixed, the ranks share it
    s1, s2 = out['strong', 1], out['strong', 2]
    assert s1['scaling'] == s2['scaling'] == 'strong' and s2['n_gpus'] == 2
    c1, c2 = s1['config'], s2['config']
    assert 'sharded over 2 GPU' in c2['workload'] and 'once per ensemble' in c2['workload']
    assert abs(c1['storms_per_step'] - 12000) < 400                       # ~12 000 seeds of the candidate block pass
    for k in ('storms_per_step', 'storm_steps_total', 'accepted_total', 'is_tc_fraction'):
        assert c1[k] == c2[k], (k, c1[k], c2[k])                          # the same storms, whoever integrates them
    assert c1['emitted_samples_per_step'] == 2 * c2['emitted_samples_per_step']      # (a per-GPU figure)
    assert abs(c2['storms_per_gpu'] * 2 - c2['storms_per_step']) < 1e-9
    assert c2['allgather_rows'] == c2['accepted_total'] and c2['allgather_rows_clipped'] == 0 and c1['allgather_rows'] is None
