"""tcr_round_dev: one round of the accept loop (util/compute.py:134-209 for a block of candidates) as ONE library call,
directly enqueued or replayed from a captured hipGraph, against the same round through the separate entry points."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TRACK_KEYS = ('n_valid', 'status', 'flags', 'nfev', 'n_accept', 'n_reject')
ROW_KEYS = ('lon', 'lat', 'v', 'm', 'vmax', 'envw')


def _staged_round(p, year, cand0, n_cand, B, stats, packed, cap):
    """The round through the separate stage methods (what round() must reproduce)."""
    p.seed_round(year, cand0, n_cand)
    p.select_passed(B)
    p.integrate(B, n_dev=p.n_passed)
    p.add_stats(stats)
    p.select_accepted()
    p.pack_accepted_meta(packed, cap, cand0)


def _snapshot(p, packed, stats, B):
    import torch
    torch.cuda.synchronize()
    n_pass = min(int(p.n_passed.item()), B)
    t = {k: p.tracks[k][:n_pass].cpu().numpy().copy() for k in TRACK_KEYS}
    is_tc = (t['flags'] & 1) != 0
    for k in ROW_KEYS:
        t[k] = p.tracks[k][:n_pass].cpu().numpy()[is_tc].copy()
    n_acc = int(p.n_accepted.item())
    return dict(n_pass=int(p.n_passed.item()), n_acc=n_acc, tracks=t, cand_idx=p.cand_idx[:n_pass].cpu().numpy().copy(),
                acc_idx=p.acc_idx[:n_acc].cpu().numpy().copy(), packed=packed[:n_acc].cpu().numpy().copy(),
                stats=stats.cpu().numpy().copy(), lon0=p.storms['lon0'][:n_pass].cpu().numpy().copy(),
                phases=p.storms['phases'][:n_pass].cpu().numpy().copy())


def _same(a, b, what):
    if not (a['n_pass'] == b['n_pass'] and a['n_acc'] == b['n_acc']):
        diff = {k: int((a['tracks'][k] != b['tracks'][k]).sum()) for k in TRACK_KEYS if a['tracks'][k].shape == b['tracks'][k].shape}
        raise AssertionError((what, a['n_acc'], b['n_acc'], 'stats', a['stats'].tolist(), b['stats'].tolist(), 'per-storm differences', diff,
                              'seeds differ', int((a['lon0'] != b['lon0']).sum()), 'phases differ', int((a['phases'] != b['phases']).sum()),
                              'flag values', np.unique(a['tracks']['flags'], return_counts=True), np.unique(b['tracks']['flags'], return_counts=True)))
    for k in ('cand_idx', 'acc_idx', 'packed', 'stats', 'lon0', 'phases'):
        assert np.array_equal(a[k], b[k], equal_nan=True), (what, k)
    for k in TRACK_KEYS + ROW_KEYS:
        assert np.array_equal(a['tracks'][k], b['tracks'][k], equal_nan=True), (what, k)


@pytest.mark.parametrize('basin,dtype,order', [('GL', 'f64', 2.0), ('NA', 'f64', False), ('GL', 'f32', 2.0)])
def test_round_equals_the_separate_calls_and_its_graph_replay(golden_env, built_lib, basin, dtype, order):
    """(a) round() direct == seed_round / select_passed / integrate / add_stats / select_accepted / pack, bit for bit;
    (b) the replayed graph gives the same for rounds with other (year, cand0) keys than the one it was captured with;
    (c) the meta columns and the n_seeds histogram are what NumPy computes from the candidate arrays."""
    import torch
    from tropical_cyclone_risk_amd import _lib
    from tropical_cyclone_risk_amd.engine import TCEngine
    from tropical_cyclone_risk_amd.pipeline import DevicePipeline
    eng = TCEngine(basin, device=0).stage_env(golden_env)
    n_cand, B, cap = 24000, 6000, 1500
    ns = eng.n_steps
    mk = lambda: DevicePipeline(eng, n_cand, B, sort_storms=order, tc_rows_only=True, dtype=dtype)
    ref, one = mk(), mk()
    dev = ref.dev
    z = lambda *s, dt=torch.int64: torch.zeros(*s, dtype=dt, device=dev)
    packed_r, packed_o = z(cap, 9 * ns + 3, dt=torch.float64), z(cap, 9 * ns + 3, dt=torch.float64)
    hist = z(84)
    keys = [(2001, 0), (2001, 3 * n_cand), (2007, 10**9 + 17), (1999, 5 * n_cand)]
    for it, (year, cand0) in enumerate(keys):
        for use_graph in (False, True):
            sr, so = z(_lib.N_STATS), z(_lib.N_STATS)
            packed_r.fill_(-7.0); packed_o.fill_(-7.0)
            _staged_round(ref, year, cand0, n_cand, B, sr, packed_r, cap)
            a = _snapshot(ref, packed_r, sr, B)
            one.round(year, cand0, n_cand, B, exact_count=True, stats=so, accepted=True, packed=packed_o, pack_cap=cap,
                      seed_hist=hist, graph=use_graph)
            b = _snapshot(one, packed_o, so, B)
            _same(a, b, (basin, dtype, year, cand0, use_graph))
            # (c) meta columns + histogram from the candidate arrays
            n_acc, width = b['n_acc'], 9 * ns
            assert 20 < n_acc <= cap and 1000 < b['n_pass']          # (NA: more seeds pass than the batch holds — clipped, stats[9])
            dense = b['acc_idx']
            slot = one.storms['slot'][:B].cpu().numpy(); bidx = one.storms['basin_idx'][:B].cpu().numpy()
            assert np.array_equal(b['packed'][:, width], (cand0 + b['cand_idx'][dense]).astype(np.float64))
            assert np.array_equal(b['packed'][:, width + 1], slot[dense] + 1.0)
            assert np.array_equal(b['packed'][:, width + 2], bidx[dense].astype(np.float64))
            fl = one.cand['seed_flags'][:n_cand].cpu().numpy(); cb = one.cand['basin_idx'][:n_cand].cpu().numpy()
            cs = one.cand['slot'][:n_cand].cpu().numpy()
            counted = (fl & 1) != 0
            want = np.bincount(cb[counted] * 12 + cs[counted], minlength=84)
            assert np.array_equal(hist.cpu().numpy(), want) and want.sum() > b['n_pass']
            cut_idx = cand0 + n_cand // 3
            cut = torch.tensor(float(cut_idx), dtype=torch.float64, device=dev)
            got = one.seed_hist(z(84), cut).cpu().numpy()
            keep = counted & (cand0 + np.arange(n_cand) <= cut_idx)
            assert np.array_equal(got, np.bincount(cb[keep] * 12 + cs[keep], minlength=84))
    gs = one.graph_stats()
    assert gs['graphs'] == 1 and gs['replays'] == len(keys) - 1, gs      # captured at first sight, replayed afterwards
    # (d) a parameter change (the step record grows) drops the graph; the round is captured again and stays right
    assert eng.grow_step_record()
    sr, so = z(_lib.N_STATS), z(_lib.N_STATS)        # (one descriptor: the graph is keyed by the buffers' addresses)
    for year, cand0 in ((2011, 0), (2012, n_cand)):
        sr.zero_(); so.zero_()
        _staged_round(ref, year, cand0, n_cand, B, sr, packed_r, cap)
        a = _snapshot(ref, packed_r, sr, B)
        one.round(year, cand0, n_cand, B, exact_count=True, stats=so, accepted=True, packed=packed_o, pack_cap=cap,
                  seed_hist=hist, graph=True)
        _same(a, _snapshot(one, packed_o, so, B), ('after grow', year))
    gs2 = one.graph_stats()
    assert gs2['graphs'] == 1 and gs2['replays'] == gs['replays'] + 1, gs2
    eng.close()


def test_round_short_of_storms_and_over_capacity(golden_env, built_lib):
    """exact_count: a round with fewer passing seeds than the batch holds integrates exactly those (flags beyond are 0,
    stats[6] counts the short round); a round with MORE passing seeds than the batch holds reports the dropped ones in
    stats[9] (bench --scaling strong must see a clipped ensemble)."""
    import torch
    from tropical_cyclone_risk_amd import _lib
    from tropical_cyclone_risk_amd.engine import TCEngine
    from tropical_cyclone_risk_amd.pipeline import DevicePipeline
    eng = TCEngine('GL', device=0).stage_env(golden_env)
    p = DevicePipeline(eng, 20000, 20000, sort_storms=2.0, tc_rows_only=True)
    st = torch.zeros(_lib.N_STATS, dtype=torch.int64, device=p.dev)
    for g in (False, True, True):
        st.zero_()
        p.round(2000, 0, 20000, 20000, stats=st, graph=g)
        s = st.cpu().numpy(); n_pass = int(p.n_passed.item())
        assert 0 < n_pass < 20000 and s[6] == 1 and s[7] == n_pass and s[9] == 0 and s[8] == 0
        assert (p.tracks['flags'][n_pass:20000] == 0).all()
    q = DevicePipeline(eng, 20000, 1000, sort_storms=2.0, tc_rows_only=True)
    st.zero_()
    q.round(2000, 0, 20000, 1000, stats=st)
    s = st.cpu().numpy()
    assert s[7] == 1000 and s[9] == n_pass - 1000 and s[6] == 0
    # the capacity argument of tcr_stats_dev is checked (v4 wrote eight counters through an unsized pointer)
    rc = eng.L.tcr_stats_dev(eng.h, 1000, None, C.byref(q._tracks_struct()), st.data_ptr(), 7, None)
    assert rc != 0 and b'n_out' in eng.L.tcr_last_error(eng.h)
    eng.close()


def test_results_do_not_depend_on_the_launch_shape(golden_env, built_lib):
    """tcr_schedule_set (storms per integrator lane) changes how many persistent waves a batch that does not fill the chip
    gets, never a storm: a round at 1, 3 and 8 storms per lane is bit-identical, direct and replayed (the graph is
    re-captured when the shape changes)."""
    import torch
    from tropical_cyclone_risk_amd import _lib
    from tropical_cyclone_risk_amd.engine import TCEngine
    from tropical_cyclone_risk_amd.pipeline import DevicePipeline
    eng = TCEngine('GL', device=0).stage_env(golden_env)
    n_cand, B, cap = 24000, 6000, 1500
    p = DevicePipeline(eng, n_cand, B, sort_storms=2.0, tc_rows_only=True)
    packed = torch.zeros(cap, 9 * eng.n_steps + 3, dtype=torch.float64, device=p.dev)
    st = torch.zeros(_lib.N_STATS, dtype=torch.int64, device=p.dev)
    ref = None
    for spl in (1, 3, 8):
        eng.schedule(spl)
        for g in (False, True, True):
            st.zero_(); packed.fill_(-7.0)
            p.round(2005, 48000, n_cand, B, stats=st, accepted=True, packed=packed, pack_cap=cap, graph=g)
            snap = _snapshot(p, packed, st, B)
            if ref is None:
                ref = snap
            _same(ref, snap, (spl, g))
    with pytest.raises(_lib.TcrError):
        eng.schedule(0)
    eng.close()


def test_compaction_single_launch_against_numpy(golden_env, built_lib):
    """tcr_compact_dev (one kernel since round 4: ticket, published tile counts, look-back) against NumPy for sizes around the
    tile boundaries, several masks, clipping at max_out, an empty input, and many calls in a row on the same scratch (the
    generation tag of the tile words)."""
    import torch
    from tropical_cyclone_risk_amd.engine import TCEngine
    eng = TCEngine('NA', device=0).stage_env(golden_env)
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(3)
    L, h = eng.L, eng.h
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    for n in (0, 1, 63, 2047, 2048, 2049, 4096, 100_003, 515_001, 7, 2048 * 3):
        flags_h = rng.integers(0, 8, size=max(n, 1)).astype(np.int32)
        flags = torch.from_numpy(flags_h).to(dev)
        idx = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
        for mask, max_out in ((2, n), (1, n), (4, max(1, n // 10)), (7, n), (8, n)):
            idx.fill_(-1)
            eng._ck(L.tcr_compact_dev(h, n, flags.data_ptr(), mask, max_out, idx.data_ptr(), count.data_ptr(), None))
            torch.cuda.synchronize()
            want = np.nonzero(flags_h[:n] & mask)[0]
            assert int(count.item()) == len(want), (n, mask)
            k = min(len(want), max_out)
            got = idx[:n].cpu().numpy() if n else np.zeros(0, np.int32)
            assert np.array_equal(got[:k], want[:k]), (n, mask, max_out)
            assert (got[k:] == -1).all(), (n, mask, max_out)              # nothing written beyond the clipped count
    # back to back without a host sync in between (stream order alone separates the launches that share the scratch)
    n = 300_000
    flags_h = rng.integers(0, 4, size=n).astype(np.int32)
    flags = torch.from_numpy(flags_h).to(dev)
    outs = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3)]
    counts = torch.zeros(3, dtype=torch.int64, device=dev)
    for rep in range(40):
        for j, mask in enumerate((1, 2, 3)):
            eng._ck(L.tcr_compact_dev(h, n, flags.data_ptr(), mask, n, outs[j].data_ptr(), counts[j:].data_ptr(), None))
    torch.cuda.synchronize()
    for j, mask in enumerate((1, 2, 3)):
        want = np.nonzero(flags_h & mask)[0]
        assert int(counts[j].item()) == len(want) and np.array_equal(outs[j][:len(want)].cpu().numpy(), want)
    eng.close()
