"""Per-year driver: the batched, multi-GPU form of the reference's ``run_tracks``
and ``run_downscaling`` (`util/compute.py:64-270`).

``run_tracks(year, n_tracks, b)`` returns the same 9-tuple the reference returns
(`compute.py:210`).  Its sequential "draw a seed, integrate, keep it if it became
a TC, until n_tracks are kept" loop (`compute.py:134-209`) is restated as rounds:

    round r covers global candidates [r*W*C, (r+1)*W*C); rank k owns the block
    [(r*W + k)*C, (r*W + k + 1)*C); every candidate has its own Philox stream, so
    the set of accepted candidates does not depend on W, C or the GPU count.

A round seeds its candidates on the device, integrates the ones whose seed
passed, and reports which were accepted.  The loop stops after the round in which
the n_tracks-th acceptance happens; tracks are the first n_tracks accepted
candidates *in candidate order*, and ``n_seeds`` counts every counted candidate
up to and including the one that completed the quota — exactly what the
sequential loop would have counted.

Deliberate deviation (documented in DESIGN.md): the reference writes a candidate's
lon/lat/v/m/env-wind rows *before* the vmax test (`compute.py:193-202`), so rows of
candidates rejected by that test leave stale tails under the next accepted
track.  Here every returned row is NaN beyond its own track end.
"""
import time

import numpy as np

from . import distributed as D
from . import namelist as default_namelist
from .basins import BASIN_IDS, TC_Basin

ROW_VARS = 9          # lon, lat, v, m, vmax, u250, v250, u850, v850 per output sample


N_META = 3          # columns appended to a survivor record: global candidate index, month (1..12), basin index


def _legacy_round(out, cand0, n_steps, torch):
    """Adapter for round functions that return host arrays (the CPU tests' fake rounds):
    counted / basin_idx / month per candidate, acc_cand / acc_rows / acc_month / acc_basin per accepted track."""
    a = len(out['acc_cand'])
    width = ROW_VARS * n_steps
    rows = np.zeros((a, width + N_META))
    if a:
        rows[:, :width] = np.asarray(out['acc_rows'], dtype=np.float64).reshape(a, width)
        rows[:, width] = np.asarray(out['acc_cand'], dtype=np.float64)
        rows[:, width + 1] = np.asarray(out['acc_month'], dtype=np.float64)
        rows[:, width + 2] = np.asarray(out['acc_basin'], dtype=np.float64)
    dev = out.get('device', 'cpu')
    counted = np.asarray(out['counted'], bool)
    key = np.asarray(out['basin_idx']).astype(np.int64) * 12 + (np.asarray(out['month']).astype(np.int64) - 1)
    idx = cand0 + np.arange(len(counted))

    def hist(cutoff=None):
        keep = counted if cutoff is None else counted & (idx <= int(cutoff))
        return torch.from_numpy(np.bincount(key[keep], minlength=len(BASIN_IDS) * 12).astype(np.float64)).to(dev)
    return dict(rows=torch.from_numpy(rows).to(dev), count=torch.tensor([a], dtype=torch.int64, device=dev),
                bad=torch.tensor([int(out.get('bad', 0))], dtype=torch.int64, device=dev), hist=hist)


def accept_loop(round_fn, n_tracks, per_rank, n_steps, max_rounds=10000, ops=None, device_result=False):
    """Order-preserving accept loop over rounds of candidates (backend-agnostic: RCCL on the GPUs, gloo in the
    CPU tests).

    round_fn(cand0, count) handles the candidates [cand0, cand0+count) of THIS rank and returns, on its device,
        rows   float64 [cap, 9*n_steps + 3]: survivor records (lon, lat, v, m, vmax, envw[ns][4]) of the accepted
               tracks in candidate order, then the columns global candidate index, month, basin index
        count  int64 [1]: how many rows are valid;  bad int64 [1] or [2]: storms that overflowed their step record and
               (optional) passing seeds that did not fit the round's storm capacity
        hist(cutoff=None) -> float64 [7*12]: candidates counting toward n_seeds (compute.py:165-167) per
               (basin, month), optionally only those with global index <= cutoff
    (or the host-array dict of `_legacy_round`).  Survivor rows never leave the device between the kernels
    and the collective; the loop synchronises with the host ONCE per round — to read the per-rank (count,
    overflow) pairs that size the all-gather — and the result is copied to the host once, at the end.
    Returns dict(rows [n_tracks, 9*n_steps], month, basin_idx, cand, n_seeds [7, 12], rounds); every rank
    returns the same result.  (With a GpuRound, `rows` is a view of its pinned host buffer: valid until that round function's next
    accept_loop — `rows_to_tuple` copies what it keeps.)  ops: the collectives (default: the process group's, `distributed`; `distributed.Local` when
    this rank works the whole year on its own).  device_result: nothing is copied to the host — the return value is
    dict(rows_dev [n_tracks, 9*n_steps + 3] (records + candidate index, month, basin index), n_seeds_dev [7*12], rounds), both on
    the round function's device (year-sharded runs keep a year's tracks in HBM until the all-gather of final tracks).
    """
    import torch
    D = ops or globals()['D']          # ops = distributed.Local: this rank works the year on its own (years sharded over the ranks)
    W, rk = D.world(), D.rank()
    width = ROW_VARS * n_steps
    got, total, hist_full, last = [], 0, None, None
    for r in range(max_rounds):
        cand0 = D.round_block(r, per_rank, rk, W)
        while True:
            out = round_fn(cand0, per_rank)
            if 'rows' not in out:
                out = _legacy_round(out, cand0, n_steps, torch)
            # ---- the one host synchronisation of the round: (accepted, overflowed) of every rank
            bad = out['bad'].reshape(-1)
            if bad.numel() < 2:
                bad = torch.cat([bad, torch.zeros(1, dtype=bad.dtype, device=bad.device)])
            pairs = D.allgather_ints(torch.cat([out['count'].reshape(1), bad[:2]]))
            counts, bads, drops = [p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs]
            if not (sum(bads) or sum(drops)):
                break
            # Decided collectively — the counts travelled in the all-gather — so every rank takes the same branch and none
            # is left waiting in a collective.
            if sum(drops):
                # more seeds passed on some rank than its dense batch holds (GpuRound sizes the batch from the measured
                # pass rate): every rank widens its batch to the whole candidate block and the round runs again — the same
                # candidates, the same Philox streams, the same results
                if not (hasattr(round_fn, 'grow_capacity') and round_fn.grow_capacity()):
                    raise RuntimeError('%d passing seeds did not fit the round\'s storm capacity' % sum(drops))
                continue
            # A storm needed more accepted RK steps than its step record holds (the reference's solve_ivp is unbounded):
            # the round function doubles the record and the round is integrated again, or, if it cannot grow, every rank
            # raises.
            # Whether the record CAN grow depends on a rank's free HBM (GpuRound.can_grow), so that is decided collectively
            # as well: every rank grows, or every rank raises — none is left waiting in the next round's all-gather.
            can = hasattr(round_fn, 'grow') and (not hasattr(round_fn, 'can_grow') or bool(round_fn.can_grow()))
            if hasattr(round_fn, 'can_grow'):
                can = all(f[0] for f in D.allgather_ints(torch.tensor([1 if can else 0], dtype=torch.int64, device=bad.device)))
            if not (can and round_fn.grow()):
                raise RuntimeError('%d storms needed more accepted RK steps than the step record holds; raise '
                                   'namelist.gpu_max_rk_steps (tcr_params.max_rk_steps)' % sum(bads))
        if counts[rk] > out['rows'].shape[0]:
            # more accepted tracks than this rank's survivor buffer holds: grow it and pack again (local, no collective)
            if not hasattr(round_fn, 'repack'):
                raise RuntimeError('accept_loop: accepted %d tracks but the round packs only %d' % (counts[rk], out['rows'].shape[0]))
            out['rows'] = round_fn.repack(counts[rk])
        # ---- the data-path collective: all-gather of this round's survivor records, device to device
        gathered, _ = D.allgather_rows(out['rows'], None, counts=counts)
        if out.get('unordered'):
            # the round integrated its storms in locality order (pipeline.select_passed): back to candidate order — rank
            # blocks are contiguous and ascending, so one sort of the round's few accepted rows by their candidate index
            gathered = gathered[torch.argsort(gathered[:, width])]
        else:
            gathered = gathered.clone() if W == 1 else gathered     # one rank: a view of the round's own (reused) buffer
        got.append(gathered)
        total += sum(counts)
        last = out
        if total >= n_tracks:
            break
        h = out['hist']()       # rounds before the last one count in full: every candidate in them precedes the cutoff
        hist_full = h if hist_full is None else hist_full + h
    else:
        raise RuntimeError('accept_loop: quota not reached after %d rounds' % max_rounds)
    rows = torch.cat(got)[:n_tracks]
    # rank blocks are contiguous and ascending, so rank order == candidate order
    cutoff = rows[-1, width]                     # the candidate that completed the quota (device scalar)
    # ---- n_seeds: counted candidates with index <= cutoff (compute.py:167), summed over ranks
    h = last['hist'](cutoff)
    t_seeds = h if hist_full is None else hist_full + h
    D.allreduce_sum_(t_seeds)
    if device_result:
        assert rows.shape[0] < 2 or bool((rows[1:, width] > rows[:-1, width]).all())
        return dict(rows_dev=rows, n_seeds_dev=t_seeds.double(), rounds=r + 1)
    # the result leaves the device here, once — into the round function's pinned buffer when it has one (GpuRound.host_rows):
    # a transfer into fresh pageable memory has the driver pin 26 MB of untouched pages per year first
    host = round_fn.host_rows(rows) if hasattr(round_fn, 'host_rows') else rows.cpu().numpy()
    cand = host[:, width].astype(np.int64)
    assert (np.diff(cand) > 0).all() if len(cand) > 1 else True
    return dict(rows=host[:, :width], month=host[:, width + 1].astype(np.int64), basin_idx=host[:, width + 2].astype(np.int64),
                cand=cand, n_seeds=t_seeds.cpu().numpy().reshape(len(BASIN_IDS), 12), rounds=r + 1)


class GpuRound:
    """round_fn backed by the device pipeline.  A round — seed → select → locality order → integrate → stats → select
    accepted → pack (+ candidate index / month / basin columns) → n_seeds histogram — is ONE library call
    (`DevicePipeline.round` = tcr_round_dev), optionally replayed from a captured hipGraph after its first use
    (`namelist.gpu_round_graph`, off by default: it saves host time only); the number of passing seeds, of accepted tracks and of overflowed step records stay
    device scalars, and nothing in a round synchronises with the host."""

    def __init__(self, engine, year, per_rank, experiment_seed=None, max_storms=None):
        import torch
        from . import _lib
        from .pipeline import DevicePipeline
        self.torch = torch
        self.eng = engine
        self.year = int(year)
        self.seed = experiment_seed
        self.per_rank = int(per_rank)
        self.unordered = bool(getattr(engine.nl, 'gpu_locality_order', True))
        self.graph = bool(getattr(engine.nl, 'gpu_round_graph', False))
        # The dense batch (rows, step records, forcing tables: ~63 kB per storm) is sized for the seeds that pass, not for
        # the candidates: the pass rate is measured once on a throw-away block of candidates (18-28 % on the synthetic
        # basins) and the batch gets 1.3x that + 1 024; a round in which more pass is run again at full width
        # (`grow_capacity`, decided collectively in accept_loop).
        if max_storms is None:
            probe = DevicePipeline(engine, min(self.per_rank, 1 << 16), 64)
            probe.seed_round(self.year, 10**15, experiment_seed=experiment_seed)
            p_pass = float(((probe.cand['seed_flags'][:probe.n_cand] & 2) != 0).double().mean().item())
            del probe
            max_storms = int(self.per_rank * min(1.0, 1.3 * p_pass)) + 1024
            self.n_expected = int(self.per_rank * p_pass)
        self.B = int(min(self.per_rank, max(64, max_storms)))
        self.n_expected = int(getattr(self, 'n_expected', 0))
        self._build()

    def _build(self):
        from . import _lib
        from .pipeline import DevicePipeline
        torch, engine, per_rank = self.torch, self.eng, self.per_rank
        self.pipe = DevicePipeline(engine, per_rank, self.B, tc_rows_only=True, dtype=getattr(engine.nl, 'gpu_dtype', 'f64'),
                                   sort_storms=self.unordered)
        dev = self.pipe.dev
        # survivor records: 26 kB per accepted track.  1-6 % of a round's candidates are accepted, so the buffer is sized
        # for a quarter of them (65 536 candidates: 0.43 GB instead of 1.7) and grows — `repack` — in the round that needs more
        self.cap = min(self.B, max(1024, per_rank // 4))
        self.packed = torch.zeros(self.cap, ROW_VARS * engine.n_steps + N_META, dtype=torch.float64, device=dev)
        self.stats = torch.zeros(_lib.N_STATS, dtype=torch.int64, device=dev)
        self.hist_round = torch.zeros(len(BASIN_IDS) * 12, dtype=torch.int64, device=dev)
        self.hist_cut = torch.zeros(len(BASIN_IDS) * 12, dtype=torch.int64, device=dev)

    def set_year(self, year):
        """Reuse the buffers (and the captured round) for another year: the year is part of the round key, not of the graph."""
        self.year = int(year)
        return self

    def grow_capacity(self):
        """A round had more passing seeds than the dense batch holds: widen it to the whole candidate block (then nothing can
        be dropped).  False when it already is that wide."""
        if self.B >= self.per_rank:
            return False
        self.B = self.per_rank
        self._build()
        return True

    def can_grow(self):
        """This rank's HBM holds a doubled step record (per_rank x max_rk_steps x ~400 B in at most half of what is free).
        A local fact: accept_loop combines it over the ranks before anyone grows."""
        cur = int(self.eng.params.max_rk_steps) or 64
        free = self.torch.cuda.mem_get_info(self.pipe.dev)[0] if self.pipe.dev.type == 'cuda' else 1 << 62
        return self.B * 2 * cur * 400 <= free // 2

    def grow(self):
        """Double the per-storm step record (tcr_params.max_rk_steps).  False once it is at the ABI's limit (the same on every
        rank); whether the memory is there is `can_grow`, which the caller has already agreed on with the other ranks."""
        return self.eng.grow_step_record()

    def repack(self, need):
        """The last round accepted more tracks than the survivor buffer holds: grow it and pack again (the tracks are
        still in the pipeline's planes).  Local to this rank, no collective."""
        self.cap = int(min(self.pipe.B, max(need, 2 * self.cap)))
        self.packed = self.torch.zeros(self.cap, self.packed.shape[1], dtype=self.torch.float64, device=self.pipe.dev)
        cand0, count = self._last
        cap = min(self.cap, count)
        self.pipe.pack_accepted_meta(self.packed, cap, cand0)
        return self.packed[:cap]

    def host_rows(self, rows):
        """Device rows [n, width] -> a NumPy view of this round function's pinned host buffer (one DMA at link speed, no page
        pinning).  The view is valid until the next call: `rows_to_tuple` copies what it keeps."""
        torch = self.torch
        if rows.device.type != 'cuda':
            return rows.numpy()
        pin = getattr(self, '_pin', None)
        if pin is None or pin.shape[0] < rows.shape[0] or pin.shape[1] != rows.shape[1]:
            pin = self._pin = torch.empty((max(rows.shape[0], 1), rows.shape[1]), dtype=rows.dtype, pin_memory=True)
        out = pin[:rows.shape[0]]
        out.copy_(rows, non_blocking=True)
        torch.cuda.current_stream(rows.device).synchronize()
        return out.numpy()

    def release(self):
        """Drop the device buffers (the engine's workspaces stay)."""
        self.pipe = self.packed = self._pin = None

    def __call__(self, cand0, count):
        p = self.pipe
        self.stats.zero_()
        cap = min(self.cap, count)
        p.round(self.year, cand0, count, min(self.B, count), self.seed, exact_count=True, stats=self.stats, accepted=True,
                packed=self.packed, pack_cap=cap, seed_hist=self.hist_round, graph=self.graph,
                n_expected=min(self.n_expected, self.B))
        self._last = (cand0, count)
        hist_full = self.hist_round.double()          # a copy: the next round overwrites the buffer

        def hist(cutoff=None):
            if cutoff is None:
                return hist_full
            return p.seed_hist(self.hist_cut, cutoff).double()
        # stats[8]: storms whose step record overflowed (status -3) among the storms of this round; stats[9]: passing seeds
        # that did not fit the dense batch
        return dict(rows=self.packed[:cap], count=p.n_accepted, bad=self.stats[8:10], hist=hist, unordered=self.unordered)


def rows_to_tuple(res, n_steps):
    """Survivor records -> the reference's 9-tuple layout (compute.py:124-133, 210)."""
    rows = res['rows']
    n = rows.shape[0]
    ns = n_steps
    tc_lon, tc_lat, tc_v, tc_m, tc_vmax = (rows[:, k * ns:(k + 1) * ns].copy() for k in range(5))
    # one copy, always: `rows` may be a view of GpuRound's reused pinned buffer, and with a single row the slice of a wider row
    # IS contiguous (np.ascontiguousarray would hand back a view that the next year's round overwrites — ADVICE r5)
    tc_env_wnds = np.array(rows[:, 5 * ns:9 * ns], dtype=np.float64, order='C', copy=True).reshape(n, ns, 4)
    tc_month = res['month'].astype(np.float64)
    tc_basin = np.array([BASIN_IDS[i] for i in res['basin_idx']], dtype='U2')
    return (tc_lon, tc_lat, tc_v, tc_m, tc_vmax, tc_env_wnds, tc_month, tc_basin, res['n_seeds'])


def default_per_rank(nl, n_tracks):
    """Candidates a rank seeds per round: enough for the quota in one or two rounds, at most namelist.gpu_candidate_round."""
    return int(max(4096, min(nl.gpu_candidate_round, 64 * n_tracks)))


def run_tracks(year, n_tracks, b, engine=None, env=None, nl=None, per_rank=None, info=None, round_fn=None, ops=None, device_result=False):
    """Generate n_tracks TC tracks in basin b for one year (reference: compute.py:64-210).

    Returns (tc_lon, tc_lat, tc_v, tc_m, tc_vmax, tc_env_wnds, tc_month, tc_basin, n_seeds).
    ``engine`` is a staged TCEngine; if omitted one is built from ``env`` (a field set shaped
    like ``synthetic.SyntheticEnv``) on this rank's GPU.  ``info`` (a dict, optional) receives what the
    reference's loop keeps implicit: ``cand`` (global candidate index of every returned track) and ``rounds``.
    ``round_fn``: a GpuRound of the same engine to reuse (its buffers and its captured round) for this year.
    ``ops``: `distributed.Local` to work the year on this rank alone (see `accept_loop`).
    ``device_result``: return accept_loop's device-resident result instead of the 9-tuple (`_allgather_years` consumes it).
    """
    nl = nl or default_namelist
    basin_id = b.basin_id if isinstance(b, TC_Basin) else b
    own = engine is None
    if own:
        from .engine import TCEngine
        import os
        engine = TCEngine(basin_id, device=D.local_device(os.environ.get('LOCAL_RANK', '0')), nl=nl).stage_env(env)
    if round_fn is not None:
        rf, per_rank = round_fn.set_year(year), round_fn.per_rank
    else:
        per_rank = int(per_rank or default_per_rank(nl, n_tracks))
        rf = GpuRound(engine, year, per_rank)
    res = accept_loop(rf, n_tracks, per_rank, engine.n_steps, ops=ops, device_result=device_result)
    if device_result:
        if own:
            engine.close()
        return res
    if info is not None:
        info.update(cand=res['cand'], rounds=res['rounds'], per_rank=per_rank)
    if own:
        engine.close()
    return rows_to_tuple(res, engine.n_steps)


def run_downscaling(basin_id, env=None, nl=None, out_dir=None):
    """Run every year of the namelist for one basin and write the track file
    (reference: compute.py:216-270).  Returns the output file name (rank 0) or None.

    The reference hands the years to dask workers, one process per year (compute.py:223-230).  Here the years are
    independent too — a year's tracks depend on (experiment seed, year) alone — and on one GPU
    ``namelist.gpu_years_in_flight`` (default 3) of them are in flight: each worker thread owns a context with its own
    month slots, stages its year's fields while the other's rounds run on the GPU, and reuses one set of round buffers
    (and one captured round) for all its years; finished years go to a background writer that puts their rows at their
    final place in the track file (`io.TrackFileWriter`), so the file is complete shortly after the last year.  With
    several ranks (one per GPU) a year's candidate blocks are sharded over them and the years run one after another:
    the per-round collectives must be issued in one order.
    """
    from . import io as tio
    nl = nl or default_namelist
    import os
    import threading
    import torch
    from . import _lib
    from .engine import TCEngine
    _lib.lib()                 # (loaded once, before the worker threads)
    b = TC_Basin(basin_id, nl)
    if env is None:
        env = tio.load_env(nl)
    s = time.time()
    years = list(range(nl.start_year, nl.end_year + 1))
    yearly = hasattr(env, 'for_year')
    device = D.local_device(os.environ.get('LOCAL_RANK', '0'))
    W, rk = D.world(), D.rank()
    # Several ranks: with at least as many years as ranks the YEARS are sharded (rank r works years r, r + W, ... on its own —
    # what the reference's one-process-per-year fan-out is — and the final tracks are all-gathered once); with fewer years
    # a year's candidate blocks are sharded and the years run one after another.
    shard_years = D.collective() and len(years) >= W and bool(getattr(nl, 'gpu_shard_years', True))
    mine = list(range(rk, len(years), W)) if shard_years else list(range(len(years)))
    ops = D.Local if shard_years else None
    n_workers = max(1, min(int(getattr(nl, 'gpu_years_in_flight', 3)), len(mine))) if (not D.collective() or shard_years) else 1
    out = [None] * len(years)
    writer = tio.TrackFileWriter(years, b, nl, out_dir) if (rk == 0 and not D.collective()) else None
    errors = []

    def work(w):
        try:
            dev = torch.device('cuda', device)
            # (a worker's launches go to its own stream; the legacy default stream would order the workers behind each other)
            stream = torch.cuda.Stream(device=dev) if n_workers > 1 else torch.cuda.current_stream(dev)
            eng = TCEngine(basin_id, device=device, nl=nl)
            eng.schedule(int(getattr(nl, 'gpu_storms_per_lane', 1 if n_workers == 1 else 2)))
            if not yearly:
                eng.stage_env(env)
            rf = None
            with torch.cuda.stream(stream):
                for i in mine[w::n_workers]:
                    if errors:
                        break
                    yr = years[i]
                    if yearly:
                        eng.stage_env(env.for_year(yr))
                    if rf is None:
                        rf = GpuRound(eng, yr, default_per_rank(nl, nl.tracks_per_year))
                        if n_workers > 1:
                            rf.graph = False         # stream capture does not tolerate the other workers' field uploads (tcrisk_hip.h)
                    # (year-sharded: the year's tracks stay in HBM until the all-gather of final tracks)
                    out[i] = run_tracks(yr, nl.tracks_per_year, b, engine=eng, nl=nl, round_fn=rf, ops=ops, device_result=shard_years)
                    if writer is not None:
                        writer.put(i, out[i])
                stream.synchronize()
            if rf is not None:
                rf.release()
            eng.close()
        except BaseException as e:          # re-raised by the caller's thread
            errors.append(e)

    if n_workers == 1:
        work(0)
    else:
        threads = [threading.Thread(target=work, args=(w,), name='year-worker-%d' % w) for w in range(n_workers)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if shard_years:
        # every rank reaches the collective, also after a failure of its own (the others must not wait for ever)
        failed = D.sum_over_ranks(1.0 if errors else 0.0, torch.device('cuda', device))
        if errors:
            raise errors[0]
        if failed:
            raise RuntimeError('run_downscaling: %d rank(s) failed' % int(failed))
        out = _allgather_years(out, mine, years, nl, torch.device('cuda', device), to_host=(rk == 0))
    elif errors:
        if writer is not None:
            writer.abort()
        raise errors[0]
    fn = None
    if rk == 0:
        fn = writer.close() if writer is not None else tio.write_tracks(out, years, b, nl, out_dir)
        print('Saved %s' % fn)
        print(time.time() - s)
    D.barrier()
    return fn


def _allgather_years(out, mine, years, nl, dev, to_host=True):
    """The all-gather of final tracks for year-sharded runs (`north_star`: "storms shard trivially across the GPUs with an
    RCCL all-gather of final tracks"; the reference collects its dask workers' 9-tuples, util/compute.py:233-242).

    out[i], i in mine: accept_loop's device result for year i — rows_dev [tracks, 9 x n_steps + 3] (survivor records +
    candidate index, month, basin index) and n_seeds_dev [7 x 12], both in HBM.  Every rank copies its years into one device
    block, ONE collective moves everything device to device, and only a rank that asks for it (`to_host`: rank 0, which
    writes the file) brings the result to the host and rebuilds the per-year 9-tuples in year order; the others drop it.
    Returns the list of 9-tuples, or None."""
    import torch
    W = D.world()
    T = int(nl.tracks_per_year)
    k_max = -(-len(years) // W)
    ns = int(nl.total_track_time_days * 24 * 3600 / nl.output_interval_s) + 1
    width = ROW_VARS * ns + N_META
    n_b = len(BASIN_IDS) * 12
    block = torch.zeros(k_max, T * width + n_b, dtype=torch.float64, device=dev)
    for j, i in enumerate(mine):
        rows = out[i]['rows_dev']
        assert tuple(rows.shape) == (T, width), (tuple(rows.shape), (T, width))
        block[j, :T * width] = rows.reshape(-1)
        block[j, T * width:] = out[i]['n_seeds_dev'].reshape(-1)
    got, counts = D.allgather_year_blocks(block, len(mine), dev)
    if not to_host:
        return None
    got = got.cpu().numpy()
    res = [None] * len(years)
    for r in range(W):
        for j in range(counts[r]):
            i = r + j * W
            rows = got[r, j, :T * width].reshape(T, width)
            res[i] = rows_to_tuple(dict(rows=rows[:, :ROW_VARS * ns], month=rows[:, ROW_VARS * ns + 1].astype(np.int64),
                                        basin_idx=rows[:, ROW_VARS * ns + 2].astype(np.int64),
                                        n_seeds=got[r, j, T * width:].reshape(len(BASIN_IDS), 12).copy()), ns)
    return res
