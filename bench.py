#!/usr/bin/env python3
"""Headline benchmark: storm-steps/s of the seed → integrate → post hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--storms B]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One *step* = one pass of the hot path over one batch on every rank: device-side
seeding of a round of candidates (compute.py:134-175), order-preserving selection
of the first B that pass, the fused DP5(4) integration with env-wind recompute,
vmax and accept flags (compute.py:176-209), and — for N > 1 — the RCCL all-gather
of the accepted tracks of that batch.  `value` is storm-steps of all ranks / wall
time of the K steps.

--scaling strong (default; BASELINE's metric and config 4 as worded: "storm-steps/sec (100k-storm ensemble) at 1/2/4/8
MI355X", "100k storms sharded across 8 GPUs"): the ENSEMBLE is fixed — one step = one ensemble drawn from a fixed block of candidates
(sized so that ~B seeds pass), the candidate block sharded over the ranks as
`compute.run_tracks` shards a round; every rank integrates all the passing seeds of
its sub-block, and the accepted tracks of the ensemble are all-gathered once.  The set
of storms of an ensemble does not depend on the number of ranks, so the storm-step
total of a run is bit-for-bit the same at every N (tests/test_seeding.py checks 1 vs 2).  At N = 1 this is the same
100 000-storm step as --scaling weak.
--scaling weak: per-GPU work is fixed — B storms per rank and step, every step's accepted tracks all-gathered.

A step is ONE library call per rank (tcr_round_dev: seed → select → locality order → forcing table → integrate →
post-processing → stats [→ select accepted → pack]); --graph on replays it from a captured hipGraph (off by default: it halves
the host time per step and changes no GPU time).

A storm-step is one hourly output interval of one live storm.
Inputs (fields) are resident in HBM before the timed region; candidates are drawn
on the device inside it.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0      # MI355X_MICROARCH.md: ~6.3 TB/s achievable
BYTES_PER_RHS = 704.0            # 20 lookups x 4 corners x 8 B + Fs 64 B
BYTES_PER_SAMPLE = 520.0         # 14 lookups x 32 B + 9 outputs x 8 B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--storms', type=int, default=100_000, help='weak: storms integrated per GPU per step; strong: storms per '
                                                                'ensemble (= per step), sharded over the GPUs')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='strong')
    ap.add_argument('--graph', choices=('auto', 'on', 'off'), default='off',
                    help='replay every step from a captured hipGraph (tcr_round_dev use_graph); auto: when a rank integrates '
                         'fewer than 50 000 storms per step.  Off by default: the replay halves the host time per step '
                         '(0.08 -> 0.04 ms) and changes no GPU time (profiles/r04_small_batches.txt)')
    ap.add_argument('--storms-per-lane', type=int, default=3,
                    help='launch shape of batches that do not fill the chip (tcr_schedule_set): 1 = one integrator lane per storm, '
                         '3 = a third of the waves, lanes take storms in turn (profiles/r04_small_batches.txt)')
    ap.add_argument('--stage-trace', action='store_true', help='record a HIP event behind every stage of every (directly enqueued) round of '
                    'the timed region and report the per-stage stream time under load (tcr_stage_trace_*; implies --graph off)')
    ap.add_argument('--staged', action='store_true', help="round 3's step: one library call per stage instead of tcr_round_dev")
    ap.add_argument('--events', choices=('auto', 'on', 'off'), default='auto',
                    help='HIP events around the table / integrator / post-processing of every batch INSIDE the timed region (they feed '
                         'roofline.pipelined_events and gpu_active_s).  Free at 100 000 storms per batch (1.188 vs 1.186 ms), 6 %% of a '
                         '12 500-storm step (0.233 vs 0.219 ms, round 6): auto = on when a rank integrates >= 50 000 storms per step.  The '
                         'headline roofline figure does not need them: it comes from isolated batches after the timed region')
    ap.add_argument('--dup-seed', type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument('--dup-select', type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument('--basin', default='GL')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--order', choices=('cells', 'candidate'), default='cells',
                    help='cells: the dense batch in locality order (2-degree genesis cells, tcr_cell_order_dev; DESIGN.md §9); '
                         'candidate: candidate order')
    ap.add_argument('--streams', type=int, default=None,
                    help='HIP streams the steps are round-robined over (each with its own context and buffers): '
                         'the low-occupancy tail of one batch overlaps the next batch.  Default 12; 16 for the small '
                         'per-rank batches of --scaling strong at N > 1 (with as many hardware queues, see below)')
    ap.add_argument('--cpu-budget', type=float, default=12.0)
    ap.add_argument('--traffic', type=float, default=None,
                    help='HBM bytes per integrate launch from a separate rocprofv3 --pmc pass (see profiles/)')
    ap.add_argument('--dtype', choices=('f64', 'f32'), default='f64',
                    help='f32: the fp32 variant of the path (BASELINE config 5): fp32 fields / state / RHS / rows, fp64 time and controller')
    ap.add_argument('--static-res', type=float, default=0.125, choices=(0.25, 0.125),
                    help='grid of the synthetic land / bathymetry planes: 0.125 (default since round 5) = the grid and type of the '
                         "reference's own intensity/data/land.nc (int8, 1440 x 2880); 0.25 = SURVEY.md section 8d's float64 planes "
                         '(rounds 1-4)')
    ap.add_argument('--shape', choices=('era5', 'gfdl'), default='era5',
                    help='era5: 1-degree wind = thermo grid; gfdl: 2 x 2.5 degree wind grid, 1 x 1.25 degree thermo grid (BASELINE config 5)')
    ap.add_argument('--bathy-kind', choices=('i16', 'f32', 'f64'), default=None,
                    help='what the synthetic bathymetry holds (default: f64 at 0.25 degrees, whole metres at 0.125)')
    ap.add_argument('--bathy-res', type=float, default=None, choices=(0.25, 0.125),
                    help='put the bathymetry on its own grid (two independent interpolators, intensity/geo.py:9-34); default: the land grid')
    ap.add_argument('--static-store', choices=('auto', 'f64'), default='auto',
                    help='auto: exact narrow storage of land / bathymetry where the values allow it (tcr_static_upload); '
                         'f64: the fp64 planes of rounds 1-4')
    ap.add_argument('--rows', choices=('tc', 'all'), default='tc',
                    help="tc: env winds / vmax / rows only for storms that pass accept test 1, as the reference does "
                         "(compute.py:190-204); all: rows for every integrated storm (round 1's workload)")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit('bench.py: --gpus must be >= 1')
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    if env_world is not None and int(env_world) != args.gpus:
        # never report a rank count other than the one asked for (`--gpus 8` without a launcher printed n_gpus 1)
        sys.exit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks' % (args.gpus, env_world))
    # A batch is latency-bound by its longest storm (~300 sequential evaluations, ~2 ms), so throughput comes from batches
    # in flight.  ROCm multiplexes a process's streams onto GPU_MAX_HW_QUEUES = 4 hardware queues by default; the small
    # per-rank batches of a sharded ensemble need more of them in flight than that (measured on one MI355X, 12 500 storms
    # per batch: 0.57 ms per step with 4 streams / 4 queues at any stream count, 0.44 with 8 / 8, 0.38 with 16 / 16;
    # at 100 000 storms per batch 8 streams on 8 queues are 3.5 % faster than 4 on 4 — 1.31 against 1.36 ms per step — and 12 on 12
    # no better; mismatched counts, e.g. 4 streams on 8 queues, are 30 % slower.  Round 4, once every small kernel fits next to an
    # integrator wave: 8 / 10 / 12 / 16 streams give 1.41 / 1.37 / 1.37 / 1.39 ms per step over the driver's 20 steps and 1.355 / 1.348 /
    # 1.326 / 1.32 over 40 — 12 is the default).  Must be set before the HIP runtime starts.
    if args.streams is None:
        args.streams = 16 if (args.scaling == 'strong' and int(os.environ.get('WORLD_SIZE', '1')) > 1) else 12
    if args.streams > 4:
        os.environ.setdefault('GPU_MAX_HW_QUEUES', str(min(args.streams, 32)))

    import numpy as np
    import torch
    from tropical_cyclone_risk_amd import distributed as D
    from tropical_cyclone_risk_amd import synthetic
    from tropical_cyclone_risk_amd.engine import TCEngine
    from tropical_cyclone_risk_amd.pipeline import DevicePipeline

    rank, world, local = D.init_from_env()
    assert world == args.gpus
    local = D.local_device(local)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    year = 2000

    env = synthetic.make_env(args.shape, static_res=args.static_res, bathy_kind=args.bathy_kind)
    if args.bathy_res is not None and args.bathy_res != args.static_res:
        import dataclasses
        e2 = synthetic.make_env(args.shape, static_res=args.bathy_res, bathy_kind=args.bathy_kind or ('i16' if args.static_res == 0.125 else 'f64'))
        env = dataclasses.replace(env, bathy=e2.bathy, blon=e2.hlon, blat=e2.hlat)
    from tropical_cyclone_risk_amd import namelist as _nl
    _nl.gpu_static_store = args.static_store
    n_str = max(1, args.streams)
    # many batches in flight: batches that do not fill the chip run with lanes that take ~3 storms in turn (tcr_schedule_set)
    engs = [TCEngine(args.basin, device=local).stage_env(env).schedule(args.storms_per_lane) for _ in range(n_str)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_str)]
    eng = engs[0]
    B = args.storms
    ns = eng.n_steps

    # ---- calibrate the seed pass rate once (untimed) so that a round of C candidates
    # contains at least B passing seeds with a wide margin
    probe = DevicePipeline(eng, 1 << 18, 1024)     # (seeding is fp64 in both modes)
    probe.seed_round(year, 10**12)
    torch.cuda.synchronize()
    p_pass = float(((probe.cand['seed_flags'] & 2) != 0).double().mean().item())
    del probe
    strong = args.scaling == 'strong'
    if strong:
        # one ensemble = the passing seeds of a fixed block of C_ens candidates (expected: args.storms of them); rank r owns
        # the r-th of `world` equal sub-blocks and integrates every passing seed in it (n_dev = the device-side count)
        C_ens = -(-int(round(args.storms / max(p_pass, 1e-3))) // 8) * 8        # the same block at 1, 2, 4 and 8 ranks
        C = -(-C_ens // world)                                                  # candidates per rank and ensemble
        C_ens = C * world
        # capacity: the number of passing seeds of a sub-block is binomial (sigma = sqrt(C p (1 - p)) < sqrt(expected)): the expected
        # count + 8 sigma + 256; a sub-block that still has more drops seeds, which the run checks on the device (acc[9])
        exp_rank = args.storms / world
        B = int(exp_rank + 8.0 * exp_rank ** 0.5) + 256
    else:
        C = int(B / max(p_pass, 1e-3) * 1.15) + 4096
    pipes = [DevicePipeline(e, C, B, sort_storms=(float(os.environ.get('TCR_CELL_DEG', '2')) if args.order == 'cells' else False), tc_rows_only=(args.rows == 'tc'), dtype=args.dtype) for e in engs]
    global BYTES_PER_RHS, BYTES_PER_SAMPLE
    if args.dtype == 'f32':
        BYTES_PER_RHS, BYTES_PER_SAMPLE = 352.0, 260.0          # SURVEY.md §8d, fp32 mode
    pipe = pipes[0]

    # storm-steps, nfev, samples, accepted, is_tc, is_tc samples, rounds with < B passing seeds, storms, step-record overflows,
    # passing seeds a batch had no room for (tcr_stats_dev)
    acc = torch.zeros(10, dtype=torch.int64, device=dev)
    n_exp = int(args.storms / world) if strong else 0      # strong: a rank's batch has 8 sigma + 256 rows to spare
    use_graph = (args.graph == 'on' or (args.graph == 'auto' and B < 50_000)) and not args.stage_trace
    row = 9 * ns
    # N > 1: all-gather of every batch's final (accepted) tracks through distributed.DeferredRowGather:
    # nothing in a step waits on the host — batch k's count is read back only when batch k + n_str is
    # issued, and only then is its row all-gather launched on RCCL's stream, n_str batches behind compute.
    cap = max(1024, int(0.2 * B))               # accepted fraction is ~6 %
    gather = D.DeferredRowGather(cap, row, dev, lag=n_str) if D.collective() else None

    def step(k):
        with torch.cuda.stream(streams[k % n_str]):
            _step(k, pipes[k % n_str])

    def _step(k, pipe, graph=None):
        # warm-up steps run exactly the same code; the accumulators are zeroed after them
        graph = use_graph if graph is None else graph
        cand0 = D.round_block(k, C, rank, world)                       # = k * C * world + rank * C
        if args.staged:
            # (--dup-seed / --dup-select: run a stage several times — same inputs, same outputs — to read its marginal cost in
            # the pipelined step off the difference; measurement aid)
            for _ in range(args.dup_seed):
                pipe.seed_round(year, cand0)
            for _ in range(args.dup_select):
                pipe.select_passed(B)
            pipe.integrate(B, n_dev=pipe.n_passed if strong else None)
            pipe.add_stats(acc, n_dev=pipe.n_passed)
            if gather is not None:
                buf = gather.buffer()                # waits (on this stream) for the gather that last read it
                pipe.select_accepted()
                pipe.pack_accepted(buf, cap)
                gather.submit(pipe.n_accepted)
        elif gather is not None:
            # (the pack goes into one of the gather's rotating buffers: kept outside the round, whose captured graph is keyed
            # by its buffers — one graph per pipe instead of one per (pipe, buffer))
            buf = gather.buffer()
            pipe.round(year, cand0, C, B, exact_count=strong, stats=acc, accepted=True, graph=graph, n_expected=n_exp)
            pipe.pack_accepted(buf, cap)
            gather.submit(pipe.n_accepted)
        else:
            pipe.round(year, cand0, C, B, exact_count=strong, stats=acc, graph=graph, n_expected=n_exp)

    def drain():
        if gather is not None:
            gather.drain()

    events_on = args.events == 'on' or (args.events == 'auto' and B >= 50_000)
    for e in engs:
        e.timing_enable(events_on)
    w_eff = max(args.warmup, n_str)          # every stream runs the full step at least once untimed
    for k in range(w_eff):
        step(k)
    drain()
    torch.cuda.synchronize()
    acc.zero_()
    D.barrier(); torch.cuda.synchronize()
    for e in engs:
        e.timing_enable(events_on)   # resets the event record: only the K timed steps count
        if args.stage_trace:
            e.stage_trace(True)
    if gather is not None:
        gather.rows_gathered = gather.rows_clipped = 0
    t0 = time.perf_counter()
    for k in range(w_eff, w_eff + args.steps):
        step(k)
    t_issue = time.perf_counter() - t0            # host time to enqueue the K steps (the GPU runs behind it)
    drain()
    torch.cuda.synchronize(); D.barrier()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, dev)
    rows_gathered, rows_clipped = (gather.rows_gathered, gather.rows_clipped) if gather is not None else (None, None)

    def sum_timings():
        tot = dict(fourier_ms=0.0, integrate_ms=0.0, post_ms=0.0, calls=0)
        for e in engs:
            try:
                m1 = e.timing_sum()
            except Exception:
                continue             # a stream that saw no timed step
            for kk in tot:
                tot[kk] += m1[kk]
        return tot
    ms = sum_timings()              # (calls == 0 when the steps were graph replays: those record no events)
    stage_ms = None
    if args.stage_trace:
        tot, nr = {}, 0
        for e in engs:
            d1, n1 = e.stage_trace_sum()
            e.stage_trace(False)
            nr += n1
            for kk, v in d1.items():
                tot[kk] = tot.get(kk, 0.0) + v
        stage_ms = {kk: v / max(nr, 1) for kk, v in tot.items()}
    # SIMD time of the integrator chains as they ran INSIDE the timed region (the last batch of every stream: wave residency
    # summed from the waves' own wall-clock stamps): under load the waves of several batches share the CUs' address paths,
    # so this is what a chain costs the pipeline; `integrate_passes` below is the same for an isolated batch
    pipe_passes = [e.pass_stats() for e in engs]
    # The same kernels once more, one launch at a time on one stream (after the timed region, not
    # part of `value`): with several streams the event-bracketed duration of a launch includes the
    # time it shared the GPU with other batches, so the per-kernel roofline is quoted both ways.
    iso = None
    if True:
        acc_keep = acc.clone()
        for e in engs:
            e.timing_enable(True)
        iso_each = []
        for k in range(w_eff + args.steps, w_eff + args.steps + 5):
            with torch.cuda.stream(streams[0]):
                _step(k, pipes[0], graph=False)          # direct enqueue: the library's timing events are recorded
            torch.cuda.synchronize()
            iso_each.append(engs[0].timing_last())
        drain()                                          # (nothing of the isolated batches stays in flight in the collective backend)
        torch.cuda.synchronize()
        # five batches, one at a time; per component the MEDIAN of the five (round 6: the chip idles between isolated batches and
        # its clock sags on some boxes — one run in four showed a 1.92 ms chain next to 1.6-1.7 in every other; the mean of three
        # carried that into roofline.frac).  `iso` keeps the fields of a timing sum over ONE call.
        med = lambda key: sorted(x[key] for x in iso_each)[len(iso_each) // 2]
        iso = dict(fourier_ms=med('fourier_ms'), integrate_ms=med('integrate_ms'), post_ms=med('post_ms'), calls=1)
        n_iso = len(iso_each)
        iso_counts = [x / n_iso for x in (acc - acc_keep).tolist()]      # per isolated batch
        acc.copy_(acc_keep)
        iso_passes = engs[0].pass_stats()
    D.allreduce_sum_(acc)
    (steps_total, nfev_total, samples_total, accepted_total, tc_total, tc_samples_total, n_short, storms_total, n_overflow,
     n_dropped) = (float(x) for x in acc.tolist())
    emitted_total = tc_samples_total if args.rows == 'tc' else samples_total     # samples k_emit actually produced
    n_short = int(n_short)
    # every passing seed of every sub-block of every step must have fitted its rank's capacity (else storms were dropped):
    # counted on the device by every step's stats kernel (acc[9]), summed over the ranks above
    if strong:
        assert int(n_dropped) == 0, 'strong scaling: %d passing seeds did not fit a rank\'s batch capacity' % int(n_dropped)
    assert int(n_overflow) == 0, '%d storms overflowed their step record (tcr_params.max_rk_steps)' % int(n_overflow)
    value = steps_total / dt

    # ---- roofline of the dominant kernel (k_integrate): algorithmic bytes per launch
    # (704 B x RHS evaluations of that launch) over its HIP-event duration
    # (events recorded on the launch stream by the library).  The per-sample share of the
    # algorithmic bytes (520 B x output samples) is k_emit's and is reported next to it.
    static_tag = '%g/%s' % (args.static_res, args.static_store)
    launches = ms['calls']
    have_events = launches > 0
    if have_events:
        k_ms = ms['integrate_ms'] / launches
        e_ms = ms['post_ms'] / launches
        int_bytes = BYTES_PER_RHS * nfev_total / (launches * world)
    # HIP-event time of every kernel of the timed region, summed over launches and streams (overlapping
    # streams make this exceed the wall time; it shows the GPU was busy even when SMI sampling misses a 50 ms region).
    # Graph replays record no events: the figure is then the isolated batches' kernel time x the timed steps.
    if have_events:
        gpu_active_s = (ms['fourier_ms'] + ms['integrate_ms'] + ms['post_ms']) * 1e-3
    else:
        gpu_active_s = (iso['fourier_ms'] + iso['integrate_ms'] + iso['post_ms']) / iso['calls'] * args.steps * 1e-3
    traffic, traffic_src = (args.traffic, 'command line') if args.traffic is not None else measured_traffic('k_integrate', args.storms if world == 1 else B, args.rows, args.dtype, args.order, static_tag, args.shape)
    # Exclusive duration: the same launch, one batch at a time on one stream right after the timed region
    # (3 batches).  With several streams the event-bracketed duration of a launch in the timed region
    # includes time it shared the GPU with other batches (it can exceed ms_per_step), so that figure is
    # kept under `pipelined_events`, not at the top level.
    ik, ie = iso['integrate_ms'] / iso['calls'], iso['post_ms'] / iso['calls']
    ib = BYTES_PER_RHS * iso_counts[1] / iso['calls']
    eb = BYTES_PER_SAMPLE * (iso_counts[5] if args.rows == 'tc' else iso_counts[2]) / iso['calls']
    achieved = ib / (ik * 1e-3) / 1e9
    e_traffic, e_src = measured_traffic('k_emit', args.storms if world == 1 else B, args.rows, args.dtype, args.order, static_tag, args.shape)
    roof = dict(bound='hbm', kernel='k_integrate', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
                note='a launch = the chain of k_integrate passes of one batch (tail compaction); achieved = %d B x RHS ' % BYTES_PER_RHS +
                     'evaluations of the batch / exclusive HIP-event duration of the chain (one batch at a time on one '
                     'stream, after the timed region: median of five); profiles/ lists k_integrate once per pass',
                algorithmic_bytes_per_launch=ib, launch_ms=ik,
                kernel_ms=dict(fourier=iso['fourier_ms'] / iso['calls'], integrate=ik, post=ie),
                pipelined_events=(dict(note='event-bracketed durations inside the timed region, %d streams overlapping' % n_str,
                                       launch_ms=k_ms, achieved=int_bytes / (k_ms * 1e-3) / 1e9,
                                       frac=int_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       kernel_ms=dict(fourier=ms['fourier_ms'] / launches, integrate=k_ms, post=e_ms))
                                  if have_events else None),
                sustained=dict(note='algorithmic bytes of all k_integrate launches / wall time of the timed region',
                               achieved=BYTES_PER_RHS * nfev_total / world / dt / 1e9,
                               frac=BYTES_PER_RHS * nfev_total / world / dt / 1e9 / HBM_PEAK_GBS),
                emit=dict(kernel='k_screen + k_dense + k_emit + k_flags' if args.rows == 'tc' else 'k_dense + k_emit + k_flags',
                          note='%d B x samples actually emitted (%s) / exclusive duration of the post-processing'
                               % (BYTES_PER_SAMPLE, 'storms that pass accept test 1' if args.rows == 'tc' else 'every storm'),
                          achieved=eb / (ie * 1e-3) / 1e9, frac=eb / (ie * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          algorithmic_bytes_per_launch=eb, launch_ms=ie, traffic=e_traffic, traffic_source=e_src))
    # k_integrate runs as a chain of passes (tail compaction); occupancy of the last isolated batch:
    # wave_ms = summed wave residency, i.e. SIMD time (one integrator wave owns a SIMD's registers)
    wc = sum(p['wave_cycles'] for p in iso_passes)
    simds = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
    st_ms = sum(p['wave_ms'] for p in iso_passes) / simds
    roof['integrate_passes'] = dict(
        passes=len(iso_passes), lane_utilisation=sum(p['lane_cycles'] for p in iso_passes) / max(1, 64 * wc),
        wave_ms=sum(p['wave_ms'] for p in iso_passes), simds=simds, simd_time_ms=st_ms,
        shader_mhz_pass0=iso_passes[0]['shader_mhz'] if iso_passes else None,
        # the same algorithmic bytes over the SIMD time the chain occupies (what it costs a pipelined run)
        achieved_over_simd_time=ib / (st_ms * 1e-3) / 1e9, frac_over_simd_time=ib / (st_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)

    # The whole step against the memory system: fabric-side bytes of ALL kernels of a step (rocprofv3 --pmc, corrected with the
    # factor calibrated on this access pattern) over the measured time per step.  Reducing the integrator's SIMD time (lane
    # utilisation 0.73 -> 0.91) does not move the step, removing the forcing table's traffic does:
    # the step sits at ~0.9 of what scattered 128-byte line fills reach on this chip (tools/calibrate_fetch.hip: 4.26 TB/s).
    # BASELINE.md section 3's figure for the whole step, from the one driver-timed number: algorithmic bytes of everything the
    # timed region did — 704 B x RHS evaluations + 520 B x samples emitted — over its wall time, against the 8 TB/s peak
    alg_gbs = (BYTES_PER_RHS * nfev_total + BYTES_PER_SAMPLE * emitted_total) / world / dt / 1e9
    roof['step_algorithmic'] = dict(achieved=alg_gbs, unit='GB/s', peak=HBM_PEAK_GBS, frac=alg_gbs / HBM_PEAK_GBS,
                                    note='(%d B x RHS evaluations + %d B x emitted samples) of the timed region / its wall time, per GPU'
                                         % (BYTES_PER_RHS, BYTES_PER_SAMPLE))
    step_bytes, step_src = measured_traffic('*', args.storms, args.rows, args.dtype, args.order, static_tag, args.shape) if world == 1 else (None, None)
    if step_bytes and world == 1:
        gbs = step_bytes / (dt / args.steps) / 1e9
        ceil_gbs, ceil_src = scattered_line_fill_ceiling()
        roof['whole_step'] = dict(traffic=step_bytes, traffic_source=step_src, achieved=gbs, unit='GB/s', peak=HBM_PEAK_GBS,
                                  frac=gbs / HBM_PEAK_GBS, achievable=HBM_ACHIEVABLE_GBS, frac_of_achievable=gbs / HBM_ACHIEVABLE_GBS,
                                  scattered_line_fill_ceiling=ceil_gbs, scattered_line_fill_ceiling_source=ceil_src,
                                  frac_of_scattered_line_fill_ceiling=(gbs / ceil_gbs) if ceil_gbs else None,
                                  note='fabric-side traffic (L2 misses incl. Infinity-Cache hits) of every kernel of a step / ms_per_step; '
                                       'achievable = MI355X_MICROARCH.md (~6.3 TB/s of the 8 TB/s peak); the line-fill ceiling is this '
                                       "project's own calibration run (1 GiB of 128-B lines gathered once each, 7 x 16 B per lane)")
    simds_ = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
    pw = [sum(p['wave_ms'] for p in ps) for ps in pipe_passes if ps]
    if pw:
        roof['integrate_passes_pipelined'] = dict(
            simd_time_ms=sum(pw) / len(pw) / simds_, batches=len(pw),
            lane_utilisation=sum(p['lane_cycles'] for ps in pipe_passes for p in ps) / max(1, 64 * sum(p['wave_cycles'] for ps in pipe_passes for p in ps)),
            note='wave residency of the k_integrate passes of the last timed batch of every stream / SIMDs')
    out = None
    backend = D.backend_name()
    # (the sample of the CPU baseline: the first storms of rank 0's last batch)
    sample = cpu_sample(pipe, B) if (rank == 0 and not args.no_cpu_baseline) else None
    if D.collective():
        D.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        # rank 0 times the CPU path at every N — after the timed region and after the process group is gone, so that no other
        # rank waits in a collective meanwhile — and the line is complete in one run
        cpu = cpu_baseline(sample, args) if sample is not None else None
        out = {
            'metric': 'storm-steps/sec (100k-storm ensemble)', 'value': value, 'unit': 'storm-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic', 'gpu_active_s': gpu_active_s,
            'config': {'workload': '%s basin, %s, synthetic %s monthly fields '
                                   '(%s deg land/bathymetry), 15-day tracks, hourly output, '
                                   'device-side seeding, %s; %s' % (args.basin, (
                                       'one ensemble of %d candidates (%.0f storms pass on average) per step, sharded over %d GPU(s), '
                                       'accepted tracks all-gathered once per ensemble' % (C_ens, storms_total / args.steps, world)) if strong
                                       else '%d storms per GPU per step' % B,
                                       'ERA5-shaped (1 deg thermo / wind)' if args.shape == 'era5' else 'GFDL-shaped (2 x 2.5 deg wind, 1 x 1.25 deg thermo)',
                                       '%g' % args.static_res, 'fp64' if args.dtype == 'f64' else 'fp32 fields/state/RHS/rows with fp64 time and step controller', (
                                       'env winds, vmax and rows only for storms that pass accept test 1, as the reference '
                                       'does (compute.py:190-204)' if args.rows == 'tc' else 'rows for every integrated storm')),
                       'static_fields': dict(zip(('storage', 'bytes'), eng.static_info()), resolution_deg=args.static_res),
                       'rows': args.rows, 'batch_order': args.order, 'is_tc_fraction': tc_total / storms_total,
                       'storms_per_step': storms_total / args.steps, 'storm_steps_total': int(steps_total),
                       'emitted_samples_per_step': emitted_total / (args.steps * world),
                       'storms_per_gpu': storms_total / (args.steps * world), 'candidates_per_round': C, 'seed_pass_rate': p_pass,
                       'n_steps_out': ns, 'rounds_short_of_storms': None if strong else n_short, 'streams': n_str,
                       'hw_queues': int(os.environ.get('GPU_MAX_HW_QUEUES', '4')), 'storms_per_lane': args.storms_per_lane,
                       'timed_region_events': bool(events_on),
                       'step_call': 'staged (one library call per stage)' if args.staged else ('tcr_round_dev, replayed from a hipGraph' if use_graph else 'tcr_round_dev, direct enqueue'),
                       'graph_replays': sum(p.graph_stats()['replays'] for p in pipes),
                       'stage_ms_under_load': stage_ms,
                       'allgather': ('accepted tracks of every %s all-gathered once (26 kB records, backend %s, one collective per step, '
                                     '%d steps behind compute)' % ('ensemble' if strong else 'step',
                                                                   backend, n_str)) if gather is not None else 'none (one GPU)',
                       'collectives': ('forced through a one-rank process group (TCR_FORCE_COLLECTIVES=1), backend %s' % backend) if (world == 1 and gather is not None)
                                      else (backend if world > 1 else 'none'),
                       'warmup_effective': w_eff, 'host_issue_ms_per_step': t_issue / args.steps * 1e3,
                       'storm_steps_per_storm': steps_total / storms_total,
                       'rhs_per_storm_step': nfev_total / max(steps_total, 1),
                       'accepted_fraction': accepted_total / storms_total, 'accepted_total': int(accepted_total),
                       'allgather_rows': rows_gathered, 'allgather_rows_clipped': rows_clipped},
            'roofline': roof,
            'cpu_baseline': cpu,
        }
        print(json.dumps(out))


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command under torch.distributed.run (one per
    GPU, RCCL) and return its exit code.  Fewer than N GPUs is an error, never a smaller run."""
    import socket
    import torch
    have = torch.cuda.device_count()
    if have < n and os.environ.get('TCR_DIST_BACKEND') != 'gloo':      # (gloo: ranks may share a GPU, functional tests)
        print('bench.py: --gpus %d needs %d GPUs on this node, %d visible; not running a smaller job under that name'
              % (n, n, have), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def scattered_line_fill_ceiling():
    """(GB/s, source) that scattered 128-byte line fills reached in this project's calibration run: true bytes of
    `cal_gather_line` (profiles/r03_fetch_calibration.json) over its average duration in the rocprofv3 kernel statistics of
    the same run (profiles/r03_fetch_calibration_kernel_stats.csv).  (None, None) when the files are missing."""
    import csv
    try:
        cal = json.load(open(os.path.join(ROOT, 'profiles', 'r03_fetch_calibration.json')))['kernels']['cal_gather_line']
        for row in csv.DictReader(open(os.path.join(ROOT, 'profiles', 'r03_fetch_calibration_kernel_stats.csv'))):
            if row['Name'].startswith('cal_gather_line'):
                return cal['true_bytes'] / float(row['AverageNs']), 'profiles/r03_fetch_calibration{.json,_kernel_stats.csv}: cal_gather_line'
    except Exception:
        pass
    return None, None


def measured_traffic(kernel, storms, rows, dtype='f64', order='cells', static='0.125/auto', shape='era5'):
    """(HBM bytes per batch, source) of a kernel from the committed rocprofv3 --pmc runs of this same workload
    (tools/collect_profiles.sh; counters are collected in separate passes from timing, as MI355X_MICROARCH.md
    prescribes, so they cannot be measured inside this process).  Only valid for the profiled size."""
    # (the file of the same static-field configuration: r06 / r05 = 0.125 degrees, narrow storage; r04 = 0.25 degrees, fp64 planes)
    fn = next((f for f in (os.path.join(ROOT, 'profiles', n) for n in ('r06_pmc_hbm.json', 'r05_pmc_hbm.json', 'r05_pmc_hbm_res0.25.json', 'r04_pmc_hbm.json'))
               if os.path.exists(f) and json.load(open(f)).get('static', '0.25/f64').replace('0.25/auto', '0.25/f64') == static.replace('0.25/auto', '0.25/f64')), '')
    tag = 'profiles/' + os.path.basename(fn)
    if storms != 100_000 or dtype != 'f64' or shape != 'era5' or not fn:
        return None, None
    try:
        d = json.load(open(fn))
        if d.get('rows') != rows or d.get('order', 'cells') != order:
            return None, None
        if kernel == '*':          # every kernel of a step
            return d['step_total']['hbm_bytes_per_batch'], tag + ': sum over the kernels of a step'
        for name, v in d['kernels'].items():
            if kernel in name:
                return v['hbm_bytes_per_batch'], tag + ': rocprofv3 --pmc, separate passes, %s' % d.get('command', '')
    except Exception:
        pass
    return None, None


def cpu_sample(pipe, B):
    """The storms the CPU baseline is timed on: the first (at most 16 384) of this rank's last GPU batch, as host arrays."""
    import numpy as np
    n = min(B, 16384)
    s = pipe.storms
    return dict(lon=s['lon0'][:n].cpu().numpy(), lat=s['lat0'][:n].cpu().numpy(), v0=s['v0'][:n].cpu().numpy(),
                m0=s['m0'][:n].cpu().numpy(), h_bl=s['h_bl'][:n].cpu().numpy(),
                month=(s['slot'][:n].cpu().numpy() + 1).astype(np.int32),
                phases=s['phases'][:n].cpu().numpy().reshape(n, 4, -1))


def cpu_baseline(host, args):
    """Time oracle/scipy_port.py on a bounded sample of the last batch's storms (rank 0, at every N)."""
    import numpy as np
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, 'storms.npz')
        np.savez(fn, **host)
        try:
            env1 = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
            r = subprocess.run([sys.executable, '-m', 'oracle.cpu_baseline', '--inputs', fn, '--basin', args.basin,
                                '--budget', str(args.cpu_budget), '--static-res', str(args.static_res), '--shape', args.shape] +
                               (['--bathy-kind', args.bathy_kind] if args.bathy_kind else []), cwd=ROOT, capture_output=True, text=True,
                               timeout=600, env=env1)
            res = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:                                        # report, never hide
            return {'value': None, 'unit': 'storm-steps/s', 'cores': 0, 'kind': 'port',
                    'sample': 'failed: %r' % (e,)}
    best = res.get('all_cores', res['one_core'])
    return {'value': best['value'], 'unit': 'storm-steps/s', 'cores': best.get('procs', 1), 'kind': 'port',
            'sample': 'oracle/scipy_port.py (solve_ivp RK45 + RectBivariateSpline.ev + numpy cholesky; integration + accept '
                      'tests + env-wind recompute and vmax for the candidates that pass accept test 1, as compute.py:176-209 '
                      'and the GPU step do) on the first %d storms of the last GPU batch, %d worker '
                      'processes, %.1f s' % (best['storms'], best.get('procs', 1), best['seconds']),
            'value_1core': res['one_core']['value'], 'storms_1core': res['one_core']['storms'], 'seconds_1core': res['one_core']['seconds'],
            'integration_only': {'value': (res.get('all_cores_integration_only') or res['one_core_integration_only'])['value'],
                                 'storms': (res.get('all_cores_integration_only') or res['one_core_integration_only'])['storms'],
                                 'value_1core': res['one_core_integration_only']['value'],
                                 'storms_1core': res['one_core_integration_only']['storms']},
            'per_core_efficiency': best.get('per_core_efficiency'), 'cgroup_cpu_quota': best.get('cgroup_cpu_quota'),
            'physical_cores_visible': best.get('physical_cores_visible'),
            'c_port_1core': res.get('c_port_one_core')}


if __name__ == '__main__':
    main()
