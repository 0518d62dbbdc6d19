"""Device-resident seed → select → integrate → pack pipeline.

PyTorch is used for what it is good at here — owning HBM buffers, exposing the
current HIP stream, and (in ``distributed.py``) driving RCCL — while every
computation on those buffers is one of the library's HIP kernels, launched
through the C ABI with raw device pointers.

One :class:`DevicePipeline` is the batched equivalent of the body of the
reference's ``while nt < n_tracks`` loop (`util/compute.py:134-209`) for a round
of candidates.
"""
import ctypes as C

import torch

from . import _lib

F64 = torch.float64
I32 = torch.int32


class DevicePipeline:
    def __init__(self, engine, max_candidates, max_storms, device=None, sort_storms=False, tc_rows_only=False,
                 dtype='f64'):
        # dtype 'f32': the fp32 variant of the path (tcr_integrate_f32_dev): float32 rows; seeds stay fp64
        assert dtype in ('f64', 'f32')
        self.dtype = dtype
        # tc_rows_only: produce rows only for storms that pass accept test 1, as the reference does
        # (compute.py:190-204, tcr_tracks.tc_rows_only); False = every row (what parity tests compare)
        self.tc_rows_only = bool(tc_rows_only)
        self.eng = engine
        self.dev = torch.device('cuda', engine.device) if device is None else device
        self.C, self.B = int(max_candidates), int(max_storms)
        ns, N = engine.n_steps, engine.n_series
        z = lambda *shape, dtype=F64: torch.empty(*shape, dtype=dtype, device=self.dev)
        seeds = lambda n, ph=True: dict(n=n, lon0=z(n), lat0=z(n), v0=z(n), m0=z(n), h_bl=z(n),
                                        slot=z(n, dtype=I32), phases=z(n, 4 * N) if ph else None,
                                        basin_idx=z(n, dtype=I32), seed_flags=z(n, dtype=I32))
        # candidates carry no Fourier phases: they are drawn at selection time, only for seeds that pass
        self.cand = seeds(self.C, ph=False)   # one seeding round
        self.storms = seeds(self.B)           # the candidates that passed, dense
        rt = F64 if dtype == 'f64' else torch.float32
        self.tracks = dict(lon=z(self.B, ns, dtype=rt), lat=z(self.B, ns, dtype=rt), v=z(self.B, ns, dtype=rt),
                           m=z(self.B, ns, dtype=rt), vmax=z(self.B, ns, dtype=rt), envw=z(self.B, ns, 4, dtype=rt),
                           n_valid=z(self.B, dtype=I32), status=z(self.B, dtype=I32),
                           flags=z(self.B, dtype=I32), nfev=z(self.B, dtype=I32),
                           n_accept=z(self.B, dtype=I32), n_reject=z(self.B, dtype=I32),
                           # rows of the planes are NaN from this sample on (-1: unknown), tcrisk_hip.h
                           pad_state=torch.full((self.B,), -1, dtype=I32, device=self.dev))
        self.cand_idx = z(self.C, dtype=I32)  # candidate index of each dense storm
        self.n_passed = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.acc_idx = z(self.B, dtype=I32)
        self.n_accepted = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.sort_storms = sort_storms
        self._rounds = {}                     # tcr_round descriptors by their configuration (round())

    # raw pointers -----------------------------------------------------------
    @staticmethod
    def _ptrs(d, keys):
        return {k: (d[k] if k == 'n' else d[k].data_ptr()) for k in keys}

    def _seeds_struct(self, d, n=None):
        s = _lib.Seeds(int(d['n'] if n is None else n),
                       *[(d[k].data_ptr() if d[k] is not None else None)
                         for k in ('lon0', 'lat0', 'v0', 'm0', 'h_bl', 'slot', 'phases', 'basin_idx', 'seed_flags')])
        return s

    def _tracks_struct(self):
        t = self.tracks
        return _lib.Tracks(*[t[k].data_ptr() for k in ('lon', 'lat', 'v', 'm', 'vmax', 'envw', 'n_valid',
                                                        'status', 'flags', 'nfev', 'n_accept', 'n_reject', 'pad_state')],
                           1 if self.tc_rows_only else 0)

    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    # stages -----------------------------------------------------------------
    def seed_round(self, year, cand0, n_cand=None, experiment_seed=None):
        """Draw candidates [cand0, cand0 + n_cand) (compute.py:136-175)."""
        n = self.C if n_cand is None else int(n_cand)
        assert n <= self.C
        seed = int(self.eng.nl.gpu_experiment_seed if experiment_seed is None else experiment_seed)
        s = self._seeds_struct(self.cand, n)
        self.eng._ck(self.eng.L.tcr_seed_dev(self.eng.h, C.c_uint64(seed), int(year), int(cand0),
                                             C.byref(s), C.c_void_p(self._stream())))
        self.n_cand = n
        self._round = (seed, int(year), int(cand0))

    def select_passed(self, n_take=None):
        """Dense batch of the first n_take candidates whose seed passed (flag bit 1),
        in candidate order.  ``self.n_passed`` (device) holds how many passed in total."""
        n_take = self.B if n_take is None else int(n_take)
        assert n_take <= self.B
        L, h, st = self.eng.L, self.eng.h, C.c_void_p(self._stream())
        self.eng._ck(L.tcr_compact_dev(h, self.n_cand, self.cand['seed_flags'].data_ptr(), 2, n_take,
                                       self.cand_idx.data_ptr(), self.n_passed.data_ptr(), st))
        if self.sort_storms:
            # Locality order (tcr_cell_order_dev): the selected candidates by the 2-degree cell of their genesis point, a
            # stable counting sort of the index list on the device — neighbours share an integrator wave and its cache
            # lines (integrator L2 misses -20 %; 100 000-storm step, 8 streams: 1.371 -> 1.345 ms, profiles/r03_bench*.json).  With it, dense order is no longer candidate order:
            # `cand_idx` is the map back, and compute.accept_loop sorts the accepted rows by their candidate index.
            cs = self._seeds_struct(self.cand, self.n_cand)
            self.eng._ck(L.tcr_cell_order_dev(h, C.byref(cs), self.cand_idx.data_ptr(), n_take, self.n_passed.data_ptr(),
                                              float(self.sort_storms if self.sort_storms is not True else 2.0), st))
        src, dst = self._seeds_struct(self.cand, self.n_cand), self._seeds_struct(self.storms, n_take)
        seed, year, cand0 = self._round
        self.eng._ck(L.tcr_gather_seeds_dev(h, C.byref(src), self.cand_idx.data_ptr(), n_take,
                                            self.n_passed.data_ptr(), C.byref(dst), C.c_uint64(seed), year, cand0, st))
        self.n_storms = n_take

    def load_storms(self, storms):
        """Upload a host batch (dict as produced by ``synthetic.draw_storm_inputs``)."""
        import numpy as np
        n = len(storms['lon'])
        assert n <= self.B
        put = lambda dst, a, dt: dst[:n].copy_(torch.from_numpy(np.ascontiguousarray(a, dtype=dt)))
        s = self.storms
        put(s['lon0'], storms['lon'], np.float64); put(s['lat0'], storms['lat'], np.float64)
        put(s['v0'], storms['v0'], np.float64); put(s['m0'], storms['m0'], np.float64)
        put(s['h_bl'], storms['h_bl'], np.float64)
        put(s['slot'], np.asarray(storms['month']) - 1, np.int32)
        s['phases'][:n].copy_(torch.from_numpy(np.ascontiguousarray(storms['phases'], dtype=np.float64).reshape(n, -1)))
        self.n_storms = n

    def integrate(self, n=None, n_dev=None):
        """gen_track + accept tests + env-wind recompute + vmax for the dense batch.  n_dev (device int64
        tensor, e.g. ``self.n_passed``): only the first min(n, n_dev) rows hold storms (tcr_storms.n_dev)."""
        n = self.n_storms if n is None else int(n)
        s = self.storms
        self._n_dev = n_dev
        si = _lib.Storms(n, *[s[k].data_ptr() for k in ('lon0', 'lat0', 'v0', 'm0', 'h_bl', 'slot', 'phases')],
                         n_dev.data_ptr() if n_dev is not None else None)
        so = self._tracks_struct()
        fn = self.eng.L.tcr_integrate_dev if self.dtype == 'f64' else self.eng.L.tcr_integrate_f32_dev
        self.eng._ck(fn(self.eng.h, C.byref(si), C.byref(so), C.c_void_p(self._stream())))
        self.n_done = n

    def add_stats(self, counters, n_dev=None):
        """counters (uint64/int64 tensor [_lib.N_STATS = 10]) += storm-steps, RHS evaluations, samples, accepted, is_tc
        storms, samples of is_tc storms, 1 if the batch was short of storms (n_dev < n), storms counted, storms whose step
        record overflowed, passing seeds the batch had no room for (tcr_stats_dev).  n_dev: device int64 scalar, default
        the one integrate() was given."""
        assert counters.numel() >= _lib.N_STATS and counters.element_size() == 8
        so = self._tracks_struct()
        nd = n_dev if n_dev is not None else getattr(self, '_n_dev', None)
        self.eng._ck(self.eng.L.tcr_stats_dev(self.eng.h, self.n_done, nd.data_ptr() if nd is not None else None,
                                              C.byref(so), counters.data_ptr(), _lib.N_STATS, C.c_void_p(self._stream())))

    def round(self, year, cand0, n_cand=None, n_take=None, experiment_seed=None, exact_count=True, stats=None,
              accepted=False, packed=None, pack_cap=0, seed_hist=None, graph=False, n_expected=0):
        """One round of the accept loop in ONE library call (tcr_round_dev): seed_round → select_passed → integrate
        (→ add_stats → select_accepted → pack with the meta columns → n_seeds histogram), nothing returning to Python in
        between; the same kernels in the same order as the separate methods, so the results are theirs.

        exact_count: integrate min(n_take, n_passed) storms (n_dev); False: the round is sized so that n_take seeds pass.
        stats: int64 tensor [_lib.N_STATS] added to.  accepted: also select the accepted tracks (acc_idx / n_accepted);
        packed [cap, >= 9*ns + 3]: their survivor records + (candidate index, month, basin index).  seed_hist: int64 tensor
        [7*12] set to the round's n_seeds contribution.  graph: replay the round from a captured hipGraph (one launch).
        n_expected: how many seeds are expected to pass (shapes the integrator's launch; 0 = n_take)."""
        n_cand = self.C if n_cand is None else int(n_cand)
        n_take = self.B if n_take is None else int(n_take)
        assert n_cand <= self.C and n_take <= self.B
        cfg = (n_cand, n_take, bool(exact_count), stats.data_ptr() if stats is not None else 0, bool(accepted),
               packed.data_ptr() if packed is not None else 0, int(pack_cap), int(packed.stride(0)) if packed is not None else 0,
               seed_hist.data_ptr() if seed_hist is not None else 0, int(n_expected))
        r = self._rounds.get(cfg)
        if r is None:
            if stats is not None:
                assert stats.numel() >= _lib.N_STATS and stats.element_size() == 8
            if seed_hist is not None:
                assert seed_hist.numel() >= 84 and seed_hist.element_size() == 8
            r = _lib.Round()
            r.n_cand, r.n_storms = n_cand, n_take
            r.cand, r.storms = self._seeds_struct(self.cand, n_cand), self._seeds_struct(self.storms, n_take)
            r.cand_idx, r.n_passed = self.cand_idx.data_ptr(), self.n_passed.data_ptr()
            r.cell_deg = 0.0 if not self.sort_storms else float(self.sort_storms if self.sort_storms is not True else 2.0)
            r.exact_count, r.f32 = int(bool(exact_count)), int(self.dtype == 'f32')
            r.tracks = self._tracks_struct()
            r.stats = stats.data_ptr() if stats is not None else None
            if accepted or packed is not None:
                r.acc_idx, r.n_accepted = self.acc_idx.data_ptr(), self.n_accepted.data_ptr()
            if packed is not None:
                r.packed, r.pack_cap, r.pack_stride = packed.data_ptr(), int(pack_cap), int(packed.stride(0))
            r.seed_hist = seed_hist.data_ptr() if seed_hist is not None else None
            r.n_expected = int(n_expected)
            self._rounds[cfg] = r
        seed = int(self.eng.nl.gpu_experiment_seed if experiment_seed is None else experiment_seed)
        self.eng._ck(self.eng.L.tcr_round_dev(self.eng.h, C.byref(r), C.c_uint64(seed), int(year), int(cand0),
                                              1 if graph else 0, C.c_void_p(self._stream())))
        self.n_cand, self.n_storms, self.n_done = n_cand, n_take, n_take
        self._round = (seed, int(year), int(cand0))
        self._n_dev = self.n_passed if exact_count else None

    def seed_hist(self, out, cutoff=None):
        """n_seeds contribution (compute.py:165-167) of the last seeded round → out (int64 tensor [7*12], set); cutoff
        (device float64 scalar): only candidates with global index <= cutoff (tcr_seed_hist_dev)."""
        assert out.numel() >= 84 and out.element_size() == 8
        cs = self._seeds_struct(self.cand, self.n_cand)
        self.eng._ck(self.eng.L.tcr_seed_hist_dev(self.eng.h, C.byref(cs), int(self._round[2]),
                                                  cutoff.data_ptr() if cutoff is not None else None, out.data_ptr(),
                                                  C.c_void_p(self._stream())))
        return out

    def graph_stats(self):
        g, r = C.c_int64(0), C.c_int64(0)
        self.eng._ck(self.eng.L.tcr_round_graph_stats(self.eng.h, C.byref(g), C.byref(r)))
        return dict(graphs=int(g.value), replays=int(r.value))

    def select_accepted(self):
        """Dense-batch rows of the accepted tracks, ascending → self.acc_idx (dense order is candidate order only without
        the locality order: `cand_idx` maps a dense row to its candidate)."""
        self.eng._ck(self.eng.L.tcr_compact_dev(self.eng.h, self.n_done, self.tracks['flags'].data_ptr(),
                                                _lib.FLAG_ACCEPTED, self.B, self.acc_idx.data_ptr(),
                                                self.n_accepted.data_ptr(), C.c_void_p(self._stream())))

    def pack_accepted(self, packed, cap):
        """Survivor records of the first ``cap`` accepted tracks into ``packed`` [cap, >= 9*ns] (extra
        columns of a wider buffer are left to the caller)."""
        so = self._tracks_struct()
        fn = self.eng.L.tcr_pack_tracks_dev if self.dtype == 'f64' else self.eng.L.tcr_pack_tracks_f32_dev
        self.eng._ck(fn(self.eng.h, C.byref(so), self.acc_idx.data_ptr(),
                        self.n_accepted.data_ptr(), int(cap), packed.data_ptr(),
                        int(packed.stride(0)), C.c_void_p(self._stream())))

    def pack_accepted_meta(self, packed, cap, cand0):
        """pack_accepted plus the three columns behind the rows: global candidate index, month, basin index
        (tcr_pack_tracks_meta_dev); packed [cap, >= 9*ns + 3]."""
        so = self._tracks_struct()
        s = self.storms
        self.eng._ck(self.eng.L.tcr_pack_tracks_meta_dev(
            self.eng.h, C.byref(so), int(self.dtype == 'f32'), self.acc_idx.data_ptr(), self.n_accepted.data_ptr(), int(cap),
            packed.data_ptr(), int(packed.stride(0)), self.cand_idx.data_ptr(), s['slot'].data_ptr(), s['basin_idx'].data_ptr(),
            int(cand0), C.c_void_p(self._stream())))

    def host_tracks(self, n=None):
        """Copy the per-storm outputs of the last integrate() back as NumPy arrays."""
        n = self.n_done if n is None else n
        out = {k: v[:n].cpu().numpy() for k, v in self.tracks.items()}
        return self.eng._finish(out)
