#!/usr/bin/env python3
"""One round of the accept loop and the exchange of its accepted tracks WITHOUT PyTorch: device memory comes from hipMalloc
through ctypes, the work from libtcrisk_hip.so's C ABI alone (tcr_round_dev; tcr_comm_* / tcr_allgather_rows_dev /
tcr_concat_rows_dev for the exchange, here a one-rank communicator).  What INTEGRATION.md's binding amounts to when the host is
ctypes-only; tests/test_multi_gpu_entry.py compares its rows with the torch-backed pipeline's, bit for bit.

    python tools/ctypes_only_round.py [--basin NA] [--cand 65536] [--storms 8192] [--year 2001] [--out rows.npz] [--no-comm]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tropical_cyclone_risk_amd import _lib, synthetic          # noqa: E402  (ctypes + NumPy only)
from tropical_cyclone_risk_amd.engine import TCEngine           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--basin', default='NA')
    ap.add_argument('--cand', type=int, default=65536)
    ap.add_argument('--storms', type=int, default=8192)
    ap.add_argument('--year', type=int, default=2001)
    ap.add_argument('--cand0', type=int, default=0)
    ap.add_argument('--out', default=None)
    ap.add_argument('--no-comm', action='store_true')
    args = ap.parse_args()

    env = synthetic.make_env('era5', static_res=0.125)
    eng = TCEngine(args.basin, device=0).stage_env(env)         # loads libtcrisk_hip.so (and the HIP runtime with it)
    L, h = eng.L, eng.h
    hip = C.CDLL(None)                                           # the HIP runtime the process already holds (global namespace) ...
    if not hasattr(hip, 'hipMalloc'):
        hip = C.CDLL('libamdhip64.so', mode=C.RTLD_GLOBAL)       # ... or, loaded as a private dependency only, by name
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    keep = []

    def dmalloc(nbytes, fill=None):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), max(16, int(nbytes))) == 0
        if fill is not None:
            assert hip.hipMemset(p, fill, max(16, int(nbytes))) == 0
        keep.append(p)
        return p.value

    def to_host(ptr, shape, dtype):
        a = np.empty(shape, dtype=dtype)
        assert hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), a.nbytes, 2) == 0        # hipMemcpyDeviceToHost
        return a
    ns, N = eng.n_steps, eng.n_series
    nc, nb = args.cand, args.storms

    def seeds(n, phases):
        return _lib.Seeds(n, dmalloc(8 * n), dmalloc(8 * n), dmalloc(8 * n), dmalloc(8 * n), dmalloc(8 * n), dmalloc(4 * n),
                          dmalloc(8 * n * 4 * N) if phases else None, dmalloc(4 * n), dmalloc(4 * n))
    r = _lib.Round()
    r.n_cand, r.n_storms = nc, nb
    r.cand, r.storms = seeds(nc, False), seeds(nb, True)
    r.cand_idx, r.n_passed = dmalloc(4 * nc), dmalloc(8, 0)
    r.cell_deg, r.exact_count, r.f32 = 2.0, 1, 0
    plane = lambda k=1: dmalloc(8 * nb * ns * k)
    pad = dmalloc(4 * nb, 0xff)                                  # pad_state = -1: unknown
    r.tracks = _lib.Tracks(plane(), plane(), plane(), plane(), plane(), plane(4), dmalloc(4 * nb), dmalloc(4 * nb), dmalloc(4 * nb),
                           dmalloc(4 * nb), dmalloc(4 * nb), dmalloc(4 * nb), pad, 1)
    stats = dmalloc(8 * _lib.N_STATS, 0)
    r.stats = stats
    r.acc_idx, r.n_accepted = dmalloc(4 * nb), dmalloc(8, 0)
    cap, stride = max(64, nb // 4), 9 * ns + 3
    r.packed, r.pack_cap, r.pack_stride = dmalloc(8 * cap * stride), cap, stride
    hist = dmalloc(8 * 84, 0)
    r.seed_hist = hist
    r.n_expected = 0
    seed = int(eng.nl.gpu_experiment_seed)
    eng._ck(L.tcr_round_dev(h, C.byref(r), C.c_uint64(seed), args.year, args.cand0, 0, None))
    eng._ck(L.tcr_sync(h, None))
    n_passed = int(to_host(r.n_passed, (1,), np.int64)[0])
    n_acc = int(to_host(r.n_accepted, (1,), np.int64)[0])
    st = to_host(stats, (_lib.N_STATS,), np.int64)
    rows = to_host(r.packed, (cap, stride), np.float64)[:min(n_acc, cap)]
    n_seeds = to_host(hist, (7, 12), np.int64)
    gathered = None
    if not args.no_comm:
        ident = (C.c_uint8 * _lib.TCR_COMM_ID_BYTES)()
        if L.tcr_comm_unique_id(ident) != 0:
            raise RuntimeError(L.tcr_last_error(None).decode())
        comm = C.c_void_p()
        eng._ck(L.tcr_comm_create(h, ident, 0, 1, C.byref(comm)))
        counts, recv, out, n_out = dmalloc(8), dmalloc(8 * cap * stride), dmalloc(8 * cap * stride), dmalloc(8, 0)
        eng._ck(L.tcr_allgather_counts_dev(comm, r.n_accepted, counts, None))
        eng._ck(L.tcr_allgather_rows_dev(comm, r.packed, cap, stride, recv, None))
        eng._ck(L.tcr_concat_rows_dev(h, 1, recv, counts, cap, stride, out, cap, n_out, None))
        hist2 = dmalloc(8 * 84)
        assert hip.hipMemcpy(C.c_void_p(hist2), C.c_void_p(hist), 8 * 84, 3) == 0                      # device to device
        eng._ck(L.tcr_allreduce_sum_i64_dev(comm, hist2, 84, None))
        eng._ck(L.tcr_sync(h, None))
        k = int(to_host(n_out, (1,), np.int64)[0])
        gathered = to_host(out, (cap, stride), np.float64)[:k]
        assert k == min(n_acc, cap) and np.array_equal(gathered, rows, equal_nan=True)
        assert np.array_equal(to_host(hist2, (7, 12), np.int64), n_seeds)
        L.tcr_comm_destroy(comm)
    for p in keep:
        hip.hipFree(p)
    eng.close()
    assert 'torch' not in sys.modules, 'this path must not import PyTorch'
    print('ctypes-only round: %d candidates, %d passed, %d integrated, %d accepted, %d storm-steps; %s; torch imported: %s'
          % (nc, n_passed, int(st[7]), n_acc, int(st[0]), 'exchange through a one-rank RCCL communicator ok' if gathered is not None else 'no exchange',
             'torch' in sys.modules))
    if args.out:
        np.savez(args.out, rows=rows, n_seeds=n_seeds, n_passed=n_passed, n_accepted=n_acc, stats=st)


if __name__ == '__main__':
    main()
