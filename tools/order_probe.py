"""Does a locality order of the dense batch pay at the level of the STEP (4 streams, memory-bound)?  The same round of
candidates is seeded every step, so the permutation of its passing seeds can be computed once, outside the timed loop, and
applied with one 400 kB device copy between the selection and the gather of the seeds — i.e. the step is timed with the order
in place but without the cost of sorting.  python tools/order_probe.py [storms]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tropical_cyclone_risk_amd import synthetic
from tropical_cyclone_risk_amd.engine import TCEngine
from tropical_cyclone_risk_amd.pipeline import DevicePipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
N_STR = 4
dev = torch.device('cuda', 0)
env = synthetic.make_env('era5')
engs = [TCEngine('GL', device=0).stage_env(env) for _ in range(N_STR)]
Cn = int(5.6 * B) + 4096
pipes = [DevicePipeline(e, Cn, B, tc_rows_only=True) for e in engs]
streams = [torch.cuda.Stream(device=dev) for _ in range(N_STR)]
p0 = pipes[0]
p0.seed_round(2000, 0); p0.select_passed(B); torch.cuda.synchronize()
base = p0.cand_idx[:B].clone()
lon, lat, slot = p0.cand['lon0'][base.long()], p0.cand['lat0'][base.long()], p0.cand['slot'][base.long()].double()


def interleave(order, n_bands=8, chunk=64):
    per = (B + n_bands - 1) // n_bands
    bands = [order[i * per:(i + 1) * per] for i in range(n_bands)]
    out, pos, c = [], [0] * n_bands, 0
    while sum(pos) < B:
        b = c % n_bands
        if pos[b] < len(bands[b]):
            t = bands[b][pos[b]:pos[b] + chunk]; out.append(t); pos[b] += len(t)
        c += 1
    return torch.cat(out)


def cells(deg, lat_major=True):
    a, b = torch.floor((lat + 90) / deg), torch.floor((lon % 360) / deg)
    return torch.argsort((a * 4096 + b) if lat_major else (b * 4096 + a), stable=True)


def morton(deg):
    a, b = torch.floor((lat + 90) / deg).long(), torch.floor((lon % 360) / deg).long()
    key = torch.zeros_like(a)
    for bit in range(10):
        key |= ((a >> bit) & 1) << (2 * bit + 1) | ((b >> bit) & 1) << (2 * bit)
    return torch.argsort(key, stable=True)


orders = {
    'candidate order': None,
    '1-degree cells (lat major)': cells(1),
    '2-degree cells (lat major)': cells(2),
    '2-degree cells (lon major)': cells(2, False),
    '3-degree cells (lat major)': cells(3),
    '4-degree cells (lat major)': cells(4),
    '2-degree cells, Morton order': morton(2),
    '1-degree cells, Morton order': morton(1),
    'slot, then 2-degree cells': torch.argsort(slot * 1e8 + torch.floor((lat + 90) / 2) * 4096 + torch.floor((lon % 360) / 2), stable=True),
    '2-degree cells, then slot': torch.argsort((torch.floor((lat + 90) / 2) * 4096 + torch.floor((lon % 360) / 2)) * 16 + slot, stable=True),
}


def step(k, perm_idx):
    with torch.cuda.stream(streams[k % N_STR]):
        p = pipes[k % N_STR]
        p.seed_round(2000, 0)
        L, h, st = p.eng.L, p.eng.h, C.c_void_p(p._stream())
        p.eng._ck(L.tcr_compact_dev(h, p.n_cand, p.cand['seed_flags'].data_ptr(), 2, B, p.cand_idx.data_ptr(), p.n_passed.data_ptr(), st))
        if perm_idx is not None:
            p.cand_idx[:B].copy_(perm_idx, non_blocking=True)
        src, dst = p._seeds_struct(p.cand, p.n_cand), p._seeds_struct(p.storms, B)
        p.eng._ck(L.tcr_gather_seeds_dev(h, C.byref(src), p.cand_idx.data_ptr(), B, p.n_passed.data_ptr(), C.byref(dst),
                                         C.c_uint64(int(p.eng.nl.gpu_experiment_seed)), 2000, 0, st))
        p.n_storms = B
        p.integrate(B)


for rep in range(2):
    for name, o in orders.items():
        perm_idx = None if o is None else base[o].contiguous()
        for k in range(8):
            step(k, perm_idx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(32):
            step(k, perm_idx)
        torch.cuda.synchronize()
        print('%-50s %.3f ms/step' % (name, (time.perf_counter() - t0) / 32 * 1e3), flush=True)
