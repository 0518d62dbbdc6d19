"""How many batches really overlap?  Steps per second of the bench's step with n streams, for streams made three ways:
torch.cuda.Stream() (torch's pool), hipStreamCreateWithFlags(non-blocking) wrapped as torch ExternalStream, and
hipStreamCreateWithPriority at alternating priorities.  python tools/stream_overlap.py [storms]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tropical_cyclone_risk_amd import synthetic
from tropical_cyclone_risk_amd.engine import TCEngine
from tropical_cyclone_risk_amd.pipeline import DevicePipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
hip = C.CDLL('libamdhip64.so')
dev = torch.device('cuda', 0)
env = synthetic.make_env('era5')


def make_streams(kind, n):
    if kind == 'torch':
        return [torch.cuda.Stream(device=dev) for _ in range(n)]
    out = []
    for i in range(n):
        h = C.c_void_p()
        if kind == 'hip':
            assert hip.hipStreamCreateWithFlags(C.byref(h), 1) == 0
        else:
            lo, hi = C.c_int(), C.c_int()
            hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
            pr = [hi.value, 0, lo.value][i % 3]
            assert hip.hipStreamCreateWithPriority(C.byref(h), 1, pr) == 0
        out.append(torch.cuda.ExternalStream(h.value, device=dev))
    return out


def run(kind, n, steps=24):
    engs = [TCEngine('GL', device=0).stage_env(env) for _ in range(n)]
    pipes = [DevicePipeline(e, int(5.6 * B) + 4096, B, tc_rows_only=True) for e in engs]
    st = make_streams(kind, n)
    def step(k):
        with torch.cuda.stream(st[k % n]):
            p = pipes[k % n]
            p.seed_round(2000, k * 10**7); p.select_passed(B); p.integrate(B)
    for k in range(2 * n):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    for e in engs:
        e.close()
    return dt


for kind in ('torch', 'hip', 'prio'):
    print(kind, ' '.join('s%d %.3f' % (n, run(kind, n)) for n in (1, 2, 3, 4, 6, 8)), flush=True)
