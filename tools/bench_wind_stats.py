"""Throughput of k_wind_stats against the HBM roofline: one month of daily fields on a
0.25-degree grid (721 x 1440 points, 31 days, 4 components = 1.03 GB; the kernel reads it twice)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tropical_cyclone_risk_amd.engine import TCEngine
T, npts = 31, 721 * 1440
eng = TCEngine('GL', device=0)
dev = torch.device('cuda', 0)
planes = [torch.randn(T, npts, dtype=torch.float64, device=dev) for _ in range(4)]
out = torch.empty(14, npts, dtype=torch.float64, device=dev)
ptrs = (C.c_void_p * 4)(*[p.data_ptr() for p in planes])
st = torch.cuda.current_stream(dev).cuda_stream
def launch():
    eng._ck(eng.L.tcr_wind_stats_dev(eng.h, T, npts, ptrs, None, 0, out.data_ptr(), C.c_void_p(st)))
for _ in range(3): launch()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 20
e0.record()
for _ in range(K): launch()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
alg = (2 * 4 * T + 14) * npts * 8          # two passes over the inputs + the 14 output planes
print('k_wind_stats: %d points x %d days: %.3f ms per month, algorithmic %.2f GB -> %.0f GB/s = %.2f of 8 TB/s'
      % (npts, T, ms, alg / 1e9, alg / ms / 1e6, alg / ms / 1e6 / 8000))
ref = torch.stack([p.mean(0) for p in planes])
assert torch.allclose(out[:4], ref, rtol=1e-12, atol=1e-12)
