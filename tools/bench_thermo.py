"""Throughput of k_potential_intensity: one time sample on a 0.25-degree grid with 37 levels."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tropical_cyclone_risk_amd import preprocess as pp
from tropical_cyclone_risk_amd.engine import TCEngine
eng = TCEngine('GL', device=0)
tab = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'entropy_table.npz'))
pp.stage_entropy_table(eng, tab['p'], tab['s'], tab['T'])
L, npts = 37, 721 * 1440
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(1)
p = torch.linspace(100000.0, 5000.0, L, dtype=torch.float64, device=dev)
sst = 285 + 17 * torch.rand(npts, dtype=torch.float64, device=dev, generator=g)
psl = 101000 + 500 * torch.randn(npts, dtype=torch.float64, device=dev, generator=g)
T = (sst - 1.0)[None] * (p[:, None] / p[0]) ** 0.19
T = torch.maximum(T, torch.tensor(205.0, dtype=torch.float64, device=dev)).contiguous()
tc = T - 273.0
es = 610.94 * torch.exp(torch.clamp(17.625 * tc / (tc + 243.04), max=10))
r = (0.75 * (p[:, None] / p[0]) ** 1.2 * 0.622 * es / (p[:, None] - es)).contiguous()
pi = torch.empty(npts, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
def launch():
    eng._ck(eng.L.tcr_potential_intensity_dev(eng.h, npts, L, p.data_ptr(), sst.data_ptr(), psl.data_ptr(), T.data_ptr(),
                                              r.data_ptr(), 1.0, pi.data_ptr(), C.c_void_p(st)))
for _ in range(3): launch()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 10
e0.record()
for _ in range(K): launch()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
alg = (2 * L + 3) * npts * 8
print('k_potential_intensity: %d columns x %d levels: %.3f ms, %.1f M columns/s, algorithmic %.2f GB -> %.0f GB/s = %.2f of 8 TB/s; PI median %.1f m/s'
      % (npts, L, ms, npts / ms / 1e3, alg / 1e9, alg / ms / 1e6, alg / ms / 1e6 / 8000, float(pi.median())))
