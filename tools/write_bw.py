"""HBM write-only bandwidth on this GPU (what bounds the forcing table): torch fill of the table's size."""
import torch, time
n = 100_000 * 361 * 4
x = torch.empty(n, dtype=torch.float64, device='cuda')
y = torch.empty(n, dtype=torch.float64, device='cuda')
for name, fn in (('fill_', lambda: x.fill_(1.5)), ('zero_', lambda: x.zero_()), ('copy_', lambda: y.copy_(x))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print('%s: %.3f ms for %.2f GB -> %.2f TB/s written' % (name, ms, n * 8 / 1e9, n * 8 / 1e9 / ms))
