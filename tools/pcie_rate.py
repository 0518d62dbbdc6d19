"""PCIe-inclusive rate of the bench's step (never bench.py's `value`): the same 100 000-storm step on four
streams, plus the hand-over of every batch's accepted tracks (9 x 361 fp64 per track) to pinned host memory.  A batch's
count is read back when its pipeline is used again (n_streams steps later), and its rows are then copied on a copy
stream, so transfers overlap the compute of later batches."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tropical_cyclone_risk_amd import synthetic
from tropical_cyclone_risk_amd.engine import TCEngine
from tropical_cyclone_risk_amd.pipeline import DevicePipeline

B, n_str, steps, warm = 100_000, 4, 40, 8
dev = torch.device('cuda', 0)
env = synthetic.make_env('era5')
engs = [TCEngine('GL', device=0).stage_env(env) for _ in range(n_str)]
ns = engs[0].n_steps
row, cap = 9 * ns, 20_000
C = int(B * 5.6)
pipes = [DevicePipeline(e, C, B, tc_rows_only=True) for e in engs]
streams = [torch.cuda.Stream(device=dev) for _ in range(n_str)]
copy_stream = torch.cuda.Stream(device=dev)
bufs = [torch.empty(cap, row, dtype=torch.float64, device=dev) for _ in range(n_str)]
hosts = [torch.empty(cap, row, dtype=torch.float64).pin_memory() for _ in range(n_str)]
done = [torch.cuda.Event() for _ in range(n_str)]
acc = torch.zeros(10, dtype=torch.int64, device=dev)
rows_out = 0


def flush(s, copy):
    """hand the rows of the batch last run on pipeline s over to the host"""
    global rows_out
    done[s].synchronize()                        # that batch was issued n_str steps ago
    n = int(pipes[s].n_accepted.item())
    rows_out += n
    if copy and n:
        copy_stream.wait_event(done[s])
        with torch.cuda.stream(copy_stream):
            hosts[s][:n].copy_(bufs[s][:n], non_blocking=True)
        streams[s].wait_stream(copy_stream)      # the buffer is reused by this stream's next batch


def run(copy):
    global rows_out
    for k in range(warm + steps):
        if k == warm:
            torch.cuda.synchronize(); acc.zero_(); rows_out = 0; t0 = time.perf_counter()
        s = k % n_str
        if k >= n_str:
            flush(s, copy)
        with torch.cuda.stream(streams[s]):
            p = pipes[s]
            p.seed_round(2000, k * C); p.select_passed(B); p.integrate(B); p.add_stats(acc)
            p.select_accepted(); p.pack_accepted(bufs[s], cap)
            done[s].record(streams[s])
    for s in range(n_str):
        flush((warm + steps + s) % n_str, copy)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return int(acc[0].item()) / dt, dt / steps * 1e3, rows_out


for copy in (False, True):
    v, ms, n = run(copy)
    print('%s: %.3g storm-steps/s, %.3f ms/step, %d accepted tracks handed over (%.0f MB per step)' % (
        'host hand-over of accepted tracks (pinned, overlapped)' if copy else 'rows stay in HBM', v, ms, n, n / steps * row * 8 / 1e6))
