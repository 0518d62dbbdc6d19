"""GPU timeline of a bench run from a rocprofv3 --kernel-trace csv: how busy was the chip during the timed steps?
    python tools/timeline.py gpurun_out/r03e/t4/t_kernel_trace.csv [n_steps] [skip_last]
Per queue the dispatches are serial; across queues they overlap.  Reports, over the window that holds N k_flags dispatches
(one per step) ending skip_last dispatches before the last one (bench.py ends with five isolated batches on one stream: skip 5): wall time per step, union-busy fraction, mean number of kernels in flight, the integrator's
SIMD demand (sum over its dispatches of min(workgroups, SIMDs) x duration / (SIMDs x wall)), and per-kernel time shares."""
import collections
import csv
import sys

fn = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 8
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
SIMDS = 1024
rows = [r for r in csv.DictReader(open(fn))]
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    r['name'] = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('tcr::', '')
    r['wgs'] = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
flags = sorted(r['e'] for r in rows if r['name'].startswith('k_flags'))
t1 = flags[-1 - skip]
t0 = flags[-1 - skip - n_last]
win = [r for r in rows if r['e'] > t0 and r['s'] < t1]
wall = (t1 - t0) / 1e6
ev = []
for r in win:
    ev.append((max(r['s'], t0), 1)); ev.append((min(r['e'], t1), -1))
ev.sort()
busy = 0; depth = 0; last = t0; area = 0
for t, d in ev:
    if depth > 0:
        busy += t - last
    area += depth * (t - last)
    depth += d; last = t
print('%d steps in %.3f ms: %.3f ms/step; some kernel running %.1f %% of the time; mean kernels in flight %.2f'
      % (n_last, wall, wall / n_last, 100.0 * busy / (t1 - t0), area / (t1 - t0)))
dem = collections.defaultdict(float); dur = collections.defaultdict(float); cnt = collections.Counter()
for r in win:
    d = (min(r['e'], t1) - max(r['s'], t0)) / 1e6
    dur[r['name']] += d; cnt[r['name']] += 1
    if r['name'].startswith('k_integrate'):
        dem['integrate'] += min(r['wgs'], SIMDS) * d
print('integrator SIMD demand: %.1f %% of the chip over the window (each of its waves owns a SIMD)' % (100.0 * dem['integrate'] / (SIMDS * wall)))
print('%-44s %6s %10s %10s' % ('kernel', 'calls', 'ms/step', 'avg us'))
for k, v in sorted(dur.items(), key=lambda kv: -kv[1])[:16]:
    print('%-44s %6d %10.3f %10.1f' % (k[:44], cnt[k], v / n_last, 1e3 * v / cnt[k]))
# queue view: gaps between consecutive dispatches of one queue inside the window
byq = collections.defaultdict(list)
for r in win:
    byq[r['Queue_Id']].append(r)
for q, rs in sorted(byq.items()):
    rs.sort(key=lambda r: r['s'])
    gap = sum(max(0, b['s'] - a['e']) for a, b in zip(rs, rs[1:])) / 1e6
    run = sum(r['e'] - r['s'] for r in rs) / 1e6
    print('queue %s: %4d dispatches, running %.3f ms, idle between its own dispatches %.3f ms (%.1f us per gap)'
          % (q, len(rs), run, gap, 1e3 * gap / max(1, len(rs) - 1)))
