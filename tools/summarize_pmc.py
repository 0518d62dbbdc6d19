"""Fold rocprofv3 --pmc counter_collection.csv files into per-kernel, per-batch HBM figures.

A batch = one bench step (one k_dense dispatch).  k_integrate is a chain of passes per batch, so
per-batch values (sum over the passes) are what compares with bench.py's per-batch algorithmic bytes;
per-dispatch averages are kept next to them."""
import collections
import csv
import glob
import json
import sys

import os

out_dir = sys.argv[1]
# FETCH_SIZE correction for THIS project's access pattern (16-byte-per-lane gathers of 128-byte grid-point lines), measured
# on a known byte count (tools/calibrate_fetch.hip -> profiles/r03_fetch_calibration.json): 1.9986 for scattered line
# gathers, 2.0000 for streaming reads and for single 16-byte pieces of distinct lines (the L2 always fills whole 128-B lines
# and the counter tallies them at 64 B).  WRITE_SIZE is exact for coalesced stores and over-reports scattered 112-of-128-B
# line writes (the step records) by 1.24x; it is used as reported.
CAL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r03_fetch_calibration.json')
FETCH_FACTOR, FETCH_SRC = 2.0, 'MI355X_MICROARCH.md (uncalibrated)'
try:
    cal = json.load(open(CAL))['kernels']
    FETCH_FACTOR, FETCH_SRC = cal['cal_gather_line']['true_over_reported'], 'profiles/r03_fetch_calibration.json: cal_gather_line'
except Exception:
    pass
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(out_dir + '/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
res = {}
for k, cs in acc.items():
    if not k.startswith('tcr::'):
        continue
    e = {}
    for c, v in cs.items():
        n = len(disp[(k, c)])
        e[c + '_per_dispatch'] = v / n
        e['dispatches_' + c] = n
    res[k] = e
batches = {c: n for (k, c), s in disp.items() if k.startswith('tcr::k_dense') for n in [len(s)]}
for k, e in res.items():
    for c in ('FETCH_SIZE', 'WRITE_SIZE', 'TCC_HIT_sum', 'TCC_MISS_sum', 'TCC_REQ_sum'):
        if c + '_per_dispatch' in e and batches.get(c):
            e[c + '_per_batch'] = e[c + '_per_dispatch'] * e['dispatches_' + c] / batches[c]
    if 'FETCH_SIZE_per_batch' in e and 'WRITE_SIZE_per_batch' in e:
        # rocprofv3 reports both in KiB.  gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE =
        # TCC_EA0_RDREQ x 64 B while the L2 fills 128-B lines, i.e. it reports half the bytes read -> doubled.
        e['hbm_bytes_per_batch_raw'] = 1024.0 * (e['FETCH_SIZE_per_batch'] + e['WRITE_SIZE_per_batch'])
        e['hbm_bytes_per_batch'] = 1024.0 * (FETCH_FACTOR * e['FETCH_SIZE_per_batch'] + e['WRITE_SIZE_per_batch'])
    if e.get('TCC_REQ_sum_per_batch'):
        e['l2_hit_rate'] = e['TCC_HIT_sum_per_batch'] / e['TCC_REQ_sum_per_batch']
tot = sum(e.get('hbm_bytes_per_batch', 0.0) for e in res.values())
tot_raw = sum(e.get('hbm_bytes_per_batch_raw', 0.0) for e in res.values())
rows = sys.argv[2] if len(sys.argv) > 2 else 'tc'
static = sys.argv[3] if len(sys.argv) > 3 else '0.125/auto'       # --static-res / --static-store of the profiled runs
command = ('rocprofv3 --kernel-trace --output-format csv --pmc <FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum> -- '
           'python bench.py --steps 3 --warmup 1 --streams 1 --rows %s --no-cpu-baseline (three separate runs)' % rows)
meta = dict(workload='GL, 100000 storms per batch; the first batch of each run pads whole plane rows (pad_state = -1)',
            units='FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; *_per_batch = summed over the dispatches of a bench step',
            fetch_factor=FETCH_FACTOR, fetch_factor_source=FETCH_SRC,
            note='hbm_bytes_per_batch = fetch_factor x FETCH_SIZE + WRITE_SIZE.  On gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B while the L2 fills '
                 '128-B lines; the factor is calibrated on this project\'s own gather pattern against a known byte count (1 GiB, every '
                 'line once, 4x the Infinity Cache): 1.9986; streaming reads 2.0000.  WRITE_SIZE is exact for coalesced stores (1 GiB fill) '
                 'and over-reports scattered 112-of-128-B line writes by 1.24x.  *_raw = FETCH_SIZE + WRITE_SIZE uncorrected.  '
                 'Infinity-Cache hits are counted, so this is fabric-side traffic.')
print(json.dumps(dict(command=command, rows=rows, order='cells', static=static, meta=meta, step_total=dict(hbm_bytes_per_batch=tot, hbm_bytes_per_batch_raw=tot_raw,
                                                                          note='sum over the tcr:: kernels of one bench step'), kernels=res), indent=1))
