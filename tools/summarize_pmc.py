"""Fold rocprofv3 --pmc counter_collection.csv files into per-kernel, per-batch HBM figures.

A batch = one bench step (one k_dense dispatch).  k_integrate is a chain of passes per batch, so
per-batch values (sum over the passes) are what compares with bench.py's per-batch algorithmic bytes;
per-dispatch averages are kept next to them."""
import collections
import csv
import glob
import json
import sys

out_dir = sys.argv[1]
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
        e['hbm_bytes_per_batch'] = 1024.0 * (2.0 * e['FETCH_SIZE_per_batch'] + e['WRITE_SIZE_per_batch'])
    if e.get('TCC_REQ_sum_per_batch'):
        e['l2_hit_rate'] = e['TCC_HIT_sum_per_batch'] / e['TCC_REQ_sum_per_batch']
rows = sys.argv[2] if len(sys.argv) > 2 else 'tc'
command = ('rocprofv3 --kernel-trace --output-format csv --pmc <FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum> -- '
           'python bench.py --steps 3 --warmup 1 --streams 1 --rows %s --no-cpu-baseline (three separate runs)' % rows)
meta = dict(workload='GL, 100000 storms per batch; the first batch of each run pads whole plane rows (pad_state = -1)',
            units='FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; *_per_batch = summed over the dispatches of a bench step',
            note='hbm_bytes_per_batch = 2 x FETCH_SIZE + WRITE_SIZE: MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B '
                 'while requests are 128-B line fills, so reads are under-reported 2x; calibrated there on 16-B/lane streaming loads, '
                 'these kernels issue 16-B/lane gathers (same request size).  WRITE_SIZE matches known byte counts (Fourier table '
                 '1.155 GB).  *_raw = FETCH_SIZE + WRITE_SIZE uncorrected.  Infinity-Cache hits are counted, so this is fabric-side traffic.')
print(json.dumps(dict(command=command, rows=rows, meta=meta, kernels=res), indent=1))
