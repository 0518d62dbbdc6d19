"""Fold the two rocprofv3 --pmc runs of build_variants/calibrate_fetch (tools/calibrate_fetch.hip) into the
measured-to-true ratios of FETCH_SIZE / WRITE_SIZE per access pattern -> profiles/r03_fetch_calibration.json."""
import collections
import csv
import glob
import json
import sys

out_dir = sys.argv[1]
GiB = float(1 << 30)
LINES = (1 << 30) // 128
TRUE = {   # kernel -> (counter, true HBM bytes, what "true" means)
    'cal_stream_read': ('FETCH_SIZE', GiB, '1 GiB read once, 16 B per lane, coalesced'),
    'cal_gather_line': ('FETCH_SIZE', LINES * 128.0, 'every 128-B line of 1 GiB once, 7 x 16 B per lane, lanes scattered (112 B of each line requested)'),
    'cal_gather_16': ('FETCH_SIZE', LINES * 128.0, 'one 16-B piece of every 128-B line of 1 GiB (16 B requested per line; "true" = whole-line fills)'),
    'cal_stream_write': ('WRITE_SIZE', GiB, '1 GiB written once, 16 B per lane, coalesced'),
    'cal_scatter_line': ('WRITE_SIZE', LINES * 112.0, '7 x 16 B of every 128-B line of 1 GiB, lanes scattered'),
    'cal_flush': ('WRITE_SIZE', GiB, '1 GiB fill between the kernels'),
}
vals = collections.defaultdict(list)
for f in glob.glob(out_dir + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        vals[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
res = {}
for k, (c, true, what) in TRUE.items():
    v = vals.get((k, c))
    if not v:
        continue
    v = sorted(v)[len(v) // 2]                    # median of the repetitions
    res[k] = dict(counter=c, pattern=what, true_bytes=true, reported_KiB=v, reported_bytes=v * 1024.0,
                  true_over_reported=true / (v * 1024.0))
    other = 'WRITE_SIZE' if c == 'FETCH_SIZE' else 'FETCH_SIZE'
    o = vals.get((k, other))
    if o:
        res[k]['other_counter_bytes'] = {other: sorted(o)[len(o) // 2] * 1024.0}
meta = dict(command='rocprofv3 --kernel-trace --output-format csv --pmc <FETCH_SIZE | WRITE_SIZE> -- build_variants/calibrate_fetch '
                    '(two runs; tools/calibrate_fetch.hip; every kernel touches a 1 GiB buffer, 4x the Infinity Cache, exactly once)',
            units='rocprofv3 reports both counters in KiB')
print(json.dumps(dict(meta=meta, kernels=res), indent=1))
