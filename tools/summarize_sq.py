"""Fold the rocprofv3 --pmc passes of tools/collect_sq.sh into one JSON: per stream count, per kernel, counter sums per bench
step (a step = one k_dense dispatch) and the ratios DESIGN.md quotes.

SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); ratios between them need no unit.  The
issue / stall split of a kernel is WAIT_ANY : WAIT_INST_ANY : ACTIVE_INST_ANY over WAVE_CYCLES (disjoint, sum ~ 1)."""
import collections
import csv
import glob
import json
import os
import re
import sys

out_dir = sys.argv[1]
res = {}
for s in (1,):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in glob.glob('%s/s%d_p*/**/*counter_collection.csv' % (out_dir, s), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').strip()
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            disp[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
    if not acc:
        continue
    batches = {c: len(v) for (k, c), v in disp.items() if k.startswith('tcr::k_dense')}
    kernels = {}
    for k, cs in acc.items():
        if not k.startswith('tcr::'):
            continue
        e = {}
        for c, v in cs.items():
            nb = batches.get(c) or 1
            e[c] = v / nb
            e['dispatches_per_step_' + c] = len(disp[(k, c)]) / nb
        wc = e.get('SQ_WAVE_CYCLES')
        if wc:
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_LDS',
                      'SQ_ACTIVE_INST_SCA', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_MISC', 'SQ_ACTIVE_INST_FLAT'):
                if c in e:
                    e['frac_' + c] = e[c] / wc
            if 'SQ_INSTS_VALU' in e:
                e['quad_cycles_per_valu_inst'] = wc / e['SQ_INSTS_VALU']
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; TA_BUSY_avr is the average over the TA instances (one per CU) of their busy
        # cycles: busy fraction of an average TA while the kernel's dispatches were on the GPU
        if e.get('GRBM_GUI_ACTIVE') and e.get('TA_BUSY_avr'):
            e['ta_busy_fraction'] = e['TA_BUSY_avr'] / (e['GRBM_GUI_ACTIVE'] / 8.0)
        if e.get('SQ_ACTIVE_INST_VALU') and e.get('SQ_INSTS_VALU'):
            e['issue_quad_cycles_per_valu_inst'] = e['SQ_ACTIVE_INST_VALU'] / e['SQ_INSTS_VALU']
        kernels[k] = e
    res['kernels'] = kernels
for s in (1, 12):
    bj = os.path.join(out_dir, 'bench_streams%d.json' % s)
    try:
        b = json.loads(open(bj).read().strip().splitlines()[-1])
        r = b['roofline']
        e = dict(ms_per_step=b['ms_per_step'], value=b['value'], chain_ms_exclusive=r['launch_ms'], integrate_passes=r.get('integrate_passes'),
                 integrate_passes_pipelined=r.get('integrate_passes_pipelined'))
        pp = (r.get('integrate_passes_pipelined') or {}).get('simd_time_ms')
        if pp:
            e['share_of_simd_time_with_an_integrator_wave_resident'] = pp / b['ms_per_step']
        res['bench_streams_%d' % s] = e
    except Exception as ex:
        res['bench_streams_%d' % s] = 'unavailable: %s' % ex
meta = dict(command='rocprofv3 --kernel-trace --output-format csv --pmc <pass> -- python bench.py --steps 3 --warmup 1 --streams S --no-cpu-baseline '
                    '(one run per pass, passes listed in tools/collect_sq.sh; rocprofv3 serialises dispatches while counters are collected)',
            units='counter sums per bench step (all dispatches of that kernel in one step, all SEs / XCDs summed as rocprofv3 reports them); '
                  'SQ cycle counters are quad-cycles',
            passes=open(os.path.join(out_dir, 'passes.txt')).read().splitlines() if os.path.exists(os.path.join(out_dir, 'passes.txt')) else None)
print(json.dumps(dict(meta=meta, **res), indent=1))
