#!/usr/bin/env python3
"""Instruction mix of one kernel of the library (device-only assembly, build.py's flags + any extra ones):
    python tools/isa_mix.py [--grep 'k_integrate<double, true, false, 2>'] [-DFLAG ...]
Prints fp64 add / mul / fma counts, IEEE division and sqrt sequences, all VALU, VMEM, LDS, waitcnt, and the register figures.
The numbers are static counts over the kernel's whole body, not executed counts."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tropical_cyclone_risk_amd import build as B        # noqa: E402

args = sys.argv[1:]
pat = 'k_integrate<double, true, false, 2>'
if '--grep' in args:
    i = args.index('--grep'); pat = args[i + 1]; del args[i:i + 2]
flags = [f for f in B.FLAGS if f not in ('-shared', '-fPIC')]
asm = '/tmp/tcr_isa_mix.s'
subprocess.check_call([B.hipcc()] + flags + args + ['--offload-device-only', '-S', '-o', asm, os.path.join(B.CSRC, 'tcr_abi.hip')], cwd=B.CSRC)
text = open(asm).read()
funcs = re.findall(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', text, flags=re.S | re.M)
names = subprocess.run(['c++filt'], input='\n'.join(f[0] for f in funcs), capture_output=True, text=True).stdout.splitlines()
for (sym, body), name in zip(funcs, names):
    if pat not in name:
        continue
    ops = collections.Counter()
    for line in body.splitlines():
        m = re.match(r'\s+([a-z][a-z0-9_]+)\b', line)
        if m:
            ops[re.sub(r'_(e32|e64|dpp|sdwa)$', '', m.group(1))] += 1
    def total(pred): return sum(n for o, n in ops.items() if pred(o))
    f64 = lambda o: o.endswith('_f64') or '_f64_' in o
    print(name[:110])
    print('  v_add_f64 %d  v_mul_f64 %d  v_fma_f64 %d  (other fma-class f64: %d)' % (
        ops['v_add_f64'], ops['v_mul_f64'], ops['v_fma_f64'], total(lambda o: f64(o) and ('fma' in o or 'mad' in o) and o != 'v_fma_f64')))
    print('  v_div_scale_f64 %d  v_div_fmas_f64 %d  v_div_fixup_f64 %d  v_rcp_f64 %d  v_rsq_f64 %d  v_sqrt_f64 %d' % (
        ops['v_div_scale_f64'], ops['v_div_fmas_f64'], ops['v_div_fixup_f64'], ops['v_rcp_f64'], ops['v_rsq_f64'], ops['v_sqrt_f64']))
    print('  fp64 VALU %d  all VALU %d  mfma %d' % (total(lambda o: o.startswith('v_') and f64(o) and 'mfma' not in o),
                                                   total(lambda o: o.startswith('v_') and 'mfma' not in o), total(lambda o: 'mfma' in o)))
    print('  global/flat loads %d  stores %d  ds %d  s_waitcnt %d  scratch %d  s_ %d' % (
        total(lambda o: re.match(r'(global|flat)_load', o) is not None), total(lambda o: re.match(r'(global|flat)_store', o) is not None),
        total(lambda o: o.startswith('ds_')), ops['s_waitcnt'], total(lambda o: o.startswith('scratch_')), total(lambda o: o.startswith('s_'))))
    regs = []
    for key in ('num_vgpr', 'num_agpr', 'numbered_sgpr', 'private_seg_size'):
        mm = re.search(r'\.set ' + re.escape(sym) + r'\.' + key + r', (\d+)', text)
        regs.append('%s %s' % (key, mm.group(1) if mm else '?'))
    print('  ' + '  '.join(regs))
