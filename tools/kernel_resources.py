#!/usr/bin/env python3
"""Compile the library with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs / AGPRs / scratch bytes per lane / occupancy (waves per SIMD) / LDS bytes.  Extra hipcc flags pass through:
    python tools/kernel_resources.py [-DTCR_EMIT_WPS=4 ...] [--grep integrate]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tropical_cyclone_risk_amd import build as B        # noqa: E402

args = sys.argv[1:]
pat = None
if '--grep' in args:
    i = args.index('--grep'); pat = args[i + 1]; del args[i:i + 2]
cmd = [B.hipcc()] + B.FLAGS + args + ['-Rpass-analysis=kernel-resource-usage', '-o', '/tmp/tcr_resources.so',
                                      os.path.join(B.CSRC, 'tcr_abi.hip')]
out = subprocess.run(cmd, cwd=B.CSRC, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r'remark: \S+ +(Function Name|Name): (\S+)', line)
    if m:
        cur = dict(name=subprocess.run(['c++filt', m.group(2)], capture_output=True, text=True).stdout.strip()[:70])
        rows.append(cur)
        continue
    m = re.search(r'(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
    if m and cur is not None:
        cur[m.group(1).split()[0]] = int(m.group(2))
print('%-70s %5s %5s %7s %4s %6s' % ('kernel', 'VGPR', 'AGPR', 'scratch', 'occ', 'LDS'))
for r in rows:
    if pat and pat not in r['name']:
        continue
    print('%-70s %5s %5s %7s %4s %6s' % (r['name'], r.get('VGPRs'), r.get('AGPRs'), r.get('ScratchSize'), r.get('Occupancy'), r.get('LDS')))
