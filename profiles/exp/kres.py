#!/usr/bin/env python3
"""Per-kernel resource usage of one .hip file: hipcc -Rpass-analysis=kernel-resource-usage, one line per kernel.
usage: kres.py file.hip [name filter] [extra hipcc flags...]"""
import re
import subprocess
import sys

src, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Rpass-analysis=kernel-resource-usage',
                    '--cuda-device-only', '-c', src, '-o', '/dev/null'] + sys.argv[3:], capture_output=True, text=True)
t = r.stderr
rows = set()
for b in re.split(r'remark: [^\n]*Function Name: ', t)[1:]:
    name = b.split('\n')[0].strip().split()[0]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if filt not in dn:
        continue
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    dn = dn.replace('cosy::', '').replace('void ', '').split('(')[0]
    rows.add('%-64s vgpr %3s agpr %3s sgpr %3s occ %s scratch %s lds %s' % (dn[:64], g('VGPRs'), g('AGPRs'), g('SGPRs'), g(r'Occupancy \[waves/SIMD\]'),
                                                                       g(r'ScratchSize \[bytes/lane\]'), g(r'LDS Size \[bytes/block\]')))
print('\n'.join(sorted(rows)))
if r.returncode:
    print(t[-3000:])
