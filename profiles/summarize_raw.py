#!/usr/bin/env python3
"""Raw per-(kernel, grid) PMC means: profiles/summarize_raw.py <dir> [kernel-substring]"""
import csv, glob, sys
from collections import defaultdict
sys.path.insert(0, 'profiles')
from summarize import short
root = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ''
ctr = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for f in glob.glob(f'{root}/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        key = (short(r['Kernel_Name']), r['Grid_Size'], r['Workgroup_Size'], r['LDS_Block_Size'], r['VGPR_Count'])
        ctr[key][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'SQ_WAVES':
            dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for key in sorted(ctr, key=lambda k: -sum(dur.get(k, [0]))):
    if filt not in key[0]:
        continue
    m = {n: sum(v) / len(v) for n, v in ctr[key].items()}
    d = sum(dur[key]) / max(len(dur[key]), 1)
    print(f'\n{key[0]} grid={key[1]} wg={key[2]} lds={key[3]} vgpr={key[4]}  ~{d:.0f} us (under PMC)')
    w = m.get('SQ_WAVES', 1)
    for n in sorted(m):
        extra = f'  per-wave {m[n] / w:10.1f}' if n.startswith('SQ_INSTS') or 'CYCLES' in n or 'ACTIVE' in n or 'WAIT' in n else ''
        print(f'   {n:28s} {m[n]:16.0f}{extra}')
