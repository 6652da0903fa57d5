#!/usr/bin/env python3
"""Batch sweep table: for 256..2048 crops per forward, the headline throughput and, per GEMM-like
kernel, mean duration, matrix-core utilisation (rocprofv3 MfmaUtil) and the MFMA FLOP rate issued
(SQ_INSTS_VALU_MFMA_MOPS_{F16,BF16} x 512 / duration).  Input: the directory profiles/collect_all.sh writes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize import short


def main(root):
    Bs = [256, 512, 1024, 2048]
    val, table = {}, defaultdict(dict)
    for B in Bs:
        try:
            val[B] = json.load(open(f'{root}/sweep_B{B}.json'))['value']
        except Exception:
            val[B] = float('nan')
        dur, util, mops = defaultdict(list), defaultdict(list), defaultdict(list)
        for f in glob.glob(f'{root}/sweep_pmc_B{B}/**/*kernel_trace.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        for f in glob.glob(f'{root}/sweep_pmc_B{B}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                if r['Counter_Name'] == 'MfmaUtil':
                    util[k].append(float(r['Counter_Value']))
                elif r['Counter_Name'] in ('SQ_INSTS_VALU_MFMA_MOPS_BF16', 'SQ_INSTS_VALU_MFMA_MOPS_F16'):
                    mops[k].append(float(r['Counter_Value']))
        for k in dur:
            if util.get(k) and mops.get(k) and sum(mops[k]) > 0:
                t = sum(dur[k]) / len(dur[k])
                table[k][B] = (t, sum(util[k]) / len(util[k]), sum(mops[k]) / len(mops[k]) * 512 / (t * 1e-6) / 1e12, len(dur[k]))
    print('pose-iterations/s by crops per forward: ' + '  '.join(f'B={B}: {val[B]:.0f}' for B in Bs))
    print('per kernel: mean us | MfmaUtil % | MFMA TFLOP/s issued (launch counts differ with the batch chunking)')
    keys = sorted(table, key=lambda k: -sum(v[0] * v[3] for v in table[k].values()))
    for k in keys:
        cells = []
        for B in Bs:
            v = table[k].get(B)
            cells.append(f'{v[0]:8.1f} {v[1]:5.1f}% {v[2]:6.1f}' if v else ' ' * 22)
        print(f'{k[:58]:58s} ' + ' | '.join(cells))


if __name__ == '__main__':
    main(sys.argv[1])
