"""Kernel concurrency of a bench run from a rocprofv3 kernel trace: span of the traced kernels, time with >= 1 / >= 2 kernels resident,
sum of kernel durations -- the evidence for what the second HIP stream buys (another chunk's kernels inside the ramp-up / drain of each launch).
usage: python profiles/overlap.py <dir with *_kernel_trace.csv> [label]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
label = sys.argv[2] if len(sys.argv) > 2 else f
rows = [r for r in csv.DictReader(open(f)) if 'copyBuffer' not in r['Kernel_Name'] and 'fillBuffer' not in r['Kernel_Name']]
ev = []
for r in rows:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
# the steady part: between the first and the last crop kernel of the run
crops = sorted(int(r['Start_Timestamp']) for r in rows if 'crop_pack' in r['Kernel_Name'] or 'crop_geometry' in r['Kernel_Name'])
lo, hi = crops[len(crops) // 4], crops[-1]
depth, prev, t1, t2, tsum = 0, None, 0, 0, 0
for t, d in ev:
    if prev is not None and depth > 0:
        a, b = max(prev, lo), min(t, hi)
        if b > a:
            t1 += b - a
            tsum += (b - a) * depth
            if depth >= 2:
                t2 += b - a
    depth += d
    prev = t
span = hi - lo
print(f'{label}: span {span / 1e6:.2f} ms | >= 1 kernel resident {100 * t1 / span:.1f} % | >= 2 kernels resident {100 * t2 / span:.1f} % | '
      f'sum of kernel durations / span {tsum / span:.3f} | {len(rows)} kernels traced')
