"""Timeline of one bench step from a rocprofv3 kernel trace: gaps (GPU idle) and non-backbone kernels.
usage: python profiles/timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'crop_pack' in r['Kernel_Name']]
i0 = idx[-10] if len(idx) >= 10 else idx[0]
seq = rows[i0:idx[-5]]          # one full step (crop_pack #1 of step k .. crop_pack #1 of step k+1)
t0 = int(seq[0]['Start_Timestamp']); prev_end = t0; busy = 0; gaps = []
for r in seq:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s - prev_end > 20000: gaps.append(((prev_end - t0) / 1e6, (s - prev_end) / 1e3, r['Kernel_Name'][:60]))
    busy += e - max(s, prev_end) if e > prev_end else 0
    prev_end = max(prev_end, e)
span = prev_end - t0
print(f'step span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms, kernels {len(seq)}')
for g in gaps: print(f'  gap at {g[0]:7.3f} ms: {g[1]:8.1f} us before {g[2]}')
