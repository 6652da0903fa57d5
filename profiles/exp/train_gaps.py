"""GPU idle gaps inside the training step: which launches the device waits for (run on the GPU box).
usage: python profiles/exp/train_gaps.py [bench_train args]   -> table on stdout"""
import json, sys, os, gzip, runpy, collections
import torch
from torch.profiler import profile, ProfilerActivity

sys.argv = ['bench_train.py'] + sys.argv[1:]
sys.path.insert(0, os.getcwd())
import bench_train

# run bench_train.main() but wrap the timed loop: simplest is to profile the whole main() with few steps
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    bench_train.main()
path = 'gpurun_out/tr4/trace.json'
os.makedirs('gpurun_out/tr4', exist_ok=True)
prof.export_chrome_trace(path)
ev = json.load(open(path))['traceEvents']
os.remove(path)
k = sorted([e for e in ev if e.get('cat') in ('kernel', 'gpu_memcpy', 'gpu_memset') and 'dur' in e], key=lambda e: e['ts'])
print('gpu events', len(k))
# steps are delimited by the adam kernel
adam = [i for i, e in enumerate(k) if 'adam_kernel' in e['name']]
print('adam launches', len(adam))
if len(adam) >= 3:
    a, b = adam[3], adam[4]            # one step of the un-synchronised timed loop (2 warm-up + 4 timed steps, then 3 event-timed ones): from the end of one adam to the end of the next
    seg = k[a:b + 1]
    t0, t1 = seg[0]['ts'] + seg[0]['dur'], seg[-1]['ts'] + seg[-1]['dur']
    busy = sum(e['dur'] for e in seg[1:])
    print('step wall %.2f ms, gpu busy (sum of durations) %.2f ms, launches %d' % ((t1 - t0) / 1e3, busy / 1e3, len(seg) - 1))
    gaps = []
    end = t0
    for prev, e in zip(seg, seg[1:]):
        g = e['ts'] - end
        gaps.append((g, prev['name'][:60], e['name'][:60], (e['ts'] - t0) / 1e3))
        end = max(end, e['ts'] + e['dur'])
    tot = sum(g for g, *_ in gaps if g > 0)
    print('idle total %.2f ms; gaps > 20 us: %.2f ms; gaps <= 20 us: %.2f ms (n=%d)' % (
        tot / 1e3, sum(g for g, *_ in gaps if g > 20) / 1e3, sum(g for g, *_ in gaps if 0 < g <= 20) / 1e3, sum(1 for g, *_ in gaps if 0 < g <= 20)))
    for g, p, n, at in sorted(gaps, reverse=True)[:25]:
        print('%8.1f us at %7.2f ms  after %-60s before %s' % (g, at, p, n))
    cpu = [e for e in ev if e.get('cat') in ('cpu_op', 'cuda_runtime', 'user_annotation') and 'dur' in e]
    for g, p, n, at in sorted(gaps, reverse=True)[:3]:
        g1 = t0 + at * 1e3            # the gap ends when the next event starts
        g0 = g1 - g
        inside = [e for e in cpu if e['ts'] < g1 and e['ts'] + e['dur'] > g0]
        agg = collections.defaultdict(lambda: [0.0, 0])
        for e in inside:
            a = agg[(e['cat'], e['name'][:70])]
            a[0] += min(e['ts'] + e['dur'], g1) - max(e['ts'], g0); a[1] += 1
        print('--- host activity inside the %.0f us gap at %.2f ms (overlap us, calls):' % (g, at))
        for (cat, name), (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
            print('   %8.1f %4d  %-14s %s' % (d, c, cat, name))
    hist = collections.Counter(min(int(g // 2) * 2, 20) for g, *_ in gaps if g > 0)
    print('gap histogram (us bucket: count):', sorted(hist.items()))
