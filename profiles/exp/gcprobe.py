import gc, time, sys, runpy
t0 = {}
def cb(phase, info):
    if phase == 'start': t0['t'] = time.perf_counter()
    else:
        dt = (time.perf_counter() - t0['t']) * 1e3
        if dt > 2: print(f'[gc] gen{info["generation"]} {dt:.1f} ms collected={info["collected"]}', file=sys.stderr, flush=True)
gc.callbacks.append(cb)
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path('bench.py', run_name='__main__')
