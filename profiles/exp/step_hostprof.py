"""host-side profile of ONE get_predictions step (steady state and right after a drain) -- run on the GPU box"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
sys.argv = ['bench.py', '--steps', '3', '--warmup', '3', '--no-cpu-baseline', '--no-other-dtypes', '--no-profile']
import torch
import bench
from cosypose_amd import pose_predictor as pp
orig = pp.CoarseRefinePosePredictor.get_predictions
state = {'n': 0}
def wrapped(self, *a, **k):
    state['n'] += 1
    if state['n'] == 4:           # a step right after a drain
        torch.cuda.synchronize()
        pr = cProfile.Profile(); t = time.perf_counter(); pr.enable()
        r = orig(self, *a, **k)
        pr.disable(); dt = time.perf_counter() - t
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
        print('host time of the step: %.2f ms' % (dt * 1e3), file=sys.stderr); print(s.getvalue()[:9000], file=sys.stderr)
        return r
    return orig(self, *a, **k)
pp.CoarseRefinePosePredictor.get_predictions = wrapped
bench.main()
