"""Build libcosyhip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'lib', 'libcosyhip.so')
SOURCES = ['kernels_geom.hip', 'kernels_dist.hip', 'kernels_raster.hip', 'kernels_train.hip', 'kernels_net.hip', 'effnet.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'cosyhip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wall', '-Wno-unused-function',
           '-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
