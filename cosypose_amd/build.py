"""Build libcosyhip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every .hip source is compiled to its own object (in parallel, only when it or a header changed), then linked:
an edit to one kernel file costs one compile, not six.
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libcosyhip.so')
SOURCES = ['kernels_geom.hip', 'kernels_dist.hip', 'kernels_raster.hip', 'kernels_train.hip', 'kernels_net.hip',
           'effnet.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    return hs + [os.path.join(HERE, '..', 'include', 'cosyhip.h')]


def _obj(src):
    return os.path.join(LIBDIR, os.path.splitext(src)[0] + '.o')


def _stale_sources(force):
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    out = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), _obj(s)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            out.append(s)
    return out


def build(force=False, verbose=False, extra_flags=()):
    os.makedirs(LIBDIR, exist_ok=True)
    stale = _stale_sources(force)
    if not stale and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(_obj(s)) for s in SOURCES):
        return LIB

    def compile_one(s):
        cmd = [HIPCC] + FLAGS + list(extra_flags) + ['-c', os.path.join(CSRC, s), '-o', _obj(s)]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(len(stale), os.cpu_count() or 1) or 1) as ex:
        list(ex.map(compile_one, stale))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + [_obj(s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    import sys
    print(build(force='--force' in sys.argv, verbose=True))
