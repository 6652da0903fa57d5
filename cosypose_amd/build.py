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
           'kernels_dw.hip', 'kernels_wave.hip', 'effnet.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# Per-file flags.  kernels_wave.hip: hipcc's SLP vectoriser pairs the depthwise FMAs of the wave front into v_pk_fma_f32 -- which
# issues no faster than two v_fma_f32 on gfx950 (profiles/exp/valu_bench.hip: 5.85 vs 2 x 3.1 cycles) -- and pays for the pairing
# with register shuffles (61 v_mov_b32 per row of the k=5 shape) and un-fused multiply + add tails.  Scalar FMAs, no moves.
FILE_FLAGS = {'kernels_wave.hip': ['-fno-slp-vectorize'], 'kernels_dw.hip': ['-fno-slp-vectorize']}   # kernels_dw.hip: see its header


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    return hs + [os.path.join(HERE, '..', 'include', 'cosyhip.h')]


TUNE_LIB = os.path.join(LIBDIR, 'libcosyhip_tune.so')   # -DCOSY_TUNE build: env knobs + phase knock-outs, experiments only


def _obj(src, tune=False):
    return os.path.join(LIBDIR, os.path.splitext(src)[0] + ('.tune.o' if tune else '.o'))


def _stale_sources(force, tune=False):
    hdr_t = max([os.path.getmtime(h) for h in _headers()] + [os.path.getmtime(os.path.abspath(__file__))])   # build.py holds the flags
    out = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), _obj(s, tune)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            out.append(s)
    return out


def build(force=False, verbose=False, tune=False):
    """The shipping library, or with tune=True the experiment build (reads COSY_* environment knobs)."""
    os.makedirs(LIBDIR, exist_ok=True)
    stale = _stale_sources(force, tune)
    LIB = TUNE_LIB if tune else globals()['LIB']
    if not stale and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(_obj(s, tune)) for s in SOURCES):
        return LIB

    def compile_one(s):
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(s, []) + (['-DCOSY_TUNE'] if tune else []) + ['-c', os.path.join(CSRC, s), '-o', _obj(s, tune)]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(len(stale), os.cpu_count() or 1) or 1) as ex:
        list(ex.map(compile_one, stale))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + [_obj(s, tune) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


ISA_STAMP = os.path.join(LIBDIR, 'wave_isa_check.json')


def _wave_src_sha():
    import hashlib
    h = hashlib.sha256()
    for f in ['kernels_wave.hip', 'net_device.h', 'kernels_net.h', 'cosy_common.h']:
        h.update(open(os.path.join(CSRC, f), 'rb').read())
    h.update(' '.join(FLAGS + FILE_FLAGS.get('kernels_wave.hip', [])).encode())
    return h.hexdigest()[:16]


def check_wave_isa(verbose=False):
    """kernels_wave.hip keeps its input fragments in registers the compiler is not told about (see the file); that is sound
    only while no compiler-generated instruction touches them.  profiles/check_wave_isa.py verifies it on the ISA hipcc
    generates with the shipping flags; the build FAILS when it is not clean (a different hipcc may allocate differently).
    The verdict is stamped next to the library (source hash + compiler version) so that a GPU-side test can tell whether
    the library it loads was checked."""
    import json
    import sys
    script = os.path.join(HERE, '..', 'profiles', 'check_wave_isa.py')
    r = subprocess.run([sys.executable, script] + (['-v'] if verbose else []), capture_output=True, text=True)
    if verbose:
        print(r.stdout[-3000:], flush=True)
    ver = subprocess.run([HIPCC, '--version'], capture_output=True, text=True).stdout.strip().split('\n')
    clean = r.returncode == 0 and 'checked 55 wave kernels' in r.stdout
    json.dump(dict(clean=clean, src_sha=_wave_src_sha(), hipcc=[l for l in ver if l][:2], summary=r.stdout.strip().split('\n')[-1]),
              open(ISA_STAMP, 'w'), indent=1)
    if not clean:
        raise RuntimeError('wave-kernel ISA check failed (reserved VGPR range touched, scratch, or wrong allocation):\n' +
                           r.stdout[-3000:] + r.stderr[-1000:])
    return True


if __name__ == '__main__':
    import sys
    print(build(force='--force' in sys.argv, verbose=True, tune='--tune' in sys.argv))
    if '--tune' not in sys.argv:
        check_wave_isa(verbose=True)
