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
SOURCES = ['kernels_geom.hip', 'kernels_dist.hip', 'kernels_raster.hip', 'kernels_train.hip', 'kernels_net.hip', 'kernels_small.hip',
           'kernels_dw.hip', 'kernels_wave.hip', 'kernels_stem.hip', 'kernels_smx.hip', 'effnet.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# No packed-fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) anywhere in the library.  Round 4: a wave's packed-fp32 results were WRONG while
# another wave on its SIMD -- another HIP stream's kernel -- issued 16-bit MFMAs with VGPR accumulators (the wave-autonomous fronts): the rasteriser, whose
# SLP-vectorised code was dense in v_pk_*_f32, lost triangles in 20-30 % of its renders beside the backbone and in NONE once these instructions were gone
# (profiles/r04_raster_streams.txt; stand-alone victim / aggressor pair: profiles/exp/pkf32_victim.hip).  The instructions issue no faster than two scalar
# ones on gfx950 (profiles/exp/valu_bench.hip), so nothing is lost; the SLP vectoriser stays on where it helps.  Round 5: the feature is off for EVERY object
# (round 4 had kept it in kernels_net.hip, whose <5,2> GEMM tiles needed 186-191 VGPRs beside their 80 AGPRs without the packed forms; their epilogue now
# walks the tile column group by column group -- 8 instead of 40 live BatchNorm registers -- and sits at 152).  tests/test_build_isa.py disassembles every
# shipped object and fails on a single v_pk_{mul,add,fma}_f32.  The host half of a compile does not know the feature and says so.
NO_PACKED_FP32 = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function'] + NO_PACKED_FP32
# Per-file flags.  kernels_wave.hip: hipcc's SLP vectoriser pairs the depthwise FMAs of the wave front into v_pk_fma_f32 -- which
# issues no faster than two v_fma_f32 on gfx950 (profiles/exp/valu_bench.hip: 5.85 vs 2 x 3.1 cycles) -- and pays for the pairing
# with register shuffles (61 v_mov_b32 per row of the k=5 shape) and un-fused multiply + add tails.  Scalar FMAs, no moves.
FILE_FLAGS = {'kernels_wave.hip': ['-fno-slp-vectorize'], 'kernels_dw.hip': ['-fno-slp-vectorize'], 'kernels_stem.hip': ['-fno-slp-vectorize'], 'kernels_smx.hip': ['-fno-slp-vectorize']}   # kernels_dw.hip: see its header


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
        if force or not os.path.exists(obj) or not os.path.exists(_res_file(s, tune)) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            out.append(s)
    return out


WAVE_ISA = os.path.join(LIBDIR, 'kernels_wave.gfx950.s')     # device assembly of THE compile that produced kernels_wave.o
ISA_STAMP = os.path.join(LIBDIR, 'wave_isa_check.json')


def _sha16(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        for blk in iter(lambda: f.read(1 << 20), b''):
            h.update(blk)
    return h.hexdigest()[:16]


def build(force=False, verbose=False, tune=False):
    """The shipping library, or with tune=True the experiment build (reads COSY_* environment knobs).
    The shipping build ALWAYS ends with the wave-kernel ISA check (on the assembly hipcc wrote while compiling the shipped
    object) and a stamp that binds the verdict to the linked library's hash: _lib.lib() refuses a library without a matching
    clean stamp, so neither a lazy rebuild nor a different hipcc can put unchecked wave kernels in front of a caller."""
    import shutil
    import tempfile
    from . import wave_isa
    os.makedirs(LIBDIR, exist_ok=True)
    if wave_isa.write_fence_include() and verbose:     # csrc/wave_fence.inc follows the variant tables of kernels_wave.hip
        print('regenerated', wave_isa.FENCE_INC, flush=True)
    stale = _stale_sources(force, tune)
    if not tune and 'kernels_wave.hip' not in stale and not os.path.exists(WAVE_ISA):
        stale.append('kernels_wave.hip')
    LIB = TUNE_LIB if tune else globals()['LIB']
    if not stale and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(_obj(s, tune)) for s in SOURCES):
        if not tune and not stamp_matches(LIB):
            check_wave_isa(verbose=verbose)
        return LIB

    def run(cmd, s):
        # -Rpass-analysis=kernel-resource-usage makes the compile that produces the object also report every kernel's registers, scratch and
        # occupancy (remarks on stderr): parsed into lib/<source>.res.json -> kernel_resources() / tests/test_build_isa.py
        r = subprocess.run(cmd + ['-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed on {s}:\n' + '\n'.join(l for l in r.stderr.split('\n') if 'remark:' not in l)[-6000:])
        _write_resources(s, r.stderr, tune)

    def compile_one(s):
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(s, []) + (['-DCOSY_TUNE'] + (['-DCOSY_WAVE_STAMPS'] if os.environ.get('COSY_WAVE_STAMPS') else []) if tune else []) + ['-c', os.path.join(CSRC, s), '-o', _obj(s, tune)]
        if s == 'kernels_wave.hip' and not tune:
            # keep the device assembly of this very compile: the ISA check reads what was linked, not a second compile
            tmp = tempfile.mkdtemp(prefix='cosy_wave_')
            try:
                cmd = cmd[:-1] + [os.path.join(tmp, 'kernels_wave.o'), '-save-temps=obj']
                if verbose:
                    print(' '.join(cmd), flush=True)
                run(cmd, s)
                shutil.move(os.path.join(tmp, 'kernels_wave-hip-amdgcn-amd-amdhsa-gfx950.s'), WAVE_ISA)
                shutil.move(os.path.join(tmp, 'kernels_wave.o'), _obj(s, tune))
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
            return
        if verbose:
            print(' '.join(cmd), flush=True)
        run(cmd, s)

    with ThreadPoolExecutor(max_workers=min(len(stale), os.cpu_count() or 1) or 1) as ex:
        list(ex.map(compile_one, stale))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + [_obj(s, tune) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    if not tune:
        check_wave_isa(verbose=verbose)
    return LIB


def _res_file(src, tune=False):
    return os.path.join(LIBDIR, os.path.splitext(src)[0] + ('.tune' if tune else '') + '.res.json')


def _write_resources(src, remarks, tune=False):
    import json
    import re
    out = {}
    for blk in re.split(r'remark: [^\n]*Function Name: ', remarks)[1:]:
        name = blk.split('\n')[0].strip().split()[0]
        g = lambda k: int(re.search(k + r': (\d+)', blk).group(1))
        out[name] = dict(vgpr=g('VGPRs'), agpr=g('AGPRs'), sgpr=g('SGPRs'), scratch=g(r'ScratchSize \[bytes/lane\]'),
                         occupancy=g(r'Occupancy \[waves/SIMD\]'), lds=g(r'LDS Size \[bytes/block\]'))
    json.dump(out, open(_res_file(src, tune), 'w'), indent=0, sort_keys=True)


def kernel_resources(demangle=True):
    """{kernel: {vgpr, agpr, sgpr, scratch (bytes per lane), occupancy (waves per SIMD), lds (static bytes)}} of every kernel of the shipping
    library, as hipcc reported them while compiling the shipped objects."""
    import json
    out = {}
    for src in SOURCES:
        if os.path.exists(_res_file(src)):
            out.update(json.load(open(_res_file(src))))
    if demangle and out:
        names = list(out)
        dm = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.strip().split('\n')
        if len(dm) == len(names):
            out = {d.replace('void ', '', 1).split('(')[0]: out[n] for d, n in zip(dm, names)}
    return out


def _csrc_sha():
    """Hash of EVERY source the library is built from (all of csrc/, the public header, the compile flags): the stamp binds the linked
    library to these, so a .so older than an edited kernels_smx.hip / kernels_stem.hip / ... can never load (round 5's stamp covered
    kernels_wave.hip and its headers only; the other hand-scheduled files were guarded by mtimes alone)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(('.hip', '.h', '.inc')):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), 'rb').read())
    h.update(open(os.path.join(HERE, '..', 'include', 'cosyhip.h'), 'rb').read())
    h.update(' '.join(FLAGS + [f'{k}:{" ".join(v)}' for k, v in sorted(FILE_FLAGS.items())]).encode())
    return h.hexdigest()[:16]


_wave_src_sha = _csrc_sha      # (old name, kept for the scripts under profiles/)


def read_stamp():
    import json
    try:
        return json.load(open(ISA_STAMP))
    except (OSError, ValueError):
        return None


def stamp_matches(lib_path=None):
    """True when the stamp says `clean` for exactly this library file (and, where the sources are present, these sources)."""
    st = read_stamp()
    lib_path = lib_path or LIB
    if not st or not st.get('clean') or not os.path.exists(lib_path) or st.get('lib_sha') != _sha16(lib_path):
        return False
    return not os.path.exists(os.path.join(CSRC, 'kernels_wave.hip')) or st.get('src_sha') == _csrc_sha()


def check_wave_isa(verbose=False):
    """kernels_wave.hip keeps its input fragments in registers the compiler is not told about (see the file); that is sound
    only while no compiler-generated instruction touches them.  wave_isa.check_file verifies it on the assembly of the
    compile that produced the shipped object; the build FAILS when it is not clean (a different hipcc may allocate
    differently).  The verdict is stamped next to the library: source hash, hash of the linked library, compiler version."""
    import json
    from . import wave_isa
    if not os.path.exists(WAVE_ISA) or os.path.getmtime(WAVE_ISA) < os.path.getmtime(os.path.join(CSRC, 'kernels_wave.hip')):
        raise RuntimeError(f'{WAVE_ISA} is missing or older than kernels_wave.hip: rebuild with build(force=True)')
    log = []
    problems, n, _ = wave_isa.check_file(WAVE_ISA, verbose=verbose, out=log.append)
    if verbose:
        print('\n'.join(log)[-3000:], flush=True)
    ver = subprocess.run([HIPCC, '--version'], capture_output=True, text=True).stdout.strip().split('\n')
    clean = not problems
    json.dump(dict(clean=clean, src_sha=_csrc_sha(), lib_sha=_sha16(LIB), kernels=n, hipcc=[l for l in ver if l][:2],
                   checked='device assembly of the compile that produced kernels_wave.o (-save-temps=obj)', summary=log[-1] if clean else problems[0]),
              open(ISA_STAMP, 'w'), indent=1)
    if not clean:
        raise RuntimeError('wave-kernel ISA check failed (reserved VGPR range touched, scratch, wrong allocation or kernel count):\n' + '\n'.join(log)[-3000:])
    return True


if __name__ == '__main__':
    import sys
    print(build(force='--force' in sys.argv, verbose=True, tune='--tune' in sys.argv))
