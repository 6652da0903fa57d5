"""cosy_effnet_b3_create's HOST side -- the blob walk, every weight packer (GEMM fragment blocks incl. bf16's hi + lo pairs, wave / small / stem-front parameter blocks,
Toeplitz fragments), the bump allocator and its host mirror -- executed on the CPU: a five-function stand-in for the HIP runtime (malloc-backed hipMalloc / hipMemcpy /
hipMemset) is preloaded in front of libamdhip64, glibc's heap checker is on (MALLOC_CHECK_=3), and every dtype x crop-size family is created and destroyed.
Round 6: a packer that wrote bf16 pairs into a buffer sized for single values (stem front) corrupted the heap on the GPU box only -- nothing on the CPU ran this code."""
import os
import subprocess
import sys
import textwrap

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
int hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? 0 : 2; }
int hipFree(void* p) { free(p); return 0; }
int hipMemcpy(void* d, const void* s, size_t n, int kind) { (void)kind; memcpy(d, s, n); return 0; }
int hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
int hipDeviceSynchronize(void) { return 0; }
'''

DRIVER = r'''
import ctypes, sys
import numpy as np
sys.path.insert(0, %(repo)r)
from cosypose_amd import _lib
l = _lib.lib()
n = l.cosy_effnet_b3_param_count()
blob = (np.random.RandomState(0).randn(n) * 0.1).astype(np.float32)
blob[blob == 0] = 0.1
for dt in (2, 1, 0):
    for (H, W) in ((256, 256), (240, 320), (224, 224), (416, 416)):
        h = ctypes.c_void_p()
        rc = l.cosy_effnet_b3_create(blob.ctypes.data, n, dt, H, W, 3, ctypes.byref(h))
        assert rc == 0, (dt, H, W, l.cosy_last_error())
        dims = (ctypes.c_int * 11)()
        assert l.cosy_effnet_b3_block_info(h, 14, dims) == 0
        assert l.cosy_effnet_b3_workspace_bytes(h) > 0
        assert l.cosy_effnet_b3_destroy(h) == 0
print('created and destroyed 12 engines')
'''


def test_create_packs_every_engine_on_a_cpu_stub(tmp_path):
    from cosypose_amd import build as hipbuild
    if not os.path.exists(hipbuild.LIB):
        pytest.skip('libcosyhip.so not built')
    src = tmp_path / 'hipstub.c'
    src.write_text(STUB)
    so = tmp_path / 'libhipstub.so'
    subprocess.check_call(['gcc', '-shared', '-fPIC', '-O1', '-o', str(so), str(src)])
    drv = tmp_path / 'drv.py'
    drv.write_text(textwrap.dedent(DRIVER % dict(repo=REPO)))
    env = dict(os.environ, LD_PRELOAD=str(so), MALLOC_CHECK_='3', MALLOC_PERTURB_='165')
    r = subprocess.run([sys.executable, str(drv)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'created and destroyed 12 engines' in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
