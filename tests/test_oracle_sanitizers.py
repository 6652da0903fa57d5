"""The C restatement (oracle/cosy_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer: the oracle is what every
parity claim rests on, so out-of-bounds reads / signed overflow / misaligned accesses in it must not hide behind plausible
numbers.  The golden tests of the geometry, roi_align, pose update, index ops and rasteriser run in a subprocess on the
-fsanitize=address,undefined build with libasan preloaded (sanitizer builds of the native code)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import REPO


@pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc')
def test_oracle_golden_tests_under_asan_ubsan():
    sys.path.insert(0, str(REPO / 'oracle'))
    import cosy_oracle
    cosy_oracle.build(sanitize=True)
    libasan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip('libasan.so not found next to gcc')
    env = dict(os.environ, COSY_ORACLE_SANITIZE='1', LD_PRELOAD=libasan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=99',
               UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1:exitcode=98', OMP_NUM_THREADS='2')
    r = subprocess.run([sys.executable, '-m', 'pytest', str(REPO / 'tests' / 'test_oracle_golden.py'), '-q', '-x', '-p', 'no:cacheprovider',
                        '-k', 'not backbone and not forward and not predictor'], capture_output=True, text=True, timeout=900, env=env, cwd=str(REPO))
    out = r.stdout[-3000:] + r.stderr[-3000:]
    assert 'AddressSanitizer' not in out and 'runtime error' not in out, out
    assert r.returncode == 0 and ' passed' in r.stdout, out
