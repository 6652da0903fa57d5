"""Static checks on the compiled wave kernels (no GPU needed: hipcc cross-compiles gfx950).

kernels_wave.hip loads its input fragments by inline asm into registers the compiler knows nothing about; that is sound
only while no compiler-generated instruction touches the reserved range and nothing spills.  profiles/check_wave_isa.py
verifies it on the ISA of the shipping build flags."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import REPO


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_wave_kernels_reserved_registers_and_no_scratch():
    r = subprocess.run([sys.executable, os.path.join(REPO, 'profiles', 'check_wave_isa.py')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert 'checked 55 wave kernels' in r.stdout
