#!/usr/bin/env python3
"""Kept for the command lines in DESIGN.md / the earlier rounds' notes: the checker lives in the package now
(cosypose_amd/wave_isa.py, run by cosypose_amd.build on every build of kernels_wave.hip)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosypose_amd.wave_isa import main  # noqa: E402

if __name__ == '__main__':
    sys.exit(main())
