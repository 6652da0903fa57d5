#!/bin/bash
# Compile the reference's only native file (cosypose/csrc/cosypose_cext.cpp, 269 lines, pybind11 + STL)
# from where it lies under /root/reference into oracle/_ref/.  No reference source is copied
# into this repo; the output .so is git-ignored (it still travels to the GPU box with gpurun).
# Used only to pin the index/assignment restatements (scatter_argmin, expand_ids_for_symmetry).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${REF:-/root/reference}
SRC="$REF/cosypose/csrc/cosypose_cext.cpp"
[ -f "$SRC" ] || { echo "reference not mounted at $REF; skipping oracle/_ref build"; exit 0; }
mkdir -p "$HERE/_ref"
g++ -O3 -shared -std=c++17 -fPIC $(python3 -m pybind11 --includes) "$SRC" \
    -o "$HERE/_ref/cosypose_cext$(python3-config --extension-suffix)"
echo "built $HERE/_ref/cosypose_cext$(python3-config --extension-suffix)"
