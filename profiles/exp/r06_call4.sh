#!/bin/bash
# round 6, call 4: the whole GPU suite on the tree with the fused wave rows
out=gpurun_out/r06d; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1; echo "tests rc $?"; tail -6 $out/tests.txt
