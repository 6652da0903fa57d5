#!/bin/bash
# round 6, call 26: the k-loop's shell: one step instead of 22 (4096), with everything else knocked out / with everything in place
out=gpurun_out/r06z; mkdir -p $out
for dbg in 4479 383 4096 0 4104; do
  COSY_TUNE_LIB=1 COSY_PW_DBG=$dbg timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_$dbg.txt
  echo "dbg $dbg: proj 19 / 24 / 25: $(grep -E '^ *(19|24|25) pw_gemm' $out/layers_$dbg.txt | awk '{print $(NF-5)}' | tr '\n' ' ')"
done
