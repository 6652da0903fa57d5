#!/bin/bash
# round 6, call 25: bisecting the prologue of the split-K project tile by early returns
out=gpurun_out/r06z; mkdir -p $out
for dbg in 2048 1024 512 383 0; do
  COSY_TUNE_LIB=1 COSY_PW_DBG=$dbg timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_$dbg.txt
  echo "dbg $dbg: proj 19 / 24 / 25: $(grep -E '^ *(19|24|25) pw_gemm' $out/layers_$dbg.txt | awk '{print $(NF-5)}' | tr '\n' ' ')"
done
