#!/bin/bash
# round 6, call 22: knock-outs of the k-loop of the split-K project GEMMs (K >= 1024: blocks 19-25), tune build, timing only
out=gpurun_out/r06z; mkdir -p $out
for dbg in 0 1 2 4 8 16 5 7 15 31; do
  COSY_TUNE_LIB=1 COSY_PW_DBG=$dbg timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_$dbg.txt
  echo "dbg $dbg: proj 19..25: $(grep -E '^ *(19|24|25) pw_gemm' $out/layers_$dbg.txt | awk '{print $(NF-5)}' | tr '\n' ' ')"
done
