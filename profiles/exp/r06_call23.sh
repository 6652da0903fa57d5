#!/bin/bash
# round 6, call 23: the fixed cost of the split-K project GEMMs: knock-outs of prologue (32: gate rows) and epilogue (64), with and without the k-loop's work (31)
out=gpurun_out/r06z; mkdir -p $out
for dbg in 0 32 64 96 31 63 95 127; do
  COSY_TUNE_LIB=1 COSY_PW_DBG=$dbg timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_$dbg.txt
  echo "dbg $dbg: proj 19 / 24 / 25: $(grep -E '^ *(19|24|25) pw_gemm' $out/layers_$dbg.txt | awk '{print $(NF-5)}' | tr '\n' ' ')  se19: $(grep -E '^ *19 se_fc1' $out/layers_$dbg.txt | awk '{print $(NF-5)}')"
done
