#!/bin/bash
# round 6, call 24: what an EMPTY launch of the split-K project tile costs (128: return at once), and the launch + first stages without any wait (127 + 256)
out=gpurun_out/r06z; mkdir -p $out
for dbg in 0 128 383 127; do
  COSY_TUNE_LIB=1 COSY_PW_DBG=$dbg timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_$dbg.txt
  echo "dbg $dbg: proj 19 / 24 / 25: $(grep -E '^ *(19|24|25) pw_gemm' $out/layers_$dbg.txt | awk '{print $(NF-5)}' | tr '\n' ' ')  se19: $(grep -E '^ *19 se_fc1' $out/layers_$dbg.txt | awk '{print $(NF-5)}') pool: $(grep -E '^ *26 pool' $out/layers_$dbg.txt | awk '{print $(NF-5)}')"
done
