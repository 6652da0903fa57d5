#!/bin/bash
# round 5, call 19: knock-out timing of the matrix-pipe small-map kernel: 1 = no tap phase, 2 = no expansion, 4 = no DMA of the next chunk
out=gpurun_out/r05u; mkdir -p $out
L="timeout 300 python bench.py --steps 6 --warmup 2 --layers --no-cpu-baseline --no-other-dtypes"
for d in 0 1 2 3 4 7; do
COSY_TUNE_LIB=1 COSY_SMALL_DBG=$d $L > $out/ko_$d.json 2> $out/ko_$d.txt
echo "smx dbg $d: $(grep 'mbconv_small' $out/ko_$d.txt | sed -n '1p;6p;7p' | awk '{print $1, $(NF-5)}' | tr '\n' ' ')"
done | tee $out/ko.txt
