#!/bin/bash
# round 5, call 17: knock-out timing of the small-map kernel (blocks 19-25): 1 = no depthwise phase, 2 = no expansion, 4 = no weight DMA
out=gpurun_out/r05s; mkdir -p $out
L="timeout 300 python bench.py --steps 6 --warmup 2 --layers --no-cpu-baseline --no-other-dtypes"
for d in 0 1 2 3 4 7; do
COSY_TUNE_LIB=1 COSY_SMALL_DBG=$d $L > $out/ko_$d.json 2> $out/ko_$d.txt
echo "small dbg $d: $(grep 'mbconv_small' $out/ko_$d.txt | sed -n '1p;6p;7p' | awk '{print $1, $(NF-5)}' | tr '\n' ' ')"
done | tee $out/ko.txt
