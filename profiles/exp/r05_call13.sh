#!/bin/bash
# round 5, call 13: where the rows of the matrix-pipe form spend their time: s_memtime timeline + timing knock-outs (tune build)
out=gpurun_out/r05n; mkdir -p $out
S=$PWD/cosypose_amd/lib/libcosyhip_stamps.so
for c in 816 576; do
COSY_TUNE_LIB=$S timeout 120 python profiles/exp/wave_timeline.py --cmid $c > $out/timeline_$c.txt 2>&1; tail -6 $out/timeline_$c.txt | cut -c1-250
done
L="timeout 300 python bench.py --steps 6 --warmup 2 --layers --no-cpu-baseline --no-other-dtypes"
for d in 0 1 2 3 64 128 256 512 960 963; do
COSY_TUNE_LIB=1 COSY_WAVE_DBG=$d $L > $out/ko_$d.json 2> $out/ko_$d.txt
echo "dbg $d: $(grep 'mbconv_wave' $out/ko_$d.txt | sed -n '8p;12p;13p' | awk '{print $1, $(NF-5)}' | tr '\n' ' ')"
done | tee $out/ko.txt
