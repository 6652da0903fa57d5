#!/bin/bash
# round 5, call 24: s_memtime timelines of the final wave kernels (stamps build)
out=gpurun_out/r05aa; mkdir -p $out
S=$PWD/cosypose_amd/lib/libcosyhip_stamps.so
for c in 816 576 192 144 288; do
COSY_TUNE_LIB=$S timeout 120 python profiles/exp/wave_timeline.py --cmid $c > $out/timeline_$c.txt 2>&1; sed -n 2,6p $out/timeline_$c.txt | cut -c1-260
done
