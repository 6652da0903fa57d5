#!/bin/bash
# round 5, call 29: is a row of the matrix-pipe form latency- or contention-bound?  Blocks 14-17 (4 waves per SIMD) and 9-12 (5) with extra LDS per workgroup
# (COSY_WAVE_LDS_PAD, experiment knob) = 3 / 2 / 1 waves per SIMD; timeline of a job (stamps build)
out=gpurun_out/r05af; mkdir -p $out
S=$PWD/cosypose_amd/lib/libcosyhip_stamps.so
for pad in 0 22000 45000 90000; do
COSY_WAVE_LDS_PAD=$pad COSY_TUNE_LIB=$S timeout 120 python profiles/exp/wave_timeline.py --cmid 816 > $out/t816_$pad.txt 2>&1; echo "816 pad $pad: $(sed -n 3p $out/t816_$pad.txt | cut -c1-80) | $(sed -n 5p $out/t816_$pad.txt | cut -c80-200)"
COSY_WAVE_MASK=0x3dffc COSY_WAVE_LDS_PAD=$pad COSY_TUNE_LIB=$S timeout 120 python profiles/exp/wave_timeline.py --cmid 576 > $out/t576_$pad.txt 2>&1; echo "576 pad $pad: $(sed -n 3p $out/t576_$pad.txt | cut -c1-80) | $(sed -n 5p $out/t576_$pad.txt | cut -c80-200)"
done | tee $out/occ.txt
