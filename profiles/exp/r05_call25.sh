#!/bin/bash
# round 5, call 25: timelines of the blocks that share a Cmid with a LATER wave launch (the stamp buffer keeps the last matching launch): the later block is
# taken off the wave kernel (COSY_WAVE_MASK) so that blocks 3 / 4, 6 / 7, 9-12 are the last launch with their Cmid
out=gpurun_out/r05aa; mkdir -p $out
S=$PWD/cosypose_amd/lib/libcosyhip_stamps.so
COSY_WAVE_MASK=0x3ffdc COSY_TUNE_LIB=$S timeout 120 python profiles/exp/wave_timeline.py --cmid 192 > $out/timeline_192_blocks34.txt 2>&1; sed -n 2,6p $out/timeline_192_blocks34.txt | cut -c1-260
COSY_WAVE_MASK=0x3fefc COSY_TUNE_LIB=$S timeout 120 python profiles/exp/wave_timeline.py --cmid 288 > $out/timeline_288_blocks67.txt 2>&1; sed -n 2,6p $out/timeline_288_blocks67.txt | cut -c1-260
COSY_WAVE_MASK=0x3dffc COSY_TUNE_LIB=$S timeout 120 python profiles/exp/wave_timeline.py --cmid 576 > $out/timeline_576_blocks9_12.txt 2>&1; sed -n 2,6p $out/timeline_576_blocks9_12.txt | cut -c1-260
