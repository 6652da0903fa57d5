#!/bin/bash
# round 5, call 31: where inside a row the cycles go (phase stamps, experiment build: profiles/exp/wave_phases.diff), at 4 waves per SIMD and with a wave alone on its SIMD
out=gpurun_out/r05ah; mkdir -p $out
S=$PWD/cosypose_amd/lib/libcosyhip_stamps.so
for pad in 0 90000; do
COSY_WAVE_PHASES=1 COSY_WAVE_LDS_PAD=$pad COSY_TUNE_LIB=$S timeout 120 python profiles/exp/wave_timeline.py --cmid 816 > $out/p816_$pad.txt 2>&1; echo "blocks 14-17 pad $pad: $(grep phases $out/p816_$pad.txt | cut -c1-400)"
COSY_WAVE_PHASES=1 COSY_WAVE_MASK=0x3dffc COSY_WAVE_LDS_PAD=$pad COSY_TUNE_LIB=$S timeout 120 python profiles/exp/wave_timeline.py --cmid 576 > $out/p576_$pad.txt 2>&1; echo "blocks 9-12 pad $pad: $(grep phases $out/p576_$pad.txt | cut -c1-400)"
done | tee $out/phases.txt
