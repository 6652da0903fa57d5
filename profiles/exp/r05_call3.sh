#!/bin/bash
# round 5, third GPU call: reproducer incl. dependent-pair forms; SE mask A/B (tune build); wave-front timelines (stamps build)
out=gpurun_out/r05c; mkdir -p $out
timeout 900 profiles/exp/pkf32_victim 12 > $out/pkf32_victim.txt 2>&1; echo "victim rc $?"
for m in 0x3fe0 0x3ffe0 0x3fe0 0x3ffe0; do
COSY_TUNE_LIB=1 COSY_SE_FUSE_MASK=$m timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/se_$m.json 2> $out/se_$m.txt
python - <<PY | tee -a $out/ab.txt
import json
d=json.load(open("$out/se_$m.json")); print("tune se mask $m", d["value"], d["roofline"]["backbone_ms_per_forward"])
PY
done
S=$PWD/cosypose_amd/lib/libcosyhip_stamps.so
for c in 816 192 576 144 288; do
COSY_TUNE_LIB=$S timeout 120 python profiles/exp/wave_timeline.py --cmid $c > $out/timeline_$c.txt 2>&1
done
COSY_TUNE_LIB=$S timeout 120 python profiles/exp/wave_timeline.py --cmid 816 --crops 1024 --stride 149 > $out/timeline_816_1024.txt 2>&1
COSY_TUNE_LIB=$S timeout 120 python profiles/exp/wave_timeline.py --cmid 816 --crops 128 --stride 19 > $out/timeline_816_128.txt 2>&1
COSY_TUNE_LIB=$S timeout 200 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/layers_stamps.json 2> $out/layers_stamps.txt
