#!/bin/bash
# round 5, call 21: block 18 (k5 s2 on the 16x16 map, Cin 136) on a wave variant <5,2,5,2,false,3> instead of expand GEMM + dwconv
out=gpurun_out/r05x; mkdir -p $out
L="timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes"
COSY_TUNE_LIB=1 $L > $out/l_base.json 2> $out/l_base.txt
COSY_TUNE_LIB=1 COSY_WAVE_MASK=0x7fffc $L > $out/l_w18.json 2> $out/l_w18.txt
grep "^ 18 " $out/l_base.txt | cut -c1-110; echo ---; grep "^ 18 " $out/l_w18.txt | cut -c1-110
python -c "
import json
for t in ('base','w18'):
    j=json.load(open('$out/l_%s.json'%t)); print(t, j['value'], j['roofline']['backbone_ms_per_forward'])"
