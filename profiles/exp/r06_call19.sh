#!/bin/bash
# round 6, call 19: does the hand-ordered interior row (row_px) pay now that the loads run at full rate?  same tree built with -DCOSY_WAVE_PIPE=0 against the shipped library
out=gpurun_out/r06t; mkdir -p $out
for lib in ship nopipe ship nopipe; do
  if [ $lib = ship ]; then unset COSY_TUNE_LIB; else export COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r06_nopipe.so; fi
  timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > $out/b_$lib.json 2> $out/layers_$lib.txt
  echo "$lib $(python -c "import json;d=json.loads(open('$out/b_$lib.json').read().strip().split(chr(10))[-1]);print(d['value'])") $(grep -E '^ *(9|13|14) mbconv_wave' $out/layers_$lib.txt | awk '{print $(NF-5)}' | tr '\n' ' ')"
done
