#!/bin/bash
# round 6, call 18: 7x10 / 10x7 maps on FOUR waves (waves 0-2 share the fifth segment's three tiles): parity, 240x320 timing
out=gpurun_out/r06s; mkdir -p $out
timeout 1700 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "storage_emulation or headline or schedule or config2 or config3 or pose_predictor or coarse_refine" > $out/tests.txt 2>&1; echo "tests rc $?"; grep -E "passed|failed|FAILED|Error" $out/tests.txt | tail -8
for c in 1 0 1 0; do
  COSY_TUNE_LIB=1 COSY_SMALL_MX=$c timeout 600 python bench.py --crop 240x320 --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --no-profile > $out/b240_$c.json 2> /dev/null
  echo "240x320 small_mx $c $(python -c "import json;d=json.loads(open('$out/b240_$c.json').read().strip().split(chr(10))[-1]);print(d['value'])")"
done
timeout 600 python bench.py --crop 240x320 --steps 6 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_240.txt
grep -E "mbconv_small" $out/layers_240.txt | head -8 | cut -c1-105
timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_256.txt
grep -E "mbconv_small" $out/layers_256.txt | head -8 | cut -c1-105
