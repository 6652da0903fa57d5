#!/bin/bash
# round 6, call 17: the matrix-pipe small front on the 7x10 / 10x7 maps (240x320 / 320x240 crops), the vectorised pool kernel: parity, then 240x320 A/B (COSY_SMALL_MX knob)
out=gpurun_out/r06r; mkdir -p $out
timeout 1700 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "storage_emulation or headline or schedule or config2 or config3 or pose_predictor or coarse_refine or backbone or full_batch" > $out/tests.txt 2>&1; echo "tests rc $?"; grep -E "passed|failed|FAILED|Error" $out/tests.txt | tail -8
for c in 1 0 1 0; do
  COSY_TUNE_LIB=1 COSY_SMALL_MX=$c timeout 600 python bench.py --crop 240x320 --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --no-profile > $out/b240_$c.json 2> /dev/null
  echo "240x320 small_mx $c $(python -c "import json;d=json.loads(open('$out/b240_$c.json').read().strip().split(chr(10))[-1]);print(d['value'])")"
done
timeout 600 python bench.py --crop 240x320 --steps 6 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_240.txt
grep -E "^ *(19|2[0-6]) " $out/layers_240.txt | cut -c1-105
timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > $out/bench256.json 2> $out/layers_256.txt
grep -E "^ *26 " $out/layers_256.txt | cut -c1-105
for i in 1 2; do timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-dtypes --no-profile 2>/dev/null | python -c "import json,sys;print('256 bench', json.loads(sys.stdin.read().strip().split(chr(10))[-1])['value'])"; done
