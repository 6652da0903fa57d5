#!/bin/bash
# round 6, call 13: (a) both squeeze-excite FCs of blocks 14-25 in one launch; (b) matrix-pipe taps for rows with a partial last segment (240x320: 15x20 / 30x40 maps)
out=gpurun_out/r06n; mkdir -p $out
timeout 1700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or headline or schedule or refiner_loop_low or config2 or config3 or pose_predictor or coarse_refine or full_batch" > $out/tests.txt 2>&1; echo "tests rc $?"; grep -E "passed|failed|FAILED|Error|assert" $out/tests.txt | tail -8
for c in 1 0 1 0; do
  COSY_TUNE_LIB=1 COSY_SE_ONE=$c timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --no-profile > $out/bench_se$c.json 2> /dev/null
  echo "se_one $c $(python -c "import json;d=json.loads(open('$out/bench_se$c.json').read().strip().split(chr(10))[-1]);print(d['value'])")"
done
COSY_TUNE_LIB=1 COSY_SE_ONE=1 timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_se1.txt
grep -E "se_f" $out/layers_se1.txt | cut -c1-100 | head -14
for i in 1 2; do
  timeout 600 python bench.py --crop 240x320 --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --no-profile > $out/bench_240_$i.json 2> /dev/null
  echo "240x320 $(python -c "import json;d=json.loads(open('$out/bench_240_$i.json').read().strip().split(chr(10))[-1]);print(d['value'])")"
done
timeout 600 python bench.py --crop 240x320 --steps 6 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_240.txt
grep -E "mbconv_wave|mbconv_small" $out/layers_240.txt | head -20 | cut -c1-105
