#!/bin/bash
# round 6, call 28: blocks 19-23's project conv on 8 waves in two K-groups with 64 x 64 per wave (tune build, COSY_PW16_ALT) against the 16-wave split-K tile
out=gpurun_out/r06ad; mkdir -p $out
COSY_TUNE_LIB=1 COSY_PW16_ALT=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation and 256x256 and fp16" > $out/tests.txt 2>&1; echo "tests rc $?"; tail -2 $out/tests.txt
for alt in 1 0 1 0; do
  COSY_TUNE_LIB=1 COSY_PW16_ALT=$alt timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > $out/b_$alt.json 2> $out/layers_$alt.txt
  echo "alt $alt: $(python -c "import json;d=json.loads(open('$out/b_$alt.json').read().strip().split(chr(10))[-1]);print(d['value'], d['roofline']['backbone_ms_per_forward'])") proj 19-23: $(grep -E '^ *(19|20|21|22|23) pw_gemm' $out/layers_$alt.txt | awk '{print $(NF-5)}' | tr '\n' ' ')"
done
