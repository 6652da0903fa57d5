#!/bin/bash
# round 5, call 11: depthwise taps on the matrix pipe (wave kernel, blocks 9-17 at 256x256): emulation parity first, then per-launch times and the
# headline against the same tree built with -DCOSY_WAVE_MX=0 (cosypose_amd/lib/r05_nomx.so)
out=gpurun_out/r05l; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "storage_emulation or bit_identical or backbone" > $out/pytest_emul.txt 2>&1; echo "pytest rc $?"; tail -15 $out/pytest_emul.txt | cut -c1-400
timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/layers_mx.json 2> $out/layers_mx.txt
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r05_nomx.so timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/layers_nomx.json 2> $out/layers_nomx.txt
grep "mbconv_wave" $out/layers_nomx.txt | head -16 | cut -c1-120
echo ---
grep "mbconv_wave" $out/layers_mx.txt | head -16 | cut -c1-120
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2 3; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r05_nomx.so $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('nomx', j['value'])"
$B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('mx  ', j['value'], j.get('pose_deviation'))"
done | tee $out/ab.txt
for d in bf16; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r05_nomx.so $B --dtype $d 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('nomx $d', j['value'])"
$B --dtype $d 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('mx   $d', j['value'], j.get('pose_deviation'))"
done | tee -a $out/ab.txt
