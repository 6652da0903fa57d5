#!/bin/bash
# round 5, call 18: the 8x8-map front with its taps on the matrix pipe (kernels_smx.hip): emulation parity, per-launch times, headline A/B
out=gpurun_out/r05t; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "storage_emulation or bit_identical or backbone or determin" > $out/pytest_emul.txt 2>&1; echo "pytest rc $?"; tail -12 $out/pytest_emul.txt | cut -c1-600
L="timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes"
$L > $out/layers_smx.json 2> $out/layers_smx.txt
grep "mbconv_small" $out/layers_smx.txt | head -8 | cut -c1-110
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r05_nomx.so $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('nomx  ', j['value'])"
$B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('smx   ', j['value'])"
done | tee $out/ab.txt
