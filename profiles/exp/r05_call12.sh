#!/bin/bash
# round 5, call 12: the expansion MFMAs read the input fragments in place (no per-row copies); with / without the weight fragments parked in LDS
out=gpurun_out/r05m; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "storage_emulation or bit_identical or backbone or determin" > $out/pytest_emul.txt 2>&1; echo "pytest rc $?"; tail -5 $out/pytest_emul.txt | cut -c1-400
L="timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes"
$L > $out/layers_xasm.json 2> $out/layers_xasm.txt
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r05_nowlds.so $L > $out/layers_nowlds.json 2> $out/layers_nowlds.txt
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r05_nomx.so $L > $out/layers_nomx.json 2> $out/layers_nomx.txt
for v in nomx xasm nowlds; do echo "--- $v"; grep "mbconv_wave" $out/layers_$v.txt | head -16 | cut -c1-100; done
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2 3; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r05_nomx.so $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('nomx  ', j['value'])"
$B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('xasm  ', j['value'])"
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r05_nowlds.so $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('nowlds', j['value'])"
done | tee $out/ab.txt
