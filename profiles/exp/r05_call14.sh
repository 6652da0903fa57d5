#!/bin/bash
# round 5, call 14: blocks 9-12 (k3, taps on the matrix pipe) at 5 waves per SIMD (96 VGPRs)
out=gpurun_out/r05q; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "storage_emulation or bit_identical or backbone or determin" > $out/pytest_emul.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_emul.txt | cut -c1-400
L="timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes"
$L > $out/layers_seg.json 2> $out/layers_seg.txt
grep "mbconv_wave" $out/layers_seg.txt | sed -n "1,16p" | cut -c1-100
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r05_nomx.so $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('nomx  ', j['value'])"
$B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('seg   ', j['value'])"
done | tee $out/ab.txt
