#!/bin/bash
# round 5, call 30: software-pipelined rows of the matrix-pipe form (16-pixel-row blocks 9-17): emulation parity, per-launch times, headline
out=gpurun_out/r05ag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "storage_emulation or bit_identical or backbone or determin or full_batch" > $out/pytest_emul.txt 2>&1; echo "pytest rc $?"; tail -5 $out/pytest_emul.txt | cut -c1-400
L="timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes"
$L > $out/layers.json 2> $out/layers.txt
grep "mbconv_wave" $out/layers.txt | sed -n "8,16p" | cut -c1-100
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2; do $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('pipe', j['value'])"; done | tee $out/ab.txt
