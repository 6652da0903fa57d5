#!/bin/bash
# round 5, call 16: the whole GPU suite with the segment form of the matrix-pipe taps (blocks 3, 4, 6, 7 too); 240x320 A/B
out=gpurun_out/r05v; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.txt | cut -c1-300
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r05_nomx.so $B --crop 240x320 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('nomx 240x320', j['value'])"
$B --crop 240x320 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('smx  240x320', j['value'])"
done | tee $out/ab.txt
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r05_nomx.so $B --dtype bf16 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('nomx bf16', j['value'])" | tee -a $out/ab.txt
$B --dtype bf16 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('smx  bf16', j['value'])" | tee -a $out/ab.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 > $out/bench_full.json; python -c "
import json; j=json.load(open('$out/bench_full.json')); print(j['value'], j.get('pose_deviation'), j.get('other_dtypes'))"
