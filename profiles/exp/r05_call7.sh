#!/bin/bash
# round 5, seventh GPU call: the whole GPU suite with the stem front at both crop widths; A/B against round 4; row-band sweep
out=gpurun_out/r05g; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2 3; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('r04', j['value'])"
$B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new', j['value'])"
done | tee $out/ab.txt
for i in 1 2; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so $B --crop 240x320 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('r04 240x320', j['value'])"
$B --crop 240x320 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new 240x320', j['value'])"
done | tee -a $out/ab.txt
for d in bf16; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so $B --dtype $d 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('r04 $d', j['value'])"
$B --dtype $d 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new $d', j['value'])"
done | tee -a $out/ab.txt
for r in 2 4 8 3 6; do
COSY_TUNE_LIB=1 COSY_STEM_RSPLIT=$r timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/rs_$r.json 2> $out/rs_$r.txt
echo "rsplit $r: $(grep stem_front $out/rs_$r.txt | head -1 | cut -c1-100)" | tee -a $out/ab.txt
done
timeout 300 python bench.py --crop 240x320 --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/layers_240.json 2> $out/layers_240.txt
head -8 $out/layers_240.txt | cut -c1-110
timeout 900 profiles/exp/pkf32_victim 12 > $out/pkf32_victim.txt 2>&1
