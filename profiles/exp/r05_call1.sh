#!/bin/bash
# round 5, first GPU call: packed-fp32 reproducer, the GPU suite, same-call A/B of the new library against round 4's
out=gpurun_out/r05a; mkdir -p $out
timeout 600 profiles/exp/pkf32_victim 12 > $out/pkf32_victim.txt 2>&1; echo "victim rc $?"; tail -5 $out/pkf32_victim.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -5 $out/pytest.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2 3; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('r04', j['value'])"
$B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new', j['value'])"
done | tee $out/ab.txt
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/layers_r04.json 2> $out/layers_r04.txt
timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/layers_new.json 2> $out/layers_new.txt
for d in bf16 fp32; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so $B --dtype $d 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('r04 $d', j['value'])"
$B --dtype $d 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new $d', j['value'])"
done | tee -a $out/ab.txt
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so $B --crop 240x320 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('r04 240x320', j['value'])" | tee -a $out/ab.txt
$B --crop 240x320 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new 240x320', j['value'])" | tee -a $out/ab.txt
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so python bench_train.py 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('r04 train', j['value'])" | tee -a $out/ab.txt
python bench_train.py 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new train', j['value'])" | tee -a $out/ab.txt
