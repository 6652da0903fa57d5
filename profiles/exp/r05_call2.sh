#!/bin/bash
# round 5, second GPU call: the packed-fp32 reproducer with the isolating instruction forms; A/B of the library with LDS-staged BatchNorm rows;
# squeeze-excite fusion mask A/B (tune build); s_memtime timeline of the wave front
out=gpurun_out/r05b; mkdir -p $out
timeout 900 profiles/exp/pkf32_victim 12 > $out/pkf32_victim.txt 2>&1; echo "victim rc $?"
timeout 900 python -m pytest tests -m gpu -x -q -k "ddp or schedule or storage_emulation or full_batch or training_step or headline" > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2 3; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('r04', j['value'])"
$B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new', j['value'])"
done | tee $out/ab.txt
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/layers_r04.json 2> $out/layers_r04.txt
timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/layers_new.json 2> $out/layers_new.txt
for i in 1 2; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so python bench_train.py 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('r04 train', j['value'])" | tee -a $out/ab.txt
python bench_train.py 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new train', j['value'])" | tee -a $out/ab.txt
done
# tune build: squeeze-excite in the project GEMM's prologue for blocks 5-13 (shipped) vs 5-17
for m in 0x3fe0 0x3ffe0 0x3fe0 0x3ffe0; do
COSY_TUNE_LIB=1 COSY_SE_FUSE_MASK=$m timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/se_$m.json 2> $out/se_$m.txt
python - <<PY | tee -a $out/ab.txt
import json
d=json.load(open("$out/se_$m.json")); print("tune se mask $m", d["value"], d["roofline"]["backbone_ms_per_forward"])
PY
done
for c in 816 192 576 144; do
COSY_TUNE_LIB=1 timeout 120 python profiles/exp/wave_timeline.py --cmid $c > $out/timeline_$c.txt 2>&1
done
COSY_TUNE_LIB=1 timeout 120 python profiles/exp/wave_timeline.py --cmid 816 --crops 1024 > $out/timeline_816_1024.txt 2>&1
