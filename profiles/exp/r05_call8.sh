#!/bin/bash
# round 5, eighth GPU call: block 0's project + block 1's depthwise as one front: parity, A/B, row-band sweep
out=gpurun_out/r05h; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q -k "storage_emulation or headline or full_batch or schedule or low_precision or forward_vs_reference or config2 or config3 or backbone_module or bench_launches or third_crop" > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.txt
for v in 0 1 0 1; do
COSY_TUNE_LIB=1 COSY_PROJ_FRONT=$v timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/projf_$v.json 2> $out/projf_$v.txt
python - <<PY | tee -a $out/ab.txt
import json
d=json.load(open("$out/projf_$v.json")); print("tune proj front $v", d["value"], d["roofline"]["backbone_ms_per_forward"])
PY
done
grep "^ *0 \|^ *1 " $out/projf_0.txt $out/projf_1.txt | cut -c1-130
for r in 2 3 4 6; do
COSY_TUNE_LIB=1 COSY_PROJ_RSPLIT=$r timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/prs_$r.json 2> $out/prs_$r.txt
echo "proj rsplit $r: $(grep proj_front $out/prs_$r.txt | head -1 | cut -c1-100)" | tee -a $out/ab.txt
done
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('r04', j['value'])"
$B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new', j['value'])"
done | tee -a $out/ab.txt
