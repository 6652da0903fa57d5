#!/bin/bash
# round 5, fourth GPU call: the fused stem + block-0 front: parity, then A/B
out=gpurun_out/r05d; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q -k "storage_emulation or headline or full_batch or schedule or low_precision or forward_vs_reference or config2 or config3 or backbone_module" > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('r04', j['value'])"
$B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new', j['value'])"
done | tee $out/ab.txt
timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/layers_new.json 2> $out/layers_new.txt
for v in 0 1 0 1; do
COSY_TUNE_LIB=1 COSY_STEM_FRONT=$v timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/stemf_$v.json 2> $out/stemf_$v.txt
python - <<PY | tee -a $out/ab.txt
import json
d=json.load(open("$out/stemf_$v.json")); print("tune stem front $v", d["value"], d["roofline"]["backbone_ms_per_forward"])
PY
done
head -12 $out/layers_new.txt
timeout 600 profiles/exp/pkf32_victim 12 > $out/pkf32_victim.txt 2>&1
