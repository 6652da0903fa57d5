#!/bin/bash
# round 5, fifth GPU call: stem front as 3-wave workgroups (L1 sharing); crops ordered by frame
out=gpurun_out/r05e; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q -k "storage_emulation and 256x256 or crop or roi_align or schedule or headline_config or render_crop" > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.txt
for v in 0 1 0 1; do
COSY_TUNE_LIB=1 COSY_STEM_FRONT=$v timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/stemf_$v.json 2> $out/stemf_$v.txt
python - <<PY | tee -a $out/ab.txt
import json
d=json.load(open("$out/stemf_$v.json")); print("tune stem front $v", d["value"], d["roofline"]["backbone_ms_per_forward"])
PY
done
grep "^ *-1\|^ *0 " $out/stemf_0.txt $out/stemf_1.txt | cut -c1-120
for v in 0 1 0 1; do
COSY_TUNE_LIB=1 COSY_CROP_BY_FRAME=$v timeout 300 python profiles/exp/crop_bench.py 2>/dev/null | sed "s/^/by_frame=$v /" | tee -a $out/crop.txt
done
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r04_ship.so $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('r04', j['value'])"
$B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new', j['value'])"
done | tee -a $out/ab.txt
