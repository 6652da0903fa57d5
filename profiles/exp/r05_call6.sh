#!/bin/bash
# round 5, sixth GPU call: stem front with consecutive-pixel fragments (DPP neighbours)
out=gpurun_out/r05f; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q -k "storage_emulation and 256x256 or schedule or headline_config or full_batch" > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.txt
for v in 0 1 0 1; do
COSY_TUNE_LIB=1 COSY_STEM_FRONT=$v timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/stemf_$v.json 2> $out/stemf_$v.txt
python - <<PY | tee -a $out/ab.txt
import json
d=json.load(open("$out/stemf_$v.json")); print("tune stem front $v", d["value"], d["roofline"]["backbone_ms_per_forward"])
PY
done
grep "^ *-1\|^ *0 " $out/stemf_0.txt $out/stemf_1.txt | cut -c1-120
