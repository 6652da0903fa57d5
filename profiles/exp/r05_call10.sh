#!/bin/bash
# round 5: mbconv_rows_kernel (consecutive-pixel fragments, branch-free row loop) against mbconv_wave_kernel on the stride-1 blocks
out=gpurun_out/r05j; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q -k "storage_emulation or headline or full_batch or schedule or low_precision or config2 or config3" > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.txt
for m in 0 0x3fed8 0x18 0xc0 0x1e00 0x2000 0x3c000 0 0x3fed8; do
COSY_TUNE_LIB=1 COSY_ROWS_MASK=$m timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/rows_$m.json 2> $out/rows_$m.txt
python - <<PY | tee -a $out/ab.txt
import json
d=json.load(open("$out/rows_$m.json")); print("rows mask $m", d["value"], d["roofline"]["backbone_ms_per_forward"])
PY
done
grep "mbconv_rows_kernel\|mbconv_wave_kernel" $out/rows_0.txt | head -16 | cut -c1-110
grep "mbconv_rows_kernel\|mbconv_wave_kernel" $out/rows_0x3fed8.txt | head -16 | cut -c1-110
