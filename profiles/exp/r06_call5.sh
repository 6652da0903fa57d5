#!/bin/bash
# round 6, call 5: bf16 with hi + lo weight pairs: the bf16 / fp16 parity tests, then bench
out=gpurun_out/r06e; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "storage_emulation or headline or refiner_loop_low or full_batch or config2 or config3 or render_crop_pack" > $out/tests.txt 2>&1; echo "tests rc $?"; grep -E "passed|failed|FAILED|Error" $out/tests.txt | tail -15
timeout 900 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06e/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], d['dtype'], 'dev', json.dumps(d['pose_deviation'])[:200])
print('other', json.dumps(d['other_dtypes']))
PY
timeout 600 python bench.py --steps 8 --warmup 3 --dtype bf16 --no-cpu-baseline --no-other-dtypes --layers > $out/bench_bf16.json 2> $out/layers_bf16.txt; echo "bench bf16 rc $?"; tail -45 $out/layers_bf16.txt | cut -c1-110
