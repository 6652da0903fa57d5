#!/bin/bash
# round 6, call 12: bf16 with the hi + lo pairs inside the 8x8-map matrix-pipe front (blocks 19-25 fused again): parity, bench, layers
out=gpurun_out/r06m; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or headline or schedule or refiner_loop_low" > $out/tests.txt 2>&1; echo "tests rc $?"; grep -E "passed|failed|FAILED|Error" $out/tests.txt | tail -8
timeout 900 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06m/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], d['dtype'], 'dev', json.dumps(d['pose_deviation'])[:200])
print('other', json.dumps(d['other_dtypes']))
PY
timeout 600 python bench.py --steps 8 --warmup 3 --dtype bf16 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > $out/bench_bf16.json 2> $out/layers_bf16.txt; echo "bench bf16 rc $?"; grep -E "^ *(19|2[0-6]) " $out/layers_bf16.txt | cut -c1-110; tail -30 $out/layers_bf16.txt | head -8
