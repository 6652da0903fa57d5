#!/bin/bash
# round 6, call 6 (after the container was re-created): state of the tree: whole GPU suite, then the driver's bench command, then layers
out=gpurun_out/r06f; mkdir -p $out
timeout 1700 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1; echo "tests rc $?"; tail -6 $out/tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06f/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], d['dtype'], 'dev', json.dumps(d.get('pose_deviation'))[:300])
print('other', json.dumps(d.get('other_dtypes')))
print('roofline', json.dumps(d.get('roofline'))[:600])
print('cpu', json.dumps(d.get('cpu_baseline'))[:400])
PY
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --layers > $out/bench_layers.json 2> $out/layers.txt; echo "layers rc $?"; tail -80 $out/layers.txt | cut -c1-120
