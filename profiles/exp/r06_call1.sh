#!/bin/bash
# round 6, call 1: the new GPU tests (f-3 on the device, large-capacity engine, packed upload) + the driver's bench command on the tree after the small closes
out=gpurun_out/r06a; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "io_formats_round_trip or engine_capacity or packed_model_upload or native_library or headline_config" > $out/tests.txt 2>&1; echo "tests rc $?"; tail -5 $out/tests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06a/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
print('dev', json.dumps(d['pose_deviation']))
print('other', json.dumps(d['other_dtypes']))
print('cpu', json.dumps({k:v for k,v in d['cpu_baseline'].items() if k in ('value','cores','all_cores','value_one_process')}))
print(d.get('notes'))
PY
