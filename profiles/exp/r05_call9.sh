#!/bin/bash
# round 5, final validation: whole GPU suite, smoke, the driver's command
out=gpurun_out/r05i; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err; python -c "
import json; d=json.load(open('$out/bench_driver_cmd.json')); print('driver cmd', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
COSY_DIST_BACKEND=gloo timeout 600 python bench_train.py --gpus 2 --steps 3 --warmup 1 --batch 16 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-900
