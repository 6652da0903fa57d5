#!/bin/bash
# (config, crops per forward, streams) sweep of bench.py: which chunking / concurrency each workload wants
OUT=gpurun_out/streams_sweep.txt; : > $OUT
run() { echo "== $*" >> $OUT; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-dtypes --no-profile "$@" 2>>$OUT.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['bsz_objects'], d['config']['streams'])" >> $OUT; }
run --streams 1
run --streams 2
run --streams 2 --bsz-objects 64
run --streams 3
run --streams 3 --bsz-objects 64
run --streams 1
run --crop 240x320 --streams 1
run --crop 240x320 --streams 2
run --config 2 --streams 1
run --config 2 --streams 2
run --config 2 --streams 3
run --config 2 --streams 3 --bsz-objects 256
run --config 2 --streams 2 --bsz-objects 256
run --config 3 --split balanced --streams 1
run --config 3 --split balanced --streams 3 --bsz-objects 128
run --config 3 --split balanced --streams 2 --bsz-objects 128
run --renderer hip --streams 1
run --renderer hip --streams 2
cat $OUT
