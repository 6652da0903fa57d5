#!/bin/bash
# round 5, call 20: the schedule knobs again on the new kernels: streams x crops per launch at the headline workload
out=gpurun_out/r05w; mkdir -p $out
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for cfg in "1 256" "2 128" "2 64" "3 86" "4 64" "3 128" "2 256"; do
set -- $cfg
$B --streams $1 --bsz-objects $2 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('streams $1 bsz $2', j['value'], j['ms_per_step'])"
done | tee $out/streams.txt
