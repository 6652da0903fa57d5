#!/bin/bash
# round 5, call 22: early-segment sample chunks again (tune build), now that the fronts are faster: D tensors of a chunk staying in the Infinity Cache
out=gpurun_out/r05y; mkdir -p $out
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for c in 0 64 32 16; do
COSY_TUNE_LIB=1 COSY_EARLY_CHUNK=$c $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('early chunk $c', j['value'], j['ms_per_step'])"
done | tee $out/chunk.txt
