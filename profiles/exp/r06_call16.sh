#!/bin/bash
# round 6, call 16: unequal chunks on the two streams (the streams then drift apart instead of running the same kernel kinds side by side)
out=gpurun_out/r06q; mkdir -p $out
for bsz in 128 144 160 176 128 144 160 192; do
  timeout 300 python bench.py --steps 12 --warmup 3 --streams 2 --bsz-objects $bsz --no-cpu-baseline --no-other-dtypes --no-profile > $out/b_$bsz.json 2>/dev/null
  echo "bsz $bsz $(python -c "import json;d=json.loads(open('$out/b_$bsz.json').read().strip().split(chr(10))[-1]);print(d['value'])")"
done
