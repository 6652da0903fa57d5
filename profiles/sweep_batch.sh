#!/bin/bash
# Batch sweep: per-launch algorithmic TFLOP/s of the backbone at 256..2048 crops per forward.
#   profiles/sweep_batch.sh <tag>     -> gpurun_out/sweep_<tag>_B<n>.{json,txt}
TAG=${1:-run}
for B in 256 512 1024 2048; do
  python bench.py --steps 3 --warmup 1 --detections $B --bsz-objects $B --no-cpu-baseline --layers \
      > gpurun_out/sweep_${TAG}_B$B.json 2> gpurun_out/sweep_${TAG}_B$B.txt
done
