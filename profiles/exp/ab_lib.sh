#!/bin/bash
# A/B of two builds of the library inside one gpurun call: ab_lib.sh <outdir> <tag>=<path.so|-> ...   ("-" = the shipping library)
out=$1; shift; mkdir -p $out
for spec in "$@"; do
  tag=${spec%%=*}; path=${spec#*=}
  if [ "$path" = "-" ]; then unset COSY_TUNE_LIB; else export COSY_TUNE_LIB=$path; fi
  timeout 200 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes > $out/$tag.json 2> $out/$tag.txt
  python - <<PY
import json
try:
    d=json.load(open("$out/$tag.json")); print("$tag", d["value"], d["roofline"]["backbone_ms_per_forward"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"])
except Exception as e: print("$tag FAILED", e)
PY
done
