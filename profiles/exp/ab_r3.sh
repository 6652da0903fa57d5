# A/B inside one gpurun call (same box): bench value + selected per-layer times; usage: ab_r3.sh <outdir> "<tag> ENV=.. ENV=.." ...
out=$1; shift; mkdir -p $out
for spec in "$@"; do
  set -- $spec; tag=$1; shift
  env COSY_TUNE_LIB=1 "$@" timeout 200 python bench.py --steps 6 --warmup 2 --layers --no-cpu-baseline --no-other-dtypes > $out/$tag.json 2> $out/$tag.txt
  python - <<PY
import json
try:
    d=json.load(open("$out/$tag.json")); print("$tag", d["value"], d["roofline"]["backbone_ms_per_forward"])
except Exception as e: print("$tag FAILED", e)
PY
done
