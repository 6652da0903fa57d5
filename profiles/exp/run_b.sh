export COSY_TUNE_LIB=1
run() { # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu-baseline --layers > gpurun_out/rb_$tag.json 2> gpurun_out/rb_$tag.txt
  echo "== $tag: $(python -c "import json;print(json.load(open('gpurun_out/rb_$tag.json'))['value'])") $(grep -E "^ *(19|2[0-5]) (mbconv_small|pw_gemm)" gpurun_out/rb_$tag.txt | awk '{for(i=1;i<=NF;i++) if($i=="us/fwd") printf "%s:%s ", $1, $(i-1)}')"
}
run old COSY_SMALL_ROWMAP=0
run new COSY_SMALL_ROWMAP=1
COSY_TUNE_LIB= python -m pytest tests -m gpu -x -q 2>&1 | tail -3
