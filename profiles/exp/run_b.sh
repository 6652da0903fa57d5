export COSY_TUNE_LIB=1
run() { # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu-baseline --layers > gpurun_out/rb_$tag.json 2> gpurun_out/rb_$tag.txt
  echo "== $tag: $(python -c "import json;print(json.load(open('gpurun_out/rb_$tag.json'))['value'])")"
  grep -E "^ *([2-9]|1[0-7]) (mbconv|pw_gemm)" gpurun_out/rb_$tag.txt | awk '{for(i=1;i<=NF;i++) if($i=="us/fwd") printf "%s:%s ", $1, $(i-1)} END {print ""}'
}
run plain COSY_WAVE_DBG=0
run nt COSY_WAVE_DBG=16
