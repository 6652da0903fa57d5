export COSY_TUNE_LIB=1
run() { # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --layers > gpurun_out/rb_$tag.json 2> gpurun_out/rb_$tag.txt
  echo "== $tag: $(python -c "import json;print(json.load(open('gpurun_out/rb_$tag.json'))['value'])")"
  grep -E "^ *[0-9]+ pw_gemm" gpurun_out/rb_$tag.txt | awk '{for(i=1;i<=NF;i++) if($i=="us/fwd") printf "%s:%s ", $1, $(i-1)} END {print ""}'
}
run ns2_2 COSY_PW_NS2_MAXKB=2
run ns2_6 COSY_PW_NS2_MAXKB=6
run ns2_99 COSY_PW_NS2_MAXKB=99
