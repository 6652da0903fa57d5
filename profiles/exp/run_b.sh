export COSY_TUNE_LIB=1
run() { # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu-baseline --layers > gpurun_out/rb_$tag.json 2> gpurun_out/rb_$tag.txt
  echo "== $tag: $(python -c "import json;print(json.load(open('gpurun_out/rb_$tag.json'))['value'])") $(grep -E "^ *[0-9]+ dwconv" gpurun_out/rb_$tag.txt | awk '{for(i=1;i<=NF;i++) if($i=="us/fwd") printf "%s:%s ", $1, $(i-1)}')"
}
run th8 COSY_DW_TH=8
run th16 COSY_DW_TH=16 COSY_DW_LDS_KB=56
run th16b COSY_DW_TH=16 COSY_DW_LDS_KB=40
run th8_56 COSY_DW_TH=8 COSY_DW_LDS_KB=56
