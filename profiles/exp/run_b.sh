export COSY_TUNE_LIB=1
run() { # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --layers > gpurun_out/rb_$tag.json 2> gpurun_out/rb_$tag.txt
  echo "== $tag: $(python -c "import json;print(json.load(open('gpurun_out/rb_$tag.json'))['value'])")"
  grep -E "^ *([2-9]|1[0-7]) (mbconv)" gpurun_out/rb_$tag.txt | awk '{for(i=1;i<=NF;i++) if($i=="us/fwd") printf "%s:%s ", $1, $(i-1)} END {print ""}'
}
run tail COSY_SE_TAIL=0x3fffc
run notail COSY_SE_TAIL=0
python profiles/exp/det.py 2>&1 | grep -v amdgpu | tail -3 | cut -c1-160
COSY_TUNE_LIB= python -m pytest tests -m gpu -x -q 2>&1 | tail -3
