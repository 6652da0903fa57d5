run() { # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu-baseline --layers > gpurun_out/rb_$tag.json 2> gpurun_out/rb_$tag.txt
  echo "== $tag: $(python -c "import json;print(json.load(open('gpurun_out/rb_$tag.json'))['value'])")"
  grep -E "^ *([2-9]|1[0-7]) (mbconv)" gpurun_out/rb_$tag.txt | awk '{for(i=1;i<=NF;i++) if($i=="us/fwd") printf "%s:%s ", $1, $(i-1)} END {print ""}'
}
run w3 X=0
python profiles/exp/det.py 2>&1 | grep -v amdgpu | tail -2 | cut -c1-160
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
