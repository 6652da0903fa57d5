run() { # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --layers > gpurun_out/rb_$tag.json 2> gpurun_out/rb_$tag.txt
  echo "== $tag: $(python -c "import json;print(json.load(open('gpurun_out/rb_$tag.json'))['value'])")"
  grep -E "^ *[0-9]+ se_kernel" gpurun_out/rb_$tag.txt | awk '{for(i=1;i<=NF;i++) if($i=="us/fwd") printf "%s:%s ", $1, $(i-1)} END {print ""}'
}
run pre X=1
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
