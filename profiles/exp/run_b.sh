export COSY_TUNE_LIB=1
python profiles/exp/det.py "COSY_WAVE_MASK=0x3fffc" 2>&1 | grep -v amdgpu | cut -c1-260
python profiles/ab_check.py "COSY_ROWS_MASK=0 COSY_WAVE_MASK=0" "COSY_WAVE_MASK=0x3fffc" 2>&1 | grep -v amdgpu.ids | grep "vs fp32"
run() { # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --layers > gpurun_out/rb_$tag.json 2> gpurun_out/rb_$tag.txt
  echo "== $tag: $(python -c "import json;print(json.load(open('gpurun_out/rb_$tag.json'))['value'])")"
  grep -E "^ *([2-9]|1[0-8]) (mbconv)" gpurun_out/rb_$tag.txt | cut -c1-120
}
run wave COSY_WAVE_MASK=0x3fffc
unset COSY_TUNE_LIB
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
