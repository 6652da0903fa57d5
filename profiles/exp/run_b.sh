export COSY_TUNE_LIB=1
run() { # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --layers > gpurun_out/rb_$tag.json 2> gpurun_out/rb_$tag.txt
  echo "== $tag: $(python -c "import json;print(json.load(open('gpurun_out/rb_$tag.json'))['value'])")"
  grep -E "^ *(1[89]|2[0-6]) pw_" gpurun_out/rb_$tag.txt | awk '{for(i=1;i<=NF;i++) if($i=="us/fwd") printf "%s:%s ", $1, $(i-1)} END {print ""}'
}
run pf2 COSY_PW_WAVE=1 COSY_PW_WAVE_PF=2
run pf3 COSY_PW_WAVE=1 COSY_PW_WAVE_PF=3
run pf4 COSY_PW_WAVE=1 COSY_PW_WAVE_PF=4
