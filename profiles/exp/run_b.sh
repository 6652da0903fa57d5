export COSY_TUNE_LIB=1
run() { # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu-baseline --layers > gpurun_out/rb_$tag.json 2> gpurun_out/rb_$tag.txt
  echo "== $tag: $(grep -E "^ *(19|24|25) mbconv_small" gpurun_out/rb_$tag.txt | awk '{for(i=1;i<=NF;i++) if($i=="us/fwd") printf "%s:%s ", $1, $(i-1)}')"
}
run base COSY_SMALL_DBG=0
run nodw COSY_SMALL_DBG=1
run noexp COSY_SMALL_DBG=2
run nodma COSY_SMALL_DBG=4
run neither COSY_SMALL_DBG=3
run cpw8 COSY_SMALL_CPW=8
run cpw29 COSY_SMALL_CPW=29
