O=gpurun_out/wk1; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-other-dtypes --steps 6 --warmup 2 --streams 1"
for crop in 256x256 240x320; do
for d in 0 1 2 3 16; do
  env COSY_TUNE_LIB=1 COSY_WAVE_DBG=$d $B --crop $crop --layers > $O/b.json 2> $O/l_${crop}_$d.txt
  echo "crop $crop dbg $d: $(grep -E '^ *(2|3|5|6|8|9|13|14) mbconv_wave' $O/l_${crop}_$d.txt | awk '{printf "%s:%s ", $1, $(NF-5)}')"
done
done
