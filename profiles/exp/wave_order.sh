O=gpurun_out/wo1; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-other-dtypes --steps 8 --warmup 3 --streams 1"
for crop in 256x256 240x320; do
for d in 0 32 0 32; do
  env COSY_TUNE_LIB=1 COSY_WAVE_DBG=$d $B --crop $crop --layers > $O/b.json 2> $O/l_${crop}_$d.txt
  echo "crop $crop dbg $d: $(python -c "import json; j=json.loads(open('$O/b.json').read().strip().split('\n')[-1]); print(j['value'])") $(grep -E '^ *(2|3|4|5|6|7|8) mbconv_wave' $O/l_${crop}_$d.txt | awk '{printf "%s:%s ", $1, $(NF-5)}')"
done
done
