O=gpurun_out/cm1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or third_crop or native or 240" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-other-dtypes --steps 8 --warmup 3 --crop 240x320"
for m in 0 1 0 1; do
  env COSY_TUNE_LIB=1 COSY_COLMAJOR=$m $B --streams 1 --layers > $O/b_$m.json 2> $O/layers_$m.txt
  python -c "import json; j=json.loads(open('$O/b_$m.json').read().strip().split('\n')[-1]); print('colmajor $m one stream', j['value'], j['roofline'].get('backbone_ms_per_forward'))"
done
for m in 0 1; do
  env COSY_TUNE_LIB=1 COSY_COLMAJOR=$m $B > $O/b2_$m.json 2>/dev/null
  python -c "import json; j=json.loads(open('$O/b2_$m.json').read().strip().split('\n')[-1]); print('colmajor $m two streams', j['value'])"
done
paste <(grep -E "^ *[0-9]+ (mbconv_wave|pixels)" $O/layers_0.txt | awk '{print $1, $(NF-5)}') <(grep -E "^ *[0-9]+ (mbconv_wave)" $O/layers_1.txt | awk '{print $(NF-5)}')
grep pixels $O/layers_1.txt | head -2
