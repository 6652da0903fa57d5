O=gpurun_out/dx1; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-other-dtypes --steps 8 --warmup 3 --streams 1"
for crop in 256x256 240x320; do
for d in 0 1 0 1; do
  env COSY_TUNE_LIB=1 COSY_DW_BY_SAMPLE=$d $B --crop $crop --layers > $O/b.json 2> $O/l_${crop}_$d.txt
  echo "crop $crop by_sample $d: $(python -c "import json; j=json.loads(open('$O/b.json').read().strip().split('\n')[-1]); print(j['value'])") $(grep -E '^ *(0|1|18) dwconv' $O/l_${crop}_$d.txt | awk '{printf "%s:%s ", $1, $(NF-5)}')"
done
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or headline_config or fp32" 2>&1 | tail -2
