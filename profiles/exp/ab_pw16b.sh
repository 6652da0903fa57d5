mkdir -p gpurun_out/r04e
B="python bench.py --no-cpu-baseline --no-other-dtypes --steps 8 --warmup 3"
for m in 1 2 1 2; do
  COSY_TUNE_LIB=1 COSY_PW16=$m $B --layers > gpurun_out/r04e/bench_$m.json 2> gpurun_out/r04e/layers_$m.txt
  python -c "import json,sys; j=json.loads(open('gpurun_out/r04e/bench_$m.json').read().strip().split('\n')[-1]); print('$m', j['value'], j['roofline']['backbone_ms_per_forward'])"
  grep -E "^ (19|20|24|25) pw_gemm" gpurun_out/r04e/layers_$m.txt
done
