mkdir -p gpurun_out/r04d
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or headline_config or full_batch_properties or refiner_loop_low or config2" > gpurun_out/r04d/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04d/tests.log; tail -4 gpurun_out/r04d/tests.log
B="python bench.py --no-cpu-baseline --no-other-dtypes --steps 10 --warmup 3"
for m in 0 1 0 1; do
  COSY_TUNE_LIB=1 COSY_PW16=$m $B --layers > gpurun_out/r04d/bench_$m.json 2> gpurun_out/r04d/layers_$m.txt
  python -c "import json,sys; j=json.loads(open('gpurun_out/r04d/bench_$m.json').read().strip().split('\n')[-1]); print('$m', j['value'], j['roofline']['backbone_ms_per_forward'])"
  grep -E "^ (19|24|25) pw_gemm" gpurun_out/r04d/layers_$m.txt
done
