mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or backbone_fp32 or headline_config or full_batch_properties or refiner_loop_low or unsupported_crop or third_crop" > gpurun_out/r04b/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04b/tests.log; tail -4 gpurun_out/r04b/tests.log
B="python bench.py --no-cpu-baseline --no-other-dtypes --steps 10 --warmup 3"
for m in 0 0x7ffff 0x1ff 0x7fe00 0 0x7ffff; do
  COSY_TUNE_LIB=1 COSY_SE_FUSE_MASK=$m $B --layers > gpurun_out/r04b/bench_$m.json 2> gpurun_out/r04b/layers_$m.txt
  python -c "import json,sys; j=json.loads(open('gpurun_out/r04b/bench_$m.json').read().strip().split('\n')[-1]); print('$m', j['value'], j['roofline']['backbone_ms_per_forward'])"
done
