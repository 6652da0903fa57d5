mkdir -p gpurun_out/r04j
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or headline_config or full_batch_properties or backbone_fp32 or training_step_vs" > gpurun_out/r04j/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04j/tests.log; tail -3 gpurun_out/r04j/tests.log
python bench.py --no-cpu-baseline --layers > gpurun_out/r04j/bench.json 2> gpurun_out/r04j/layers.txt
python -c "import json; j=json.loads(open('gpurun_out/r04j/bench.json').read().strip().split('\n')[-1]); print(j['value'], j['config']['single_stream'], j['roofline']['backbone_ms_per_forward'], {k:v['value'] for k,v in j['other_dtypes'].items()})"
grep -E "^ +(3|13|14) pw_gemm" gpurun_out/r04j/layers.txt | cut -c1-110
python bench.py --no-cpu-baseline --no-other-dtypes --no-profile --crop 240x320 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('240x320', j['value'])"
