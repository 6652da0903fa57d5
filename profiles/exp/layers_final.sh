mkdir -p gpurun_out/r04q
B="python bench.py --no-cpu-baseline --no-other-dtypes --steps 8 --warmup 3 --streams 1 --layers"
$B --crop 240x320 > gpurun_out/r04q/bench_240x320.json 2> gpurun_out/r04q/layers_events_240x320.txt
$B --dtype fp32 > gpurun_out/r04q/bench_fp32.json 2> gpurun_out/r04q/layers_events_fp32.txt
$B --dtype fp32 --crop 240x320 > gpurun_out/r04q/bench_fp32_240x320.json 2> gpurun_out/r04q/layers_events_fp32_240x320.txt
$B --dtype bf16 > gpurun_out/r04q/bench_bf16.json 2> gpurun_out/r04q/layers_events_bf16.txt
for f in 240x320 fp32 fp32_240x320 bf16; do python -c "import json; j=json.loads(open('gpurun_out/r04q/bench_$f.json').read().strip().split('\n')[-1]); print('$f', j['value'], j['roofline']['backbone_ms_per_forward'])"; done
