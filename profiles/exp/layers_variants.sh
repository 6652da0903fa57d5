mkdir -p gpurun_out/r04h
B="python bench.py --no-cpu-baseline --no-other-dtypes --steps 8 --warmup 3"
$B --layers --crop 240x320 > gpurun_out/r04h/bench_240.json 2> gpurun_out/r04h/layers_240.txt
$B --layers --dtype fp32 > gpurun_out/r04h/bench_fp32.json 2> gpurun_out/r04h/layers_fp32.txt
$B --layers --dtype fp32 --crop 240x320 > gpurun_out/r04h/bench_fp32_240.json 2> gpurun_out/r04h/layers_fp32_240.txt
$B --layers > gpurun_out/r04h/bench_256.json 2> gpurun_out/r04h/layers_256.txt
python bench_train.py --kernels > gpurun_out/r04h/train.json 2> gpurun_out/r04h/train_kernels.txt
for f in 240 fp32 fp32_240 256; do python -c "import json; j=json.loads(open('gpurun_out/r04h/bench_$f.json').read().strip().split('\n')[-1]); print('$f', j['value'], j['roofline']['backbone_ms_per_forward'])"; done
python -c "import json; print(json.load(open('gpurun_out/r04h/train.json'))['value'])"
