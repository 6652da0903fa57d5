O=gpurun_out/rg2; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or third_crop or train or packed or prefetch" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-other-dtypes --steps 10 --warmup 3"
for i in 1 2; do
$B --crop 240x320 > $O/b240_$i.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/b240_$i.json').read().strip().split('\n')[-1]); print('240x320', j['value'], j['config'].get('single_stream',{}).get('value'))"
done
$B --crop 240x320 --streams 1 --layers > $O/b240_layers.json 2> $O/layers_240.txt
$B > $O/b256.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/b256.json').read().strip().split('\n')[-1]); print('256', j['value'], j['config'].get('single_stream',{}).get('value'))"
python bench_train.py > $O/train.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/train.json').read().strip().split('\n')[-1]); print('train', j['value'], j['split_ms'])"
python bench_train.py --steps 3 --warmup 2 --kernels > /dev/null 2> $O/train_kernels.txt; grep -E "combine|final" $O/train_kernels.txt | cut -c1-60,150-260
