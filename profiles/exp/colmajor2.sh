O=gpurun_out/cm2; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or third_crop or native or 240" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-other-dtypes --steps 8 --warmup 3 --crop 240x320"
$B --streams 1 --layers > $O/b.json 2> $O/layers.txt
python -c "import json; j=json.loads(open('$O/b.json').read().strip().split('\n')[-1]); print('one stream', j['value'], j['roofline'].get('backbone_ms_per_forward'))"
for i in 1 2; do $B > $O/b2.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/b2.json').read().strip().split('\n')[-1]); print('two streams', j['value'])"; done
grep -E "^ *(2|3) " $O/layers.txt | cut -c1-110
python bench.py --no-cpu-baseline --no-other-dtypes --steps 8 --warmup 3 > $O/b256.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/b256.json').read().strip().split('\n')[-1]); print('256x256', j['value'])"
