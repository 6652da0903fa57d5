for i in 1 2 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('default', j['value'], j['config']['single_stream']['value'], j['other_dtypes']['bf16']['value'])"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-other-dtypes --no-profile | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('streams1', j['value'])"
done
