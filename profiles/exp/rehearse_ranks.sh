export COSY_DIST_BACKEND=gloo
A="--steps 2 --warmup 1 --no-cpu-baseline --no-other-dtypes --no-profile"
for cfg in "--config 1 --gpus 2" "--config 2 --gpus 2" "--config 3 --gpus 3" "--config 3 --split balanced --gpus 4"; do
  python bench.py $A $cfg 2>/tmp/err.txt | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg ->', j['n_gpus'], j['value'], j['config']['candidates_per_rank'], j['config']['process_group'])" || tail -5 /tmp/err.txt
done
python bench_train.py --gpus 2 --steps 2 --warmup 1 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('train 2 ranks', j['value'], j['config']['process_group'])"
