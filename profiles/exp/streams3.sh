A="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for cfg in "--streams 2" "--streams 3" "--streams 2 --bsz-objects 64" "--streams 3 --bsz-objects 64" "--streams 2"; do
python bench.py $A $cfg | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg', j['value'], j['config']['bsz_objects'])"
done
