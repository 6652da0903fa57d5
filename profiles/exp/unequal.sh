A="--gpus 1 --steps 16 --warmup 4 --no-cpu-baseline --no-other-dtypes --no-profile --streams 2"
for b in 128 136 144 160 176 192 128; do
  python bench.py $A --bsz-objects $b | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('chunks $b +', 256-$b, j['value'])"
done
