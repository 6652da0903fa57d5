for i in 1 2 3; do
for m in 1 0; do
COSY_BENCH_GC_LATE=$m python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('gc_late=$m steps20', j['value'])"
COSY_BENCH_GC_LATE=$m python bench.py --no-cpu-baseline --no-other-dtypes --no-profile | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('gc_late=$m default', j['value'])"
done
done
