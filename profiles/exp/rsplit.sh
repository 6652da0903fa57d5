A="--gpus 1 --steps 16 --warmup 4 --no-cpu-baseline --no-other-dtypes --no-profile"
for r in 0 1 2 4 0; do
  if [ $r = 0 ]; then unset COSY_WAVE_RSPLIT; else export COSY_WAVE_RSPLIT=$r; fi
  COSY_TUNE_LIB=1 python bench.py $A | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('rsplit $r streams2', j['value'])"
  COSY_TUNE_LIB=1 python bench.py $A --streams 1 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('rsplit $r streams1', j['value'])"
done
