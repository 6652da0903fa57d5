A="--gpus 1 --steps 16 --warmup 4 --no-cpu-baseline --no-other-dtypes --no-profile"
for c in 0 64 32 0; do
  COSY_TUNE_LIB=1 COSY_EARLY_CHUNK=$c python bench.py $A | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('early chunk $c streams2', j['value'])"
  COSY_TUNE_LIB=1 COSY_EARLY_CHUNK=$c python bench.py $A --streams 1 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('early chunk $c streams1', j['value'])"
done
COSY_TUNE_LIB=1 COSY_EARLY_CHUNK=128 python bench.py $A --streams 1 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('early chunk 128 streams1', j['value'])"
