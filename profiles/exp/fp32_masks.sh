mkdir -p gpurun_out/r04n
A="--no-cpu-baseline --no-other-dtypes --no-profile --steps 6 --warmup 2 --streams 1 --dtype fp32"
for m in 0x13c 0x138 0x0; do
  echo "tile mask $m 256:"; COSY_TUNE_LIB=1 COSY_TILE_MASK=$m python bench.py $A | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(j['value'])"
  echo "tile mask $m 240x320:"; COSY_TUNE_LIB=1 COSY_TILE_MASK=$m python bench.py $A --crop 240x320 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(j['value'])"
done
