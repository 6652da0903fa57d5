O=gpurun_out/tc1; mkdir -p $O
for v in "COSY_X=0" "COSY_RED_CAP=512" "COSY_RED_CAP=256" "COSY_RED_CAP=2048" "COSY_WG_CAP=128" "COSY_WG_CAP=64" "COSY_X=0"; do
  env COSY_TUNE_LIB=1 $v python bench_train.py > $O/t.json 2>/dev/null
  python -c "import json; j=json.loads(open('$O/t.json').read().strip().split('\n')[-1]); print('$v', j['value'], j['split_ms'])"
done
