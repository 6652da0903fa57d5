mkdir -p gpurun_out/r04i
B="python bench.py --no-cpu-baseline --no-other-dtypes --no-profile --steps 12 --warmup 3"
for i in 1 2 3; do
for st in 1 2; do
  $B --streams $st > gpurun_out/r04i/s${st}_$i.json 2>/dev/null
  python -c "import json; j=json.loads(open('gpurun_out/r04i/s${st}_$i.json').read().strip().split('\n')[-1]); print('streams $st', j['value'], j['config']['bsz_objects'])"
done; done
$B --crop 240x320 > gpurun_out/r04i/c240.json 2>/dev/null; python -c "import json; j=json.loads(open('gpurun_out/r04i/c240.json').read().strip().split('\n')[-1]); print('240x320', j['value'])"
$B --crop 240x320 --streams 2 > gpurun_out/r04i/c240s2.json 2>/dev/null; python -c "import json; j=json.loads(open('gpurun_out/r04i/c240s2.json').read().strip().split('\n')[-1]); print('240x320 s2', j['value'])"
