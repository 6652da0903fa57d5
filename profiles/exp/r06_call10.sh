#!/bin/bash
# round 6, call 10: inputs of the fp32-FMA wave fronts (blocks 2 / 5 / 8) chunked with the pixels of a row permuted for a lane's run: parity tests, same-call A/B
out=gpurun_out/r06k; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or headline or refiner_loop_low or full_batch or config2 or config3 or backbone or schedule or pose_predictor" > $out/tests.txt 2>&1; echo "tests rc $?"; grep -E "passed|failed|FAILED|Error" $out/tests.txt | tail -8
for c in 1 0 1 0; do
  COSY_TUNE_LIB=1 COSY_X_PERM=$c timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --no-profile > $out/bench_perm$c.json 2> /dev/null
  echo "perm $c $(python -c "import json;d=json.loads(open('$out/bench_perm$c.json').read().strip().split(chr(10))[-1]);print(d['value'])")"
done
for c in 1 0; do
  COSY_TUNE_LIB=1 COSY_X_PERM=$c timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_perm$c.txt
done
python - <<'PY'
import re
def rows(f):
    out=[]
    for ln in open(f):
        m=re.match(r'\s*(\d+) (\S.*?)\s+n=\s*\d+\s+([\d.]+) us/fwd', ln)
        if m: out.append((int(m.group(1)), m.group(2).strip(), float(m.group(3))))
    return out
a=rows('gpurun_out/r06k/layers_perm1.txt'); b=rows('gpurun_out/r06k/layers_perm0.txt')
print('backbone us: perm', round(sum(r[2] for r in a),1), 'no perm', round(sum(r[2] for r in b),1))
for x,y in zip(a,b):
    if abs(x[2]-y[2])>1.0: print(f'{x[0]:3d} {x[1][:60]:60s} {y[2]:7.1f} -> {x[2]:7.1f}')
PY
