#!/bin/bash
# round 6, call 11: chunked inputs for the one-pixel-per-lane fp32-FMA fronts too (240x320 crops: the 15x20 maps of blocks 9-17): parity at six crop sizes, A/B at 240x320
out=gpurun_out/r06l; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or schedule or pose_predictor or coarse_refine" > $out/tests.txt 2>&1; echo "tests rc $?"; grep -E "passed|failed|FAILED|Error" $out/tests.txt | tail -8
for c in 1 0 1 0; do
  COSY_TUNE_LIB=1 COSY_X_CHUNKED=$c timeout 600 python bench.py --crop 240x320 --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --no-profile > $out/bench_240_chk$c.json 2> /dev/null
  echo "240x320 chunked $c $(python -c "import json;d=json.loads(open('$out/bench_240_chk$c.json').read().strip().split(chr(10))[-1]);print(d['value'])")"
done
for c in 1 0; do
  COSY_TUNE_LIB=1 COSY_X_CHUNKED=$c timeout 600 python bench.py --crop 240x320 --steps 6 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_240_chk$c.txt
done
python - <<'PY'
import re
def rows(f):
    out=[]
    for ln in open(f):
        m=re.match(r'\s*(\d+) (\S.*?)\s+n=\s*\d+\s+([\d.]+) us/fwd', ln)
        if m: out.append((int(m.group(1)), m.group(2).strip(), float(m.group(3))))
    return out
a=rows('gpurun_out/r06l/layers_240_chk1.txt'); b=rows('gpurun_out/r06l/layers_240_chk0.txt')
print('backbone us: chunked', round(sum(r[2] for r in a),1), 'nhwc', round(sum(r[2] for r in b),1))
for x,y in zip(a,b):
    if abs(x[2]-y[2])>1.0: print(f'{x[0]:3d} {x[1][:60]:60s} {y[2]:7.1f} -> {x[2]:7.1f}')
PY
