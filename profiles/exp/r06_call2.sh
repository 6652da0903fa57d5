#!/bin/bash
# round 6, call 2: the fused (hand-ordered) interior row of the matrix-pipe wave kernels, dev build vs the shipped library: bit-level A/B of D / gates / poses of
# blocks 9-17, then the per-launch table of both
out=gpurun_out/r06b; mkdir -p $out
DEV=$PWD/cosypose_amd/lib/libcosyhip_dev.so
timeout 300 python profiles/exp/ab_bits.py --out $out/base.npz > $out/ab_base.txt 2>&1; echo "base rc $?"; tail -2 $out/ab_base.txt
COSY_TUNE_LIB=$DEV timeout 300 python profiles/exp/ab_bits.py --out $out/dev.npz > $out/ab_dev.txt 2>&1; echo "dev rc $?"; tail -2 $out/ab_dev.txt
python profiles/exp/ab_bits.py --compare $out/base.npz $out/dev.npz | tail -25
for v in base dev base dev; do
  if [ $v = dev ]; then export COSY_TUNE_LIB=$DEV; else unset COSY_TUNE_LIB; fi
  timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --layers > $out/bench_$v.json 2> $out/layers_$v.txt; echo "bench $v rc $?"
  python - <<PY
import json,re
d=json.loads(open('$out/bench_$v.json').read().strip().split('\n')[-1])
print('$v', 'value', d['value'], 'backbone ms', d['roofline']['backbone_ms_per_forward'])
for l in open('$out/layers_$v.txt'):
    if re.match(r'\s*(9|13|14) mbconv_wave', l): print('   ', l.strip()[:110])
PY
done
