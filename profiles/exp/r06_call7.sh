#!/bin/bash
# round 6, call 7: the LDS-tiled crop + pack kernel: bit-identity / oracle tests, then the launch alone against round 5's per-pixel kernel (tune build, same tables)
out=gpurun_out/r06g; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crop or roi_align or render_crop" > $out/tests.txt 2>&1; echo "tests rc $?"; tail -5 $out/tests.txt
for sz in "256 256" "240 320"; do
  for t in 1 0 1 0; do COSY_TUNE_LIB=1 COSY_CROP_TILED=$t timeout 300 python profiles/exp/crop_bench.py $sz; done
done 2>&1 | tee $out/crop_bench.txt
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-dtypes > $out/bench.json 2> $out/bench.err; echo "bench rc $?"
python -c "
import json
d=json.loads(open('gpurun_out/r06g/bench.json').read().strip().split('\n')[-1]); print('value', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --steps 8 --warmup 3 --dtype bf16 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > $out/bench_bf16.json 2> $out/layers_bf16.txt; echo "bench bf16 rc $?"; tail -48 $out/layers_bf16.txt | cut -c1-120
