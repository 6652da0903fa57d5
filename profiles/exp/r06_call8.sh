#!/bin/bash
# round 6, call 8: (a) tiled crop kernel with box-derived window extents (its loads no longer wait for a tap entry): tests + launch time;
# (b) wave kernels with 1 / 2 / 4 jobs (waves) per workgroup: a workgroup's slots are recycled when its slowest job ends
out=gpurun_out/r06h; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crop or roi_align or render_crop" > $out/tests.txt 2>&1; echo "tests rc $?"; tail -3 $out/tests.txt
for sz in "256 256" "240 320"; do
  for t in 1 0 1; do COSY_TUNE_LIB=1 COSY_CROP_TILED=$t timeout 300 python profiles/exp/crop_bench.py $sz 2>/dev/null; done
done | tee $out/crop_bench.txt
for wpb in 4 1 2 4 1; do
  COSY_TUNE_LIB=1 COSY_WAVE_WPB=$wpb timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --no-profile > $out/bench_wpb$wpb.json 2> /dev/null
  echo "wpb $wpb $(python -c "import json;d=json.loads(open('$out/bench_wpb$wpb.json').read().strip().split(chr(10))[-1]);print(d['value'])")"
done
for wpb in 4 1; do
  COSY_TUNE_LIB=1 COSY_WAVE_WPB=$wpb timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > /dev/null 2> $out/layers_wpb$wpb.txt
  echo "== wpb $wpb"; grep "mbconv_wave_kernel" $out/layers_wpb$wpb.txt | tail -14 | cut -c1-110
done
