#!/bin/bash
# round 6, call 3: s_memtime timelines of the fused-row dev kernel vs the base kernel at 4 / 2 / 1 waves per SIMD (LDS pad), and the layer table of the grouped-loop dev build
out=gpurun_out/r06c; mkdir -p $out
L=$PWD/cosypose_amd/lib
for pad in 0 45000 90000; do
  for v in base dev; do
    for cm in 816 576; do
      COSY_WAVE_LDS_PAD=$pad COSY_TUNE_LIB=$L/libcosyhip_${v}st.so timeout 120 python profiles/exp/wave_timeline.py --cmid $cm > $out/tl_${v}_${cm}_$pad.txt 2>&1
      echo "== $v cmid $cm pad $pad"; grep -E "rows per job|prologue|row start|last row|launch:" $out/tl_${v}_${cm}_$pad.txt | cut -c1-260
    done
  done
done
COSY_TUNE_LIB=$L/libcosyhip_dev.so timeout 300 python profiles/exp/ab_bits.py --out $out/dev.npz > $out/ab_dev.txt 2>&1; timeout 300 python profiles/exp/ab_bits.py --out $out/base.npz > $out/ab_base.txt 2>&1
python profiles/exp/ab_bits.py --compare $out/base.npz $out/dev.npz | tail -5
for v in base dev; do
  if [ $v = dev ]; then export COSY_TUNE_LIB=$L/libcosyhip_dev.so; else unset COSY_TUNE_LIB; fi
  timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --layers > $out/bench_$v.json 2> $out/layers_$v.txt; echo "bench $v rc $?"
  grep -E "^\s*(9|13|14) mbconv_wave" $out/layers_$v.txt | cut -c1-100
done
