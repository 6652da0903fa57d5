# 240x320 crops: 8-wave / split-K / 32-row-per-wave GEMM tiles under the row-side gate (tune build)
O=gpurun_out/rg1; mkdir -p $O
B="python bench.py --crop 240x320 --no-cpu-baseline --no-other-dtypes --steps 8 --warmup 3 --streams 1"
run() { # name env...
  n=$1; shift
  env COSY_TUNE_LIB=1 "$@" $B --layers > $O/bench_$n.json 2> $O/layers_$n.txt
  python -c "import json,sys; j=json.loads(open('$O/bench_$n.json').read().strip().split('\n')[-1]); print('$n', j['value'], j['roofline'].get('backbone_ms_per_forward'))"
  grep -E "^ *(8|9|13|14|18|19|24|25|26) pw_gemm" $O/layers_$n.txt | cut -c1-100
  env COSY_TUNE_LIB=1 "$@" timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation and 240x320 and fp16" 2>&1 | tail -1
}
run base COSY_X=0
run pw8rg COSY_PW8_RG=1
run pw16rg COSY_PW16_RG=1
run mi2 COSY_PW_MI=2 COSY_PW_MI_RG=1
run base2 COSY_X=0
