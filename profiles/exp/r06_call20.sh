#!/bin/bash
# round 6, call 20: chunked inputs for the GEMM consumers too -- block 18's expand conv (272-byte NHWC pixels) and the head conv (768-byte pixels; needs the chunked epilogue in the split-K tiles)
out=gpurun_out/r06u; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or headline or schedule or backbone or pose_predictor" > $out/tests.txt 2>&1; echo "tests rc $?"; grep -E "passed|failed|FAILED|Error" $out/tests.txt | tail -6
for cfg in "1 1" "0 0" "1 0" "1 1" "0 0"; do
  set -- $cfg
  COSY_TUNE_LIB=1 COSY_X_CHUNKED_EXPAND=$1 COSY_X_CHUNKED_HEAD=$2 timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > $out/b_$1$2.json 2> $out/layers_$1$2.txt
  echo "expand $1 head $2: $(python -c "import json;d=json.loads(open('$out/b_$1$2.json').read().strip().split(chr(10))[-1]);print(d['value'], d['roofline']['backbone_ms_per_forward'])") b18: $(grep -E '^ *18 pw_gemm' $out/layers_$1$2.txt | awk '{print $(NF-5)}' | tr '\n' ' ') b25 proj: $(grep -E '^ *25 pw_gemm' $out/layers_$1$2.txt | awk '{print $(NF-5)}') b24: $(grep -E '^ *24 pw_gemm' $out/layers_$1$2.txt | awk '{print $(NF-5)}') head: $(grep -E '^ *26 pw_gemm' $out/layers_$1$2.txt | awk '{print $(NF-5)}')"
done
