#!/bin/bash
# round 6, call 21: four-stage LDS-DMA ring for the split-K project tiles of blocks 19-23 (one workgroup per CU: bytes in flight = the ring)
out=gpurun_out/r06x; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation and 256x256 or headline_config" > $out/tests.txt 2>&1; echo "tests rc $?"; grep -E "passed|failed|FAILED|Error" $out/tests.txt | tail -4
for ns in 4 3 4 3; do
  COSY_TUNE_LIB=1 COSY_PW16_NS=$ns timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > $out/b_$ns.json 2> $out/layers_$ns.txt
  echo "ns $ns: $(python -c "import json;d=json.loads(open('$out/b_$ns.json').read().strip().split(chr(10))[-1]);print(d['value'], d['roofline']['backbone_ms_per_forward'])") proj 19-23: $(grep -E '^ *(19|20|21|22|23) pw_gemm' $out/layers_$ns.txt | awk '{print $(NF-5)}' | tr '\n' ' ')"
done
