#!/bin/bash
# round 6, call 27: 1x1-conv GEMM with its DMA sources hoisted out of the k-loop (a per-lane pointer + a constant advance per stage): parity, layers, bench
out=gpurun_out/r06aa; mkdir -p $out
timeout 1700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or headline or schedule or backbone or config2 or config3 or pose_predictor or full_batch" > $out/tests.txt 2>&1; echo "tests rc $?"; grep -E "passed|failed|FAILED|Error" $out/tests.txt | tail -6
timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-dtypes --streams 1 --layers > $out/b.json 2> $out/layers.txt
grep -E "pw_gemm" $out/layers.txt | head -28 | awk '{print $1, $(NF-5)}' | tr '\n' ' '; echo
tail -40 $out/layers.txt | grep -E "^pw_gemm_dma_kernel  |^mbconv_wave_kernel  " | cut -c1-100
python -c "import json;d=json.loads(open('$out/b.json').read().strip().split(chr(10))[-1]);print('one stream', d['value'], d['roofline']['backbone_ms_per_forward'])"
for i in 1 2; do timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-dtypes --no-profile 2>/dev/null | python -c "import json,sys;print('bench', json.loads(sys.stdin.read().strip().split(chr(10))[-1])['value'])"; done
