#!/bin/bash
# round 5, call 26: wave-kernel workgroup order (job quad, sample) instead of (sample, job quad): parameters / weights of what runs together shared in L2
out=gpurun_out/r05ab; mkdir -p $out
L="timeout 300 python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline --no-other-dtypes"
for o in 0 1; do
COSY_TUNE_LIB=1 COSY_WAVE_ORDER=$o $L > $out/l_$o.json 2> $out/l_$o.txt
echo "--- order $o"; grep "mbconv_wave" $out/l_$o.txt | head -16 | awk '{print $1, $(NF-5)}' | tr '\n' ' '; echo
done
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2; do for o in 0 1; do
COSY_TUNE_LIB=1 COSY_WAVE_ORDER=$o $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('order $o', j['value'])"
done; done | tee $out/ab.txt
