for B in 768 1024 1280 1536; do
python bench.py --steps 4 --warmup 2 --detections $B --bsz-objects $B --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print($B, d['value'], d['ms_per_step'])"
done
python bench.py --steps 4 --warmup 2 --detections 1024 --bsz-objects 1024 --no-cpu-baseline --layers 2>&1 >/dev/null | grep -E "host enqueue|^mbconv_wave_kernel  |^pw_gemm_dma_kernel  " | head -5
