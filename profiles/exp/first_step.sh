P='import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(sys.argv[1], j["value"], j["config"]["step_ms"]["each"][:6])'
B="python bench.py --gpus 1 --steps 12 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
$B | python -c "$P" streams2
$B --streams 1 | python -c "$P" streams1
COSY_BENCH_NOSYNC=1 $B | python -c "$P" streams2_nosync
COSY_BENCH_NOSYNC=1 $B --streams 1 | python -c "$P" streams1_nosync
