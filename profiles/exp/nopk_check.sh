python profiles/exp/race_hunt6.py 2>&1 | grep -v amdgpu | tail -1
SYN=0 RS=240x320 NDET=96 CH=32 python profiles/exp/race_hunt.py steady 15 2>&1 | grep -v amdgpu | tail -1
SYN=0 RS=256x256 NDET=256 CH=128 python profiles/exp/race_hunt.py steady 10 2>&1 | grep -v amdgpu | tail -1
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-profile"
for i in 1 2; do
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/old_ship.so $B | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('old', j['value'])"
$B | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new', j['value'])"
done
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/old_ship.so python bench_train.py | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('old train', j['value'])"
python bench_train.py | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new train', j['value'])"
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/old_ship.so python bench.py --renderer hip --no-cpu-baseline --no-other-dtypes --no-profile | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('old renderer-in-loop', j['value'])"
python bench.py --renderer hip --no-cpu-baseline --no-other-dtypes --no-profile | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('new renderer-in-loop', j['value'])"
