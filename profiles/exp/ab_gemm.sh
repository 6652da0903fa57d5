mkdir -p gpurun_out/r04l
A="--no-cpu-baseline --no-other-dtypes --steps 10 --warmup 3 --streams 1 --layers"
(cd .r03tree && python bench.py --no-cpu-baseline --no-other-dtypes --steps 10 --warmup 3 --layers) > gpurun_out/r04l/r03.json 2> gpurun_out/r04l/layers_r03.txt
python bench.py $A > gpurun_out/r04l/cur.json 2> gpurun_out/r04l/layers_cur.txt
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/libcosyhip_v1.so python bench.py $A > gpurun_out/r04l/v1.json 2> gpurun_out/r04l/layers_v1.txt
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/libcosyhip_v2.so python bench.py $A > gpurun_out/r04l/v2.json 2> gpurun_out/r04l/layers_v2.txt
for f in r03 cur v1 v2; do python -c "import json; j=json.loads(open('gpurun_out/r04l/$f.json').read().strip().split('\n')[-1]); print('$f', j['value'], j['roofline']['backbone_ms_per_forward'])"; done
