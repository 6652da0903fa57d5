# Same-box A/B against the round-3 tree.  Prepare once: git worktree add .r03tree 4741c64 && (cd .r03tree && python -m cosypose_amd.build)
# (.r03tree is git-ignored and travels to the GPU box with the snapshot).
mkdir -p gpurun_out/r04m
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "storage_emulation or headline_config or full_batch_properties" > gpurun_out/r04m/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04m/tests.log; tail -3 gpurun_out/r04m/tests.log
A="--no-cpu-baseline --no-other-dtypes --steps 10 --warmup 3"
(cd .r03tree && python bench.py $A --layers) > gpurun_out/r04m/r03.json 2> gpurun_out/r04m/layers_r03.txt
python bench.py $A --streams 1 --layers > gpurun_out/r04m/cur.json 2> gpurun_out/r04m/layers_cur.txt
for i in 1 2; do
(cd .r03tree && python bench.py $A --no-profile) > gpurun_out/r04m/r03_$i.json 2>/dev/null
python bench.py $A --no-profile --streams 1 > gpurun_out/r04m/s1_$i.json 2>/dev/null
python bench.py $A --no-profile --streams 2 > gpurun_out/r04m/s2_$i.json 2>/dev/null
done
for f in r03 cur r03_1 s1_1 s2_1 r03_2 s1_2 s2_2; do python -c "import json; j=json.loads(open('gpurun_out/r04m/$f.json').read().strip().split('\n')[-1]); print('$f', j['value'], (j['roofline'] or {}).get('backbone_ms_per_forward'))"; done
