# Same-box A/B against the round-3 tree.  Prepare once: git worktree add .r03tree 4741c64 && (cd .r03tree && python -m cosypose_amd.build)
# (.r03tree is git-ignored and travels to the GPU box with the snapshot).
mkdir -p gpurun_out/r04k
A="--no-cpu-baseline --no-other-dtypes --no-profile --steps 12 --warmup 3"
for i in 1 2 3; do
  (cd .r03tree && python bench.py $A) > gpurun_out/r04k/r03_$i.json 2>/dev/null
  python bench.py $A --streams 1 > gpurun_out/r04k/r04_s1_$i.json 2>/dev/null
  python bench.py $A --streams 2 > gpurun_out/r04k/r04_s2_$i.json 2>/dev/null
  for f in r03 r04_s1 r04_s2; do python -c "import json; j=json.loads(open('gpurun_out/r04k/${f}_$i.json').read().strip().split('\n')[-1]); print('$f', j['value'])"; done
done
(cd .r03tree && python bench.py $A --crop 240x320) | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('r03 240x320', j['value'])"
python bench.py $A --crop 240x320 --streams 1 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('r04 240x320 s1', j['value'])"
(cd .r03tree && python bench_train.py) | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('r03 train', j['value'])"
python bench_train.py | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('r04 train', j['value'])"
