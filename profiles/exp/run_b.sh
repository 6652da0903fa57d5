export COSY_DIST_BACKEND=gloo
for c in "--config 1" "--config 2" "--config 3" "--config 3 --split balanced"; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --no-profile $c > gpurun_out/b2.json 2> gpurun_out/b2.err
  echo "rc=$? $c"; cut -c1-200 gpurun_out/b2.json; python -c "import json;d=json.load(open('gpurun_out/b2.json'));print(d['config']['candidates_per_rank'], d['config']['all_gather_us'], d['scaling'])"; grep -i "error\|Traceback" gpurun_out/b2.err | head -3
done
