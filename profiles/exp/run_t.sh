export COSY_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29515 bench.py --gpus 4 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --config 2 2>/dev/null | grep "^{" | cut -c1-230
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29516 bench.py --gpus 4 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --config 3 2>/dev/null | grep "^{" | cut -c1-230
