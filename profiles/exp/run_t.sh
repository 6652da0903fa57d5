python -m pytest tests -m gpu -x -q -k "train or wgrad or ddp or bn_ or dw_" 2>&1 | tail -8
python bench_train.py --kernels 2>&1 | grep -E "time by family|^\{|gemm " | cut -c1-400
