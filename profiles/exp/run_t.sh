python -m pytest tests -m gpu -x -q -k "train_gemm" 2>&1 | tail -8
