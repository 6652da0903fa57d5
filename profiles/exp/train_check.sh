mkdir -p gpurun_out/r04g
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training or train_gemm or ddp or c_abi" > gpurun_out/r04g/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04g/tests.log; tail -5 gpurun_out/r04g/tests.log
python bench_train.py --kernels > gpurun_out/r04g/train.json 2> gpurun_out/r04g/train_kernels.txt; cat gpurun_out/r04g/train.json; grep -E "time by family" gpurun_out/r04g/train_kernels.txt
