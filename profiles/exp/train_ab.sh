# training step: upload order / batched weight packing A/B + the training parity tests
mkdir -p gpurun_out/tr1
python -m pytest tests -m gpu -x -q -k "train or packed or adam" 2>&1 | tail -5 > gpurun_out/tr1/tests.txt
python bench_train.py --steps 10 --warmup 3 > gpurun_out/tr1/prefetch.json 2> gpurun_out/tr1/prefetch.err
python bench_train.py --steps 10 --warmup 3 --upload in_step > gpurun_out/tr1/in_step.json 2> gpurun_out/tr1/in_step.err
python bench_train.py --steps 3 --warmup 2 --kernels > gpurun_out/tr1/kern.json 2> gpurun_out/tr1/kern.txt
