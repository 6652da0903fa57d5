# training step: upload order A/B + the training parity tests
mkdir -p gpurun_out/tr2
python -m pytest tests -m gpu -x -q -k "train or packed or adam or prefetch" 2>&1 | tail -5 > gpurun_out/tr2/tests.txt
python bench_train.py --steps 20 --warmup 3 > gpurun_out/tr2/prefetch.json 2> gpurun_out/tr2/prefetch.err
python bench_train.py --steps 20 --warmup 3 --upload in_step > gpurun_out/tr2/in_step.json 2> gpurun_out/tr2/in_step.err
python bench_train.py --steps 20 --warmup 3 > gpurun_out/tr2/prefetch_b.json 2> gpurun_out/tr2/prefetch_b.err
python bench_train.py --steps 3 --warmup 2 --kernels > gpurun_out/tr2/kern.json 2> gpurun_out/tr2/kern.txt
