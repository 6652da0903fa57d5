# training step: upload order A/B
mkdir -p gpurun_out/tr5
python bench_train.py --steps 20 --warmup 3 > gpurun_out/tr5/prefetch.json 2> gpurun_out/tr5/prefetch.err
python bench_train.py --steps 20 --warmup 3 --upload in_step > gpurun_out/tr5/in_step.json 2> gpurun_out/tr5/in_step.err
python bench_train.py --steps 20 --warmup 3 > gpurun_out/tr5/prefetch_b.json 2> gpurun_out/tr5/prefetch_b.err
python bench_train.py > gpurun_out/tr5/default.json 2> gpurun_out/tr5/default.err
