python bench.py --steps 4 --warmup 1 > gpurun_out/b_c1.json 2> gpurun_out/b_c1.err; tail -c 1500 gpurun_out/b_c1.json; tail -3 gpurun_out/b_c1.err
python bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_c2.json 2> gpurun_out/b_c2.err; cut -c1-900 gpurun_out/b_c2.json; tail -3 gpurun_out/b_c2.err
python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_c3.json 2> gpurun_out/b_c3.err; cut -c1-900 gpurun_out/b_c3.json; tail -3 gpurun_out/b_c3.err
python bench.py --config 3 --split balanced --steps 2 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/b_c3b.json 2> gpurun_out/b_c3b.err; cut -c1-400 gpurun_out/b_c3b.json; tail -3 gpurun_out/b_c3b.err
