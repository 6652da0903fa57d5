mkdir -p gpurun_out/tr4
python profiles/exp/train_gaps.py --steps 4 --warmup 2 > gpurun_out/tr4/gaps_prefetch.txt 2> gpurun_out/tr4/gaps_prefetch.err
python profiles/exp/train_gaps.py --steps 4 --warmup 2 --upload in_step > gpurun_out/tr4/gaps_in_step.txt 2> gpurun_out/tr4/gaps_in_step.err
