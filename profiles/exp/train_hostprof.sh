# host-side profile of the training step (which Python / ctypes calls the ~1100 launches per step cost)
mkdir -p gpurun_out/tr3
python -m cProfile -o gpurun_out/tr3/prof.out bench_train.py --steps 10 --warmup 3 > gpurun_out/tr3/bench.json 2> gpurun_out/tr3/bench.err
python - <<'P' > gpurun_out/tr3/top.txt
import pstats
p = pstats.Stats('gpurun_out/tr3/prof.out')
p.sort_stats('tottime').print_stats(45)
p.sort_stats('cumulative').print_stats(60)
P
