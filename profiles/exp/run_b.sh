python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile | cut -c1-150
export TMPDIR=/tmp
rocprofv3 -M --kernel-trace --stats -f csv -d gpurun_out/px -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
grep -E "crop_" gpurun_out/px/t_kernel_stats.csv | cut -c1-120
