python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --renderer hip | cut -c1-330
export TMPDIR=/tmp
rocprofv3 -M --kernel-trace --stats -f csv -d gpurun_out/px -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --renderer hip > /dev/null 2>&1
grep -E "crop_pack|raster|render" gpurun_out/px/t_kernel_stats.csv | cut -c1-160
