#!/bin/bash
# Everything the round's profiles/ artefacts are made from, in one gpurun call (run from the repo root on the GPU box):
#   profiles/collect_all.sh <tag>
# writes gpurun_out/all_<tag>/ : bench lines (default, --layers, configs 2/3, fp16, 240x320, on-device renderer, training),
# the rocprofv3 stats + PMC summaries (profiles/collect.sh) and the batch sweep with matrix-core utilisation.
# Copy what is to be judged into profiles/ afterwards (profiles/publish.sh <tag>).
set -u
TAG=${1:-r06}
OUT=gpurun_out/all_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# 2. per-layer HIP-event table
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --streams 1 --layers > $OUT/bench_layers.json 2> $OUT/layers_events.txt
# 3. other configurations on one GPU
python bench.py --config 2 --no-cpu-baseline > $OUT/bench_config2.json 2> /dev/null
python bench.py --config 3 --no-cpu-baseline > $OUT/bench_config3_skewed.json 2> /dev/null
python bench.py --config 3 --split balanced --no-cpu-baseline > $OUT/bench_config3_balanced.json 2> /dev/null
python bench.py --crop 240x320 --no-cpu-baseline > $OUT/bench_240x320.json 2> /dev/null
python bench.py --renderer hip --no-cpu-baseline > $OUT/bench_renderer_hip.json 2> /dev/null
# 3b. the default is two HIP streams per rank (CoarseRefinePosePredictor n_streams); the single-stream schedule and three streams on config 3 beside it
python bench.py --streams 1 --no-cpu-baseline --no-other-dtypes --no-profile > $OUT/bench_streams1.json 2> /dev/null
python bench.py --crop 240x320 --streams 1 --no-cpu-baseline --no-other-dtypes --no-profile > $OUT/bench_240x320_streams1.json 2> /dev/null
python bench.py --config 3 --split balanced --streams 3 --bsz-objects 128 --no-cpu-baseline --no-other-dtypes --no-profile > $OUT/bench_config3_balanced_streams3.json 2> /dev/null
python bench_train.py --kernels > $OUT/bench_train.json 2> $OUT/bench_train_kernels.txt
# 4. rocprofv3 stats + PMC passes of the headline command
bash profiles/collect.sh $TAG > $OUT/collect.log 2>&1
cp -r gpurun_out/prof_$TAG/summary.txt gpurun_out/prof_$TAG/pmc_traffic.json gpurun_out/prof_$TAG/rccl_kernels.csv $OUT/ 2>/dev/null
cp gpurun_out/prof_$TAG/stats/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
# 1. headline line, exactly the driver's command -- after the PMC passes, with this call's traffic file in place (bench.py reports
#    roofline.traffic only from a file collected for the same kernel sources)
cp $OUT/pmc_traffic.json profiles/${TAG%[a-z]}_pmc_traffic.json 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
# 5. batch sweep: throughput and matrix-core utilisation of the GEMM kernels at 256..2048 crops per forward
for B in 256 512 1024 2048; do
  python bench.py --steps 4 --warmup 2 --detections $B --bsz-objects $B --streams 1 --no-cpu-baseline --no-profile --no-other-dtypes > $OUT/sweep_B$B.json 2> /dev/null
  rocprofv3 -M --kernel-trace --pmc MfmaUtil SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES -f csv -d $OUT/sweep_pmc_B$B -o t -- \
      python bench.py --steps 1 --warmup 1 --detections $B --bsz-objects $B --streams 1 --no-cpu-baseline --no-profile --no-other-dtypes > /dev/null 2> $OUT/sweep_pmc_B$B.err
done
python profiles/sweep_table.py $OUT > $OUT/batch_sweep.txt 2>&1
rm -rf $OUT/sweep_pmc_B*/  # raw counter dumps are large; the table is what is kept
tail -40 $OUT/batch_sweep.txt
cat $OUT/bench.json
