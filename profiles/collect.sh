#!/bin/bash
# Profile `bench.py` on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh <tag> [bench args...]
# 1. rocprofv3 --kernel-trace --stats  (per-kernel time)            -> gpurun_out/prof_<tag>/stats
# 2. PMC passes, each in its own run with counters only (never combined with sys/hip tracing)
#    -> gpurun_out/prof_<tag>/pmc_*
# Summaries for the judge are produced by profiles/summarize.py and copied into profiles/.
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# --streams 1: the per-kernel numbers (rocprofv3 averages, PMC traffic per launch) are for ONE stream of full-size launches, like the HIP-event
# pass that bench.py's `roofline` comes from (under concurrency a kernel's counters and duration include its neighbours')
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-other-dtypes --streams 1 $*"
rocprofv3 -M --kernel-trace --stats -f csv -d $OUT/stats -o t -- python bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.err
pass() { # name counters...
  local name=$1; shift
  rocprofv3 -M --kernel-trace --pmc "$@" -f csv -d $OUT/pmc_$name -o t -- python bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err || echo "pmc pass $name failed"
}
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass mfma MfmaUtil SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum
# the collectives through RCCL on this one GPU (forced 1-rank group).  With a single rank RCCL completes an all-gather as a device copy
# (no ncclDevKernel appears in a kernel trace), so the evidence is RCCL's own log of the communicator and of every collective call.
COSY_FORCE_DIST=1 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL python bench.py $ARGS > $OUT/rccl.log 2>&1
tail -1 $OUT/rccl.log > $OUT/bench_rccl.json
grep -E "RCCL version|HIP version|ROCm version|Init COMPLETE|comm 0x[0-9a-f]+ rank|Init timings|AllGather|AllReduce|Abort COMPLETE" $OUT/rccl.log | cut -c1-260 | awk '!/AllGather/ || ++n<=6' > $OUT/rccl_kernels.csv
python profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
