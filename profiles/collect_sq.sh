#!/bin/bash
# Extra SQ-level PMC passes (instruction mix, LDS, occupancy) for kernel tuning: profiles/collect_sq.sh <tag> [bench args]
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-profile $*"
pass() { local name=$1; shift
  rocprofv3 -M --kernel-trace --pmc "$@" -f csv -d $OUT/pmc_$name -o t -- python bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err || echo "pmc pass $name failed"; }
pass mix SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE
pass lds SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass occ SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_CYCLES
python profiles/summarize_raw.py $OUT ${KFILTER:-dwconv} > $OUT/summary_sq.txt 2>&1
cat $OUT/summary_sq.txt | head -80
