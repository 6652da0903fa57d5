#!/bin/bash
# SQ-level counters of the wave kernels (separate counter-only passes): profiles/exp/pmc_wave.sh <outdir> [env assignments for the bench...]
OUT=$1; shift
mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-other-dtypes --streams 1"
pass() { local name=$1; shift
  timeout 300 rocprofv3 -M --kernel-trace --pmc "$@" -f csv -d $OUT/pmc_$name -o t -- python bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err || echo "pmc pass $name failed"; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
pass b SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_IFETCH
pass c SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
pass d SQ_WAVE_CYCLES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
pass e SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_VMEM GRBM_GUI_ACTIVE
python - <<PY
import csv, glob, collections
for name in 'abcde':
    fs = glob.glob('$OUT/pmc_%s/**/*counter_collection.csv' % name, recursive=True)
    if not fs:
        print('pass', name, 'no output'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if 'mbconv_wave_kernel' not in k: continue
        key = k[k.index('kernelI') + 7:].split('EEv')[0]      # mangled template arguments (rocprofv3 -M)
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key in sorted(agg):
        if not (key.startswith('DF16_') and any(t in key for t in ('Li5ELi1ELi5ELi1ELi1ELb1', 'Li3ELi1ELi3ELi1ELi1ELb1', 'Li5ELi1ELi3ELi1ELi1ELb1'))): continue
        c = {n: sum(v) / len(v) for n, v in agg[key].items()}
        wc = c.get('SQ_WAVE_CYCLES')
        print(name, key, ' '.join(f'{n[3:] if n.startswith("SQ_") else n}={v:.4g}' + (f'({v / wc:.2f})' if wc and n != 'SQ_WAVE_CYCLES' else '') for n, v in sorted(c.items())))
PY
