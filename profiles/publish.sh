#!/bin/bash
# Copy the judged artefacts of a profiles/collect_all.sh run from gpurun_out/ (scratch) into profiles/ (tracked):
#   profiles/publish.sh <tag> [round-prefix]      e.g. profiles/publish.sh r02c r02
TAG=${1:?tag}; R=${2:-r06}
S=gpurun_out/all_$TAG
cp $S/bench.json profiles/${R}_bench.json
cp $S/layers_events.txt profiles/${R}_layers_events.txt
cp $S/summary.txt profiles/${R}_pmc_summary.txt
cp $S/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp $S/kernel_stats.csv profiles/${R}_rocprofv3_kernel_stats.csv
cp $S/batch_sweep.txt profiles/${R}_batch_sweep.txt
cp $S/rccl_kernels.csv profiles/${R}_rccl_1rank_log.txt
cp $S/bench_train.json profiles/${R}_bench_train.json
grep -E "time by family|  gemm " $S/bench_train_kernels.txt > profiles/${R}_bench_train_families.txt
python - <<PY
import json, glob, os
out = {}
for f in sorted(glob.glob('$S/bench_*.json')) + sorted(glob.glob('$S/sweep_B*.json')):
    try:
        d = json.load(open(f))
    except Exception:
        continue
    out[os.path.basename(f)[:-5]] = {k: d.get(k) for k in ('metric', 'value', 'unit', 'ms_per_step', 'dtype', 'n_gpus') } | {'workload': (d.get('config') or {}).get('workload'), 'split_ms': d.get('split_ms')}
json.dump(out, open('profiles/${R}_bench_variants.json', 'w'), indent=1)
print(json.dumps({k: v['value'] for k, v in out.items()}, indent=1))
PY
