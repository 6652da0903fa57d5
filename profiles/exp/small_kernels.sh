cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sk; rocprofv3 -M --kernel-trace --stats -f csv -d gpurun_out/sk -o t -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-other-dtypes > /dev/null 2> gpurun_out/sk.err
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/sk/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in sorted(rows,key=lambda r:-int(r['Calls'])):
    if float(r['AverageNs'])<30000: print(f"{int(r['Calls']):6d} calls  avg {float(r['AverageNs'])/1e3:7.1f} us  total {float(r['TotalDurationNs'])/1e6:7.3f} ms  {r['Name'][:100]}")
PY
rm -rf gpurun_out/sk
