cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tri -o tri -- python tools/bench_tri.py > gpurun_out/prof_tri.log 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof_tri/tri_kernel_stats.csv')):
    n=r['Name'].split('(')[0][-60:]
    print("%-62s %6s avg %9.0f min %8s"%(n,r['Calls'],float(r['AverageNs']),r['MinNs']))
PY
