cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/kt_small -o x --output-format csv -- python3 tools/small_launches.py les128_aer 1000000 16 > gpurun_out/kt_small.log 2>&1
cut -d, -f1-4,6-7 gpurun_out/kt_small/x_kernel_stats.csv | sed 's/"void mi3d::\([a-z_]*\)<\([^>]*\)>[^"]*"/\1<\2>/; s/"mi3d::\([a-z_]*\)([^"]*"/\1/' | head -12
python3 - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/kt_small/x_kernel_trace.csv'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=None
for r in rows[-14:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if t0 is None: t0=s
    print('%-40s start %8.1f us  dur %8.1f us' % (r['Kernel_Name'][:40], (s-t0)/1e3, (e-s)/1e3))
PY
