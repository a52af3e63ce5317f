#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/c48; rm -rf $O; mkdir -p $O
for w in les480_mv9 les480_mv9_lambert; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o k --output-format csv -- python3 tools/pmc_run.py 1e8 $w > $O/kt_$w.log 2>&1
  echo "== $w"; tail -1 $O/kt_$w.log
  python3 - $O/kt_$w <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if float(row['TotalDurationNs']) > 1e6: print('  %-70s calls %3s  total %8.2f ms' % (row['Name'].split('(')[0][-70:], row['Calls'], float(row['TotalDurationNs'])/1e6))
PY
done
