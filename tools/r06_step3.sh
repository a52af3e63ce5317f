#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/c3; rm -rf $O; mkdir -p $O
export MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0
for w in les128_flux les480_flux; do
 for r in 0 1; do
  MI3D_TALLY_RUNS=$r timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt_${w}_runs$r -o p --output-format csv -- python3 tools/pmc_run.py 5e7 $w > $O/kt_${w}_runs$r.log 2>&1 || echo "failed $w $r"
  echo "== $w runs=$r"; tail -1 $O/kt_${w}_runs$r.log
  python3 - $O/kt_${w}_runs$r <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if float(row['TotalDurationNs']) > 2e5: print('  %-60s calls %3s  total %8.2f ms' % (row['Name'].split('(')[0][-60:], row['Calls'], float(row['TotalDurationNs'])/1e6))
PY
 done
done
