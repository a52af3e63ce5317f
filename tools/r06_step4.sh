#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/$1; rm -rf $O; mkdir -p $O
timeout -k 10 300 python tools/r06_flux_ab.py les128_flux 1e8 4 > $O/ab_les128_flux.log 2>&1; echo "les128 rc $?"; tail -7 $O/ab_les128_flux.log
timeout -k 10 300 python tools/r06_flux_ab.py les480_flux 5e7 4 > $O/ab_les480_flux.log 2>&1; echo "les480 rc $?"; tail -7 $O/ab_les480_flux.log
export MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0
for w in les128_flux les480_flux; do
 for r in 1; do
  MI3D_TALLY_RUNS=$r timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt_${w}_runs$r -o p --output-format csv -- python3 tools/pmc_run.py 5e7 $w > $O/kt_${w}_runs$r.log 2>&1 || echo "failed $w $r"
  echo "== $w runs=$r"
  python3 - $O/kt_${w}_runs$r <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if float(row['TotalDurationNs']) > 2e5: print('  %-60s calls %3s  total %8.2f ms' % (row['Name'].split('(')[0][-60:], row['Calls'], float(row['TotalDurationNs'])/1e6))
PY
 done
done
