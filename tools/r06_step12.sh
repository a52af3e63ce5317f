#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/c25; rm -rf $O; mkdir -p $O
for m in 1 2; do
  timeout -k 10 100 rocprofv3 --kernel-trace --stats -d $O/m$m -o t --output-format csv -- python3 tools/r06_dbg_b2b2.py $m > $O/m$m.log 2>&1
  tail -1 $O/m$m.log
  python3 - $O/m$m <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_t' in row['Name']: print('  %-50s calls %5s' % (row['Name'].split('(')[0][-50:], row['Calls']))
PY
done
