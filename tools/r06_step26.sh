#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/c51; rm -rf $O; mkdir -p $O
export MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0
for lib in er3t_amd/libmi3drt.so tools/ab_nohist128.so tools/ab_nohist64.so; do
  n=$(basename $lib .so)
  MI3D_LIBRARY=$PWD/$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt_$n -o k --output-format csv -- python3 tools/pmc_run.py 1e8 les480_flux > $O/kt_$n.log 2>&1
  echo "== $n"; python3 - $O/kt_$n <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_transport_flux' in row['Name']: print('  %-50s calls %3s  total %8.2f ms' % (row['Name'].split('(')[0][-50:], row['Calls'], float(row['TotalDurationNs'])/1e6))
PY
done
