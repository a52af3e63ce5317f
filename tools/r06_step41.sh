#!/bin/bash
# the ray kernel: what does its walk pay per gathered line?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s41; rm -rf $O; mkdir -p $O
for v in base rgather2; do
MI3D_LIBRARY=$PWD/tools/ab_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt_$v -o k --output-format csv -- python3 tools/pmc_run.py 4e7 les480_mv9 > $O/kt_$v.log 2>&1
echo "== $v $(tail -1 $O/kt_$v.log)"; python3 - $O/kt_$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/k_kernel_stats.csv', recursive=True)[0]
for row in list(csv.DictReader(open(f)))[:4]:
    print('%-70s calls %3s total %8.2f ms' % (row['Name'][:70], row['Calls'], float(row['TotalDurationNs'])/1e6))
PY
done
MI3D_LIBRARY=$PWD/tools/ab_base.so timeout -k 10 200 python tools/sched_diag_mv.py 2>&1 | tail -12
