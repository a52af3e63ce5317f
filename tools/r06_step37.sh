#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s37; rm -rf $O; mkdir -p $O
for v in base keys; do
echo "== $v"
MI3D_TL_VERBOSE=1 MI3D_LIBRARY=$PWD/tools/ab_$v.so MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0 timeout -k 10 200 python3 tools/pmc_run.py 1e8 les480_flux 2>&1 | tail -6
for w in les128_flux; do
MI3D_LIBRARY=$PWD/tools/ab_$v.so MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt_${v}_$w -o k --output-format csv -- python3 tools/pmc_run.py 1e8 $w > $O/kt_${v}_$w.log 2>&1
python3 - $O/kt_${v}_$w <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/k_kernel_stats.csv', recursive=True)[0]
for row in list(csv.DictReader(open(f)))[:6]:
    print('%-60s calls %3s total %8.2f ms' % (row['Name'][:60], row['Calls'], float(row['TotalDurationNs'])/1e6))
PY
done
done
