#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s38; rm -rf $O; mkdir -p $O
for v in base keys keysA; do
MI3D_LIBRARY=$PWD/tools/ab_$v.so MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt_$v -o k --output-format csv -- python3 tools/pmc_run.py 1e8 les480_flux > $O/kt_$v.log 2>&1
echo "== $v"; python3 - $O/kt_$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/k_kernel_stats.csv', recursive=True)[0]
for row in list(csv.DictReader(open(f)))[:6]:
    print('%-60s calls %3s total %8.2f ms' % (row['Name'][:60], row['Calls'], float(row['TotalDurationNs'])/1e6))
PY
done
echo "== les480_flux 5e7, kernels on one stream" | tee -a $O/ab.log
AB_WORKLOAD=les480_flux MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0 timeout -k 10 300 python tools/ab.py 5e7 tools/ab_base.so tools/ab_keys.so tools/ab_base.so tools/ab_keys.so 2>&1 | tee -a $O/ab.log
echo "== les480_flux 5e7 x 4 back to back, default streams" | tee -a $O/ab.log
AB_WORKLOAD=les480_flux AB_STEPS=4 timeout -k 10 300 python tools/ab.py 5e7 tools/ab_base.so tools/ab_keys.so 2>&1 | tee -a $O/ab.log
