#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s40; rm -rf $O; mkdir -p $O
timeout -k 10 700 python -m pytest tests -x -q -m gpu -k "flux or tally or record or heat or overflow or config3 or config4 or list" > $O/pytest_flux.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -2 $O/pytest_flux.log
[ $rc -eq 0 ] || exit 1
for v in base keys2 keys3; do
MI3D_LIBRARY=$PWD/tools/ab_$v.so MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt_$v -o k --output-format csv -- python3 tools/pmc_run.py 1e8 les480_flux > $O/kt_$v.log 2>&1
echo "== $v"; python3 - $O/kt_$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/k_kernel_stats.csv', recursive=True)[0]
for row in list(csv.DictReader(open(f)))[:6]:
    print('%-60s calls %3s total %8.2f ms' % (row['Name'][:60], row['Calls'], float(row['TotalDurationNs'])/1e6))
PY
done
echo "== les480_flux 5e7, kernels on one stream" | tee -a $O/ab.log
AB_WORKLOAD=les480_flux MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0 timeout -k 10 300 python tools/ab.py 5e7 tools/ab_base.so tools/ab_keys3.so tools/ab_base.so tools/ab_keys3.so 2>&1 | tee -a $O/ab.log
echo "== les480_flux 5e7 x 4 back to back, default streams" | tee -a $O/ab.log
AB_WORKLOAD=les480_flux AB_STEPS=4 timeout -k 10 300 python tools/ab.py 5e7 tools/ab_base.so tools/ab_keys3.so 2>&1 | tee -a $O/ab.log
echo "== by grid size" | tee -a $O/ab.log
MI3D_LIBRARY=$PWD/tools/ab_keys3.so timeout -k 10 300 python tools/flux_grid_sizes.py 5e7 2>&1 | tee -a $O/ab.log
