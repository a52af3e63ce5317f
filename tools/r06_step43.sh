#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s43; rm -rf $O; mkdir -p $O
for w in les480_flux les128_flux; do
echo "== $w 5e7, kernels on one stream" | tee -a $O/ab.log
AB_WORKLOAD=$w MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0 timeout -k 10 300 python tools/ab.py 5e7 tools/ab_base.so tools/ab_dyn.so tools/ab_base.so tools/ab_dyn.so 2>&1 | tee -a $O/ab.log
echo "== $w 5e7 x 4 back to back, default streams" | tee -a $O/ab.log
AB_WORKLOAD=$w AB_STEPS=4 timeout -k 10 300 python tools/ab.py 5e7 tools/ab_base.so tools/ab_dyn.so 2>&1 | tee -a $O/ab.log
done
MI3D_LIBRARY=$PWD/tools/ab_dyn.so timeout -k 10 300 python tools/r06_flux_ab.py les480_flux 2e7 2 2>&1 | tail -8
