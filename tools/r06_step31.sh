#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s31; mkdir -p $O
L=tools
echo "== les480_flux 5e7, kernels on one stream" | tee -a $O/ab.log
AB_WORKLOAD=les480_flux MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0 timeout -k 10 400 python tools/ab.py 5e7 $L/ab_base.so $L/ab_agg2.so $L/ab_base1024.so $L/ab_base.so $L/ab_agg2.so 2>&1 | tee -a $O/ab.log
echo "== les480_flux 5e7 x 4 back to back, default streams" | tee -a $O/ab.log
AB_WORKLOAD=les480_flux AB_STEPS=4 timeout -k 10 400 python tools/ab.py 5e7 $L/ab_base.so $L/ab_agg2.so $L/ab_base1024.so 2>&1 | tee -a $O/ab.log
