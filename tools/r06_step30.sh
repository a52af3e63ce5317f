#!/bin/bash
# A/B: wave-aggregated places in k_tl_runs (480 x 480), 1024-thread workgroups of it, s_setprio in the photon loops
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s30; mkdir -p $O
L=tools
echo "== les480_flux 5e7, kernels on one stream" | tee -a $O/ab.log
AB_WORKLOAD=les480_flux MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0 timeout -k 10 400 python tools/ab.py 5e7 $L/ab_base.so $L/ab_agg.so $L/ab_agg1024.so $L/ab_priow.so $L/ab_prioc.so 2>&1 | tee -a $O/ab.log
echo "== les480_flux 5e7 x 4 back to back, default streams" | tee -a $O/ab.log
AB_WORKLOAD=les480_flux AB_STEPS=4 timeout -k 10 400 python tools/ab.py 5e7 $L/ab_base.so $L/ab_agg.so $L/ab_agg1024.so 2>&1 | tee -a $O/ab.log
echo "== les480 5e8" | tee -a $O/ab.log
timeout -k 10 400 python tools/ab.py 5e8 $L/ab_base.so $L/ab_priow.so $L/ab_priow1.so $L/ab_prioc.so $L/ab_base.so 2>&1 | tee -a $O/ab.log
echo "== les480_mv9 4e7" | tee -a $O/ab.log
AB_WORKLOAD=les480_mv9 timeout -k 10 400 python tools/ab.py 4e7 $L/ab_base.so $L/ab_priow.so $L/ab_prioc.so 2>&1 | tee -a $O/ab.log
echo "== les128_flux 5e7" | tee -a $O/ab.log
AB_WORKLOAD=les128_flux timeout -k 10 300 python tools/ab.py 5e7 $L/ab_base.so $L/ab_priow.so $L/ab_prioc.so 2>&1 | tee -a $O/ab.log
