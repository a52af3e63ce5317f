#!/bin/bash
# compiler scheduling strategies (-mllvm: no change of floating-point semantics) on the photon loops
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s53; rm -rf $O; mkdir -p $O
L=tools
echo "== les480 5e8" | tee -a $O/ab.log
timeout -k 10 600 python tools/ab.py 5e8 $L/ab_base.so $L/ab_maxilp.so $L/ab_memcl.so $L/ab_bias100.so $L/ab_bias0.so $L/ab_trackers.so $L/ab_noalign.so $L/ab_base.so 2>&1 | tee -a $O/ab.log
echo "== les128_flux 5e7 x 4 back to back" | tee -a $O/ab.log
AB_WORKLOAD=les128_flux AB_STEPS=4 timeout -k 10 600 python tools/ab.py 5e7 $L/ab_base.so $L/ab_maxilp.so $L/ab_memcl.so $L/ab_bias100.so $L/ab_bias0.so $L/ab_trackers.so 2>&1 | tee -a $O/ab.log
echo "== les480_mv9 4e7" | tee -a $O/ab.log
AB_WORKLOAD=les480_mv9 timeout -k 10 600 python tools/ab.py 4e7 $L/ab_base.so $L/ab_maxilp.so $L/ab_memcl.so $L/ab_bias100.so $L/ab_bias0.so $L/ab_trackers.so 2>&1 | tee -a $O/ab.log
echo "== les128_mie 2e8" | tee -a $O/ab.log
AB_WORKLOAD=les128_mie timeout -k 10 600 python tools/ab.py 2e8 $L/ab_base.so $L/ab_maxilp.so $L/ab_memcl.so $L/ab_bias100.so $L/ab_bias0.so $L/ab_trackers.so 2>&1 | tee -a $O/ab.log
