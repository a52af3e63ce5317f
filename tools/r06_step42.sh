#!/bin/bash
# the headline loop's walk threshold and full-pass cadence swept again on the round's final tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s42; rm -rf $O; mkdir -p $O
L=tools
echo "== les480 5e8" | tee -a $O/ab.log
timeout -k 10 600 python tools/ab.py 5e8 $L/ab_base.so $L/ab_thr12.so $L/ab_thr20.so $L/ab_thr24.so $L/ab_fp6.so $L/ab_fp10.so $L/ab_fp12.so $L/ab_thr20fp10.so $L/ab_base.so 2>&1 | tee -a $O/ab.log
echo "== les128 2e8" | tee -a $O/ab.log
AB_WORKLOAD=les128 timeout -k 10 600 python tools/ab.py 2e8 $L/ab_base.so $L/ab_thr12.so $L/ab_thr20.so $L/ab_fp6.so $L/ab_fp10.so $L/ab_thr20fp10.so 2>&1 | tee -a $O/ab.log
