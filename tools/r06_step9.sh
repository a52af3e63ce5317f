#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/c17; rm -rf $O; mkdir -p $O
for lib in er3t_amd/libmi3drt.so tools/ab_r05.so; do
  n=$(basename $lib .so)
  MI3D_LIBRARY=$PWD/$lib AB_WORKLOAD=les128_flux AB_STEPS=2 timeout -k 10 200 rocprofv3 --kernel-trace -d $O/tr_$n -o trace --output-format csv -- python3 tools/overlap_trace.py 1e8 > $O/trace_$n.log 2>&1
  f=$(find $O/tr_$n -name "trace_kernel_trace.csv" | head -1)
  TRACE_ROWS=400 python tools/overlap_trace.py --read $f | cut -c1-120 > $O/timeline_$n.txt
done
