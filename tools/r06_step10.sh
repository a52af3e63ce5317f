#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/$1; rm -rf $O; mkdir -p $O
timeout -k 10 300 python tools/r06_flux_ab.py les128_flux 1e8 4 > $O/ab_les128_flux.log 2>&1; echo "les128 rc $?"; grep -v atomics $O/ab_les128_flux.log | tail -4
timeout -k 10 300 python tools/r06_flux_ab.py les480_flux 5e7 4 > $O/ab_les480_flux.log 2>&1; echo "les480 rc $?"; grep -v atomics $O/ab_les480_flux.log | tail -4
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "flux or heat or record or pre_pass" > $O/pytest_flux.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_flux.log
AB_WORKLOAD=les128_flux AB_STEPS=2 timeout -k 10 200 rocprofv3 --kernel-trace -d $O/tr -o trace --output-format csv -- python3 tools/overlap_trace.py 1e8 > $O/trace.log 2>&1
f=$(find $O/tr -name "trace_kernel_trace.csv" | head -1)
TRACE_ROWS=400 python tools/overlap_trace.py --read $f | cut -c1-120 > $O/timeline.txt
