#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/c52; rm -rf $O; mkdir -p $O
timeout -k 10 300 python tools/r06_flux_ab.py les480_flux 5e7 4 > $O/ab_les480_flux.log 2>&1; echo "les480 rc $?"; grep -v atomics $O/ab_les480_flux.log | tail -4
timeout -k 10 300 python tools/r06_flux_ab.py les128_flux 1e8 4 > $O/ab_les128_flux.log 2>&1; echo "les128 rc $?"; grep -v atomics $O/ab_les128_flux.log | tail -4
timeout -k 10 900 python -m pytest tests/test_lib_abi.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "stream_order or flux or heat or record or pre_pass" 2>&1 | tail -3
