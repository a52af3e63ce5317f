#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== round-6 library"; timeout -k 10 300 python tools/time_dropin.py 2>&1 | tail -5
echo "== round-5 library"; MI3D_LIBRARY=$PWD/tools/ab_r05.so timeout -k 10 300 python tools/time_dropin.py 2>&1 | tail -5
AB_WORKLOAD=les128_flux timeout -k 10 120 python tools/small_runs.py 6e6 24 2>&1
MI3D_LIBRARY=$PWD/tools/ab_r05.so AB_WORKLOAD=les128_flux timeout -k 10 120 python tools/small_runs.py 6e6 24 2>&1
