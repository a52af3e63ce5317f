#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/c29; rm -rf $O; mkdir -p $O
for lib in er3t_amd/libmi3drt.so tools/ab_st128.so; do
  echo "== $lib"
  MI3D_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/r06_rate.py les128_flux 1e8 4 2>&1 || exit 1
  MI3D_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/r06_rate.py les480_flux 5e7 4 2>&1 || exit 1
done
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
