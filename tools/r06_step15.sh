#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in er3t_amd/libmi3drt.so tools/ab_ray1.so; do
  echo "== $lib"
  MI3D_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/r06_rate.py les128_mie 2e8 4 2>&1 || exit 1
done
timeout -k 10 120 python tools/r06_rate.py les128 2e8 4 2>&1
MI3D_FORCE_GEN=1 timeout -k 10 120 python tools/r06_rate.py les128 2e8 4 2>&1
