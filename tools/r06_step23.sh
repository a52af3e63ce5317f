#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in er3t_amd/libmi3drt.so tools/ab_NT1.so tools/ab_NT8.so tools/ab_NT9.so tools/ab_NT11.so; do
  echo "== $lib"
  MI3D_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/r06_rate.py les128_flux 1e8 4 2>&1 || exit 1
  MI3D_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/r06_rate.py les480_flux 5e7 4 2>&1 || exit 1
done
