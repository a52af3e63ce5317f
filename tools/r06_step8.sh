#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MI3D_TL_VERBOSE=1
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 120 python -u tools/r06_rate.py les128_flux 1e8 4 2>&1 || exit 1
done
for i in 1 2 3 4; do
  timeout -k 10 120 python -u tools/r06_rate.py les480_flux 5e7 4 2>&1 || exit 1
done
