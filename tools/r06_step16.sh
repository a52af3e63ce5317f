#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "X=1" "MI3D_TL_LAZY=0" "MI3D_OVERLAP_SORT=0"; do
  echo "== $spec"
  env $spec timeout -k 10 120 python tools/r06_rate.py les128_flux 1e8 4 2>&1 || exit 1
  env $spec timeout -k 10 120 python tools/r06_rate.py les480_flux 5e7 4 2>&1 || exit 1
  env $spec AB_WORKLOAD=les128_flux timeout -k 10 120 python tools/small_runs.py 6e6 24 2>&1 || exit 1
done
