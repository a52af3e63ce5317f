#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "X=1" "MI3D_FLUX_GRID_WG=3" "MI3D_FLUX_GRID_WG=2" "MI3D_OVERLAP_PRE=0" "MI3D_FLUX_GRID_WG=3 MI3D_OVERLAP_PRE=0" "MI3D_FLUX_GRID_WG=3 MI3D_TL_SPLIT=8"; do
  echo "== $spec"
  env $spec timeout -k 10 120 python tools/r06_rate.py les128_flux 1e8 4 2>&1 || exit 1
  env $spec timeout -k 10 120 python tools/r06_rate.py les480_flux 5e7 4 2>&1 || exit 1
done
