#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "X=1" "MI3D_PRE_NADIR=1" "MI3D_PRE_NADIR=2"; do
  echo "== $spec"
  env $spec timeout -k 10 200 python tools/r06_rate.py les480 1e9 4 2>&1 || exit 1
done
