#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MI3D_LIBRARY=$PWD/tools/ab_runsdiag.so python tools/r06_runsdiag.py les128_flux 5e7
MI3D_LIBRARY=$PWD/tools/ab_runsdiag.so python tools/r06_runsdiag.py les480_flux 5e7
bash tools/r06_step4.sh $1
