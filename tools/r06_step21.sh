#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 2 3 4; do echo "== MI3D_FUSED_SLOTS=$n"; MI3D_FUSED_SLOTS=$n timeout -k 10 300 python tools/time_dropin.py 2>&1 | tail -4; MI3D_FUSED_SLOTS=$n python tools/r06_case.py 2>&1 | grep '"seconds"\|seconds_mcarats'; done
