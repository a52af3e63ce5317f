#!/bin/bash
# level-aligned tally bins: the flux tests, then A/B against the tree before by grid size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s35; rm -rf $O; mkdir -p $O
timeout -k 10 700 python -m pytest tests -x -q -m gpu -k "flux or tally or record or heat or overflow or config3 or config4 or list" > $O/pytest_flux.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -15 $O/pytest_flux.log
[ $rc -eq 0 ] || exit 1
for w in les480_flux les128_flux; do
echo "== $w 5e7, kernels on one stream" | tee -a $O/ab.log
AB_WORKLOAD=$w MI3D_OVERLAP_SORT=0 MI3D_OVERLAP_PRE=0 timeout -k 10 300 python tools/ab.py 5e7 tools/ab_base.so tools/ab_keys.so tools/ab_base.so tools/ab_keys.so 2>&1 | tee -a $O/ab.log
echo "== $w 5e7 x 4 back to back, default streams" | tee -a $O/ab.log
AB_WORKLOAD=$w AB_STEPS=4 timeout -k 10 300 python tools/ab.py 5e7 tools/ab_base.so tools/ab_keys.so 2>&1 | tee -a $O/ab.log
done
echo "== by grid size, level-aligned bins" | tee -a $O/ab.log
MI3D_LIBRARY=$PWD/tools/ab_keys.so timeout -k 10 300 python tools/flux_grid_sizes.py 5e7 2>&1 | tee -a $O/ab.log
