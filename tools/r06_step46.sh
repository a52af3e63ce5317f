#!/bin/bash
# small flux jobs read one by one: launches and wall time per job, the tree before (ab_base) and with the launch rule that tolerates a few per cent
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s46; rm -rf $O; mkdir -p $O
for lib in tools/ab_base.so er3t_amd/libmi3drt.so; do
 echo "== $lib" | tee -a $O/small.log
 for w in les128_flux les480_flux; do
  for n in 2e6 6e6 2e7 5e7; do MI3D_LIBRARY=$PWD/$lib AB_WORKLOAD=$w timeout -k 10 100 python tools/small_runs_launches.py $n 10 2>&1 | tail -1 | sed "s/^/$w /" | tee -a $O/small.log; done
 done
done
python tools/time_dropin.py 2>&1 | tail -6 | tee -a $O/small.log
