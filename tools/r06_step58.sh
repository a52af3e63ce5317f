#!/bin/bash
# photons per lane of a small launch (MI3D_PHOTONS_PER_LANE): the retrieval curve, the published case, config-3 jobs through files
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s58; rm -rf $O; mkdir -p $O
for n in 1 4 8 16; do echo "== MI3D_PHOTONS_PER_LANE=$n" | tee -a $O/ppl.log
 MI3D_PHOTONS_PER_LANE=$n timeout -k 10 200 python tools/profile_ref_vs_cot.py 1e7 2>&1 | grep "optical thicknesses" | tee -a $O/ppl.log
 MI3D_PHOTONS_PER_LANE=$n timeout -k 10 200 python tools/time_dropin.py 2>&1 | grep "files\|fused" | tee -a $O/ppl.log
 for w in les128 les128_flux; do for p in 6e5 2e6; do MI3D_PHOTONS_PER_LANE=$n AB_WORKLOAD=$w timeout -k 10 100 python tools/small_runs.py $p 24 2>&1 | tail -1 | cut -c1-120 | sed "s/^/$w /" | tee -a $O/ppl.log; done; done
done
