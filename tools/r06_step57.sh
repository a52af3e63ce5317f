#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s57; rm -rf $O; mkdir -p $O
for n in 2 3 4; do echo "== MI3D_FILE_SLOTS=$n" | tee -a $O/slots.log; MI3D_FILE_SLOTS=$n MI3D_FUSED_SLOTS=$n timeout -k 10 200 python tools/profile_ref_vs_cot.py 1e7 2>&1 | grep "optical thicknesses" | tee -a $O/slots.log; MI3D_FILE_SLOTS=$n MI3D_FUSED_SLOTS=$n timeout -k 10 200 python tools/profile_dropin_flux.py 2>&1 | grep "flux files" | tail -1 | tee -a $O/slots.log; done
