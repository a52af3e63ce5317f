cd $GRAFT_REPO_ROOT
python tools/time_dropin.py 2>&1 | tail -4
MI3D_FUSED_SLOTS=1 python tools/time_dropin.py 2>&1 | tail -4
