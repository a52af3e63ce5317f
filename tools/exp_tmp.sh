cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_dropin.py -x -q 2>&1 | tail -5 || exit 1
python tools/time_dropin.py 2>&1 | tail -4
MI3D_FUSED_SLOTS=1 python tools/time_dropin.py 2>&1 | tail -4
