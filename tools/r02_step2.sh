set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_s2; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
MI3D_KERNEL=generic timeout -k 10 300 python tools/tile_sweep.py 1e8 les480 0 64 > $O/sweep_generic.log 2>&1; cat $O/sweep_generic.log
timeout -k 10 300 python tools/tile_sweep.py 1e8 les480 0 32 64 > $O/sweep_col.log 2>&1; cat $O/sweep_col.log
