cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k pooled 2>&1 | tail -5 || exit 1
export MI3D_KERNEL=pool
for l in pool_s32 pool_s40 pool_w4; do
  export MI3D_LIBRARY=$GRAFT_REPO_ROOT/tools/ab_$l.so
  echo $l $(timeout -k 10 200 python3 tools/pmc_run.py 5e8 les480 | tail -1)
done
unset MI3D_LIBRARY
echo pool_default $(python3 tools/pmc_run.py 5e8 les480 | tail -1)
echo mv9 pool $(python3 tools/pmc_run.py 4e7 les480_mv9 | tail -1)
unset MI3D_KERNEL
echo mv9 lean $(python3 tools/pmc_run.py 4e7 les480_mv9 | tail -1)
