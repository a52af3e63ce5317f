cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5 || exit 1
for w in les480_mv9 les480_mv9_lambert; do python3 tools/pmc_run.py 4e7 $w; python3 tools/pmc_run.py 4e7 $w; done
