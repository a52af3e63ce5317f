cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -6 || exit 1
python3 tools/pmc_run.py 5e8 les128_aer; python3 tools/pmc_run.py 5e8 les128_aer
python3 tools/pmc_run.py 5e8 les480
