export AB_WORKLOAD=les128_flux
python tools/ab.py 2e7 tools/ab_base.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
