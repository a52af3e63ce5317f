#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/r06_rate.py les128_mie 2e8 4 2>&1 || exit 1
MI3D_NO_MIX3=1 timeout -k 10 120 python tools/r06_rate.py les128_mie 2e8 4 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dropin.py tests/test_k16.py -x -q -m gpu -k "single_histories or mie or tabulated or table or ref_vs_cot or k16" 2>&1 | tail -4
