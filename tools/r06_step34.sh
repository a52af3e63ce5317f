#!/bin/bash
# the GPU suite, the smoke run and the driver's bench command on the tree as it stands
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/s34; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench.json.log 2> $O/bench.err; echo "bench rc $?"; cut -c1-600 $O/bench.json.log
